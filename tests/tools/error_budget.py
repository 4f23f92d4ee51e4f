"""Error budget of the 16-bit throughput mode on the LOOP's return value, taken on the CPU with
oracle/lowprec_model.py (a model of the GPU's rounding sites) against the fp32 oracle: face model, T=600, S=1998+2, B=1,
guidance scale 10, ddim10 -- the configuration tests/test_hip_round2.py::test_T600_forward_and_ddim10_vs_oracle measures on
the GPU (fp16: fwd 5.3e-4, ddim10 3.1e-3; bf16: 4.3e-3 / 2.5e-2), which validates the model.

usage: python tests/tools/error_budget.py [--steps 10] [--out profiles/r03_error_budget.json] [--only NAME ...]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from audio2photoreal_amd.spec import face_spec, pose_spec                      # noqa: E402
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict   # noqa: E402
from oracle import a2p_oracle as O                                   # noqa: E402
from oracle import lowprec_model as LP                               # noqa: E402

GROUPS = {
    "input x_t + weight": ["in.a", "in.w"],
    "LN->QKV act": ["qkv.a"], "QKV weight": ["qkv.w"],
    "self Q,K stored": ["self.q", "self.k"], "self V stored": ["self.v"], "self P": ["self.p"],
    "self attn out (act of out_proj)": ["self.o"], "self out_proj weight": ["oself.w"],
    "LN->Qcross act": ["qc.a"], "Qcross weight": ["qc.w"], "cross Q stored": ["cross.q"],
    "cross K/V cache storage": ["cross.kv"], "cross time-token tails": ["cross.tail"], "cross P": ["cross.p"],
    "cross attn out": ["cross.o"], "cross out_proj weight": ["ocross.w"],
    "LN->linear1 act": ["ff1.a"], "linear1 weight": ["ff1.w"], "GELU hidden": ["ff2.a"], "linear2 weight": ["ff2.w"],
    "final_layer act": ["fin.a"], "final_layer weight": ["fin.w"],
    "conditioning path operands": ["cond.a", "cond.w"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--T", type=int, default=600)
    ap.add_argument("--base", default="fp16")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_error_budget.json"))
    ap.add_argument("--scenarios", default="")
    ap.add_argument("--model", default="face")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    spec = face_spec() if a.model == "face" else pose_spec()
    fmt = a.model
    if fmt == "pose":
        GROUPS.update({"keyframe attn: LN->Q act + weight": ["qc2.a", "qc2.w"], "keyframe attn: stored Q, K/V, P": ["cross2.q", "cross2.kv", "cross2.p"],
                       "keyframe attn out + out_proj weight": ["cross2.o", "ocross2.w"], "conv tail act": ["tail.a"], "conv tail weight": ["tail.w"]})
    sd = synthetic_state_dict(spec, 10)
    B, T = 1, a.T
    inp = synthetic_inputs(spec, B, T, 10)
    scale = torch.full((B,), 10.0 if fmt == "face" else 2.0)
    kf, mk = (inp["keyframes"], inp["mask"]) if spec.is_pose else (None, None)
    ref = O.OracleDenoiser(sd, fmt, spec.num_layers, spec.num_heads)
    samp = O.OracleSampler("ddim10")
    t700 = torch.tensor([700])

    def run(den):
        with torch.no_grad():
            fwd = den.forward_cfg(inp["x_T"], t700, inp["cond_embed"], scale, kf, mk)
            fn = lambda x, ts: den.forward_cfg(x, ts, inp["cond_embed"], scale, kf, mk)
            x0, _ = samp.ddim_sample_loop(fn, inp["x_T"], max_steps=a.steps)
        return fwd, x0

    t0 = time.time()
    want_fwd, want_x0 = run(ref)
    print(f"oracle: {time.time() - t0:.1f} s", flush=True)
    rel = lambda g, w: float((g - w).norm() / w.norm())
    res = {"config": {"model": fmt, "T": T, "S": 2000, "B": B, "scale": float(scale[0]), "sampler": f"ddim10 x {a.steps} steps", "base": a.base,
                      "tool": "oracle/lowprec_model.py (CPU model of the GPU rounding sites)"}, "rows": {}}

    def case(name, rounding):
        t0 = time.time()
        fwd, x0 = run(LP.LowPrecDenoiser(sd, fmt, spec.num_layers, spec.num_heads, rounding))
        r = {"fwd_rel_l2": rel(fwd, want_fwd), "loop_rel_l2": rel(x0, want_x0)}
        res["rows"][name] = r
        print(f"{name:55s} fwd {r['fwd_rel_l2']:.3e}  loop {r['loop_rel_l2']:.3e}   ({time.time() - t0:.0f} s)", flush=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
        return r

    sc = [s for s in a.scenarios.split(",") if s]
    if not sc or "base" in sc:
        case("all fp32 (model == oracle check)", LP.Rounding("fp32"))
        allr = case(f"all {a.base}", LP.Rounding(a.base))
    if not sc or "only" in sc:
        for g, sites in GROUPS.items():      # ONLY this class rounded: its own contribution (variances add)
            case(f"only: {g}", LP.Rounding("fp32", {s: a.base for s in sites}))
    if not sc or "fix" in sc:
        acts = ["qkv.a", "self.o", "qc.a", "cross.o", "ff1.a", "ff2.a", "fin.a", "in.a"]
        wts = ["qkv.w", "oself.w", "qc.w", "ocross.w", "ff1.w", "ff2.w", "fin.w", "in.w"]
        attn = ["self.q", "self.k", "self.v", "self.p", "cross.q", "cross.kv", "cross.tail", "cross.p"]
        case("fix: GEMM activations exact", LP.Rounding(a.base, {s: "fp32" for s in acts}))
        case("fix: GEMM weights exact", LP.Rounding(a.base, {s: "fp32" for s in wts}))
        case("fix: GEMM activations + weights exact", LP.Rounding(a.base, {s: "fp32" for s in acts + wts}))
        case("fix: attention operands exact", LP.Rounding(a.base, {s: "fp32" for s in attn}))
        case("fix: everything per-step exact (only cond path rounded)", LP.Rounding(a.base, {s: "fp32" for s in acts + wts + attn}))
    if not sc or "plan" in sc:   # the candidates for the shipped mode
        case("plan: final_layer act exact", LP.Rounding(a.base, {"fin.a": "fp32"}))
        case("plan: final_layer act as fp16 hi+lo pair", LP.Rounding(a.base, {"fin.a": "fp16x2"}))
        case("plan: final_layer act + weight exact", LP.Rounding(a.base, {"fin.a": "fp32", "fin.w": "fp32"}))
        case("plan: final_layer + input_projection exact", LP.Rounding(a.base, {s: "fp32" for s in ("fin.a", "fin.w", "in.a", "in.w")}))
        if fmt == "pose":
            case("plan: final_layer + input_projection + conv tail exact", LP.Rounding(a.base, {s: "fp32" for s in ("fin.a", "fin.w", "in.a", "in.w", "tail.a", "tail.w")}))
        case("plan: final + input exact, cond path exact", LP.Rounding(a.base, {s: "fp32" for s in ("fin.a", "fin.w", "in.a", "in.w", "cond.a", "cond.w")}))


if __name__ == "__main__":
    main()
