"""The 16-bit throughput modes OUTSIDE O(1) activations ("what's weak" 1 / next-round item 3), taken on the CPU
with oracle/lowprec_model.py (the model of the GPU's rounding sites that predicts its measured errors to three digits) against
the fp32 oracle: the guided forward and the ddim10 loop of the bench configuration with synthetic weights pushed towards trained
statistics (audio2photoreal_amd.synthetic.trained_like_state_dict) -- every Linear weight x {2, 4, 8}, peaky attention logits,
a residual stream of 1e3 .. 1e4 units.  Reports, per scenario: the largest |logit| and the largest 16-bit-stored operand the
fp32 oracle sees (65504 is IEEE half's ceiling), whether the 16-bit model stays finite, and the errors.

usage: python tests/tools/trained_like_budget.py [--model face] [--T 600] [--steps 10] [--out profiles/r04_trained_like_budget.json]
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
from audio2photoreal_amd.synthetic import synthetic_inputs, trained_like_state_dict   # noqa: E402
from oracle import a2p_oracle as O                                   # noqa: E402
from oracle import lowprec_model as LP                               # noqa: E402

SCENARIOS = {
    "xavier (the round-3 fixtures)": {},
    "weights x2": {"weight_gain": 2.0},
    "weights x4": {"weight_gain": 4.0},
    "weights x8": {"weight_gain": 8.0},
    "q,k rows x1.5": {"qk_gain": 1.5},
    "q,k rows x2": {"qk_gain": 2.0},
    "q,k rows x3": {"qk_gain": 3.0},
    "q,k rows x4": {"qk_gain": 4.0},
    "q,k rows x5": {"qk_gain": 5.0},
    "residual stream x1e2": {"resid_gain": 1e2},
    "residual stream x1e3": {"resid_gain": 1e3},
    "residual stream x1e4": {"resid_gain": 1e4},
    "weights x4 + q,k x2 + residual x1e3": {"weight_gain": 4.0, "qk_gain": 2.0, "resid_gain": 1e3},
}


class Probe(LP.Rounding):
    """fp32 pass-through that records the largest magnitude per rounding site (what the 16-bit formats would have to hold)."""

    def __init__(self):
        super().__init__("fp32")
        self.peak = {}
        self.logit_peak = 0.0

    def __call__(self, site, x):
        self.peak[site] = max(self.peak.get(site, 0.0), float(x.abs().max()))
        return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="face")
    ap.add_argument("--T", type=int, default=600)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_trained_like_budget.json"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    fmt = a.model
    spec = face_spec() if fmt == "face" else pose_spec()
    B, T = 1, a.T
    inp = synthetic_inputs(spec, B, T, 10)
    scale = torch.full((B,), 10.0 if fmt == "face" else 2.0)
    kf, mk = (inp["keyframes"], inp["mask"]) if spec.is_pose else (None, None)
    samp = O.OracleSampler("ddim10")
    t700 = torch.tensor([700])
    rel = lambda g, w: float((g - w).norm() / w.norm())
    res = {"config": {"model": fmt, "T": T, "B": B, "scale": float(scale[0]), "sampler": f"ddim10 x {a.steps} steps",
                      "tool": "oracle/lowprec_model.py vs oracle/a2p_oracle.py (CPU)"}, "rows": {}}
    for name, kw in SCENARIOS.items():
        if a.only and a.only not in name:
            continue
        t0 = time.time()
        sd = trained_like_state_dict(spec, 10, **kw)
        # a trained denoiser predicts normalised motion: rescale the output head so that the GUIDED output of the fp32 oracle has
        # unit scale whatever the scenario did to the stack (otherwise the ddim loop itself diverges, in fp32 too)
        with torch.no_grad():
            probe_out = O.OracleDenoiser(sd, fmt, spec.num_layers, spec.num_heads).forward_cfg(inp["x_T"], t700, inp["cond_embed"], scale, kf, mk)
        g = float(probe_out.std())
        head = "final_layer" if fmt == "face" else [k for k in sd if k.startswith("final_layer") or k.startswith("pose_tail") or "conv" in k.split(".")[0]][0].split(".")[0]
        for k in ("final_layer.weight", "final_layer.bias"):
            if k in sd:
                sd[k] = sd[k] / g

        def run(den):
            with torch.no_grad():
                fwd = den.forward_cfg(inp["x_T"], t700, inp["cond_embed"], scale, kf, mk)
                fn = lambda x, ts: den.forward_cfg(x, ts, inp["cond_embed"], scale, kf, mk)
                x0, _ = samp.ddim_sample_loop(fn, inp["x_T"], max_steps=a.steps)
            return fwd, x0

        want_fwd, want_x0 = run(O.OracleDenoiser(sd, fmt, spec.num_layers, spec.num_heads))
        probe = Probe()
        with torch.no_grad():
            LP.LowPrecDenoiser(sd, fmt, spec.num_layers, spec.num_heads, probe).forward_cfg(inp["x_T"], t700, inp["cond_embed"], scale, kf, mk)
        row = {"scenario": kw, "head_rescale": 1.0 / g, "oracle_out_absmax": float(want_fwd.abs().max()),
               "peak_16bit_operand": max(probe.peak.values()), "peak_site": max(probe.peak, key=probe.peak.get),
               "peak_logit_operands": {k: probe.peak[k] for k in ("self.q", "self.k", "cross.q") if k in probe.peak},
               "peak_logit": getattr(probe, "logit_peak", None)}
        for base in ("fp16", "bf16"):
            fwd, x0 = run(LP.LowPrecDenoiser(sd, fmt, spec.num_layers, spec.num_heads, LP.Rounding(base, {s: "fp16x2" if base == "fp16" else "fp32" for s in ("fin.a", "fin.w", "in.a", "in.w", "tail.a", "tail.w")})))
            fin = bool(torch.isfinite(fwd).all() and torch.isfinite(x0).all())
            row[base] = {"finite": fin, "fwd_rel_l2": rel(fwd, want_fwd) if fin else None, "loop_rel_l2": rel(x0, want_x0) if fin else None}
        res["rows"][name] = row
        print(f"{name:42s} |logit| <= {row['peak_logit']:.1f}  peak operand {row['peak_16bit_operand']:.3g} ({row['peak_site']})  fp16 {row['fp16']}  bf16 {row['bf16']}  ({time.time() - t0:.0f} s)", flush=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
