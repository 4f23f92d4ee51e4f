"""The FULL sampling chains of the benchmarked workloads run by the REFERENCE ITSELF (/root/reference, imported read-only, CPU fp32; tests/golden/ref_import.py's
three patches), on the inputs of tests/tools/chain_vs_oracle.py:

    body   BASELINE configs[2]: SpacedDiffusion("ddim100").ddim_sample_loop_progressive over ClassifierFreeSampleModel(FiLMTransformer pose), keyframes + mask, scale 2
    face   BASELINE configs[1]: SpacedDiffusion("").p_sample_loop_progressive (1000 DDPM steps, the restored p_sample noise injected per step), scale 10

-> tests/golden/golden_chain_<workload>_ref_v1.npz: the chain state after the same steps the oracle fixture (golden_chain_<workload>_v1.npz) holds.  The GPU suite gates
the product's chains against THESE (reference-produced) states; tests/test_oracle_golden.py checks that the oracle's states agree with them.  Build container only
(the reference tree does not travel):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_chain.py --workload body        # ~1 min
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_chain.py --workload face        # ~25 min on 4 threads
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

import chain_vs_oracle as CVO  # noqa: E402
import ref_import as ri  # noqa: E402
from make_golden import load_synth  # noqa: E402


class _NoiseFeed:
    """What ref_import's restored p_sample pops its noise from: step n gets chain_vs_oracle.noise_of(n) (no 600 MB list)."""

    def __init__(self, shape):
        self.shape, self.n = shape, 0

    def __bool__(self):
        return True

    def pop(self, _):
        z = CVO.noise_of(self.n, self.shape)
        self.n += 1
        return z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True, choices=sorted(CVO.WORKLOADS))
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    w = CVO.WORKLOADS[a.workload]
    spec, inp = CVO.inputs(a.workload)
    ns = ri.import_reference()
    out_path = a.out or os.path.join(HERE, f"golden_chain_{a.workload}_ref_v1.npz")
    steps = min(a.steps or w["steps"], w["steps"])
    saved, t0 = {}, time.time()
    with ri.cpu_cuda(), torch.no_grad():
        model, diff = ri.build_reference_model(ns, w["fmt"], spec.num_layers, spec.num_heads, w["respacing"])
        load_synth(model, spec)
        cfg_model = ns.cfg.ClassifierFreeSampleModel(model)
        model._a2p_cond_embed = inp["cond"]
        y = {"audio": torch.zeros(1, 1, 2), "scale": torch.full((1,), w["scale"])}
        if spec.is_pose:
            y["keyframes"], y["mask"] = inp["kf"].clone(), inp["mask"].clone()
        shape = tuple(inp["x"].shape)
        if w["sampler"] == "ddpm":
            diff._a2p_step_noise = _NoiseFeed(shape)
            gen = diff.p_sample_loop_progressive(cfg_model, shape, clip_denoised=False, model_kwargs={"y": y}, noise=inp["x"].clone())
        else:
            gen = diff.ddim_sample_loop_progressive(cfg_model, shape, clip_denoised=False, model_kwargs={"y": y}, noise=inp["x"].clone())
        for n, o in enumerate(gen):
            if (n + 1) in w["save"] or n + 1 == steps:
                saved[f"step{n + 1}"] = o["sample"].numpy().copy()
                np.savez(out_path, seconds=time.time() - t0, threads=a.threads, **saved)
                print(f"reference {a.workload} step {n + 1}: {time.time() - t0:.0f} s", flush=True)
            if n + 1 >= steps:
                break


if __name__ == "__main__":
    main()
