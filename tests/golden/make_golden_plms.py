"""Golden vectors for the PLMS sampler and the reverse DDIM step, produced by the REFERENCE itself
(/root/reference, CPU fp32) on the same synthetic weights/inputs as make_golden.py.  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_plms.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_import as ri  # noqa: E402
from make_golden import SEED, load_synth, y_dict  # noqa: E402
from audio2photoreal_amd.spec import face_spec, pose_spec  # noqa: E402
from audio2photoreal_amd.synthetic import synthetic_inputs  # noqa: E402

CASES = {"face": dict(B=1, frames=240, orders=(2, 4)), "pose": dict(B=2, frames=240, orders=(3,))}


def main():
    torch.manual_seed(SEED)
    torch.set_num_threads(8)
    ns = ri.import_reference()
    out = {}
    with ri.cpu_cuda(), torch.no_grad():
        for fmt, case in CASES.items():
            spec = face_spec() if fmt == "face" else pose_spec()
            model, diff10 = ri.build_reference_model(ns, fmt, spec.num_layers, spec.num_heads, "ddim10")
            load_synth(model, spec)
            cfg_model = ns.cfg.ClassifierFreeSampleModel(model)
            scale = 10.0 if fmt == "face" else 2.0
            B, frames = case["B"], case["frames"]
            inp = synthetic_inputs(spec, B, frames, SEED)
            shape = (B, spec.nfeats, 1, frames)
            for order in case["orders"]:
                y = y_dict(model, spec, inp, B, frames, scale)
                res = diff10.plms_sample_loop(cfg_model, shape, clip_denoised=False, model_kwargs={"y": y},
                                              noise=inp["x_T"].clone(), order=order)
                out[f"{fmt}/plms10_order{order}"] = res.numpy()
            # first two PLMS steps (order 2): Euler start + one Adams-Bashforth step, with the intermediate state
            y = y_dict(model, spec, inp, B, frames, scale)
            gen = diff10.plms_sample_loop_progressive(cfg_model, shape, clip_denoised=False, model_kwargs={"y": y},
                                                      noise=inp["x_T"].clone(), order=2)
            for i, o in zip(range(2), gen):
                out[f"{fmt}/plms_step{i}/sample"] = o["sample"].numpy()
                out[f"{fmt}/plms_step{i}/pred_xstart"] = o["pred_xstart"].numpy()
            # one reverse-ODE step at spaced index 5
            y = y_dict(model, spec, inp, B, frames, scale)
            r = diff10.ddim_reverse_sample(cfg_model, inp["x_T"].clone(), torch.tensor([5] * B), clip_denoised=False,
                                           model_kwargs={"y": y})
            out[f"{fmt}/ddim_reverse_t5"] = r["sample"].numpy()
            print(fmt, "done", flush=True)
    np.savez(os.path.join(HERE, "golden_plms_v1.npz"), **out)
    print("wrote", len(out), "arrays,", sum(v.nbytes for v in out.values()) / 1e6, "MB")


if __name__ == "__main__":
    main()
