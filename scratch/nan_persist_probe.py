"""Why does a healthy forward produce inf / nan after a poisoned one on the same context?  (round 4, tests/test_hip_round4.py)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio2photoreal_amd import _lib  # noqa: E402
from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel  # noqa: E402
from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model  # noqa: E402
from audio2photoreal_amd.spec import face_spec  # noqa: E402
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict  # noqa: E402


def read(model, name, n):
    host = np.empty(n, np.float32)
    _lib.check(model._lib().a2p_debug_read(model._ctx, name.encode(), host.ctypes.data_as(C.c_void_p), host.nbytes), name)
    return host


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    dev = torch.device("cuda:0")
    spec = face_spec(num_layers=2)
    B, T = 1, 64
    inp = synthetic_inputs(spec, B, T, 10)
    model, diffusion = create_model_and_diffusion(default_args("face", layers=2, timestep_respacing="ddim5"), "test", precision=precision, max_batch=B)
    sd = synthetic_state_dict(spec, 10)
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0, device=dev)}
    x, t = inp["x_T"].to(dev), torch.tensor([3], device=dev)
    load_model(model, sd)
    ok0 = cfg(x, t, y)
    print("healthy first:", bool(torch.isfinite(ok0).all()))
    load_model(model, {**sd, "input_projection.weight": sd["input_projection.weight"] * 1e30})
    bad = cfg(x, t, y)
    print("poisoned finite:", bool(torch.isfinite(bad).all()))
    load_model(model, sd)
    ok1 = cfg(x, t, y)
    print("healthy again finite:", bool(torch.isfinite(ok1).all()), "equal to first:", bool(torch.equal(ok0, ok1)))
    for name, n in (("x", 2 * T * 512), ("film", 2 * 2 * 3 * 1024), ("ktail", 2 * 2 * 512), ("vtail", 2 * 2 * 512), ("tvec", 2 * 512), ("tokr", 2 * 512), ("mo", 2 * T * 256)):
        a = read(model, name, n)
        print(f"  {name}: non-finite {int((~np.isfinite(a)).sum())} of {n}")
    ok2 = cfg(x, t, y)
    print("third healthy call finite:", bool(torch.isfinite(ok2).all()))


if __name__ == "__main__":
    main()
