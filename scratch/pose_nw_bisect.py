"""Next-round bisect of the open item in DESIGN.md section 6: where do the 4- and 8-wave chain shapes first disagree for the body
model (d=256) on a small forward?  One-layer model, buffers read back with a2p_debug_read after a forward with each shape:
  vt  = V^T of layer 0, written only by the PRE kernel      -> PRE (LayerNorm + rotary + QKV) differs or not
  qk  = last writer is MID2's query projection              -> everything up to the second cross attention
  x   = residual stream after POST                          -> feed-forward / final FiLM
Run under gpurun:  python scratch/pose_nw_bisect.py
"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio2photoreal_amd import _lib
from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
from audio2photoreal_amd.spec import pose_spec
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict

dev = torch.device("cuda:0")
for layers in (1, 2, 6):
    spec = pose_spec(num_layers=layers)
    model, _ = create_model_and_diffusion(default_args("pose", layers=layers), "test", precision="bf16", max_batch=4)
    load_model(model, synthetic_state_dict(spec, 10))
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    B, T = 3, 448
    inp = synthetic_inputs(spec, B, T, 10)
    y = {"cond_embed": inp["cond_embed"].to(dev), "keyframes": inp["keyframes"].to(dev), "mask": inp["mask"].to(dev),
         "scale": torch.full((B,), 2.0, device=dev)}
    x, t = inp["x_T"].to(dev), torch.tensor([901, 417, 33], device=dev)
    lib, snaps = _lib.load(), {}
    for nw in ("4", "8"):
        for mt in ("2", "3", "4"):
            os.environ["A2P_CHAIN_NW"], os.environ["A2P_CHAIN_MT"] = nw, mt
            out = cfg(x, t, y).float().cpu().numpy()
            bufs = {"out": out}
            for name, nbytes in (("vt", 2 * B * 256 * 448 * 2), ("qk", 2 * B * T * 512 * 2), ("x", 2 * B * T * 256 * 4)):
                a = np.empty(nbytes, np.uint8)
                _lib.check(lib.a2p_debug_read(model._ctx, name.encode(), a.ctypes.data_as(C.c_void_p), nbytes), name)
                bufs[name] = a
            snaps[(nw, mt)] = bufs
    ref = snaps[("4", "4")]
    for key, b in snaps.items():
        print(f"L={layers} NW={key[0]} MT={key[1]}:", {k: ("same" if np.array_equal(v, ref[k]) else "DIFF") for k, v in b.items()}, flush=True)
        if layers == 1 and key != ("4", "4"):
            for name, dt, cols in (("x", np.float32, 256), ("vt", np.uint16, 448), ("qk", np.uint16, 512)):
                a, r = b[name].view(dt).reshape(-1, cols), ref[name].view(dt).reshape(-1, cols)
                bad = np.argwhere(a != r)
                if len(bad):
                    rows, cs = np.unique(bad[:, 0]), np.unique(bad[:, 1])
                    print(f"     {name}: {len(bad)} differing elements, rows {rows[:12].tolist()}..{rows[-3:].tolist()} ({len(rows)} rows), cols {cs[:12].tolist()}..{cs[-3:].tolist()} ({len(cs)} cols)", flush=True)
