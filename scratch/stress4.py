"""Who is the victim of the two-stream nondeterminism: the time-path outputs (film / ktail / vtail) or the main path?"""
import os, sys, torch, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio2photoreal_amd import _lib
from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
from audio2photoreal_amd.spec import face_spec
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict
dev = torch.device("cuda:0")
spec = face_spec()
model, _ = create_model_and_diffusion(default_args("face"), "test", precision="bf16", max_batch=8)
load_model(model, synthetic_state_dict(spec, 10))
cfg = ClassifierFreeSampleModel(model.to(dev).eval())
inp = synthetic_inputs(spec, 8, 600, 10)
y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((8,), 10.0, device=dev)}
x = inp["x_T"].to(dev)
t = torch.tensor([999, 750, 500, 250, 100, 10, 1, 0], device=dev)
os.environ["A2P_SIDE_STREAM"] = "1"
lib = _lib.load()
sizes = {"film": 16 * 8 * 3 * 1024 * 4, "ktail": 16 * 8 * 512 * 4, "vtail": 16 * 8 * 512 * 4, "tvec": 16 * 512 * 4, "tokr": 16 * 512 * 4, "tokn": 16 * 512 * 4, "tct": 8 * 3 * 512 * 4}
def snap():
    out = {}
    for k, n in sizes.items():
        a = np.empty(n // 4, np.float32)
        _lib.check(lib.a2p_debug_read(model._ctx, k.encode(), a.ctypes.data_as(C.c_void_p), n), k)
        out[k] = a
    return out
ref_out = cfg(x, t, y).clone(); ref_tp = snap()
nbad = 0
for i in range(int(os.environ.get("ITERS", "300"))):
    out = cfg(x, t, y); tp = snap()
    same_out = torch.equal(out, ref_out)
    diff_tp = [k for k in sizes if not np.array_equal(tp[k], ref_tp[k])]
    if not same_out or diff_tp:
        nbad += 1
        if nbad <= 6:
            rows = {k: (sorted(set((np.nonzero(tp[k] != ref_tp[k])[0] // (tp[k].size // (8 if k == "tct" else 16))).tolist())), int((tp[k] != ref_tp[k]).sum()),
                        float(np.abs(tp[k] - ref_tp[k]).max())) for k in diff_tp}
            if "tokr" in diff_tp:
                idx = np.nonzero(tp["tokr"] != ref_tp["tokr"])[0]
                os.makedirs("gpurun_out", exist_ok=True)
                np.savez(f"gpurun_out/tokr_bad_{nbad}.npz", got=tp["tokr"], ref=ref_tp["tokr"], tokn=tp["tokn"], idx=idx)
                print("  cols", (idx % 512).tolist())
                print("  got ", np.round(tp["tokr"][idx], 4).tolist())
                print("  ref ", np.round(ref_tp["tokr"][idx], 4).tolist())
                print("  tokn", np.round(tp["tokn"][idx], 4).tolist())
            print(f"iter {i}: output {'same' if same_out else 'DIFFERENT'}; time-path buffers that differ: {rows}", flush=True)
print(f"{nbad} anomalous forwards", flush=True)
