"""Locate the mismatching elements of a non-reproducible forward (chain path, MT=2 is the most failure-prone)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for env in ({"A2P_SIDE_STREAM": "1"}, {"A2P_SIDE_STREAM": "1", "A2P_SIDE_JOIN": "1"}, {"A2P_SIDE_STREAM": "1", "A2P_SIDE_JOIN": "2"}, {"A2P_SIDE_STREAM": "1"}, {"A2P_SIDE_STREAM": "1", "A2P_SIDE_JOIN": "1"}, {"A2P_SIDE_STREAM": "1", "A2P_SIDE_JOIN": "2"}):
    for k in ("A2P_NO_SIDE_STREAM", "A2P_CHAIN_MT", "A2P_NO_CHAIN", "A2P_SIDE_EARLY_JOIN", "A2P_SIDE_STREAM", "A2P_SIDE_JOIN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ref = cfg(x, t, y).clone()
    nbad = 0
    for i in range(100):
        out = cfg(x, t, y)
        if not torch.equal(out, ref):
            nbad += 1
            if nbad <= 3:
                d = (out != ref)                       # [B, T, C]
                bs = d.any(dim=2).any(dim=1).nonzero().flatten().tolist()
                for b in bs[:2]:
                    ts = d[b].any(dim=1).nonzero().flatten()
                    cs = d[b].any(dim=0).nonzero().flatten()
                    print(f"  {env} iter {i}: sample {b}: frames {int(ts.min())}..{int(ts.max())} ({len(ts)} rows), channels {int(cs.min())}..{int(cs.max())} ({len(cs)}), "
                          f"max|diff| {float((out[b]-ref[b]).abs().max()):.3e}", flush=True)
                print(f"  samples affected: {bs}", flush=True)
    print(f"{env}: {nbad}/100 mismatching", flush=True)
