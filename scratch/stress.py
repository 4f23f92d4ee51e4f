"""Race hunt: repeated guided forwards at the bench size, count bitwise mismatches per configuration (env is read per call)."""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name, env in (("default", {}), ("no side stream", {"A2P_NO_SIDE_STREAM": "1"}), ("no chain", {"A2P_NO_CHAIN": "1"}),
                  ("chain MT=2", {"A2P_CHAIN_MT": "2"}), ("default again", {})):
    for k in ("A2P_NO_SIDE_STREAM", "A2P_NO_CHAIN", "A2P_CHAIN_MT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ref = cfg(x, t, y).clone()
    bad, worst = 0, 0.0
    for i in range(N):
        out = cfg(x, t, y)
        if not torch.equal(out, ref):
            bad += 1
            worst = max(worst, float((out - ref).abs().max()))
    torch.cuda.synchronize()
    print(f"{name:16s}: {bad}/{N} mismatching forwards, worst |diff| {worst:.3e}", flush=True)
