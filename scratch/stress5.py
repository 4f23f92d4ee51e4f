"""Round 6, session 4: determinism of the barrier-free feed-forward block (LDS counters), the fused final_layer and the CHAIN_IN kernel under load -- the same guided forward
ITERS times (after the family calibration), every output compared with the first BIT FOR BIT; a second process-level stream runs a copy loop to perturb timing (STRESS_BG=1).
STRESS_BATCH = samples (8: 48-row kernels, 32: 80-row kernels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
from audio2photoreal_amd.spec import face_spec
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict
dev = torch.device("cuda:0")
B = int(os.environ.get("STRESS_BATCH", "8"))
os.environ["A2P_CHAIN_V"] = "4"
spec = face_spec()
model, _ = create_model_and_diffusion(default_args("face"), "test", precision=os.environ.get("STRESS_PRECISION", "fp16"), max_batch=B)
load_model(model, synthetic_state_dict(spec, 10))
cfg = ClassifierFreeSampleModel(model.to(dev).eval())
inp = synthetic_inputs(spec, B, 600, 10)
y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0, device=dev)}
x = inp["x_T"].to(dev)
t = torch.tensor(([999, 750, 500, 250, 100, 10, 1, 0] * 4)[:B], device=dev)
bg = torch.cuda.Stream(device=dev) if os.environ.get("STRESS_BG") else None
junk = torch.empty(64 << 20, device=dev, dtype=torch.uint8) if bg else None
ref = cfg(x, t, y).clone()
nbad = 0
for i in range(int(os.environ.get("ITERS", "300"))):
    if bg is not None:
        with torch.cuda.stream(bg):
            junk[: 32 << 20].copy_(junk[32 << 20:], non_blocking=True)
    out = cfg(x, t, y)
    if not torch.equal(out, ref):
        nbad += 1
        if nbad <= 5:
            d = (out - ref).abs()
            print(f"iter {i}: DIFFERENT, {int((d > 0).sum())} elements, max {float(d.max()):.3e}, samples {sorted(set((d.flatten(1).amax(1) > 0).nonzero().flatten().tolist()))}", flush=True)
model.check_finite()
print(f"B={B} {os.environ.get('STRESS_PRECISION', 'fp16')} bg={'on' if bg else 'off'}: {nbad} of {int(os.environ.get('ITERS', '300'))} forwards differ from the first", flush=True)
