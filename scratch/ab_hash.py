"""Output digest of a guided bf16 forward (face B=8 T=600, pose B=16 T=600) for A/B library builds: A2P_LIB=<so> python scratch/ab_hash.py"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
from audio2photoreal_amd.spec import face_spec, pose_spec
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict
dev = torch.device("cuda:0")
for fmt, B in (("face", 8), ("pose", 16), ("face", 32)):
    spec = face_spec() if fmt == "face" else pose_spec()
    model, _ = create_model_and_diffusion(default_args(fmt), "test", precision="bf16", max_batch=B)
    load_model(model, synthetic_state_dict(spec, 10))
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    inp = synthetic_inputs(spec, B, 600, 10)
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0 if fmt == "face" else 2.0, device=dev)}
    if spec.is_pose:
        y["keyframes"], y["mask"] = inp["keyframes"].to(dev), inp["mask"].to(dev)
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(3)).to(dev)
    out = cfg(inp["x_T"].to(dev), t, y)
    torch.cuda.synchronize()
    print(fmt, B, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], float(out.abs().mean()), flush=True)
    del model, cfg
