"""Open item: do the 4- and 8-wave chain shapes agree bit-for-bit on pose B=3, T=450 (ragged last panel, T % 4 != 0)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
from audio2photoreal_amd.spec import pose_spec
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict
dev = torch.device("cuda:0"); spec = pose_spec()
model, _ = create_model_and_diffusion(default_args("pose"), "test", precision="bf16", max_batch=4)
load_model(model, synthetic_state_dict(spec, 10)); cfg = ClassifierFreeSampleModel(model.to(dev).eval())
for B, T in ((3, 450), (3, 448), (4, 450), (3, 452)):
    inp = synthetic_inputs(spec, B, T, 10)
    y = {"cond_embed": inp["cond_embed"].to(dev), "keyframes": inp["keyframes"].to(dev), "mask": inp["mask"].to(dev), "scale": torch.full((B,), 2.0, device=dev)}
    x, t = inp["x_T"].to(dev), torch.tensor([901, 417, 33, 0][:B], device=dev)
    o = {}
    for nw in ("4", "8", "4"):
        os.environ["A2P_CHAIN_NW"] = nw
        o.setdefault(nw, []).append(cfg(x, t, y).clone())
    d48 = (o["4"][0] - o["8"][0]).abs().max().item(); d44 = (o["4"][0] - o["4"][1]).abs().max().item()
    print(f"B={B} T={T}: |nw4-nw8|max={d48:.3e} |nw4-nw4'|max={d44:.3e} nan={bool(torch.isnan(o['4'][0]).any())}", flush=True)
