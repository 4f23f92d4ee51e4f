"""Generate the golden fixtures in tests/golden/ by running the REFERENCE ITSELF
(/root/reference, imported read-only, CPU fp32) on deterministic synthetic
weights/inputs (audio2photoreal_amd.synthetic).  Runs only in the build container;
the committed .npz files travel to the GPU box, this script's inputs do not.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_import as ri  # noqa: E402
from audio2photoreal_amd.spec import face_spec, pose_spec  # noqa: E402
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict, synthetic_tensor  # noqa: E402

SEED = 10  # reference default seed (utils/diff_parser_utils.py:82)


def load_synth(model, spec):
    sd = synthetic_state_dict(spec, SEED)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    bad = [k for k in missing if not (k.startswith("audio_model.") or k.startswith("lip_model.")
                                      or k.endswith("rotary.freqs"))]
    assert not bad, bad
    return sd


def y_dict(model, spec, inp, B, frames, scale):
    model._a2p_cond_embed = inp["cond_embed"]
    y = {"audio": torch.zeros(B, 1, 2), "scale": torch.full((B,), scale)}
    if spec.is_pose:
        y["keyframes"] = inp["keyframes"].clone()
        y["mask"] = inp["mask"].clone()
    return y


def main():
    torch.manual_seed(SEED)
    torch.set_num_threads(8)
    ns = ri.import_reference()
    out = {}

    # ---- schedule tables (a1-a4) ------------------------------------------------
    for name, resp in (("full", ""), ("ddim10", "ddim10"), ("ddim100", "ddim100"), ("ddim500", "ddim500")):
        args = ri.ref_args("face", 8, 8, resp)
        diff = ns.mu.create_gaussian_diffusion(args)
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                  "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                  "posterior_mean_coef1", "posterior_mean_coef2", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod"):
            out[f"sched/{name}/{k}"] = getattr(diff, k)
        out[f"sched/{name}/timestep_map"] = np.array(diff.timestep_map, dtype=np.int64)
    out["sched/space/ddim50"] = np.array(sorted(ns.rs.space_timesteps(1000, "ddim50")), dtype=np.int64)
    out["sched/space/10,15,20"] = np.array(sorted(ns.rs.space_timesteps(300, "10,15,20")), dtype=np.int64)

    with ri.cpu_cuda(), torch.no_grad():
        for fmt in ("face", "pose"):
            spec = face_spec() if fmt == "face" else pose_spec()
            model, diff10 = ri.build_reference_model(ns, fmt, spec.num_layers, spec.num_heads, "ddim10")
            sd = load_synth(model, spec)
            cfg_model = ns.cfg.ClassifierFreeSampleModel(model)
            scale = 10.0 if fmt == "face" else 2.0

            # ---- one decoder layer (a18-a20), small T/S, real widths ----------------
            d = spec.latent_dim
            lay = model.seqTransDecoder.stack[0]
            lx = synthetic_tensor(SEED, "layer_x", (2, 48, d))
            lmem = synthetic_tensor(SEED, "layer_mem", (2, 80, d))
            lt = synthetic_tensor(SEED, "layer_t", (2, d))
            lmem2 = synthetic_tensor(SEED, "layer_mem2", (2, 8, d)) if spec.is_pose else None
            out[f"{fmt}/layer0"] = lay(lx, lmem, lt, memory2=lmem2).numpy()

            # ---- single forwards (a14-a17, a22-a24): T=240, B=2, mixed timesteps ----
            B, frames = 2, 240
            inp = synthetic_inputs(spec, B, frames, SEED)
            if spec.is_pose:
                inp["mask"][1, :, :, 90:] = False   # exercise the masked-keyframe path
            times = torch.tensor([937, 12])
            y = y_dict(model, spec, inp, B, frames, scale)
            out[f"{fmt}/fwd_cond"] = model(inp["x_T"], times, y, cond_drop_prob=0.0).numpy()
            y = y_dict(model, spec, inp, B, frames, scale)
            out[f"{fmt}/fwd_uncond"] = model(inp["x_T"], times, y, cond_drop_prob=1.0).numpy()
            y = y_dict(model, spec, inp, B, frames, scale)
            out[f"{fmt}/fwd_cfg"] = cfg_model(inp["x_T"], times, y).numpy()

            # ---- ddim10 end to end (a5-a13): BASELINE config 0 shape for face --------
            B, frames = (1, 240) if fmt == "face" else (2, 600)
            inp = synthetic_inputs(spec, B, frames, SEED)
            y = y_dict(model, spec, inp, B, frames, scale)
            res = diff10.ddim_sample_loop(cfg_model, (B, spec.nfeats, 1, frames), clip_denoised=False,
                                          model_kwargs={"y": y}, noise=inp["x_T"].clone())
            out[f"{fmt}/ddim10"] = res.numpy()

            # ---- DDPM p_sample_loop with the restored noise, 10 respaced steps -------
            B, frames = 1, 240
            inp = synthetic_inputs(spec, B, frames, SEED, steps_of_noise=10)
            y = y_dict(model, spec, inp, B, frames, scale)
            diff10._a2p_step_noise = [inp["step_noise"][i].clone() for i in range(10)]
            res = diff10.p_sample_loop(cfg_model, (B, spec.nfeats, 1, frames), clip_denoised=False,
                                       model_kwargs={"y": y}, noise=inp["x_T"].clone())
            out[f"{fmt}/ddpm10"] = res.numpy()
            diff10._a2p_step_noise = None

            # ---- first 3 steps of the full 1000-step DDPM chain ----------------------
            model_f, diff1000 = ri.build_reference_model(ns, fmt, spec.num_layers, spec.num_heads, "")
            inp = synthetic_inputs(spec, B, frames, SEED, steps_of_noise=3)
            y = y_dict(model, spec, inp, B, frames, scale)
            diff1000._a2p_step_noise = [inp["step_noise"][i].clone() for i in range(3)]
            gen = diff1000.p_sample_loop_progressive(cfg_model, (B, spec.nfeats, 1, frames), clip_denoised=False,
                                                     model_kwargs={"y": y}, noise=inp["x_T"].clone())
            for i, o in zip(range(3), gen):
                last = o
            out[f"{fmt}/ddpm1000_first3"] = last["sample"].numpy()
            print(fmt, "done", flush=True)

    np.savez(os.path.join(HERE, "golden_v1.npz"), **out)
    tot = sum(v.nbytes for v in out.values())
    print("wrote", len(out), "arrays,", tot / 1e6, "MB")


if __name__ == "__main__":
    main()
