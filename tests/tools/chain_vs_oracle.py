"""The FULL 1000-step DDPM chain of the headline workload (face, T=600, S=1998+2, guidance scale 10, B=1) through the ORACLE, so that
the full-chain parity claim is product-vs-oracle and not GPU-16-bit-vs-GPU-fp32 ("Next round" item 8).  The oracle
needs ~20-40 CPU-minutes for the chain, so this is an offline tool with three sides that share every input:

  --side oracle   (build container, CPU)  oracle/a2p_oracle.py p_sample x 1000 -> profiles/r06_chain_oracle.npz
                  (the final sample and the state every 100 steps)
  --side gpu      (GPU box)               the product's p_sample_loop_progressive in fp32 / fp16 / bf16 under the same noise
                  -> gpurun_out/r06_chain_gpu.npz
  --side compare  (anywhere)              rel-L2 of every GPU mode against the oracle at every saved step -> profiles/r06_chain_vs_oracle.json

Weights, x_T and conditioning are the bench's (audio2photoreal_amd/synthetic.py, seed 10, sample id 0); the noise of step n is
torch.randn on the CPU from Generator(seed 70000 + n) on BOTH sides (the CPU generator is bit-reproducible across machines).
Follows /root/reference/diffusion/gaussian_diffusion.py:434-477 (p_sample) and :525-607 (p_sample_loop_progressive)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SAVE_EVERY = 100


def inputs(T):
    from audio2photoreal_amd.spec import face_spec
    from audio2photoreal_amd.synthetic import cond_tokens_for_frames, synthetic_state_dict, synthetic_tensor
    spec = face_spec()
    S0 = cond_tokens_for_frames(T)
    sd = synthetic_state_dict(spec, 10)
    cond = synthetic_tensor(10, "cond_embed/0", (S0, spec.cond_feature_dim))[None]
    x = synthetic_tensor(10, "x_T/0", (spec.nfeats, 1, T))[None]
    return spec, sd, cond, x


def noise_of(n, shape):
    return torch.randn(shape, generator=torch.Generator().manual_seed(70000 + n))


def side_oracle(a):
    from oracle import a2p_oracle as O
    torch.set_num_threads(a.threads)
    spec, sd, cond, x = inputs(a.T)
    den = O.OracleDenoiser(sd, "face", spec.num_layers, spec.num_heads)
    scale = torch.full((1,), 10.0)
    fn = lambda xx, ts: den.forward_cfg(xx, ts, cond, scale)
    smp = O.OracleSampler("")
    saved = {}
    cur = x
    t0 = time.time()
    with torch.no_grad():
        for n in range(a.steps):
            t = 999 - n
            cur = smp.p_sample(fn, cur, torch.tensor([t]), noise_of(n, x.shape))["sample"]
            if (n + 1) % SAVE_EVERY == 0 or n + 1 == a.steps:
                saved[f"step{n + 1}"] = cur.numpy().copy()
                np.savez(a.out or os.path.join(ROOT, "profiles", "r06_chain_oracle.npz"), seconds=time.time() - t0, threads=a.threads, **saved)
                print(f"oracle step {n + 1}: {time.time() - t0:.0f} s", flush=True)


def side_gpu(a):
    from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
    from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
    dev = torch.device("cuda:0")
    spec, sd, cond, x = inputs(a.T)
    shape = tuple(x.shape)
    out_all = {}
    for precision in ("fp32", "fp16", "bf16"):
        m, diff = create_model_and_diffusion(default_args("face", timestep_respacing=""), "test", precision=precision, max_batch=1)
        load_model(m, sd)
        m.auto_escalate = False          # each mode must run AS that mode (the bench shape stays inside the envelope anyway)
        cfg = ClassifierFreeSampleModel(m.to(dev).eval())
        y = {"cond_embed": cond.to(dev), "scale": torch.full((1,), 10.0, device=dev)}
        t0 = time.time()
        with torch.no_grad():
            gen = diff.p_sample_loop_progressive(cfg, shape, noise=x.to(dev), clip_denoised=False, model_kwargs={"y": y},
                                                 step_noise=lambda n: noise_of(n, shape).to(dev))
            for n, out in enumerate(gen):
                if (n + 1) % SAVE_EVERY == 0 or n + 1 == a.steps:
                    out_all[f"{precision}_step{n + 1}"] = out["sample"].float().cpu().numpy().copy()
                if n + 1 >= a.steps:
                    break
        torch.cuda.synchronize()
        out_all[f"{precision}_seconds"] = time.time() - t0
        out_all[f"{precision}_ran_as"] = str(getattr(m, "precision", precision))
        print(f"gpu {precision}: {time.time() - t0:.1f} s (ran as {out_all[f'{precision}_ran_as']})", flush=True)
        m.release()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(a.out or os.path.join(ROOT, "gpurun_out", "r06_chain_gpu.npz"), **out_all)


def side_compare(a):
    o = np.load(a.oracle or os.path.join(ROOT, "profiles", "r06_chain_oracle.npz"))
    g = np.load(a.gpu or os.path.join(ROOT, "gpurun_out", "r06_chain_gpu.npz"))
    rec = {"what": "face B=1 T=600 S=2000 guidance 10, 1000-step DDPM chain (p_sample), product (HIP, through the C ABI) vs oracle/a2p_oracle.py under "
                   "identical x_T / conditioning / per-step noise; rel-L2 and max-norm error of the chain state after n steps",
           "oracle_seconds": float(o["seconds"]), "oracle_threads": int(o["threads"]), "modes": {}}
    for precision in ("fp32", "fp16", "bf16"):
        rows = {}
        for k in sorted((k for k in o.files if k.startswith("step")), key=lambda s: int(s[4:])):
            if f"{precision}_{k}" not in g.files:
                continue
            want, got = o[k].astype(np.float64), g[f"{precision}_{k}"].astype(np.float64)
            rows[k] = {"rel_l2": float(f"{np.linalg.norm(got - want) / np.linalg.norm(want):.3e}"),
                       "max_norm": float(f"{np.abs(got - want).max() / np.abs(want).max():.3e}")}
        rec["modes"][precision] = {"ran_as": str(g[f"{precision}_ran_as"]), "gpu_seconds": round(float(g[f"{precision}_seconds"]), 2), "steps": rows}
    out = a.out or os.path.join(ROOT, "profiles", "r06_chain_vs_oracle.json")
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", required=True, choices=["oracle", "gpu", "compare"])
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--T", type=int, default=600)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--out", default="")
    ap.add_argument("--oracle", default="")
    ap.add_argument("--gpu", default="")
    a = ap.parse_args()
    {"oracle": side_oracle, "gpu": side_gpu, "compare": side_compare}[a.side](a)
