#!/usr/bin/env python
"""bench.py -- denoise steps/sec of the MI355X-native sampling hot path.

Workload (BASELINE.json configs[1]): face diffusion, 1000-step DDPM (p_sample_loop with the
restored noise), classifier-free guidance (2 denoiser passes per step), batch 8 samples per GPU,
600-frame sequences, 1998 audio tokens (+2 time tokens), bf16 operands / fp32 accumulate.
Synthetic weights + inputs (no checkpoints/datasets offline).  One "step" = one p_sample:
2 x FiLMTransformer forward over 8 samples + guidance + posterior update.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--precision bf16|fp32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU = sample parallel: every rank denoises its own 8 samples, no per-step
communication (weak scaling); one RCCL all_gather of the final samples after the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs (2*MACs) of the decoder stack per forward per sample at T=600, S=2000 with the
# audio-token K/V hoisted: face 50.77 GF, i.e. 101.5 GF per denoise step per sample.
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3     # fp32 MFMA
PEAK_HBM_GBS = 8000.0


def algorithmic_flops(spec, T, S, nseq):
    """Per forward of `nseq` sequences, by kernel class (attention + FFN + projections of the decoder
    stack; hoisted audio-token K/V projections excluded)."""
    d, ff, L = spec.latent_dim, spec.ff_size, spec.num_layers
    sa_attn = 4.0 * T * T * d                      # QK^T + PV
    ca_attn = 4.0 * T * S * d
    gemm = 2.0 * T * d * (3 * d + d) + 2.0 * T * d * (d + d) + 4.0 * T * d * ff   # SA qkv+o, CA q+o, FFN
    if spec.is_pose:
        ca_attn2 = 4.0 * T * 20 * d
        gemm += 2.0 * T * d * (d + d)
    else:
        ca_attn2 = 0.0
    io = 2.0 * T * d * spec.nfeats * 1.5      # input_projection (once per sample) + final_layer (per sequence)
    return {"decoder_gemm": nseq * L * gemm, "io_gemm": nseq * io, "attn_self": nseq * L * sa_attn,
            "attn_cross": nseq * L * (ca_attn + ca_attn2)}


def run_pipeline(a, dev):
    """One subject of BASELINE configs[4] on one GPU: everything after the (out-of-scope) wav2vec front end."""
    from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
    from audio2photoreal_amd.model.guide import GuideTransformer
    from audio2photoreal_amd.model.vqvae import TemporalVertexCodec
    from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
    from audio2photoreal_amd.sample.generate import _replace_keyframes
    from audio2photoreal_amd.spec import GuideSpec, TokenizerSpec, face_spec, pose_spec
    from audio2photoreal_amd.synthetic import (cond_tokens_for_frames, synthetic_guide_state_dict, synthetic_state_dict, synthetic_tensor,
                                               synthetic_tokenizer_state_dict)
    B, T = a.batch, a.frames
    S0, gs, ts = cond_tokens_for_frames(T), GuideSpec(), TokenizerSpec()
    guide = GuideTransformer(tokens=gs.tokens, num_layers=gs.num_layers, dim=gs.dim, emb_len=gs.emb_len,
                             num_audio_layers=gs.num_audio_layers, max_batch=B, max_positions=96)
    guide.load_state_dict(synthetic_guide_state_dict(gs, 10), strict=False)
    tok = TemporalVertexCodec(ts.n_vertices, ts.latent_dim, ts.categories, ts.residual_depth)
    tok.load_state_dict(synthetic_tokenizer_state_dict(ts, 10), strict=False)
    models = {}
    for fmt, spec in (("pose", pose_spec()), ("face", face_spec())):
        m, d = create_model_and_diffusion(default_args(fmt, timestep_respacing="ddim100"), "test", precision=a.precision, max_batch=B)
        load_model(m, synthetic_state_dict(spec, 10))
        if fmt == "pose":
            m.setup_guide_predictor(guide.to(dev).eval(), tok.to(dev))
        models[fmt] = (spec, ClassifierFreeSampleModel(m.to(dev).eval()), d)
    nk = len(range(T)[::30])
    feats = synthetic_tensor(10, "pipeline_audio_feats", (B, S0, 1024)).to(dev)
    lip = synthetic_tensor(10, "pipeline_lip_feats", (B, S0, 1014)).to(dev)

    def once():
        guide._prepared_for = None                     # every run pays the hoisted conditioning of all three models
        for _, cfg_m, _ in models.values():
            cfg_m.model._cond_key = None
        torch.cuda.synchronize()
        st, t0 = {}, time.perf_counter()

        def mark(name):
            nonlocal t0
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            st[name], t0 = (t1 - t0) * 1e3, t1
        spec, cfg, diff = models["pose"]
        y = {"cond_embed": feats, "keyframes": torch.zeros(B, nk, 104, device=dev), "mask": torch.ones(B, 1, 1, T, dtype=torch.bool, device=dev),
             "scale": torch.full((B,), 2.0, device=dev)}
        y["keyframes"] = _replace_keyframes({"y": y}, cfg).to(dev)
        mark("guide_tokens_and_vq_decode_ms")
        body = diff.ddim_sample_loop(cfg, (B, spec.nfeats, 1, T), clip_denoised=False, model_kwargs={"y": y})
        mark("body_ddim100_ms")
        spec, cfg, diff = models["face"]
        yf = {"cond_embed": torch.cat([feats, lip], -1), "scale": torch.full((B,), 10.0, device=dev)}
        face = diff.ddim_sample_loop(cfg, (B, spec.nfeats, 1, T), clip_denoised=False, model_kwargs={"y": yf})
        mark("face_ddim100_ms")
        assert bool(torch.isfinite(body).all()) and bool(torch.isfinite(face).all())
        return st
    once()                                    # contexts, weight upload, allocator warm-up
    st = once()
    total = sum(st.values()) / 1e3
    print(json.dumps({"metric": "end-to-end sec/sample: guide transformer -> body ddim100 -> face ddim100, 600 frames, from audio features "
                                "(BASELINE configs[4] shape, one subject, one GPU; the wav2vec front end is out of scope)",
                      "value": round(total / B, 5), "unit": "s/sample", "higher_is_better": False, "n_gpus": 1, "batch": B,
                      "dtype": a.precision, "data": "synthetic", "total_s": round(total, 4),
                      "stages_ms": {k: round(v, 2) for k, v in st.items()}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--model", default="face", choices=["face", "pose"],
                    help="face = BASELINE configs[1] (the metric's config, default); pose = configs[2] shape (body model, keyframes, scale 2)")
    ap.add_argument("--pipeline", action="store_true",
                    help="instead of the step benchmark: BASELINE configs[4] shape on this GPU for one subject -- audio features -> "
                         "guide transformer tokens -> VQ keyframes -> body ddim100 -> face ddim100 (demo/demo.py:156-216), sec/sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the product has no CPU path)"
    # test hooks (single-GPU boxes): A2P_BENCH_SHARE_GPU=1 maps every rank to cuda:0 and A2P_BENCH_BACKEND=gloo carries the
    # three collectives (barrier, max-reduce of the time, final gather) over host memory, so the N>1 control flow can be
    # exercised without N GPUs.  Production: one rank per GPU, backend "nccl" (= RCCL over xGMI on ROCm).
    share = bool(os.environ.get("A2P_BENCH_SHARE_GPU"))
    backend = os.environ.get("A2P_BENCH_BACKEND", "nccl")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from audio2photoreal_amd import _lib
    from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
    from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
    from audio2photoreal_amd.spec import face_spec, pose_spec
    from audio2photoreal_amd.synthetic import cond_tokens_for_frames, synthetic_state_dict, synthetic_tensor

    if a.pipeline:
        return run_pipeline(a, dev)
    spec = face_spec() if a.model == "face" else pose_spec()
    B, T = a.batch, a.frames
    S0 = cond_tokens_for_frames(T)
    sd = synthetic_state_dict(spec, 10)
    model, diffusion = create_model_and_diffusion(default_args(a.model, timestep_respacing=""), "test",
                                                  precision=a.precision, max_batch=B)
    load_model(model, sd)
    model = model.to(dev).eval()
    cfg = ClassifierFreeSampleModel(model)

    # per-rank inputs indexed by GLOBAL sample id so results do not depend on the world size
    g0 = rank * B
    cond = torch.stack([synthetic_tensor(10, f"cond_embed/{g0 + i}", (S0, spec.cond_feature_dim)) for i in range(B)]).to(dev)
    x = torch.stack([synthetic_tensor(10, f"x_T/{g0 + i}", (spec.nfeats, 1, T)) for i in range(B)]).to(dev)
    y = {"cond_embed": cond, "scale": torch.full((B,), 10.0 if a.model == "face" else 2.0, device=dev)}
    if spec.is_pose:
        nk = len(range(T)[:: spec.keyframe_step])
        y["keyframes"] = torch.stack([synthetic_tensor(10, f"keyframes/{g0 + i}", (nk, spec.keyframe_dim)) for i in range(B)]).to(dev)
        y["mask"] = torch.ones(B, 1, 1, T, dtype=torch.bool, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)

    model.prepare(x, y)                       # context + weight upload + first conditioning pass (untimed setup)
    torch.cuda.synchronize()
    model._cond_key = None
    t0 = time.perf_counter()
    model.prepare(x, y)                       # hoisted conditioning: once per sample, outside the loop
    torch.cuda.synchronize()
    prepare_s = time.perf_counter() - t0

    n_chain = diffusion.num_timesteps
    steps_idx = diffusion._step_index_tensor(dev, B)
    state = {"x": x, "i": n_chain - 1}

    def run_steps(n):
        for _ in range(n):
            noise = torch.randn(x.shape, device=dev, generator=gen)        # randn_like(x) of p_sample
            out = diffusion.p_sample(cfg, state["x"], steps_idx[state["i"]], clip_denoised=False,
                                     model_kwargs={"y": y}, noise=noise)
            state["x"] = out["sample"]
            state["i"] = state["i"] - 1 if state["i"] > 0 else n_chain - 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # one-off setup like the weight upload above: the library measures which chain-kernel workgroup shape is faster on THIS
        # box during the first 6 forwards of a given size (csrc/a2p_lib_run.h chain_pick_nw; both shapes give identical bits)
        for _ in range(6):
            cfg(x, steps_idx[0], y)
        torch.cuda.synchronize()
        run_steps(a.warmup)
        barrier()
        t0 = time.perf_counter()
        run_steps(a.steps)
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(state["x"]).all(), "non-finite samples"

    # ---- single end-of-run gather of the samples over RCCL/xGMI (outside the timed region) ----
    gather_ms = None
    if world > 1:
        mine = state["x"].contiguous().to(coll_dev)
        outs = [torch.empty_like(mine) for _ in range(world)]
        barrier()
        t0 = time.perf_counter()
        dist.all_gather(outs, mine)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t0) * 1e3

    # ---- per-kernel time of the dominant kernel classes, HIP events on the launch stream ----
    kernels, roofline = {}, None
    if rank == 0 and not a.no_kernel_timing:
        lib = _lib.load()
        import ctypes as C
        flops = algorithmic_flops(spec, T, S0 + 2, 2 * B)
        ksteps = min(a.steps, 5)
        # bf16 mode runs the decoder-layer GEMMs inside the fused "chain" kernels (projections + FiLM + LayerNorm + FFN);
        # fp32 mode (and A2P_NO_CHAIN=1) runs them as separate GEMM launches
        chained = a.precision == "bf16" and not os.environ.get("A2P_NO_CHAIN")
        flops["chain" if chained else "gemm"] = flops.pop("decoder_gemm") + (0.0 if chained else flops["io_gemm"])
        if chained:
            flops["gemm"] = flops["io_gemm"]
        for name, kind in (("chain", _lib.KERNEL_CHAIN), ("gemm", _lib.KERNEL_GEMM), ("attn_self", _lib.KERNEL_ATTN_SELF),
                           ("attn_cross", _lib.KERNEL_ATTN_CROSS), ("ln_rope", _lib.KERNEL_LNROPE)):
            _lib.check(lib.a2p_kernel_timing(model._ctx, kind, 1), "a2p_kernel_timing")
            with torch.no_grad():
                run_steps(ksteps)
            ms, n = C.c_double(), C.c_int64()
            _lib.check(lib.a2p_kernel_time_ms(model._ctx, C.byref(ms), C.byref(n)), "a2p_kernel_time_ms")
            _lib.check(lib.a2p_kernel_timing(model._ctx, kind, 0), "a2p_kernel_timing")
            if n.value == 0:
                continue
            per_step_ms = ms.value / ksteps
            ent = {"ms_per_step": round(per_step_ms, 4), "launches_per_step": n.value // ksteps,
                   "avg_launch_us": round(1e3 * ms.value / max(n.value, 1), 2)}
            if name in flops:
                ent["algorithmic_gflop_per_step"] = round(flops[name] / 1e9, 2)
                ent["tflops"] = round(flops[name] / (per_step_ms * 1e-3) / 1e12, 2)
            kernels[name] = ent
        dom = max((k for k in kernels if k in flops), key=lambda k: kernels[k]["ms_per_step"])
        peak = PEAK_BF16_TFLOPS if a.precision == "bf16" else PEAK_F32_TFLOPS
        # HBM bytes per launch of that kernel class from the rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 note
        # of MI355X_MICROARCH.md + WRITE_SIZE; scratch/run_pmc.sh writes the file) -- null when not collected
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and a.model == "face" and B == 8 and T == 600 and a.precision == "bf16":   # the profiled workload
            traffic = json.load(open(tpath)).get(dom)
        roofline = {"kernel": dom, "bound": "mfma", "achieved": kernels[dom]["tflops"], "peak": peak, "unit": "TFLOP/s",
                    "frac": round(kernels[dom]["tflops"] / peak, 4), "traffic": traffic and traffic["total_bytes"],
                    "traffic_detail": traffic,
                    "avg_launch_us": kernels[dom]["avg_launch_us"],
                    "algorithmic_gflop_per_launch": round(flops[dom] / 1e9 / kernels[dom]["launches_per_step"], 3)}

    # ---- CPU baseline: the oracle (port of the reference algorithm) on this box's host cores ----
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.model == "face":   # reported at N=1 only (other ranks would idle in the final barrier)
        from oracle import a2p_oracle as O
        cores = min(os.cpu_count() or 1, 32)   # torch CPU matmuls at these sizes stop scaling (and thrash) past ~32 threads
        torch.set_num_threads(cores)
        cb = 1   # bounded sample: 1 sample (of the 8), 2 DDPM steps, same T/S, fp32
        den = O.OracleDenoiser(sd, "face", spec.num_layers, spec.num_heads)
        ce, xc = cond[:cb].cpu(), x[:cb].cpu()
        fn = lambda xx, ts: den.forward_cfg(xx, ts, ce, torch.full((cb,), 10.0))
        smp = O.OracleSampler("")
        CPU_STEPS = 8   # ~10-20 s on the box's host cores
        nz = [torch.randn(xc.shape) for _ in range(CPU_STEPS + 1)]
        with torch.no_grad():
            smp.p_sample_loop(fn, xc, nz, max_steps=1)      # warm-up
            t0 = time.perf_counter()
            smp.p_sample_loop(fn, xc, nz, max_steps=CPU_STEPS)
            cdt = time.perf_counter() - t0
        sample_steps_per_s = cb * CPU_STEPS / cdt
        cpu = {"value": round(sample_steps_per_s / B, 5), "unit": f"denoise steps/sec at batch {B} (scaled from sample-steps/sec)",
               "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"oracle (torch CPU fp32 restatement, conditioning path recomputed every forward like the reference's "
                         f"decoder-only path), {cb} sample x {CPU_STEPS} DDPM steps, T={T}, S={S0 + 2}: {cdt:.2f} s"}

    if rank == 0:
        value = world * a.steps / dt
        fl = algorithmic_flops(spec, T, S0 + 2, 2 * B)
        step_flops = fl["decoder_gemm"] + fl["attn_self"] + fl["attn_cross"]   # decoder attention + FFN + projections
        line = {
            "metric": f"diffusion denoise steps/sec ({a.model}, {T}-frame seq, batch {B} per GPU, CFG)", "value": round(value, 4),
            "unit": "steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"{a.model} FiLM denoiser {spec.num_layers}L/{spec.num_heads}H d{spec.latent_dim}, 1000-step DDPM p_sample chain, B={B}/GPU x2 CFG, "
                                   f"T={T}, {S0}+2 cond tokens", "global_batch": B * world, "parallelism": f"sample-parallel x{world}"},
            "sample_steps_per_sec": round(value * B, 3),
            "decoder_tflops": round(world * step_flops * a.steps / dt / 1e12, 2),
            "decoder_mfma_frac": round(step_flops * a.steps / dt / 1e12 / (PEAK_BF16_TFLOPS if a.precision == "bf16" else PEAK_F32_TFLOPS), 4),
            "prepare_s": round(prepare_s, 3), "gather_ms": gather_ms,
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()   # rank 0 may still be in its post-run measurement legs
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
