#!/usr/bin/env python
"""bench.py -- denoise steps/sec of the MI355X-native sampling hot path.

Headline workload (BASELINE.json configs[1]): face diffusion, 1000-step DDPM (p_sample_loop with the
restored noise), classifier-free guidance (2 denoiser passes per step), batch 8 samples per GPU,
600-frame sequences, 1998 audio tokens (+2 time tokens), 16-bit operands (IEEE half by default, bf16 as a leg) / fp32 accumulate.
Synthetic weights + inputs (no checkpoints/datasets offline).  One "step" = one p_sample:
2 x FiLMTransformer forward over 8 samples + guidance + posterior update.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--precision fp16|bf16|fp32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Defaults: N=1, K=200 timed steps per region behind W=20 warm-up steps, 3 regions (the median is reported); the whole default run,
legs, parity and CPU baseline included, takes ~35 s on the GPU box.

The ONE JSON line rank 0 prints carries, next to the contract fields:
  roofline      dominant kernel class, algorithmic FLOPs / measured launch time (HIP events on the launch stream) vs the bf16 peak
  cpu_baseline  the oracle (CPU port of the reference algorithm) timed on this box's host cores, bounded sample
  parity        all three modes (fp16 = benchmarked, bf16, fp32 = parity mode) against that same oracle run at the bench shape
                (face, T=600, S=2000), plus the drift of the full 1000-step chain of both 16-bit modes vs GPU-fp32 under identical noise
  legs          the same workload with bf16 operands ("bf16") and in the exact-fp32 parity mode ("fp32"), and the other north-star
                shapes on this GPU: face B=32 ("b32"), the body model B=16 with keyframes ("body") and the reference's CPU-runnable
                config 0 shape ("cfg0")

Multi-GPU = sample parallel: rank r denoises the global samples shard_bounds(N*B, N, r) with its own
replica, no per-step communication (weak scaling); one all_gather (sample_parallel.gather_samples; RCCL over xGMI) of the
final samples after the timed region.  `value` = denoise steps all ranks ran / max-over-ranks time: at N GPUs one "step" of
the job advances N*B samples (`sample_steps_per_sec` = value * B is the size-independent figure).
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3     # fp32 MFMA
PEAK_HBM_GBS = 8000.0


def algorithmic_flops(spec, T, S, nseq):
    """Per forward of `nseq` sequences, by kernel class (attention + FFN + projections of the decoder
    stack; hoisted audio-token K/V projections excluded; SURVEY.md §8d: face 50.77 GF per sequence at T=600, S=2000)."""
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


def box_record(dev):
    """What distinguishes one leased MI355X from another (the two-group box spread): device properties the HIP
    runtime reports, plus rocm-smi's clocks / power cap / partition modes when the tool is there.  Every bench line carries it, so the
    boxes of a round can be correlated with their numbers afterwards (profiles/r05_boxes.md)."""
    import subprocess
    pr = torch.cuda.get_device_properties(dev)
    rec = {"name": pr.name, "gcn_arch": getattr(pr, "gcnArchName", None), "cus": pr.multi_processor_count,
           "hbm_gb": round(pr.total_memory / 2 ** 30, 1), "l2_mb": round(getattr(pr, "L2_cache_size", 0) / 2 ** 20, 1),
           "clock_mhz": getattr(pr, "clock_rate", 0) // 1000 or None, "mem_clock_mhz": getattr(pr, "memory_clock_rate", 0) // 1000 or None}
    try:   # what the memory system of THIS box delivers: 1 GiB device-to-device copies (read + write), best of 5
        n = 1 << 28
        a, b = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
        a.fill_(1.0)
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            b.copy_(a)
            torch.cuda.synchronize(dev)
            best = min(best, time.perf_counter() - t0)
        rec["hbm_copy_gbs"] = round(2 * 4 * n / best / 1e9, 1)
        del a, b
    except Exception as e:   # noqa: BLE001 -- a report field
        rec["hbm_copy_gbs"] = f"unavailable ({type(e).__name__})"
    for key, cmd in (("smi", ["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--showcomputepartition",
                              "--showmemorypartition", "--json"]),):
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout
            j = json.loads(out[out.index("{"):])
            card = j.get("card0", j)
            rec[key] = {k: v for k, v in card.items() if any(w in k.lower() for w in ("sclk", "mclk", "fclk", "socclk", "power", "partition", "performance"))}
        except Exception as e:   # noqa: BLE001 -- a report field
            rec[key] = f"unavailable ({type(e).__name__})"
    # KFD topology of the GPU node(s) (harvesting shows up as simd_count / cu_count / array geometry; num_xcc; firmware versions) and the
    # board's VBIOS / serial-independent ids: the per-box data the review asked to correlate with the two speed groups
    try:
        import glob
        nodes = []
        for f in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
            try:
                kv = dict(l.split(None, 1) for l in open(f).read().splitlines() if " " in l)
            except OSError:   # nodes of devices outside this container's cgroup are not readable
                continue
            if int(kv.get("simd_count", "0")) == 0:
                continue
            keep = ("simd_count", "cu_count", "array_count", "simd_arrays_per_engine", "cu_per_simd_array", "simd_per_cu", "num_xcc", "max_waves_per_simd",
                    "lds_size_in_kb", "max_engine_clk_fcompute", "fw_version", "sdma_fw_version", "gfx_target_version", "num_sdma_engines",
                    "num_cp_queues", "local_mem_size", "capability", "debug_prop", "unique_id", "location_id", "domain", "drm_render_minor")
            nodes.append({k: kv[k].strip() for k in keep if k in kv})
        rec["kfd"] = nodes
    except Exception as e:   # noqa: BLE001 -- a report field
        rec["kfd"] = f"unavailable ({type(e).__name__})"
    try:
        import glob
        vb = [open(f).read().strip() for f in sorted(glob.glob("/sys/class/drm/card*/device/vbios_version"))]
        rec["vbios"] = sorted(set(vb))
    except Exception as e:   # noqa: BLE001
        rec["vbios"] = f"unavailable ({type(e).__name__})"
    try:
        out = subprocess.run(["rocm-smi", "--showhw", "--showuniqueid", "--showserial"], capture_output=True, text=True, timeout=20).stdout
        rows = [" ".join(l.split()) for l in out.splitlines() if l.strip() and not set(l.strip()) <= set("=-")]
        rec["showhw"] = [r for r in rows if "ROCm System" not in r and "End of ROCm" not in r][:8]
    except Exception as e:   # noqa: BLE001
        rec["showhw"] = f"unavailable ({type(e).__name__})"
    return rec


def rel_errors(got, want):
    got, want = got.double().cpu(), want.double().cpu()
    return {"rel_l2": float((got - want).norm() / want.norm()), "max_norm": float((got - want).abs().max() / want.abs().max())}


# ----------------------------------------------------------------------------------------------------------------------
# one workload = model + diffusion + resident inputs for this rank's samples
# ----------------------------------------------------------------------------------------------------------------------
class Case:
    def __init__(self, fmt, B, T, precision, dev, sample_ids, respacing="", sampler="ddpm", max_batch=None):
        from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
        from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
        from audio2photoreal_amd.spec import face_spec, pose_spec
        from audio2photoreal_amd.synthetic import cond_tokens_for_frames, synthetic_state_dict, synthetic_tensor
        self.fmt, self.B, self.T, self.precision, self.dev, self.sampler = fmt, B, T, precision, dev, sampler
        self.spec = face_spec() if fmt == "face" else pose_spec()
        self.S0 = cond_tokens_for_frames(T)
        self.sd = synthetic_state_dict(self.spec, 10)
        model, self.diffusion = create_model_and_diffusion(default_args(fmt, timestep_respacing=respacing), "test",
                                                           precision=precision, max_batch=max_batch or B)
        load_model(model, self.sd)
        self.model = model.to(dev).eval()
        self.cfg = ClassifierFreeSampleModel(self.model)
        spec = self.spec
        # inputs indexed by GLOBAL sample id so results do not depend on the world size
        self.cond = torch.stack([synthetic_tensor(10, f"cond_embed/{g}", (self.S0, spec.cond_feature_dim)) for g in sample_ids]).to(dev)
        self.x = torch.stack([synthetic_tensor(10, f"x_T/{g}", (spec.nfeats, 1, T)) for g in sample_ids]).to(dev)
        self.y = {"cond_embed": self.cond, "scale": torch.full((B,), 10.0 if fmt == "face" else 2.0, device=dev)}
        if spec.is_pose:
            nk = len(range(T)[:: spec.keyframe_step])
            self.y["keyframes"] = torch.stack([synthetic_tensor(10, f"keyframes/{g}", (nk, spec.keyframe_dim)) for g in sample_ids]).to(dev)
            self.y["mask"] = torch.ones(B, 1, 1, T, dtype=torch.bool, device=dev)
        self.gen = torch.Generator(device=dev).manual_seed(1234 + sample_ids[0])
        self.n_chain = self.diffusion.num_timesteps
        self.steps_idx = self.diffusion._step_index_tensor(dev, B)
        self.state = {"x": self.x, "i": self.n_chain - 1}

    def setup(self):
        """Untimed one-off work: context, weight upload, hoisted conditioning (timed separately), per-box shape calibration."""
        self.model.prepare(self.x, self.y)
        torch.cuda.synchronize()
        self.model.invalidate_cond()
        t0 = time.perf_counter()
        self.model.prepare(self.x, self.y)            # hoisted conditioning: once per clip, outside the loop
        torch.cuda.synchronize()
        self.prepare_s = time.perf_counter() - t0
        with torch.no_grad():
            # warm-up forwards: the library measures which chain-kernel family (MID / POST: kernels_chain.h or the tall kernels) is faster on
            # THIS box during the first 11 forwards of a given size (csrc/a2p_lib_run.h chain_pick_family; with A2P_CHAIN_TUNE=1 another 6
            # for the workgroup shape, chain_pick_nw) -- all of them here, so that no timed step carries a calibration event wait whatever
            # --warmup the caller passes
            for _ in range(18):
                self.cfg(self.x, self.steps_idx[0], self.y)
        torch.cuda.synchronize()

    def run_steps(self, n):
        st, d = self.state, self.diffusion
        for _ in range(n):
            if self.sampler == "ddpm":
                noise = torch.randn(self.x.shape, device=self.dev, generator=self.gen)        # randn_like(x) of p_sample
                out = d.p_sample(self.cfg, st["x"], self.steps_idx[st["i"]], clip_denoised=False, model_kwargs={"y": self.y}, noise=noise)
            else:
                out = d.ddim_sample(self.cfg, st["x"], self.steps_idx[st["i"]], clip_denoised=False, model_kwargs={"y": self.y})
            st["x"] = out["sample"]
            st["i"] = st["i"] - 1 if st["i"] > 0 else self.n_chain - 1

    def step_flops(self):
        fl = algorithmic_flops(self.spec, self.T, self.S0 + 2, 2 * self.B)
        return fl["decoder_gemm"] + fl["attn_self"] + fl["attn_cross"]   # decoder attention + FFN + projections


def time_case(case, steps, warmup, repeats, barrier):
    """`repeats` x [EXACTLY `steps` steps bracketed by barrier + synchronize]; returns the per-repeat wall times."""
    dts = []
    with torch.no_grad():
        case.run_steps(warmup)
        for _ in range(repeats):
            barrier()
            t0 = time.perf_counter()
            case.run_steps(steps)
            barrier()
            dts.append(time.perf_counter() - t0)
    return dts


def kernel_breakdown(case, ksteps):
    """Per-kernel-class time inside the step from dispatch-packet events on the launch stream (a2p_kernel_timing), and the
    roofline record of the dominant class."""
    import ctypes as C
    from audio2photoreal_amd import _lib
    lib = case.model._lib()
    flops = algorithmic_flops(case.spec, case.T, case.S0 + 2, 2 * case.B)
    # the decoder-layer GEMMs run inside the fused "chain" kernels (16-bit modes at >= 1100 rows) or as separate GEMM launches (fp32
    # mode, A2P_NO_CHAIN=1, the small-forward kernels below 1100 rows): the FLOPs go to whichever class actually launched (below)
    dec_gemm = flops.pop("decoder_gemm")
    kernels = {}
    for name, kind in (("chain", _lib.KERNEL_CHAIN), ("gemm", _lib.KERNEL_GEMM), ("attn_self", _lib.KERNEL_ATTN_SELF),
                       ("attn_cross", _lib.KERNEL_ATTN_CROSS), ("ln_rope", _lib.KERNEL_LNROPE)):
        _lib.check(lib.a2p_kernel_timing(case.model._ctx, kind, 1), "a2p_kernel_timing")
        with torch.no_grad():
            case.run_steps(ksteps)
        ms, n = C.c_double(), C.c_int64()
        _lib.check(lib.a2p_kernel_time_ms(case.model._ctx, C.byref(ms), C.byref(n)), "a2p_kernel_time_ms")
        _lib.check(lib.a2p_kernel_timing(case.model._ctx, kind, 0), "a2p_kernel_timing")
        if n.value == 0:
            continue
        per_step_ms = ms.value / ksteps
        kernels[name] = {"ms_per_step": round(per_step_ms, 4), "launches_per_step": n.value // ksteps,
                         "avg_launch_us": round(1e3 * ms.value / max(n.value, 1), 2)}
    if "chain" in kernels:
        flops["chain"], flops["gemm"] = dec_gemm, flops["io_gemm"]
    else:
        flops["gemm"] = dec_gemm + flops["io_gemm"]
    # finer classes of the chain launches (PRE / MID / POST / MIDPOST) and the body model's fused output tail, timed the same way
    # (include/a2p_hip.h A2P_KERNEL_CHAIN_*; a launch is timed when its class OR its sub-class is selected)
    sub = {}
    if "chain" in kernels or case.spec.is_pose:
        d, ff, T, ns, L = case.spec.latent_dim, case.spec.ff_size, case.T, 2 * case.B, case.spec.num_layers
        pre_fl, q_fl, o_fl, ffn_fl = 2.0 * T * d * 3 * d, 2.0 * T * d * d, 2.0 * T * d * d, 4.0 * T * d * ff
        # algorithmic FLOPs per LAUNCH over all `ns` sequences: PRE = [Q|K|V]; MID = out_proj + Q; POST = out_proj + FFN (+ the next
        # layer's PRE work for all but the last layer: averaged); MIDPOST (body) = MID2 + keyframe attention + POST
        sub_fl = {"chain_pre": ns * pre_fl, "chain_mid": ns * (o_fl + q_fl), "chain_post": ns * (o_fl + ffn_fl + pre_fl * (L - 1) / L),
                  "chain_midpost": ns * (2 * o_fl + q_fl + ffn_fl + pre_fl * (L - 1) / L + 4.0 * T * 20 * d), "pose_tail": None}
        for name, kind in (("chain_pre", _lib.KERNEL_CHAIN_PRE), ("chain_mid", _lib.KERNEL_CHAIN_MID), ("chain_post", _lib.KERNEL_CHAIN_POST),
                           ("chain_midpost", _lib.KERNEL_CHAIN_MIDPOST), ("pose_tail", _lib.KERNEL_POSE_TAIL)):
            if name == "pose_tail" and not case.spec.is_pose:
                continue
            if name != "pose_tail" and "chain" not in kernels:
                continue
            _lib.check(lib.a2p_kernel_timing(case.model._ctx, kind, 1), "a2p_kernel_timing")
            with torch.no_grad():
                case.run_steps(ksteps)
            ms, n = C.c_double(), C.c_int64()
            _lib.check(lib.a2p_kernel_time_ms(case.model._ctx, C.byref(ms), C.byref(n)), "a2p_kernel_time_ms")
            _lib.check(lib.a2p_kernel_timing(case.model._ctx, kind, 0), "a2p_kernel_timing")
            if n.value == 0:
                continue
            ent = {"ms_per_step": round(ms.value / ksteps, 4), "launches_per_step": n.value // ksteps,
                   "avg_launch_us": round(1e3 * ms.value / n.value, 2)}
            if sub_fl[name]:
                ent["algorithmic_gflop_per_launch"] = round(sub_fl[name] / 1e9, 3)
                ent["tflops"] = round(sub_fl[name] / (ent["avg_launch_us"] * 1e-6) / 1e12, 2)
                ent["mfma_frac"] = round(ent["tflops"] / (PEAK_F32_TFLOPS if case.precision == "fp32" else PEAK_BF16_TFLOPS), 4)
            sub[name] = ent
    for name, ent in kernels.items():
        if name in flops:
            ent["algorithmic_gflop_per_step"] = round(flops[name] / 1e9, 2)
            ent["tflops"] = round(flops[name] / (ent["ms_per_step"] * 1e-3) / 1e12, 2)
    # each class is timed in its own pass (dispatch-packet events serialise the launches of that class against the side stream);
    # the sum of the classes against the untimed step says how much that inflates
    kernels["_sum_of_classes_ms_per_step"] = round(sum(v["ms_per_step"] for v in kernels.values()), 4)
    if sub:
        kernels["_sub_classes"] = sub     # (not part of the sum: every entry is a subset of "chain" / "gemm")
    dom = max((k for k in kernels if k in flops and not k.startswith("_")), key=lambda k: kernels[k]["ms_per_step"])
    peak = PEAK_F32_TFLOPS if case.precision == "fp32" else PEAK_BF16_TFLOPS
    roofline = {"kernel": dom, "bound": "mfma", "achieved": kernels[dom]["tflops"], "peak": peak, "unit": "TFLOP/s",
                "frac": round(kernels[dom]["tflops"] / peak, 4), "traffic": None,
                "avg_launch_us": kernels[dom]["avg_launch_us"],
                "algorithmic_gflop_per_launch": round(flops[dom] / 1e9 / kernels[dom]["launches_per_step"], 3)}
    if dom == "chain" and case.fmt == "face":
        # since round 6 the first and the last chain launch of a face forward also compute input_projection and final_layer (split-operand islands, three 16-bit products each):
        # their MFMA work is NOT in the algorithmic figure above, their time IS in the class
        roofline["note"] = ("the 17 chain launches of a step include input_projection and final_layer (chain4_kernel<MT, CHAIN_IN> / <MT, POST, 2>); "
                            "only the decoder layers' GEMM flops are counted")
    return kernels, roofline


def chain_workgroup_waves(case):
    """Waves per workgroup of the chain kernels the library launched last (4 | 8).  Deterministic since round 4 (8 unless
    A2P_CHAIN_NW / A2P_CHAIN_TUNE say otherwise; csrc/a2p_lib_run.h chain_pick_nw); both shapes produce identical bits."""
    import ctypes as C
    from audio2photoreal_amd import _lib
    nw = C.c_int32(0)
    try:
        _lib.check(case.model._lib().a2p_debug_read(case.model._ctx, b"chain_nw", C.byref(nw), 4), "a2p_debug_read")
    except Exception:   # noqa: BLE001 -- a report field, never a reason to lose the line
        return None
    return int(nw.value)


def chain_family(case):
    """Chain kernel family the library settled on for this workload on this box (two digits, MID then POST: 1 = kernels_chain.h, 4 = the tall kernels_chain4.h;
    csrc/a2p_lib_run.h chain_pick_family: measured during the first forwards; both produce identical bits)."""
    import ctypes as C
    from audio2photoreal_amd import _lib
    v = C.c_int32(0)
    try:
        _lib.check(case.model._lib().a2p_debug_read(case.model._ctx, b"chain_family", C.byref(v), 4), "a2p_debug_read")
    except Exception:   # noqa: BLE001 -- a report field
        return None
    return int(v.value)


def load_probe(case, dev, seconds=0.7):
    """Board power / core clock / junction temperature WHILE this workload's steps run (never inside a timed region): the host enqueues
    `seconds` worth of steps, samples the hwmon sensors (sysfs; rocm-smi as the fallback) while the GPU works through them, then waits.
    Round 5 found B=32 power-capped (1330 W of 1400, 2.19 GHz instead of 2.39: profiles/r05_power_clock_b8_b32.txt); with this record every
    bench line says what ITS box did under ITS load (sampled by a second host thread while the steps run) -- the datum to correlate with the two speed groups of the leased boxes."""
    import glob
    import subprocess

    # the hwmon directory of THIS device: a node's sysfs lists all of its GPUs, also those outside the container's cgroup (first version
    # read a neighbour's idle sensors), so the card is matched by PCI address; no match -> rocm-smi (which only sees the visible device)
    hwmon = None
    try:
        pr = torch.cuda.get_device_properties(dev)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        for card in glob.glob("/sys/class/drm/card*/device"):
            if os.path.basename(os.path.realpath(card)).lower() == bdf:
                hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
                hwmon = hw[0] if hw else None
    except Exception:   # noqa: BLE001 -- a report field
        hwmon = None

    def sample():
        out = {}
        if hwmon is not None:
            try:
                for name, key, scale in (("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6), ("freq1_input", "sclk_mhz", 1e-6),
                                         ("temp2_input", "junction_c", 1e-3), ("temp1_input", "edge_c", 1e-3)):
                    f = os.path.join(hwmon, name)
                    if key not in out and os.path.exists(f):
                        out[key] = round(float(open(f).read()) * scale, 1)
            except (OSError, ValueError):
                out = {}
            if out:
                return out
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
            import re
            for key, pat in (("power_w", r"Power \(W\): ([\d.]+)"), ("sclk_mhz", r"sclk clock level: \d+: \((\d+)Mhz\)"), ("junction_c", r"junction\) \(C\): ([\d.]+)")):
                m = re.search(pat, txt)
                if m:
                    out[key] = float(m.group(1))
        except Exception:   # noqa: BLE001 -- a report field
            pass
        return out

    try:
        import threading
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        case.run_steps(10)
        torch.cuda.synchronize(dev)
        per = (time.perf_counter() - t0) / 10
        n = max(20, min(2000, int(seconds / max(per, 1e-5))))
        idle = sample()
        samples, stop = [], threading.Event()

        def sampler():   # a host thread beside the enqueueing one (which blocks on the launch queue's depth, so it cannot sample itself)
            while not stop.is_set():
                samples.append((time.perf_counter(), sample()))
                time.sleep(0.03)

        th = threading.Thread(target=sampler, daemon=True)
        t_start = time.perf_counter()
        th.start()
        case.run_steps(n)
        torch.cuda.synchronize(dev)
        t_end = time.perf_counter()
        stop.set()
        th.join(timeout=2.0)
        warm = [smp for t, smp in samples if t_start + 0.35 * (t_end - t_start) <= t <= t_end - 0.02 and smp]   # the power manager has settled
        keys = sorted({k for smp in warm for k in smp})
        return {"steps": n, "seconds": round(t_end - t_start, 3), "source": "hwmon sysfs" if hwmon else "rocm-smi", "idle_before": idle, "samples": len(warm),
                **{k: {"min": min(smp[k] for smp in warm if k in smp), "max": max(smp[k] for smp in warm if k in smp)} for k in keys}}
    except Exception as e:   # noqa: BLE001 -- a report field
        return f"unavailable ({type(e).__name__}: {e})"


def leg_record(case, steps, warmup, repeats, ksteps=3):
    """Sub-record of a secondary workload (same measurement as the headline, fewer fields)."""
    case.setup()
    dts = time_case(case, steps, warmup, repeats, torch.cuda.synchronize)
    dt = statistics.median(dts)
    kernels, roofline = kernel_breakdown(case, ksteps)
    peak = PEAK_F32_TFLOPS if case.precision == "fp32" else PEAK_BF16_TFLOPS
    return {"workload": f"{case.fmt} B={case.B} x2 CFG, T={case.T}, {case.S0}+2 cond tokens, {case.sampler} step, {case.precision}",
            "value": round(steps / dt, 3), "unit": "steps/s", "ms_per_step": round(1e3 * dt / steps, 4),
            "sample_steps_per_sec": round(case.B * steps / dt, 2), "repeats_ms_per_step": [round(1e3 * t / steps, 4) for t in dts],
            "decoder_tflops": round(case.step_flops() * steps / dt / 1e12, 2),
            "decoder_mfma_frac": round(case.step_flops() * steps / dt / 1e12 / peak, 4),
            "roofline": roofline, "kernels": kernels, "prepare_s": round(case.prepare_s, 4),
            "chain_workgroup_waves": chain_workgroup_waves(case) if "chain" in kernels else None,
            "chain_family": chain_family(case) if "chain" in kernels else None,
            "under_load": load_probe(case, case.dev) if "chain" in kernels else None}


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle) + parity of both GPU modes against that same oracle run
# ----------------------------------------------------------------------------------------------------------------------
def cpu_and_parity(case, dev, want_parity=True, chain_steps=1000, chain_batch=2, all_cores_live=False):
    """The oracle (CPU restatement of the reference algorithm, oracle/a2p_oracle.py) on ONE sample of the bench workload at
    the bench shape: 4 DDPM steps at the head of the 1000-step chain (t = 999..996) and 4 at its tail (t = 3..0, where the
    model's x0 carries the whole update).  The wall time of those 8 steps is the `cpu_baseline`; their outputs are the
    reference both GPU modes are compared with (`parity.short`).  `parity.chain` is the drift of the FULL 1000-step chain:
    the benchmarked bf16 mode vs the GPU fp32 mode (which is parity-pinned against the oracle / reference goldens) under
    identical per-step noise."""
    from oracle import a2p_oracle as O
    from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
    from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
    spec, T, S0 = case.spec, case.T, case.S0
    ncpu = os.cpu_count() or 1
    threads = min(ncpu, 32)       # torch CPU matmuls at these sizes stop scaling (and thrash) past ~32 threads
    torch.set_num_threads(threads)
    den = O.OracleDenoiser(case.sd, case.fmt, spec.num_layers, spec.num_heads)
    ce, xc = case.cond[:1].cpu(), case.x[:1].cpu()
    scale = float(case.y["scale"][0])
    fn = lambda xx, ts: den.forward_cfg(xx, ts, ce, torch.full((1,), scale))
    smp = O.OracleSampler("")
    t_list = [999, 998, 997, 996, 3, 2, 1, 0]
    g = torch.Generator().manual_seed(4321)
    nz = [torch.randn(xc.shape, generator=g) for _ in t_list]
    x_tail = torch.randn(xc.shape, generator=g) * 0.5          # a plausible late-chain state for the tail steps
    want = {}
    with torch.no_grad():
        smp.p_sample(fn, xc, torch.tensor([999]), nz[0])       # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        cur = xc
        for k, t in enumerate(t_list):
            if t == 3:
                want["head"] = out
                cur = x_tail
            out = smp.p_sample(fn, cur, torch.tensor([t]), nz[k])
            cur = out["sample"]
        want["tail"] = out
        cdt = time.perf_counter() - t0
    # cpu_baseline: the SAME step the GPU is timed on -- the whole batch (B samples, both guidance passes) through the oracle, 2 DDPM
    # steps at the head of the chain after one warm-up step.  (Rounds 1-4 timed ONE sample and divided by B: CPU matmul efficiency
    # at B is not B x the B=1 efficiency; the 1-sample figure stays as `one_sample_scaled` for continuity.)
    Bc = case.B
    xb, ceb, scb = case.x.cpu(), case.cond.cpu(), case.y["scale"].cpu()
    fnb = lambda xx, ts: den.forward_cfg(xx, ts, ceb, scb)
    nzb = [torch.randn(xb.shape, generator=g) for _ in range(3)]
    n_b = 2
    tb_list = [999, 998, 997]
    want_b = []                                            # the oracle's outputs at the BENCHMARKED batch: parity.b<B> below
    with torch.no_grad():
        out = smp.p_sample(fnb, xb, torch.full((Bc,), 999), nzb[0])     # warm-up at the real batch
        want_b.append(out)
        warm = out["sample"]
        cur = warm
        t0 = time.perf_counter()
        for k in range(n_b):
            out = smp.p_sample(fnb, cur, torch.full((Bc,), 998 - k), nzb[1 + k])
            want_b.append(out)
            cur = out["sample"]
        bdt = time.perf_counter() - t0
        # SURVEY section 8d asks for the box's cores.  Every host CPU as a torch thread is measured ON REQUEST only (--cpu-all-cores): on
        # the 256-CPU GPU boxes these matmul sizes thrash (round 6: ONE sample x ONE step took 231 s at 256 threads against 2.7 s for the
        # whole batch at 32), i.e. the leg alone would add ~8 minutes to the default run.  The default line carries the measured record
        # (profiles/r06_cpu_all_cores.json) with its source; the 32-thread figure stays the headline of this record.
        all_cores = None
        if ncpu > threads and all_cores_live:
            torch.set_num_threads(ncpu)
            fn1 = lambda xx, ts: den.forward_cfg(xx, ts, ceb[:1], scb[:1])
            cur = smp.p_sample(fn1, xb[:1], torch.full((1,), 999), nzb[0][:1])["sample"]     # re-warm the larger pool
            t0 = time.perf_counter()
            smp.p_sample(fn1, cur, torch.full((1,), 998), nzb[1][:1])
            adt = time.perf_counter() - t0
            all_cores = {"value": round(1.0 / adt / Bc, 5), "cores": ncpu, "seconds": round(adt, 2), "measured": "this run",
                         "what": f"ONE sample x 1 step with {ncpu} torch threads, divided by the batch {Bc} (the {threads}-thread figure above times the whole batch)"}
            torch.set_num_threads(threads)
        elif ncpu > threads:
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_cpu_all_cores.json")) as f:
                    all_cores = json.load(f)
                all_cores["measured"] = "profiles/r06_cpu_all_cores.json (a 256-CPU GPU box, round 6; NOT this run: --cpu-all-cores re-measures, ~8 min)"
            except (OSError, ValueError):
                all_cores = None
    cpu = {"value": round(n_b / bdt, 5),
           "unit": f"denoise steps/sec at batch {Bc} (the whole batch timed: {n_b} steps)",
           "cores": threads, "host_cpus": ncpu, "all_host_cpus_as_threads": all_cores, "kind": "port",
           "sample": f"oracle (torch CPU fp32 restatement of the reference algorithm, conditioning path recomputed every forward like "
                     f"the reference's decoder-only path), {Bc} samples x {n_b} DDPM steps (t = 998, 997) after one warm-up step, "
                     f"T={T}, S={S0 + 2}: {bdt:.2f} s",
           "one_sample_scaled": {"value": round(len(t_list) / cdt / case.B, 5),
                                 "what": f"rounds 1-4's figure: 1 sample x {len(t_list)} steps ({cdt:.2f} s), divided by {case.B}"}}
    # The reference tree is not on the GPU box: the build container timed the reference ITSELF beside this port on one sample of the
    # same workload (tests/tools/cpu_reference_vs_port.py --variant-a); the ratios travel with the repo.  Two variants of BASELINE.md
    # section 3: decoder-only (audio features fed in: what this port computes) and variant (A), what a user of the reference gets --
    # its own encode_audio / encode_lip re-run in every forward of every step.
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r04_cpu_reference_variants.json")) as f:
            ab = json.load(f)
        src = ("profiles/r04_cpu_reference_variants.json: the reference itself vs this port, build container, "
               f"{ab['threads']} threads, {ab['shape']} (NOT measured in this run)")
        cpu["reference_decoder_only"] = {"port_over_reference_speed": ab["port_over_reference_speed"],
                                         "estimated_value": round(cpu["value"] / ab["port_over_reference_speed"], 5), "source": src}
        va = ab["variant_a_reference_with_its_audio_front_end"]
        cpu["reference_variant_a"] = {"slowdown_vs_decoder_only_reference": va["slowdown_vs_decoder_only_reference"],
                                      "estimated_value": round(cpu["value"] / ab["port_over_reference_speed"] / va["slowdown_vs_decoder_only_reference"], 5),
                                      "what": va["note"], "source": src}
    except (OSError, KeyError, ValueError):
        pass
    if not want_parity:
        return cpu, None

    def gpu_short(precision):
        if precision == case.precision:
            cfg, diff = case.cfg, case.diffusion
        else:
            m, diff = create_model_and_diffusion(default_args(case.fmt, timestep_respacing=""), "test", precision=precision, max_batch=2)
            load_model(m, case.sd)
            cfg = ClassifierFreeSampleModel(m.to(dev).eval())
        y = {"cond_embed": case.cond[:1].contiguous(), "scale": case.y["scale"][:1].contiguous()}
        idx = diff._step_index_tensor(dev, 1)
        got = {}
        with torch.no_grad():
            cur = case.x[:1].contiguous()
            for k, t in enumerate(t_list):
                if t == 3:
                    got["head"] = out
                    cur = x_tail.to(dev)
                out = diff.p_sample(cfg, cur, idx[t], clip_denoised=False, model_kwargs={"y": y}, noise=nz[k].to(dev))
                cur = out["sample"]
            got["tail"] = out
        rec = {}
        for part in ("head", "tail"):
            for key in ("sample", "pred_xstart"):
                rec[f"{part}_{key}"] = {k: float(f"{v:.3e}") for k, v in rel_errors(got[part][key], want[part][key]).items()}
        if cfg is not case.cfg:
            cfg.model.release()
        return rec

    def gpu_batch(precision):
        """The benchmarked batch itself against the oracle: the SAME three p_sample steps (t = 999, 998, 997; same x_T, conditioning and
        noise for all B samples) the cpu_baseline run above pushed through the oracle, chained on the GPU's own outputs."""
        if precision == case.precision:
            cfg, diff = case.cfg, case.diffusion
        else:
            m, diff = create_model_and_diffusion(default_args(case.fmt, timestep_respacing=""), "test", precision=precision, max_batch=Bc)
            load_model(m, case.sd)
            cfg = ClassifierFreeSampleModel(m.to(dev).eval())
        idx = diff._step_index_tensor(dev, Bc)
        rec = {}
        with torch.no_grad():
            cur = case.x.contiguous()
            for k, t in enumerate(tb_list):
                out = diff.p_sample(cfg, cur, idx[t], clip_denoised=False, model_kwargs={"y": case.y}, noise=nzb[k].to(dev))
                cur = out["sample"]
                for key in ("sample", "pred_xstart"):
                    rec[f"t{t}_{key}"] = {kk: float(f"{v:.3e}") for kk, v in rel_errors(out[key], want_b[k][key]).items()}
            per_sample = [rel_errors(out["sample"][b], want_b[-1]["sample"][b])["rel_l2"] for b in range(Bc)]
        rec["last_step_worst_single_sample_rel_l2"] = float(f"{max(per_sample):.3e}")
        if cfg is not case.cfg:
            cfg.model.release()
        return rec

    parity = {"reference": "oracle/a2p_oracle.py (CPU fp32), pinned to reference-generated goldens by tests/test_oracle_golden.py",
              "shape": f"{case.fmt} B=1 T={T} S={S0 + 2}, p_sample at t={t_list}",
              "short": {"fp32": gpu_short("fp32"), "bf16": gpu_short("bf16"), "fp16": gpu_short("fp16")}}
    parity[f"b{case.B}"] = {"what": f"the benchmarked batch ({case.fmt} B={Bc} x2 CFG, T={T}, S={S0 + 2}) through the oracle AND the product: "
                                f"chained p_sample at t={tb_list}, identical x_T / conditioning / noise (the oracle run is the cpu_baseline's)",
                        "chain_family": chain_family(case),
                        case.precision: gpu_batch(case.precision), "fp32": gpu_batch("fp32")}
    if chain_steps:
        finals = {}
        Bc = chain_batch
        shape = (Bc, spec.nfeats, 1, T)
        for precision in ("fp32", "bf16", "fp16"):
            m, diff = create_model_and_diffusion(default_args(case.fmt, timestep_respacing=""), "test", precision=precision, max_batch=Bc)
            load_model(m, case.sd)
            cfg = ClassifierFreeSampleModel(m.to(dev).eval())
            y = {"cond_embed": case.cond[:Bc].contiguous(), "scale": case.y["scale"][:Bc].contiguous()}
            gg = torch.Generator(device=dev)

            def step_noise(n):
                gg.manual_seed(50000 + n)
                return torch.randn(shape, device=dev, generator=gg)
            t0 = time.perf_counter()
            with torch.no_grad():
                gen = diff.p_sample_loop_progressive(cfg, shape, noise=case.x[:Bc].contiguous(), clip_denoised=False,
                                                     model_kwargs={"y": y}, step_noise=step_noise)
                for n, out in enumerate(gen):
                    if n + 1 >= chain_steps:
                        break
            torch.cuda.synchronize()
            finals[precision] = (out["sample"].clone(), time.perf_counter() - t0)
            m.release()
        parity["chain"] = {"what": f"final sample of the {chain_steps}-step DDPM chain, B={Bc}, T={T}: 16-bit modes vs GPU fp32 mode, identical noise",
                           "bf16": {k: float(f"{v:.3e}") for k, v in rel_errors(finals["bf16"][0], finals["fp32"][0]).items()},
                           "fp16": {k: float(f"{v:.3e}") for k, v in rel_errors(finals["fp16"][0], finals["fp32"][0]).items()},
                           **{f"{k}_chain_s": round(v[1], 2) for k, v in finals.items()}}
    # the same full chain against the ORACLE (20 CPU-minutes: offline, tests/tools/chain_vs_oracle.py; the record travels with the repo)
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_chain_vs_oracle.json")) as f:
            co = json.load(f)
        parity["chain_vs_oracle"] = {"what": co["what"], "source": "profiles/r06_chain_vs_oracle.json (builder-run: oracle in the build container, product on an MI355X; NOT measured in this run -- tests/test_hip_round6.py::test_full_sampling_chain_vs_oracle_states re-runs the product side against the committed oracle states in every GPU test run)",
                                     **{prec: {"after_1000_steps": m["steps"]["step1000"], "after_500_steps": m["steps"]["step500"]} for prec, m in co["modes"].items()}}
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_chain_vs_oracle_body.json")) as f:
            cb = json.load(f)
        parity["chain_vs_oracle"]["body_ddim100"] = {"what": cb["what"], **{prec: m["steps"]["step100"] for prec, m in cb["modes"].items()}}
        if "vs_reference" in co and "vs_reference" in cb:   # the same chains run by the reference itself (tests/golden/make_golden_chain.py): rel-L2 of the final state
            parity["chain_vs_reference"] = {"what": co["vs_reference"]["what"], "source": parity["chain_vs_oracle"]["source"],
                                            "face_1000_steps": {k: v["step1000"] for k, v in co["vs_reference"].items() if isinstance(v, dict)},
                                            "body_ddim100": {k: v["step100"] for k, v in cb["vs_reference"].items() if isinstance(v, dict)}}
    except (OSError, KeyError, ValueError):
        pass
    bar = 1e-3
    worst = {}
    for prec in ("fp32", "fp16", "bf16"):
        vals = [v["rel_l2"] for v in parity["short"][prec].values()]
        if prec in parity[f"b{case.B}"]:
            vals += [v["rel_l2"] for v in parity[f"b{case.B}"][prec].values() if isinstance(v, dict)]
        if "chain" in parity and prec in parity["chain"]:
            vals.append(parity["chain"][prec]["rel_l2"])
        worst[prec] = max(vals)
    parity["bar"] = {"rel_l2": bar, "worst_rel_l2": {k: float(f"{v:.3e}") for k, v in worst.items()},
                     "meets_bar": {k: bool(v <= bar) for k, v in worst.items()}}
    return cpu, parity


# ----------------------------------------------------------------------------------------------------------------------
class PipelineSubject:
    """One subject of BASELINE configs[4] (its own weight set: synthetic seed 10 + subject) on one GPU, from raw 48 kHz stereo audio:
    native front end (vq-wav2vec conv features + lip regressor, stub geometry / synthetic weights) -> guide tokens -> VQ keyframes
    -> body ddim -> face ddim, for the samples `sample_ids` of that subject.  Every random draw is a function of (subject, global
    sample id): how the samples of a subject are split over ranks does not change them."""

    def __init__(self, a, dev, subject, sample_ids, subject_samples=None):
        from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
        from audio2photoreal_amd.model.guide import GuideTransformer
        from audio2photoreal_amd.model.vqvae import TemporalVertexCodec
        from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
        from audio2photoreal_amd.sample_parallel import derive_seed, per_sample_noise
        from audio2photoreal_amd.spec import GuideSpec, TokenizerSpec, face_spec, pose_spec
        from audio2photoreal_amd.synthetic import (synthetic_frontend_state_dict, synthetic_guide_state_dict, synthetic_state_dict,
                                                   synthetic_tensor, synthetic_tokenizer_state_dict)
        self.dev, self.subject, self.ids = dev, subject, list(sample_ids)
        B, T = len(self.ids), a.frames
        self.B, self.T = B, T
        seed = 10 + subject
        respacing = a.respacing or "ddim100"
        gs, ts = GuideSpec(), TokenizerSpec()
        self.guide = GuideTransformer(tokens=gs.tokens, num_layers=gs.num_layers, dim=gs.dim, emb_len=gs.emb_len,
                                      num_audio_layers=gs.num_audio_layers, max_batch=B, max_positions=96)
        self.guide.load_state_dict(synthetic_guide_state_dict(gs, seed), strict=False)
        tok = TemporalVertexCodec(ts.n_vertices, ts.latent_dim, ts.categories, ts.residual_depth)
        tok.load_state_dict(synthetic_tokenizer_state_dict(ts, seed), strict=False)
        self.models = {}
        for fmt, spec in (("pose", pose_spec()), ("face", face_spec())):
            m, d = create_model_and_diffusion(default_args(fmt, timestep_respacing=respacing), "test", precision=a.precision, max_batch=B,
                                              audio_frontend="native")
            load_model(m, {**synthetic_state_dict(spec, seed), **synthetic_frontend_state_dict(seed, lip=fmt == "face")})
            if fmt == "pose":
                m.setup_guide_predictor(self.guide.to(dev).eval(), tok.to(dev))
            if subject_samples and subject_samples > B:     # a block of a subject's samples: take the kernel family of the whole subject
                m.global_batch_hint = subject_samples       # (include/a2p_hip.h a2p_set_batch_hint; sample_parallel does the same)
            self.models[fmt] = (spec, ClassifierFreeSampleModel(m.to(dev).eval()), d)
        self.nk = len(range(T)[::30])
        # y["audio"]: z-normalised 48 kHz stereo [B, T * 1600, 2], one stream per (subject, global sample id)
        self.audio = torch.stack([synthetic_tensor(seed, f"audio/{g}", (T * 1600, 2)) for g in self.ids]).to(dev)
        n_uni = self.nk * tok.residual_depth
        self.uniforms = torch.stack([torch.rand(n_uni, generator=torch.Generator().manual_seed(derive_seed(4321, subject, 1, g)))
                                     for g in self.ids], dim=1)                                   # [n, B]: column b belongs to sample ids[b]
        self.noise = {fmt: per_sample_noise((B, spec.nfeats, 1, T), [derive_seed(4321, subject, 2 + k, g) for g in self.ids]).to(dev)
                      for k, (fmt, (spec, _, _)) in enumerate(self.models.items())}

    def _invalidate(self):
        self.guide.invalidate_cond()                       # every run pays the hoisted conditioning of all three models
        for _, cfg_m, _ in self.models.values():
            cfg_m.model.invalidate_cond()

    def _body(self, feats):
        from audio2photoreal_amd.sample.generate import _replace_keyframes
        dev, B, T = self.dev, self.B, self.T
        spec, cfg, diff = self.models["pose"]
        y = {"cond_embed": feats, "keyframes": torch.zeros(B, self.nk, 104, device=dev), "mask": torch.ones(B, 1, 1, T, dtype=torch.bool, device=dev),
             "scale": torch.full((B,), 2.0, device=dev)}
        y["keyframes"] = _replace_keyframes({"y": y}, cfg, self.uniforms).to(dev)
        if os.environ.get("A2P_BENCH_DEBUG"):     # diagnosis of run-to-run differences: which stage moved?
            import hashlib
            hh = lambda t: hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:10]
            print(f"[debug] subject {self.subject} ids {self.ids}: feats {hh(feats)} keyframes {hh(y['keyframes'])}", file=sys.stderr, flush=True)
        return y, lambda: diff.ddim_sample_loop(cfg, (B, spec.nfeats, 1, T), noise=self.noise["pose"], clip_denoised=False, model_kwargs={"y": y})

    def _face(self, face_ce):
        spec, cfg, diff = self.models["face"]
        yf = {"cond_embed": face_ce, "scale": torch.full((self.B,), 10.0, device=self.dev)}
        return diff.ddim_sample_loop(cfg, (self.B, spec.nfeats, 1, self.T), noise=self.noise["face"], clip_denoised=False, model_kwargs={"y": yf})

    def sequential(self):
        """Stage by stage with a synchronise in between: the per-stage times."""
        self._invalidate()
        torch.cuda.synchronize()
        st, t0 = {}, time.perf_counter()

        def mark(name):
            nonlocal t0
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            st[name], t0 = (t1 - t0) * 1e3, t1
        feats = self.models["pose"][1].model.audio_frontend.encode_audio(self.audio)     # encode_audio: shared by the guide, body and face models
        face_ce = self.models["face"][1].model.audio_frontend.encode_lip(self.audio, feats)  # encode_lip: + the lip regressor's 1014 channels
        mark("audio_front_end_ms")
        _, run_body = self._body(feats)
        mark("guide_tokens_and_vq_decode_ms")
        body = run_body()
        mark("body_ddim_ms")
        face = self._face(face_ce)
        mark("face_ddim_ms")
        assert bool(torch.isfinite(body).all()) and bool(torch.isfinite(face).all())
        return st, body, face

    def overlapped(self):
        """The same work with the dependency graph it actually has: the face model needs only the audio features, the body model
        needs the guide's keyframes.  Face runs on one HIP stream; guide (one CU per sequence, latency-bound) -> VQ decode -> body on
        another, so the guide transformer's 54 ms pass under the face model's denoising steps."""
        dev = self.dev
        self._invalidate()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        feats = self.models["pose"][1].model.audio_frontend.encode_audio(self.audio)
        face_ce = self.models["face"][1].model.audio_frontend.encode_lip(self.audio, feats)
        main = torch.cuda.current_stream(dev)
        s_face, s_body = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        s_face.wait_stream(main)
        s_body.wait_stream(main)
        # the loops' once-per-call non-finite check reads a device flag (= waits for its stream): deferred until both streams are
        # loaded, or the host would sit in the face loop's check while the body stream has nothing to do
        for _, _, diff in self.models.values():
            diff.defer_finite_check = True
        try:
            with torch.cuda.stream(s_face):           # enqueued first: nothing on this stream blocks the host
                face = self._face(face_ce)
            with torch.cuda.stream(s_body):
                body = self._body(feats)[1]()
            with torch.cuda.stream(s_face):
                self.models["face"][1].a2p_check_finite()
            with torch.cuda.stream(s_body):
                self.models["pose"][1].a2p_check_finite()
        finally:
            for _, _, diff in self.models.values():
                diff.defer_finite_check = False
        main.wait_stream(s_face)
        main.wait_stream(s_body)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, body, face

    def release(self):
        for _, cfg_m, _ in self.models.values():
            cfg_m.model.release()


def run_pipeline(a, dev):
    """One subject on one GPU (`bench.py --pipeline`): sequential stage times, the two-stream schedule, and that both give the
    same samples bit for bit."""
    B = a.batch
    subj = PipelineSubject(a, dev, 0, list(range(B)))
    subj.sequential()                         # contexts, weight upload, allocator warm-up
    st, body_seq, face_seq = subj.sequential()
    total = sum(st.values()) / 1e3
    _, body_ov, face_ov = subj.overlapped()
    same = bool(torch.equal(body_ov, body_seq)) and bool(torch.equal(face_ov, face_seq))
    assert same, "the two-stream schedule changed the samples"
    ov = min(subj.overlapped()[0] for _ in range(2))
    line = {"metric": "end-to-end sec/sample: audio front end -> guide transformer -> body ddim -> face ddim, from raw "
                      "48 kHz audio (BASELINE configs[4] shape, one subject, one GPU)",
            "value": round(min(total, ov) / B, 5), "unit": "s/sample", "higher_is_better": False, "n_gpus": 1, "batch": B, "frames": a.frames,
            "respacing": a.respacing or "ddim100", "dtype": a.precision, "data": "synthetic", "total_s": round(min(total, ov), 4),
            "sequential_total_s": round(total, 4), "overlapped_total_s": round(ov, 4), "overlapped_equals_sequential": same,
            "value_note": "overlapped = face on one HIP stream, guide -> VQ -> body on another (the face model does not depend on "
                          "the guide); stages_ms are the per-stage times of the sequential run",
            "stages_ms": {k: round(v, 2) for k, v in st.items()}}
    print(json.dumps(line))
    return line


def pipeline_placement(subjects, samples, world):
    """rank -> [(subject, first sample, one-past-last sample)] for BASELINE configs[4] (`subjects` weight sets x `samples` samples over
    `world` GPUs).  world >= subjects: world // subjects ranks per weight set (the reference ships one checkpoint pair per subject, so a
    rank holds ONE subject's weights), each taking a contiguous block of that subject's samples; the world % subjects leftover ranks join
    the first subjects.  world < subjects: subjects are dealt round-robin, all samples of a subject on its rank."""
    from audio2photoreal_amd.sample_parallel import shard_bounds
    plan = [[] for _ in range(world)]
    if world >= subjects:
        base, extra = divmod(world, subjects)
        r = 0
        for s in range(subjects):
            n = base + (1 if s < extra else 0)
            for k in range(n):
                lo, hi = shard_bounds(samples, n, k)
                plan[r].append((s, lo, hi))
                r += 1
    else:
        for s in range(subjects):
            plan[s % world].append((s, 0, samples))
    return plan


def run_pipeline_job(a, dev, rank, world, dist, coll_dev):
    """BASELINE configs[4] as a job: `--subjects` weight sets x `--batch` samples each over `--gpus` ranks (placement above), the
    two-stream schedule per (rank, subject), NO communication until the one gather of all [body | face] samples at the end.
    value = end-to-end seconds per sample = max-over-ranks wall time / (subjects x samples)."""
    from audio2photoreal_amd.sample_parallel import gather_blocks
    S, N = a.subjects, a.batch
    plan = pipeline_placement(S, N, world)
    mine = [(s, lo, hi) for (s, lo, hi) in plan[rank] if hi > lo]
    units = [PipelineSubject(a, dev, s, list(range(lo, hi)), subject_samples=N) for (s, lo, hi) in mine]
    for u in units:
        u.overlapped()                        # contexts, weight upload, allocator warm-up

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dts, outs = [], None
    for _ in range(max(1, a.repeats)):
        barrier()
        t0 = time.perf_counter()
        outs = [u.overlapped()[1:] for u in units]
        barrier()
        dts.append(time.perf_counter() - t0)
    if world > 1:
        tt = torch.tensor(dts, device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dts = [float(v) for v in tt.tolist()]
    dt = statistics.median(dts)
    T = a.frames
    local = (torch.cat([torch.cat([b, f], dim=1) for (b, f) in outs], dim=0) if outs else torch.zeros(0, 104 + 256, 1, T, device=dev)).contiguous()
    sizes = [sum(hi - lo for (_, lo, hi) in plan[r]) for r in range(world)]
    barrier()
    t0 = time.perf_counter()
    allx = gather_blocks(local.to(coll_dev), sizes) if world > 1 else local
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t0) * 1e3 if world > 1 else None
    assert allx.shape[0] == S * N and bool(torch.isfinite(allx).all())
    if rank == 0:
        order = [(s, g) for r in range(world) for (s, lo, hi) in plan[r] for g in range(lo, hi)]   # (subject, sample) of every gathered row
        digest = {f"{s}/{g}": [float(allx[i, :104].double().abs().sum().cpu()), float(allx[i, 104:].double().abs().sum().cpu())]
                  for i, (s, g) in enumerate(order)}      # [body, face] per (subject, sample)
        line = {"metric": "end-to-end sec/sample: audio front end -> guide transformer -> body ddim -> face ddim, from raw 48 kHz audio "
                          f"(BASELINE configs[4]: {S} subjects x {N} samples)",
                "value": round(dt / (S * N), 5), "unit": "s/sample", "higher_is_better": False, "n_gpus": world, "steps": 1, "warmup": 1,
                "scaling": "strong", "dtype": a.precision, "data": "synthetic", "total_s": round(dt, 4), "repeats_s": [round(t, 4) for t in dts],
                "gather_ms": gather_ms, "vs_baseline": None,
                "collectives": ({"backend": dist.get_backend(), "ranks": dist.get_world_size(), "gather": os.environ.get("A2P_GATHER", "ring"),
                                 "gather_rows": int(allx.shape[0])} if world > 1 else None),
                "box": box_record(dev),
                "config": {"workload": f"{S} subjects (weight sets) x {N} samples, T={T}, {a.respacing or 'ddim100'} body + face, native audio front end",
                           "placement": {str(r): [[s, lo, hi] for (s, lo, hi) in plan[r]] for r in range(world)},
                           "parallelism": f"subject/sample-parallel x{world}: no communication before the final gather"},
                "sample_digests": digest}
        print(json.dumps(line), flush=True)
    for u in units:
        u.release()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed steps per region (default 200: a region costs ~0.6 ms on top of its steps -- clocks and queues leaving idle -- which is 1.5 %% of a 20-step region; profiles/r03_steps_sweep.txt)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of exactly --steps steps each; the median is reported")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--precision", default="fp16", choices=["bf16", "fp16", "fp32"],
                    help="fp16 (default): 16-bit throughput mode with IEEE-half operands -- same MFMA rate as bf16, 8x smaller operand rounding "
                         "error (parity record below); bf16: the dtype BASELINE's configs name, reported as the `bf16` leg of a default run; "
                         "fp32: the parity mode (exact fp32 MFMA)")
    ap.add_argument("--model", default="face", choices=["face", "pose"],
                    help="face = BASELINE configs[1] (the metric's config, default); pose = configs[2] shape (body model, keyframes, scale 2)")
    ap.add_argument("--pipeline", action="store_true",
                    help="instead of the step benchmark: BASELINE configs[4] shape on this GPU for one subject -- audio features -> "
                         "guide transformer tokens -> VQ keyframes -> body ddim100 -> face ddim100 (demo/demo.py:156-216), sec/sample")
    ap.add_argument("--subjects", type=int, default=1,
                    help="with --pipeline: BASELINE configs[4] as a job -- this many subjects (weight sets) x --batch samples each, placed over "
                         "--gpus ranks (world // subjects ranks per subject), one gather at the end")
    ap.add_argument("--respacing", default=None, help="with --pipeline: timestep respacing of both diffusion models (default ddim100; tests use ddim5)")
    ap.add_argument("--total-samples", type=int, default=0,
                    help="STRONG scaling (BASELINE configs[3]): a fixed number of samples split over --gpus ranks (contiguous blocks; a rank may "
                         "hold none) instead of --batch samples per GPU; the pose model then steps its ddim100 chain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the oracle with EVERY host CPU as a torch thread (one sample x one step; ~8 minutes on a 256-CPU box, where it "
                         "thrashes: the default line quotes profiles/r06_cpu_all_cores.json instead)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity legs (the oracle still times the cpu_baseline)")
    ap.add_argument("--no-legs", action="store_true", help="skip the face B=32 and body B=16 sub-records")
    ap.add_argument("--chain-steps", type=int, default=1000, help="length of the bf16-vs-fp32 drift chain (0 = skip)")
    ap.add_argument("--write-parity", default=None, help="also write the parity record to this JSON file (profiles/r02_parity.json)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` started directly (the form the driver uses at N=1): launch the N ranks ourselves, one process per
        # GPU, exactly as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <args>`
        # would; rank 0 of that job prints the one JSON line on this process's stdout
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))
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
    dist = None
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
    from audio2photoreal_amd.sample_parallel import gather_samples, shard_bounds

    if a.pipeline:
        if world == 1 and a.subjects == 1:
            return run_pipeline(a, dev)
        run_pipeline_job(a, dev, rank, world, dist, coll_dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    B, T = a.batch, a.frames
    strong = a.total_samples > 0
    total = a.total_samples if strong else world * B
    lo, hi = shard_bounds(total, world, rank)              # this rank's block of the global batch (weak scaling: B per GPU; strong: total / world)
    B = hi - lo
    assert not (rank == 0 and B == 0), "no samples at all"
    case = None
    if B > 0:                                              # a rank without samples (strong scaling, total < world) only joins the collectives
        if strong and a.model == "pose":
            case = Case(a.model, B, T, a.precision, dev, list(range(lo, hi)), respacing="ddim100", sampler="ddim")
        else:
            case = Case(a.model, B, T, a.precision, dev, list(range(lo, hi)))
        if world > 1:
            case.model.global_batch_hint = total           # every shard takes the kernel family of the unsharded batch (a2p_set_batch_hint)
        case.setup()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if case is not None:
        dts = time_case(case, a.steps, a.warmup, a.repeats, barrier)
    else:
        dts = []
        for _ in range(a.repeats):
            barrier()
            barrier()
            dts.append(0.0)
    if world > 1:   # max over ranks per repeat, then the median repeat
        tt = torch.tensor(dts, device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dts = [float(v) for v in tt.tolist()]
    dt = statistics.median(dts)
    assert case is None or torch.isfinite(case.state["x"]).all(), "non-finite samples"

    # ---- the single end-of-run collective (outside the timed region): all ranks receive all samples, in global order ----
    gather_ms = None
    if world > 1:
        if case is not None:
            mine = case.state["x"].contiguous().to(coll_dev)
        else:
            from audio2photoreal_amd.spec import face_spec, pose_spec
            mine = torch.zeros(0, (face_spec() if a.model == "face" else pose_spec()).nfeats, 1, T, device=coll_dev)
        barrier()
        t0 = time.perf_counter()
        allx = gather_samples(mine, total)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t0) * 1e3
        assert allx.shape[0] == total and bool(torch.isfinite(allx).all())

    kernels, roofline = {}, None
    under_load = load_probe(case, dev) if rank == 0 and not a.no_kernel_timing else None   # after the timed regions (not under a profiler): power / clock / temperature while the steps run
    if rank == 0 and not a.no_kernel_timing:
        kernels, roofline = kernel_breakdown(case, min(a.steps, 5))
        if "chain" in kernels:
            roofline["chain_workgroup_waves"] = chain_workgroup_waves(case)
            roofline["chain_family"] = chain_family(case)
        # HBM bytes per launch of the dominant class from the rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 note of
        # MI355X_MICROARCH.md + WRITE_SIZE; scratch/run_pmc.sh writes the file) -- null when not collected for this workload
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and a.model == "face" and B == 8 and T == 600 and a.precision != "fp32":   # 16-bit modes: same bytes
            tj = json.load(open(tpath))
            traffic = tj.get(roofline["kernel"])
            # the entry of the kernel FAMILY that was timed (chain_family = 10 x MID + POST), not the mix of a calibration run
            fam = (tj.get("chain_by_family") or {}).get(str(roofline.get("chain_family"))) if roofline["kernel"] == "chain" else None
            if fam:
                traffic = dict(fam, chain_family=roofline.get("chain_family"))
            if traffic:
                roofline["traffic"], roofline["traffic_detail"] = traffic["total_bytes"], traffic
                roofline["traffic_source"] = ("profiles/pmc_traffic.json: builder-run rocprofv3 --pmc passes of this workload (scratch/run_pmc.sh), "
                                              "NOT collected in this run")

    cpu = parity = None
    legs = {}
    if rank == 0 and world == 1:   # reported at N=1 only (other ranks would idle in the final barrier)
        if not a.no_cpu_baseline and a.model == "face":
            cpu, parity = cpu_and_parity(case, dev, want_parity=not a.no_parity, chain_steps=a.chain_steps, all_cores_live=a.cpu_all_cores)
            if parity and a.write_parity:
                with open(a.write_parity, "w") as f:
                    json.dump(parity, f, indent=1)
        if not a.no_legs and a.model == "face" and B == 8 and a.precision != "fp32" and not strong:
            case.model.release()           # free the headline context before the larger ones
            if a.precision != "bf16":
                legs["bf16"] = leg_record(Case("face", 8, T, "bf16", dev, list(range(8))), a.steps, a.warmup, a.repeats)
                legs["bf16"]["note"] = "the headline workload with bfloat16 operands (the dtype BASELINE configs[1] names); parity: parity.*.bf16"
            legs["fp32"] = leg_record(Case("face", 8, T, "fp32", dev, list(range(8))), max(a.steps // 2, 5), 2, a.repeats)
            legs["fp32"]["note"] = "the headline workload in the parity mode (exact fp32 MFMA everywhere; parity.*.fp32 ~1e-6)"
            legs["b32"] = leg_record(Case("face", 32, T, a.precision, dev, list(range(32))), max(a.steps // 2, 5), 2, a.repeats)
            legs["b32"]["note"] = "north_star batch-32 roofline leg (face FiLM denoiser, p_sample step)"
            body = Case("pose", 16, T, a.precision, dev, list(range(16)), respacing="ddim100", sampler="ddim")
            legs["body"] = leg_record(body, a.steps, a.warmup, a.repeats)
            legs["body"]["note"] = "BASELINE configs[2]: body diffusion, keyframe conditioning + CFG scale 2, batch 16, 600 frames, ddim100 step"
            cfg0 = Case("face", 1, 240, a.precision, dev, [0], respacing="ddim10", sampler="ddim")
            legs["cfg0"] = leg_record(cfg0, max(500, a.steps), max(20, a.warmup), a.repeats)
            legs["cfg0"]["note"] = ("BASELINE configs[0] shape: face, batch 1, 240 frames, ddim10 step.  480 rows: below the chain kernels' "
                                    "break-even; the small-forward kernels (csrc/kernels_small.h: LayerNorm fused into the A load, whole K "
                                    "resident; attn_ksplit_kernel: keys split over the waves) make it 76 dependent launches of 5-12 us: "
                                    "latency bound, not compute bound")

    if rank == 0:
        # weak scaling: every rank steps its own B samples -> world x steps; strong scaling: ONE job of `total` samples -> steps
        value = (1 if strong else world) * a.steps / dt
        peak = PEAK_F32_TFLOPS if a.precision == "fp32" else PEAK_BF16_TFLOPS
        spec, S0 = case.spec, case.S0
        line = {
            "metric": (f"diffusion denoise steps/sec ({a.model}, {T}-frame seq, {total} samples over {world} GPU(s), CFG)" if strong else
                       f"diffusion denoise steps/sec ({a.model}, {T}-frame seq, batch {B} per GPU, CFG)"), "value": round(value, 4),
            "unit": "steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"{a.model} FiLM denoiser {spec.num_layers}L/{spec.num_heads}H d{spec.latent_dim}, "
                                   f"{'ddim100 step' if case.sampler == 'ddim' else '1000-step DDPM p_sample chain'}, B={B}/GPU x2 CFG, "
                                   f"T={T}, {S0}+2 cond tokens", "global_batch": total,
                       "parallelism": f"sample-parallel x{world}" + (f" (strong scaling: {total} samples in contiguous blocks, {B} on rank 0)" if strong else ""),
                       "dtype_note": ("16-bit MFMA operands are IEEE half instead of the bfloat16 BASELINE configs[1] names: same MFMA rate and "
                                      "kernels, and the only 16-bit format that meets the north_star's 1e-3 on the loop's return value "
                                      "(parity record); the bfloat16 run of the same workload is legs.bf16") if a.precision == "fp16" else None},
            "repeats": a.repeats, "repeats_ms_per_step": [round(1e3 * t / a.steps, 4) for t in dts],
            "value_note": ("steps of the whole job / max-over-ranks time (median repeat); one step advances all samples of the job" if strong else
                           "steps of every rank / max-over-ranks time (median repeat); one step advances B samples per GPU"),
            "sample_steps_per_sec": round(value * (total if strong else B), 3),
            "decoder_tflops": round((total / B) * case.step_flops() * a.steps / dt / 1e12, 2),
            "decoder_mfma_frac": round(case.step_flops() * a.steps / dt / 1e12 / peak, 4),
            "prepare_s": round(case.prepare_s, 4), "gather_ms": gather_ms,
            "collectives": ({"backend": dist.get_backend(), "ranks": dist.get_world_size(), "gather": os.environ.get("A2P_GATHER", "ring"),
                             "gather_rows": int(allx.shape[0]), "gather_bytes_per_rank": int(mine.numel() * mine.element_size())}
                            if world > 1 else None),
            "box": box_record(dev), "under_load": under_load,
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu, "parity": parity, "legs": legs or None,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()   # rank 0 may still be in its post-run measurement legs
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
