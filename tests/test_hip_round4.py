"""Round-4 GPU tests (`pytest -m gpu`): the multi-GPU job shapes of BASELINE configs[3] / [4] on the GPU that exists (ranks sharing
cuda:0, gloo collectives; the driver's multi-GPU node runs the same code one rank per GPU over backend "nccl" = RCCL), the
non-finite flag, and the 16-bit modes outside O(1) activations."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from audio2photoreal_amd import _lib
from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
from audio2photoreal_amd.spec import face_spec, pose_spec
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict, trained_like_state_dict
from conftest import ROOT, record

pytestmark = pytest.mark.gpu
SEED = 10


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _bench(nproc, *args, timeout=1200):
    env = dict(os.environ, A2P_BENCH_SHARE_GPU="1", A2P_BENCH_BACKEND="gloo")
    if nproc > 1:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)]
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"]
    r = subprocess.run(cmd + list(args), env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


# ----------------------------------------------------------------------------- BASELINE configs[3]: strong scaling
@pytest.mark.parametrize("nproc,total", [(2, 3), (3, 2)])
def test_strong_scaling_mode_of_the_bench(nproc, total):
    """`bench.py --total-samples N --model pose`: a FIXED number of body samples in contiguous blocks over the ranks (configs[3]:
    64 samples over 1/2/4/8 GPUs), "scaling": "strong", value = steps of the one job per second.  (3 ranks, 2 samples): the last
    rank holds no sample and still takes part in the barriers, the max-reduce and the gather."""
    line = _bench(nproc, "--model", "pose", "--total-samples", str(total), "--frames", "240", "--steps", "3", "--warmup", "1", "--repeats", "1",
                  "--no-cpu-baseline", "--no-legs", "--no-kernel-timing")
    assert line["scaling"] == "strong" and line["n_gpus"] == nproc and line["config"]["global_batch"] == total
    assert line["gather_ms"] is not None and line["gather_ms"] > 0.0 and line["value"] > 0 and line["steps"] == 3
    assert "ddim100" in line["config"]["workload"]
    record(f"bench_strong/{nproc}ranks_{total}samples", value=float(line["value"]), gather_ms=float(line["gather_ms"]))


# ----------------------------------------------------------------------------- BASELINE configs[4]: subjects x samples
def test_pipeline_job_placements_agree_sample_for_sample():
    """configs[4] as a job (`bench.py --pipeline --subjects S`): 2 subjects x 2 samples on 1 rank (both subjects one after the other),
    on 2 ranks (one subject each) and on 4 ranks (two ranks per weight set, one sample each).  Every random draw is a function of
    (subject, global sample id) and a block of a subject's samples takes the kernel family of the whole subject, so the gathered
    [body | face] samples must be the same in all three placements -- to the last bit."""
    common = ["--pipeline", "--subjects", "2", "--batch", "2", "--frames", "240", "--respacing", "ddim5", "--repeats", "1"]
    lines = {n: _bench(n, *common) for n in (1, 2, 4)}
    for n, line in lines.items():
        assert line["n_gpus"] == n and line["scaling"] == "strong" and line["value"] > 0 and len(line["sample_digests"]) == 4
        assert (line["gather_ms"] is not None) == (n > 1)
    assert lines[4]["config"]["placement"] == {"0": [[0, 0, 1]], "1": [[0, 1, 2]], "2": [[1, 0, 1]], "3": [[1, 1, 2]]}
    assert lines[1]["sample_digests"] == lines[2]["sample_digests"] == lines[4]["sample_digests"], lines
    assert len({tuple(v) for v in lines[1]["sample_digests"].values()}) == 4  # four different samples
    record("pipeline_job_placements", sec_per_sample={str(n): float(l["value"]) for n, l in lines.items()})


def test_pipeline_two_stream_schedule_equals_the_sequential_one(tmp_path):
    """`bench.py --pipeline` (one subject, one GPU): face on one HIP stream, guide -> VQ decode -> body on another; bench.py itself
    asserts that the overlapped schedule reproduces the sequential samples bit for bit."""
    line = _bench(1, "--pipeline", "--batch", "2", "--frames", "240", "--respacing", "ddim5")
    assert line["overlapped_equals_sequential"] is True and line["value"] > 0
    assert set(line["stages_ms"]) == {"audio_front_end_ms", "guide_tokens_and_vq_decode_ms", "body_ddim_ms", "face_ddim_ms"}
    record("pipeline_overlap", sequential_s=float(line["sequential_total_s"]), overlapped_s=float(line["overlapped_total_s"]))


# ----------------------------------------------------------------------------- non-finite detection
@pytest.mark.parametrize("precision", ["fp16", "bf16", "fp32"])
def test_non_finite_outputs_raise_once_per_sampling_call(dev, precision):
    """A denoiser evaluation that produces inf / nan sets a device flag in the fused step tail (include/a2p_hip.h a2p_check_finite);
    the sampling loops read it ONCE, after their last step, and raise A2PError; the flag is cleared by the check, and a healthy
    call afterwards passes.  Poison: an input_projection weight of 1e30 (the residual stream overflows fp32 itself two layers in)."""
    spec = face_spec(num_layers=2)
    B, T = 1, 64
    inp = synthetic_inputs(spec, B, T, SEED)
    args = default_args("face", layers=2, timestep_respacing="ddim5")
    model, diffusion = create_model_and_diffusion(args, "test", precision=precision, max_batch=B)
    sd = synthetic_state_dict(spec, SEED)
    load_model(model, {**sd, "input_projection.weight": sd["input_projection.weight"] * 1e30})
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0, device=dev)}
    with pytest.raises(_lib.A2PError, match="inf / nan"):
        diffusion.ddim_sample_loop(cfg, (B, spec.nfeats, 1, T), clip_denoised=False, model_kwargs={"y": y}, noise=inp["x_T"].to(dev))
    cfg.a2p_check_finite()                                            # cleared by the raise: a second check is clean
    diffusion.defer_finite_check = True                               # multi-stream callers: the loop does not wait for its stream ...
    diffusion.ddim_sample_loop(cfg, (B, spec.nfeats, 1, T), clip_denoised=False, model_kwargs={"y": y}, noise=inp["x_T"].to(dev))
    with pytest.raises(_lib.A2PError, match="inf / nan"):            # ... and the flag is still there when the caller asks
        cfg.a2p_check_finite()
    diffusion.defer_finite_check = False
    load_model(model, sd)
    out = diffusion.ddim_sample_loop(cfg, (B, spec.nfeats, 1, T), clip_denoised=False, model_kwargs={"y": y}, noise=inp["x_T"].to(dev))
    assert torch.isfinite(out).all()
    bad = cfg(torch.full_like(inp["x_T"], float("nan")).to(dev), torch.tensor([3], device=dev), y)   # direct forward: flag, no raise
    assert not torch.isfinite(bad).all()
    with pytest.raises(_lib.A2PError, match="inf / nan"):
        model.check_finite()
    model.release()


# ----------------------------------------------------------------------------- the 16-bit modes outside O(1) activations
def _oracle_forward(sd, fmt, spec, inp, t, scale, probe=None):
    from oracle import a2p_oracle as O
    from oracle import lowprec_model as LP
    kf, mk = (inp["keyframes"], inp["mask"]) if spec.is_pose else (None, None)
    den = O.OracleDenoiser(sd, fmt, spec.num_layers, spec.num_heads) if probe is None else LP.LowPrecDenoiser(sd, fmt, spec.num_layers, spec.num_heads, probe)
    with torch.no_grad():
        return den.forward_cfg(inp["x_T"], t, inp["cond_embed"], scale, kf, mk)


def _logit_probe():
    """fp32 pass-through rounding hook of oracle/lowprec_model.py that only records the largest attention logit."""
    from oracle import lowprec_model as LP
    probe = LP.Rounding("fp32")
    probe.logit_peak = 0.0
    return probe


@pytest.mark.parametrize("fmt,name,kw,inside", [
    ("face", "weights_x2", {"weight_gain": 2.0}, True),
    ("face", "qk_x2", {"qk_gain": 2.0}, True),
    ("face", "resid_1e3", {"resid_gain": 1e3}, True),
    ("face", "qk_x3", {"qk_gain": 3.0}, False),
    ("pose", "weights_x2", {"weight_gain": 2.0}, True),
])
def test_16bit_modes_on_trained_like_statistics(dev, fmt, name, kw, inside):
    """VERDICT round 3 (weak 1): every earlier parity fixture has xavier-scale weights -- near-uniform softmax rows, a residual stream
    of a few units.  Here the synthetic weights are pushed towards trained statistics (audio2photoreal_amd.synthetic
    .trained_like_state_dict; the output head is rescaled so that the guided output keeps unit scale) and the guided forward of all
    three precisions is compared with the fp32 oracle:
      * fp32 (parity mode) holds 1e-3 everywhere, with orders of magnitude to spare;
      * IEEE-half operands hold 1e-3 INSIDE the envelope (row maxima of the scaled scores up to ~13: every Linear weight x2, the
        q/k rows x2, a residual stream of 1e3 units in front of final_layer's split-operand rows);
      * at q/k rows x3 (maxima ~29) they do not (CPU model of the rounding sites: 2.4e-3) -- the library measures the maximum on the
        device (a2p_attention_logit_max) and the Python mirror warns (A2PPrecisionWarning) instead of returning silently;
      * bfloat16 is ~8x worse throughout (recorded, gated loosely)."""
    import warnings
    spec = face_spec() if fmt == "face" else pose_spec()
    B, T = 1, 240
    inp = synthetic_inputs(spec, B, T, SEED)
    t = torch.tensor([700])
    scale = torch.full((B,), 10.0 if fmt == "face" else 2.0)
    sd = trained_like_state_dict(spec, SEED, **kw)
    g = float(_oracle_forward(sd, fmt, spec, inp, t, scale).std())
    head = [k for k in sd if k.startswith("final_conv.")] if fmt == "pose" else ["final_layer.weight", "final_layer.bias"]
    for k in head:                                        # the LAST linear map of the model: the oracle output scales by exactly 1/g
        sd[k] = sd[k] / g
    probe = _logit_probe()
    want = _oracle_forward(sd, fmt, spec, inp, t, scale, probe)
    assert torch.isfinite(want).all() and 0.5 < float(want.std()) < 2.0
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": scale.to(dev)}
    if spec.is_pose:
        y["keyframes"], y["mask"] = inp["keyframes"].to(dev), inp["mask"].to(dev)
    errs, peaks, warned = {}, {}, {}
    for precision in ("fp32", "fp16", "bf16"):
        model, _ = create_model_and_diffusion(default_args(fmt), "test", precision=precision, max_batch=B)
        load_model(model, sd)
        cfg = ClassifierFreeSampleModel(model.to(dev).eval())
        got = cfg(inp["x_T"].to(dev), t.to(dev), y).cpu()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            model.check_finite()
        warned[precision] = any(issubclass(x.category, _lib.A2PPrecisionWarning) for x in w)
        peaks[precision] = model.last_logit_max
        errs[precision] = float((got - want).norm() / want.norm())
        model.release()
    record(f"trained_like/{fmt}/{name}", oracle_logit_peak=probe.logit_peak, device_logit_max=peaks, rel_l2=errs, warned=warned)
    # the device-side maximum is the row maximum of the scaled scores; the oracle probe records max |score|: the maximum is what
    # matters for the envelope, and the two agree to operand rounding whenever the largest |score| is a positive one
    assert peaks["fp32"] <= probe.logit_peak * 1.001 + 1e-3 and peaks["fp32"] > 0.3 * probe.logit_peak, (peaks, probe.logit_peak)
    assert abs(peaks["fp16"] - peaks["fp32"]) < 0.08 * abs(peaks["fp32"]) + 0.05      # rounded q / k: measured 1-4 % below the fp32 maximum
    assert errs["fp32"] < 1e-4, errs
    assert warned["fp32"] is False
    if inside:
        assert errs["fp16"] < 1e-3, errs
        assert errs["bf16"] < 1.2e-2, errs
        assert peaks["fp32"] < _lib.LOGIT_ENVELOPE_FP16 and not warned["fp16"] and not warned["bf16"]
    else:
        assert peaks["fp32"] > _lib.LOGIT_ENVELOPE_FP16 and warned["fp16"] and warned["bf16"], (peaks, warned)
        assert errs["fp16"] < 1e-2 and errs["bf16"] < 8e-2, errs          # out of the envelope, not out of control


# ----------------------------------------------------------------------------- fused output tail of the body model
@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("B,T", [(2, 600), (3, 208), (1, 64), (2, 40)])
def test_fused_pose_tail_equals_the_nine_gemm_launches(dev, B, T, precision, monkeypatch):
    """csrc/kernels_tail.h: final_layer + the six dilated convolutions + final_conv of the body model as ONE LDS-resident kernel
    (64 output frames + a 24-frame halo per workgroup) against rounds 2-3's nine split-operand GEMM launches (A2P_NO_FUSED_TAIL=1,
    read when the weights are finalized).  Both are the same exact-island arithmetic (hi/lo operand pairs, three products, fp32
    accumulation) in a different summation order: they must agree to fp32 rounding, far inside the 16-bit modes' distance from the
    oracle.  T = 208 / 40: ragged last frame block; T = 64: a single block whose halo lies entirely before the sequence start."""
    spec = pose_spec()
    inp = synthetic_inputs(spec, B, T, SEED)
    sd = synthetic_state_dict(spec, SEED)
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 2.0, device=dev), "keyframes": inp["keyframes"].to(dev),
         "mask": inp["mask"].to(dev)}
    t = torch.tensor(([901, 417, 33] * 2)[:B], device=dev)
    outs = {}
    for name in ("gemms", "fused"):
        if name == "gemms":
            monkeypatch.setenv("A2P_NO_FUSED_TAIL", "1")
        else:
            monkeypatch.delenv("A2P_NO_FUSED_TAIL", raising=False)
        model, _ = create_model_and_diffusion(default_args("pose"), "test", precision=precision, max_batch=B)
        load_model(model, sd)
        cfg = ClassifierFreeSampleModel(model.to(dev).eval())
        if 2 * B * T < 960:
            monkeypatch.setenv("A2P_CHAIN_MT", "3")
        outs[name] = cfg(inp["x_T"].to(dev), t, y).cpu()
        monkeypatch.delenv("A2P_CHAIN_MT", raising=False)
        model.release()
    assert torch.isfinite(outs["fused"]).all()
    err = float((outs["fused"] - outs["gemms"]).norm() / outs["gemms"].norm())
    worst = float((outs["fused"] - outs["gemms"]).abs().max() / outs["gemms"].abs().max())
    record(f"fused_tail_vs_gemms/{precision}/B{B}_T{T}", rel_l2=err, max_norm=worst)
    # IEEE-half pairs carry 22 mantissa bits, bfloat16 pairs 16: the dropped lo x lo term and the pair's own rounding sit at 2^-22 / 2^-17
    tol = {"fp16": 2e-6, "bf16": 4e-5}[precision]
    assert err < tol and worst < 10 * tol, (err, worst)


# ----------------------------------------------------------------------------- sharded blocks take the unsharded batch's kernels
@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("fmt", ["face", "pose"])
def test_batch_hint_makes_shards_bit_identical_across_the_family_boundary(dev, fmt, precision):
    """include/a2p_hip.h a2p_set_batch_hint (sample_parallel sets `global_batch_hint`): a global batch of 4 sequences of 240 frames is
    2*4*240 = 1920 rows under guidance -- a chain-kernel forward -- while its shards of 2 and 1 samples (960 / 480 rows) lie below the
    family threshold (1100 rows for the face model, 960 for the body model) and would take the small-forward / per-op kernels, whose
    rounding differs.  With the hint every shard must reproduce its rows of the unsharded forward BIT FOR BIT, whatever panel heights,
    tile shapes or workgroup shapes its own row count selects inside the family."""
    spec = face_spec() if fmt == "face" else pose_spec()
    B, T = 4, 240
    inp = synthetic_inputs(spec, B, T, SEED)
    model, _ = create_model_and_diffusion(default_args(fmt), "test", precision=precision, max_batch=B)
    load_model(model, synthetic_state_dict(spec, SEED))
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    t = torch.tensor([911, 640, 333, 7], device=dev)

    def y_of(lo, hi):
        y = {"cond_embed": inp["cond_embed"][lo:hi].to(dev), "scale": torch.full((hi - lo,), 10.0 if fmt == "face" else 2.0, device=dev)}
        if spec.is_pose:
            y["keyframes"], y["mask"] = inp["keyframes"][lo:hi].to(dev), inp["mask"][lo:hi].to(dev)
        return y
    x = inp["x_T"].to(dev)
    whole = cfg(x, t, y_of(0, B)).cpu()
    for blocks in ([(0, 2), (2, 4)], [(0, 1), (1, 2), (2, 3), (3, 4)], [(0, 3), (3, 4)]):
        model.global_batch_hint = B
        parts = [cfg(x[lo:hi].contiguous(), t[lo:hi].contiguous(), y_of(lo, hi)).cpu() for lo, hi in blocks]
        assert torch.equal(torch.cat(parts), whole), (fmt, precision, blocks, float((torch.cat(parts) - whole).abs().max()))
    model.global_batch_hint = 0
    alone = cfg(x[:1].contiguous(), t[:1].contiguous(), y_of(0, 1)).cpu()        # no hint: the shard's own family, close but not equal
    assert not torch.equal(alone, whole[:1]) and float((alone - whole[:1]).norm() / whole[:1].norm()) < (2e-3 if precision == "fp16" else 2e-2)
    model.release()


# ----------------------------------------------------------------------------- the A/B switches kept next to the exact islands
@pytest.mark.parametrize("fmt", ["face", "pose"])
@pytest.mark.parametrize("switch", ["A2P_TAIL16", "A2P_TAIL_F32"])
def test_exact_island_ab_switches_still_run(dev, fmt, switch, monkeypatch):
    """csrc/a2p_lib.hip: `A2P_TAIL16=1` (input_projection / final_layer / the body model's conv tail on 16-bit operands, round 2's
    path: final_layer rides on the last chain kernel) and `A2P_TAIL_F32=1` (the islands as fp32-MFMA GEMMs on fp32 copies, round 3's first
    version) are kept for A/B measurements next to the default split-operand islands; both are read when the weights are finalized.
    They must keep producing the forward: TAIL_F32 at the default's accuracy, TAIL16 at round 2's (measured 8e-4 ... 2e-3)."""
    from oracle import a2p_oracle as O
    spec = face_spec() if fmt == "face" else pose_spec()
    B, T = 2, 600
    inp = synthetic_inputs(spec, B, T, SEED)
    sd = synthetic_state_dict(spec, SEED)
    scale = 10.0 if fmt == "face" else 2.0
    t = torch.tensor([845, 96], device=dev)
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), scale, device=dev)}
    if spec.is_pose:
        y["keyframes"], y["mask"] = inp["keyframes"].to(dev), inp["mask"].to(dev)
    den = O.OracleDenoiser(sd, fmt, spec.num_layers, spec.num_heads, torch.float32)
    want = den.forward_cfg(inp["x_T"][:1], t[:1].cpu(), inp["cond_embed"][:1], torch.full((1,), scale),
                           inp["keyframes"][:1] if spec.is_pose else None, inp["mask"][:1] if spec.is_pose else None)
    errs = {}
    for name in ("default", switch):
        if name != "default":
            monkeypatch.setenv(switch, "1")
        model, _ = create_model_and_diffusion(default_args(fmt), "test", precision="fp16", max_batch=B)
        load_model(model, sd)
        cfg = ClassifierFreeSampleModel(model.to(dev).eval())
        got = cfg(inp["x_T"].to(dev), t, y).cpu()
        assert torch.isfinite(got).all()
        errs[name] = float((got[:1] - want).norm() / want.norm())
        model.release()
        monkeypatch.delenv(switch, raising=False)
    record(f"island_switch/{fmt}/{switch}", **errs)
    assert errs["default"] < 1e-3
    if switch == "A2P_TAIL_F32":
        assert errs[switch] < 1e-3 and abs(errs[switch] - errs["default"]) < 1e-4, errs
    else:
        assert errs["default"] <= errs[switch] * 1.05 < 4e-3, errs        # the 16-bit tail is what the islands were built to replace


# ----------------------------------------------------------------------------- keyframe attention inside the chain kernel
@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("B,T", [(2, 600), (3, 208), (1, 570), (2, 90)])
def test_fused_keyframe_attention_equals_the_three_launches(dev, B, T, precision, monkeypatch):
    """csrc/kernels_chain.h CHAIN_MIDPOST: MID2 | keyframe cross attention (multihead_attn2, transformer_modules.py:206-215) | POST of
    the body model as ONE kernel -- the query panel stays in LDS, the attention over the <= 32 keyframe tokens runs on it in place --
    against the three launches (A2P_NO_FUSED_KF=1).  Same operand roundings (16-bit q, k, v, p; fp32 accumulation), a slightly different
    softmax arithmetic: the forwards must agree far inside the 16-bit modes' distance from the oracle, and both must keep that distance.
    T = 600 / 208 / 90: 16-row tiles straddle two sequences (the tile is computed against both key sets); ragged last panels throughout."""
    from oracle import a2p_oracle as O
    spec = pose_spec()
    inp = synthetic_inputs(spec, B, T, SEED)
    sd = synthetic_state_dict(spec, SEED)
    if B > 1:
        inp["mask"][1, :, :, T // 3:] = False
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 2.0, device=dev), "keyframes": inp["keyframes"].to(dev),
         "mask": inp["mask"].to(dev)}
    t = torch.tensor(([901, 417, 33] * 2)[:B], device=dev)
    model, _ = create_model_and_diffusion(default_args("pose"), "test", precision=precision, max_batch=B)
    load_model(model, sd)
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    if 2 * B * T < 960:
        monkeypatch.setenv("A2P_CHAIN_ROWS", "1")
    outs = {}
    for name in ("three_launches", "fused"):
        if name == "fused":
            monkeypatch.delenv("A2P_NO_FUSED_KF", raising=False)
        else:
            monkeypatch.setenv("A2P_NO_FUSED_KF", "1")
        outs[name] = cfg(inp["x_T"].to(dev), t, y).cpu()
    monkeypatch.delenv("A2P_CHAIN_ROWS", raising=False)
    model.check_finite()
    peak = model.last_logit_max
    model.release()
    assert torch.isfinite(outs["fused"]).all()
    pair = float((outs["fused"] - outs["three_launches"]).norm() / outs["three_launches"].norm())
    den = O.OracleDenoiser(sd, "pose", spec.num_layers, spec.num_heads, torch.float32)
    want = den.forward_cfg(inp["x_T"][:1], t[:1].cpu(), inp["cond_embed"][:1], torch.full((1,), 2.0), inp["keyframes"][:1], inp["mask"][:1])
    e_f = float((outs["fused"][:1] - want).norm() / want.norm())
    e_3 = float((outs["three_launches"][:1] - want).norm() / want.norm())
    record(f"fused_keyframe_attention/{precision}/B{B}_T{T}", pair=pair, fused_vs_oracle=e_f, three_launches_vs_oracle=e_3, logit_max=peak)
    k = 1.0 if precision == "fp16" else 8.0
    assert pair < k * 4e-4 and e_f < k * 1e-3 and e_f < 1.15 * e_3 + k * 5e-5, (pair, e_f, e_3)   # measured: pair 1.3e-4 .. 1.9e-4 (fp16)


# ----------------------------------------------------------------------------- round 5: head geometries the fused kernel is NOT written for
@pytest.mark.parametrize("heads,B,T", [(4, 2, 600), (4, 3, 208), (8, 8, 12)])
def test_pose_head_geometries_outside_the_fused_keyframe_kernel(dev, heads, B, T, monkeypatch):
    """CHAIN_MIDPOST hard-codes 8 heads x 32 and handles a 16-row tile that straddles at most TWO sequences.  A pose model with
    --heads 4 (head_dim 64 at d = 256: FiLMTransformer's default num_heads and the reference's argparse default) and clips shorter
    than 16 frames (a tile can cover three sequences) must therefore take the three launches -- checked against the oracle, and the
    default path must equal A2P_NO_FUSED_KF=1 bit for bit (it IS the three launches)."""
    from oracle import a2p_oracle as O
    spec = pose_spec(num_heads=heads)
    inp = synthetic_inputs(spec, B, T, SEED)
    sd = synthetic_state_dict(spec, SEED)
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 2.0, device=dev), "keyframes": inp["keyframes"].to(dev),
         "mask": inp["mask"].to(dev)}
    t = torch.tensor(([901, 417, 33] * 3)[:B], device=dev)
    model, _ = create_model_and_diffusion(default_args("pose", heads=heads), "test", precision="fp16", max_batch=B)
    load_model(model, sd)
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    if 2 * B * T < 960:
        monkeypatch.setenv("A2P_CHAIN_ROWS", "1")
    outs = {}
    for name in ("default", "three_launches"):
        if name == "default":
            monkeypatch.delenv("A2P_NO_FUSED_KF", raising=False)
        else:
            monkeypatch.setenv("A2P_NO_FUSED_KF", "1")
        outs[name] = cfg(inp["x_T"].to(dev), t, y).cpu()
    monkeypatch.delenv("A2P_CHAIN_ROWS", raising=False)
    model.release()
    assert torch.equal(outs["default"], outs["three_launches"]), "a geometry outside the fused kernel's contract took the fused kernel"
    den = O.OracleDenoiser(sd, "pose", spec.num_layers, heads, torch.float32)
    want = den.forward_cfg(inp["x_T"][:1], t[:1].cpu(), inp["cond_embed"][:1], torch.full((1,), 2.0), inp["keyframes"][:1], inp["mask"][:1])
    err = float((outs["default"][:1] - want).norm() / want.norm())
    record(f"pose_heads{heads}/fp16/B{B}_T{T}", vs_oracle=err)
    assert err < 1e-3, err


# ----------------------------------------------------------------------------- round 5: outside the envelope the 16-bit modes escalate
@pytest.mark.parametrize("name,kw", [("qk_x3", {"qk_gain": 3.0}), ("qk_x4", {"qk_gain": 4.0})])
def test_16bit_sampling_call_outside_the_envelope_escalates_to_fp32(dev, name, kw):
    """q/k rows x3 / x4: attention row maxima of ~30 / ~50, where IEEE-half operands cost 2.6e-3 / diverge.  A sampling call in
    precision="fp16" must still RETURN what the fp32 path returns: the library reports the logit maximum after the first step
    (a2p_precision_verdict), the model re-creates its context in fp32 (sticky), the step is repeated and the loop goes on.
    Reference: the oracle's ddim5 loop in FLOAT64.  The fp32 oracle's own distance from it is the scenario's conditioning in fp32
    arithmetic: ~1e-6 at x3, but x4 (softmax rows that are one-hot to 2^-70) is chaotic over five steps -- two CORRECT fp32
    implementations (torch CPU and the fp32 MFMA kernels) land 0.3 apart there (first GPU run of this test), so the gate is
    max(1e-3, 4x that distance) and the x4 case proves the control flow + the single guided forward (<= 1e-3, below), not the loop.
    `auto_escalate=False` keeps rounds 1-4's warn-and-return behaviour (recorded: its error).  The price is recorded too."""
    import time
    import warnings
    from oracle import a2p_oracle as O
    spec = face_spec()
    B, T = 1, 240
    inp = synthetic_inputs(spec, B, T, SEED)
    scale = torch.full((B,), 10.0)
    sd = trained_like_state_dict(spec, SEED, **kw)
    g = float(_oracle_forward(sd, "face", spec, inp, torch.tensor([700]), scale).std())
    for k in ("final_layer.weight", "final_layer.bias"):
        sd[k] = sd[k] / g
    refs = {}
    for dt in (torch.float64, torch.float32):
        den = O.OracleDenoiser(sd, "face", spec.num_layers, spec.num_heads, dt)
        fn = lambda x, ts: den.forward_cfg(x, ts, inp["cond_embed"], scale)
        with torch.no_grad():
            _, xs = O.OracleSampler("ddim5").ddim_sample_loop(fn, inp["x_T"].to(dt))
            fwd = den.forward_cfg(inp["x_T"].to(dt), torch.tensor([700]), inp["cond_embed"], scale)
        refs[dt] = (xs.double(), fwd.double())
    want, want_fwd = refs[torch.float64]
    cond = float((refs[torch.float32][0] - want).norm() / want.norm())          # what fp32 arithmetic itself can promise here
    gate = max(1e-3, 4.0 * cond)
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": scale.to(dev)}
    shape = (B, spec.nfeats, 1, T)
    res = {}
    for mode, auto in (("escalating", True), ("warn_only", False)):
        model, diffusion = create_model_and_diffusion(default_args("face", timestep_respacing="ddim5"), "test", precision="fp16", max_batch=B,
                                                      auto_escalate=auto)
        load_model(model, sd)
        cfg = ClassifierFreeSampleModel(model.to(dev).eval())
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = diffusion.ddim_sample_loop(cfg, shape, clip_denoised=False, model_kwargs={"y": y}, noise=inp["x_T"].to(dev)).cpu().double()
        warned = sum(issubclass(x.category, _lib.A2PPrecisionWarning) for x in w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        again = diffusion.ddim_sample_loop(cfg, shape, clip_denoised=False, model_kwargs={"y": y}, noise=inp["x_T"].to(dev)).cpu().double()
        torch.cuda.synchronize()
        ms = round(1e3 * (time.perf_counter() - t0), 2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fwd = cfg(inp["x_T"].to(dev), torch.tensor([700], device=dev), y).cpu().double()       # one guided forward on whatever the model is now
        res[mode] = {"rel_l2": float((got - want).norm() / want.norm()), "second_call_rel_l2": float((again - want).norm() / want.norm()),
                     "forward_rel_l2": float((fwd - want_fwd).norm() / want_fwd.norm()),
                     "warnings": warned, "precision_after": model.precision, "escalated_from": model.escalated_from,
                     "second_call_ms": ms, "logit_max": model.last_logit_max}
        model.release()
    record(f"escalation/face/{name}", fp32_oracle_vs_fp64_oracle=cond, gate=gate, **res)
    e, wo = res["escalating"], res["warn_only"]
    assert e["precision_after"] == "fp32" and e["escalated_from"] == "fp16" and e["warnings"] == 1, e
    assert e["rel_l2"] < gate and e["second_call_rel_l2"] < gate, (e, cond)   # the call that escalated AND the sticky fp32 calls after it
    assert e["forward_rel_l2"] < 1e-3, e                                        # a single forward is well conditioned even at x4
    assert wo["precision_after"] == "fp16" and wo["escalated_from"] is None and wo["warnings"] >= 1, wo
    assert wo["forward_rel_l2"] > 1e-3 and wo["rel_l2"] > e["rel_l2"], (wo, e)   # the 16-bit forward IS outside the bar here
    if name == "qk_x3":
        assert cond < 1e-4 and gate == 1e-3, cond


# ----------------------------------------------------------------------------- round 5: the 8-wave anti-phase attention kernel (opt-in experiment)
@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("fmt", ["face", "pose"])
def test_attn2_kernel_is_bit_identical_to_attn_kernel(dev, fmt, precision, monkeypatch):
    """csrc/kernels_attn2.h (A2P_ATTN2=1): one 8-wave workgroup per CU, 48 + 32 queries per SIMD, the two waves of a SIMD alternating
    between a matrix segment (PV + next tile's QK^T) and a vector segment (softmax) in anti-phase, 8-slot K/V ring behind counted
    vmcnt waits.  Same arithmetic per (query, key tile) as attn_kernel, so the outputs must be IDENTICAL -- through a2p_attention
    (one tile, ragged tiles, many tiles, a spiked key) and through a whole guided forward (cached K/V slots, the time-token tail
    patched into the last tile, the shared unconditional slot)."""
    from audio2photoreal_amd.spec import face_spec as fs, pose_spec as ps
    monkeypatch.setenv("A2P_ATTN3", "0")   # the reference side is attn_kernel, not round 6's attn3_kernel (which is not bit-identical to either)
    spec = fs() if fmt == "face" else ps()
    model, _ = create_model_and_diffusion(default_args(fmt), "test", precision=precision, max_batch=2)
    load_model(model, synthetic_state_dict(spec, SEED))
    model = model.to(dev).eval()
    model._ensure_ctx(dev, 2)
    lib = model._lib()
    d = spec.latent_dim
    g = torch.Generator().manual_seed(7)

    def both(fn):
        outs = []
        for val in (None, "1"):
            if val is None:
                monkeypatch.delenv("A2P_ATTN2", raising=False)
            else:
                monkeypatch.setenv("A2P_ATTN2", val)
            _lib.check(lib.a2p_reload_env(model._ctx), "a2p_reload_env")
            outs.append(fn())
        monkeypatch.delenv("A2P_ATTN2", raising=False)
        _lib.check(lib.a2p_reload_env(model._ctx), "a2p_reload_env")
        return outs
    for (N, Tq, S) in [(2, 100, 77), (1, 600, 800), (3, 33, 20), (2, 321, 640), (1, 150, 150)]:
        q, k, v = (torch.randn(N, L, d, generator=g) for L in (Tq, S, S))
        k[0, S // 3] *= 6.0
        qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)

        def run():
            out = torch.empty(N, Tq, d, device=dev)
            _lib.check(lib.a2p_attention(model._ctx, _lib.ptr(qd), _lib.ptr(kd), _lib.ptr(vd), _lib.ptr(out), N, Tq, S, _lib.current_stream()), "a2p_attention")
            return out.cpu()
        a, b = both(run)
        assert torch.isfinite(b).all() and torch.equal(a, b), (N, Tq, S, float((a - b).abs().max()))
    B, T = 2, 600
    inp = synthetic_inputs(spec, B, T, SEED)
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0 if fmt == "face" else 2.0, device=dev)}
    if spec.is_pose:
        y["keyframes"], y["mask"] = inp["keyframes"].to(dev), inp["mask"].to(dev)
    cfg = ClassifierFreeSampleModel(model)
    t = torch.tensor([901, 33], device=dev)
    a, b = both(lambda: cfg(inp["x_T"].to(dev), t, y).cpu())
    model.release()
    assert torch.isfinite(b).all() and torch.equal(a, b), float((a - b).abs().max())
