"""Round 6 GPU tests: the BENCHMARKED batch sizes directly against the oracle ("What's weak" 4: every oracle
comparison stopped at B <= 2 and the tall chain kernels / the B >= 3 K/V-slot arithmetic were proven only by bit-identity to another
HIP kernel), and `python bench.py --gpus N` starting its own ranks (the form the driver uses)."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest
import torch

from audio2photoreal_amd import _lib
from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
from audio2photoreal_amd.spec import face_spec, pose_spec
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict
from conftest import ROOT, record, rel_l2, rel_max

pytestmark = pytest.mark.gpu
SEED = 10


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _debug_i64(model, name):
    n = C.c_int64(0)
    _lib.check(model._lib().a2p_debug_read(model._ctx, name, C.byref(n), 8), "a2p_debug_read")
    return int(n.value)


def _worst_sample(got, want):
    return max(rel_l2(got[b], want[b]) for b in range(got.shape[0]))


# ----------------------------------------------------------------------------- face, B = 32 and B = 8: the tall kernels vs the oracle
@pytest.mark.parametrize("B,rows", [(32, 80), (8, 48)])
def test_face_guided_forward_at_the_benchmarked_batch_vs_oracle_tall_kernels(dev, B, rows, monkeypatch):
    """One guided forward of the face model at T=600, S=1998+2 with B = 32 (north_star's roofline batch: 80-row panels) and B = 8
    (the headline batch: 48-row panels) through the tall chain kernels (A2P_CHAIN_V=4 forced, 16 tall launches asserted) against the
    oracle's forward of the SAME batch (model/cfg_sampler.py:30-33 over model/diffusion.py:338-403): fp16 <= 1e-3, fp32 <= 1e-5
    (rel-L2 over the batch AND for the worst single sample: a wrong K/V slot or panel of one sample cannot hide in the norm)."""
    from oracle import a2p_oracle as O
    T = 600
    spec = face_spec()
    sd = synthetic_state_dict(spec, SEED)
    inp = synthetic_inputs(spec, B, T, SEED)
    times = torch.tensor(([901, 417, 33, 650, 999, 0, 250, 777] * 4)[:B])
    den = O.OracleDenoiser(sd, "face", spec.num_layers, spec.num_heads)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    with torch.no_grad():
        want = torch.cat([den.forward_cfg(inp["x_T"][b:b + 8], times[b:b + 8], inp["cond_embed"][b:b + 8], torch.full((min(8, B - b),), 10.0))
                          for b in range(0, B, 8)])          # 8 samples at a time: the CPU attention scores of 32 would be 2.5 GB per pass
    monkeypatch.setenv("A2P_CHAIN_V", "4")
    for precision, tol in (("fp16", 1e-3), ("fp32", 1e-5)):
        model, _ = create_model_and_diffusion(default_args("face"), "test", precision=precision, max_batch=B)
        load_model(model, sd)
        cfg = ClassifierFreeSampleModel(model.to(dev).eval())
        y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0, device=dev)}
        x, t = inp["x_T"].to(dev), times.to(dev)
        cfg(x, t, y)                                          # context + hoisted conditioning
        before, before_fin = _debug_i64(model, b"chain4_launches"), _debug_i64(model, b"final_fused_launches")
        got = cfg(x, t, y).cpu()
        launched = _debug_i64(model, b"chain4_launches") - before
        fused_final = _debug_i64(model, b"final_fused_launches") - before_fin
        model.check_finite()
        model.release()
        e = {"rel_l2": rel_l2(got, want), "max_norm": rel_max(got, want), "worst_sample_rel_l2": _worst_sample(got, want)}
        record(f"oracle_at_bench_batch/face_B{B}_{rows}row/{precision}", tall_launches=launched, **e)
        if precision != "fp32":                               # fp32 parity mode runs the per-op exact-fp32 kernels, not the chain kernels
            assert launched == 17, launched                   # 8 MID + 8 POST + the input projection / PRE kernel of layer 0
            assert fused_final == 1, fused_final             # the last POST kernel computes final_layer too (split-operand island inside the kernel)
        assert e["rel_l2"] < tol and e["worst_sample_rel_l2"] < tol * (1.0 if precision == "fp32" else 1.5), e


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("B,T,mt", [(8, 600, 0), (3, 208, 4), (2, 88, 3), (16, 600, 0), (2, 328, 5)])
def test_final_layer_inside_the_last_post_kernel_is_bit_identical(dev, B, T, mt, precision, monkeypatch):
    """model/diffusion.py:397 (final_layer) as a split-operand exact island INSIDE the last decoder layer's tall POST kernel (chain4_kernel<MT, POST, 2>: the rows never
    leave the registers, hi / lo panels in LDS, [W_hi | W_hi | W_lo] on the weight stream) against the launches it replaces (the rows stored, split3_kernel, gemm_kernel;
    A2P_NO_FUSED_FINAL=1): accumulators from zero, the same k order, the bias last -- the SAME BITS, for 48-, 64- and 80-row panels (80 rows: the lo pieces go through
    LDS half of K at a time), ragged last panels and panels that straddle two sequences.  (The kernel family of a forward is chosen per box; it must stay invisible in the results.)"""
    spec = face_spec()
    inp = synthetic_inputs(spec, B, T, SEED)
    model, _ = create_model_and_diffusion(default_args("face"), "test", precision=precision, max_batch=B)
    load_model(model, synthetic_state_dict(spec, SEED))
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0, device=dev)}
    t = torch.tensor(([901, 417, 33, 650] * 4)[:B], device=dev)
    monkeypatch.setenv("A2P_CHAIN_V", "4")
    if mt:
        monkeypatch.setenv("A2P_CHAIN_MT", str(mt))
    if 2 * B * T < 1100:
        monkeypatch.setenv("A2P_CHAIN_ROWS", "1")
    outs, fused = {}, {}
    for name, flag in (("fused", None), ("launches", "1")):
        if flag:
            monkeypatch.setenv("A2P_NO_FUSED_FINAL", flag)
        before = _debug_i64(model, b"final_fused_launches") if model._ctx is not None else 0
        outs[name] = cfg(inp["x_T"].to(dev), t, y).cpu()
        fused[name] = _debug_i64(model, b"final_fused_launches") - before
    for k in ("A2P_CHAIN_V", "A2P_CHAIN_MT", "A2P_CHAIN_ROWS", "A2P_NO_FUSED_FINAL"):
        monkeypatch.delenv(k, raising=False)
    model.check_finite()
    model.release()
    assert fused == {"fused": 1, "launches": 0}, fused
    assert torch.isfinite(outs["fused"]).all()
    diff = float((outs["fused"] - outs["launches"]).abs().max())
    record(f"fused_final_vs_launches/B{B}_T{T}_mt{mt}/{precision}", max_abs_diff=diff)
    assert diff == 0.0, diff


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("B,T,mt,guided", [(8, 600, 0, True), (3, 208, 4, True), (2, 88, 3, True), (16, 600, 0, True), (2, 328, 5, True), (3, 600, 0, False)])
def test_input_projection_inside_the_first_chain_kernel_is_bit_identical(dev, B, T, mt, guided, precision, monkeypatch):
    """model/diffusion.py:345-346,364 (permute + input_projection) and layer 0's PRE work (norm1 -> rotary -> [Q|K], V^T) as ONE tall kernel (chain4_kernel<MT, CHAIN_IN>: the
    noisy input is read in its [B, C, T] layout, split into hi / lo 16-bit panels in LDS, projected as a split-operand island in gemm_kernel's k order, the rows stored once and
    normalised from registers) against the three launches it replaces (pack_input_split3_kernel, gemm_kernel, the gen-1 PRE kernel; A2P_NO_FUSED_IN=1): the SAME BITS for 48- /
    64- / 80-row panels, ragged last panels, panels that straddle two samples, with guidance (shared layer-0 half) and without (one conditional pass)."""
    spec = face_spec()
    inp = synthetic_inputs(spec, B, T, SEED)
    model, _ = create_model_and_diffusion(default_args("face"), "test", precision=precision, max_batch=B)
    load_model(model, synthetic_state_dict(spec, SEED))
    model = model.to(dev).eval()
    fwd = ClassifierFreeSampleModel(model) if guided else model
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0, device=dev)}
    t = torch.tensor(([901, 417, 33, 650] * 4)[:B], device=dev)
    monkeypatch.setenv("A2P_CHAIN_V", "4")
    if mt:
        monkeypatch.setenv("A2P_CHAIN_MT", str(mt))
    if (2 if guided else 1) * B * T < 1100:
        monkeypatch.setenv("A2P_CHAIN_ROWS", "1")
    outs, fused = {}, {}
    for name, flag in (("fused", None), ("launches", "1")):
        if flag:
            monkeypatch.setenv("A2P_NO_FUSED_IN", flag)
        before = _debug_i64(model, b"chain_in_launches") if model._ctx is not None else 0
        outs[name] = fwd(inp["x_T"].to(dev), t, y).cpu()
        fused[name] = _debug_i64(model, b"chain_in_launches") - before
    for k in ("A2P_CHAIN_V", "A2P_CHAIN_MT", "A2P_CHAIN_ROWS", "A2P_NO_FUSED_IN"):
        monkeypatch.delenv(k, raising=False)
    model.check_finite()
    model.release()
    assert fused == {"fused": 1, "launches": 0}, fused
    assert torch.isfinite(outs["fused"]).all()
    diff = float((outs["fused"] - outs["launches"]).abs().max())
    record(f"fused_input_vs_launches/B{B}_T{T}_mt{mt}_{'cfg' if guided else 'cond'}/{precision}", max_abs_diff=diff)
    assert diff == 0.0, diff


@pytest.mark.parametrize("fmt", ["face", "pose"])
def test_time_mlp_table_gives_the_bits_of_the_per_forward_time_mlp(dev, fmt, monkeypatch):
    """model/diffusion.py:349-353 (time_mlp -> to_time_cond / to_time_tokens) depends on the timestep VALUE alone: a2p_finalize_weights tabulates it for t = 0 .. 999 with the kernels
    that otherwise run in every forward, and tpath_post_kernel picks sample b's row by t[b].  Same bits as the three launches (A2P_TIME_TABLE=0), for mixed timesteps in one batch;
    a timestep outside the table is REPORTED by check_finite (never silently clamped)."""
    from audio2photoreal_amd._lib import A2PError
    spec = face_spec() if fmt == "face" else pose_spec()
    B, T = 3, 208
    inp = synthetic_inputs(spec, B, T, SEED)
    model, _ = create_model_and_diffusion(default_args(fmt), "test", precision="fp16", max_batch=B)
    load_model(model, synthetic_state_dict(spec, SEED))
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0 if fmt == "face" else 2.0, device=dev)}
    if spec.is_pose:
        y["keyframes"], y["mask"] = inp["keyframes"].to(dev), inp["mask"].to(dev)
    t = torch.tensor([999, 0, 417], device=dev)
    x = inp["x_T"].to(dev)
    table = cfg(x, t, y).cpu()
    monkeypatch.setenv("A2P_TIME_TABLE", "0")
    computed = cfg(x, t, y).cpu()
    monkeypatch.delenv("A2P_TIME_TABLE", raising=False)
    model.check_finite()
    assert torch.isfinite(table).all() and torch.equal(table, computed), float((table - computed).abs().max())
    cfg(x, torch.tensor([1000, 3, 4], device=dev), y)
    with pytest.raises(A2PError, match="timestep outside"):
        model.check_finite()
    model.release()


def test_last_layer_takes_the_tall_fused_kernel_under_the_mixed_family(dev, monkeypatch):
    """On the slow GPU type the in-situ calibration keeps the gen-1 POST kernels (family 41: tall MID, gen-1 POST).  The LAST layer's POST kernel is tall regardless -- with
    final_layer inside it replaces three launches -- and the result is the same bits as the all-tall forward: 8 tall MID + 1 tall POST launches, one fused final_layer."""
    B, T = 8, 600
    spec = face_spec()
    inp = synthetic_inputs(spec, B, T, SEED)
    model, _ = create_model_and_diffusion(default_args("face"), "test", precision="fp16", max_batch=B)
    load_model(model, synthetic_state_dict(spec, SEED))
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0, device=dev)}
    t = torch.tensor(([901, 417, 33, 650] * 4)[:B], device=dev)
    outs, tall, fused = {}, {}, {}
    for v in ("4", "41"):
        monkeypatch.setenv("A2P_CHAIN_V", v)
        b0 = (_debug_i64(model, b"chain4_launches"), _debug_i64(model, b"final_fused_launches")) if model._ctx is not None else (0, 0)
        outs[v] = cfg(inp["x_T"].to(dev), t, y).cpu()
        tall[v] = _debug_i64(model, b"chain4_launches") - b0[0]
        fused[v] = _debug_i64(model, b"final_fused_launches") - b0[1]
    monkeypatch.delenv("A2P_CHAIN_V", raising=False)
    model.check_finite()
    model.release()
    assert tall == {"4": 17, "41": 10} and fused == {"4": 1, "41": 1}, (tall, fused)   # (8 MID + 8 | 1 POST + the input / PRE kernel of layer 0)
    assert torch.equal(outs["4"], outs["41"])


# ----------------------------------------------------------------------------- body, B = 16 (BASELINE configs[2])
def test_body_B16_T600_two_ddim_steps_vs_oracle(dev):
    """BASELINE configs[2] at its own batch: body model, keyframe conditioning, CFG scale 2, B = 16, T = 600 -- the first two steps of
    the ddim100 chain (diffusion/gaussian_diffusion.py:667-718 over model/cfg_sampler.py:30-33) chained on each side's own outputs,
    against the oracle: fp16 <= 1e-3, fp32 <= 2e-5 on `sample` and `pred_xstart` of both steps, and for the worst single sample."""
    from oracle import a2p_oracle as O
    B, T = 16, 600
    spec = pose_spec()
    sd = synthetic_state_dict(spec, SEED)
    inp = synthetic_inputs(spec, B, T, SEED)
    den = O.OracleDenoiser(sd, "pose", spec.num_layers, spec.num_heads)
    smp = O.OracleSampler("ddim100")
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    fn = lambda x, ts: den.forward_cfg(x, ts, inp["cond_embed"], torch.full((B,), 2.0), inp["keyframes"].clone(), inp["mask"])
    want = []
    with torch.no_grad():
        cur = inp["x_T"]
        for i in (99, 98):
            out = smp.ddim_sample(fn, cur, torch.full((B,), i))
            want.append(out)
            cur = out["sample"]
    for precision, tol in (("fp16", 1e-3), ("fp32", 2e-5)):
        model, diffusion = create_model_and_diffusion(default_args("pose", timestep_respacing="ddim100"), "test", precision=precision, max_batch=B)
        load_model(model, sd)
        cfg = ClassifierFreeSampleModel(model.to(dev).eval())
        y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 2.0, device=dev),
             "keyframes": inp["keyframes"].clone().to(dev), "mask": inp["mask"].clone().to(dev)}
        idx = diffusion._step_index_tensor(dev, B)
        e = {}
        with torch.no_grad():
            cur = inp["x_T"].to(dev)
            for k, i in enumerate((99, 98)):
                out = diffusion.ddim_sample(cfg, cur, idx[i], clip_denoised=False, model_kwargs={"y": y})
                cur = out["sample"]
                for key in ("sample", "pred_xstart"):
                    e[f"step{k}_{key}"] = rel_l2(out[key].cpu(), want[k][key])
            e["worst_sample_rel_l2"] = _worst_sample(out["sample"].cpu(), want[-1]["sample"])
        model.check_finite()
        model.release()
        record(f"oracle_at_bench_batch/body_B16_ddim100_2steps/{precision}", **e)
        assert max(e.values()) < tol * (1.0 if precision == "fp32" else 1.5) and e["step1_sample"] < tol, e


# ----------------------------------------------------------------------------- `python bench.py --gpus 2`, started the way the driver starts N = 1
def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun in front of it ("What's weak" 3: the assert on WORLD_SIZE made that
    command die): bench.py re-executes itself under torch.distributed.run with one process per GPU.  Here the two ranks share the one
    GPU of the test box and the three collectives run over gloo; rank 0's JSON line arrives on the launcher's stdout."""
    env = dict(os.environ, A2P_BENCH_SHARE_GPU="1", A2P_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--repeats", "1", "--batch", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["collectives"]["ranks"] == 2 and line["config"]["global_batch"] == 4
    assert line["gather_ms"] is not None and line["value"] > 0 and line["steps"] == 4
    record("bench_gpus2_self_launch", value=float(line["value"]), gather_ms=float(line["gather_ms"]))


# ----------------------------------------------------------------------------- a 16-bit OVERFLOW escalates too
def test_fp16_overflow_escalates_to_fp32_instead_of_raising(dev):
    """A checkpoint whose feed-forward hidden activations exceed IEEE half's range (layer 0: linear1 x 1e5, linear2 x 1e-5 -- harmless in
    fp32) makes the fp16 mode produce inf / nan.  That is the hardest way of leaving the 16-bit envelope, and with auto_escalate the caller
    must get the fp32 answer, not an A2PError: check_finite() absorbs the non-finite flag, moves the model to fp32 (sticky, one
    A2PPrecisionWarning) and the loop repeats the call.  The result is compared with the oracle on the same checkpoint."""
    from oracle import a2p_oracle as O
    spec = face_spec(num_layers=2)
    B, T = 1, 64
    inp = synthetic_inputs(spec, B, T, SEED)
    sd = dict(synthetic_state_dict(spec, SEED))
    p = "seqTransDecoder.stack.0."
    sd[p + "linear1.weight"] = sd[p + "linear1.weight"] * 1e5
    sd[p + "linear1.bias"] = sd[p + "linear1.bias"] * 1e5
    sd[p + "linear2.weight"] = sd[p + "linear2.weight"] * 1e-5
    args = default_args("face", layers=2, timestep_respacing="ddim5")
    model, diffusion = create_model_and_diffusion(args, "test", precision="fp16", max_batch=B)
    load_model(model, sd)
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0, device=dev)}
    with pytest.warns(_lib.A2PPrecisionWarning, match="inf / nan"):
        got = diffusion.ddim_sample_loop(cfg, (B, spec.nfeats, 1, T), clip_denoised=False, model_kwargs={"y": y}, noise=inp["x_T"].to(dev)).cpu()
    assert model.precision == "fp32" and model.escalated_from == "fp16"
    assert torch.isfinite(got).all()
    den = O.OracleDenoiser(sd, "face", spec.num_layers, spec.num_heads)
    with torch.no_grad():
        fn = lambda x, ts: den.forward_cfg(x, ts, inp["cond_embed"], torch.full((B,), 10.0))
        want, _ = O.OracleSampler("ddim5").ddim_sample_loop(fn, inp["x_T"])
    e = rel_l2(got, want)
    record("fp16_overflow_escalation/ddim5", rel_l2=e)
    assert e < 1e-3, e
    model.release()


# ----------------------------------------------------------------------------- attn3_kernel: the ISA-level attention tile body
def _attention_fp64(q, k, v, heads):
    """softmax(q k^T / sqrt(dh)) v per head in float64 (nn.MultiheadAttention's core, transformer_modules.py:239-246, no projections)."""
    N, Tq, d = q.shape
    dh = d // heads
    qh, kh, vh = (x.double().view(N, -1, heads, dh).transpose(1, 2) for x in (q, k, v))
    w = torch.softmax(qh @ kh.transpose(-1, -2) / dh ** 0.5, dim=-1)
    return (w @ vh).transpose(1, 2).reshape(N, Tq, d)


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("fmt", ["face", "pose"])
def test_attn3_kernel_vs_fp64_and_vs_attn_kernel(dev, fmt, precision, monkeypatch):
    """csrc/kernels_attn3.h through a2p_attention (A2P_ATTN3=2: wherever legal; 0: attn_kernel): one wave per SIMD, 80 queries per
    wave, asm tile body on owned registers, Q pre-scaled, the softmax reference carried LAZILY through the MFMA C operand.  Not
    bit-identical to attn_kernel (different reference values, one more rounding of Q): both are compared with a float64 attention,
    and with each other.  Shapes: one partial tile, full tiles only (S = 128: no drain), many tiles + partial, ragged query blocks
    (321 = one workgroup + 1 query), 600 x 2000 (the benchmarked cross shape); `spike` multiplies one key row by 6 -- scores jump
    by far more than the lazy window (8 in log2 units) in the MIDDLE of the key range, so the fix-up path (move_refs + the recomputed
    score tile) runs -- and `neg` shifts all scores far below zero (a reference that must go DOWN from its initial 0)."""
    spec = face_spec() if fmt == "face" else pose_spec()
    model, _ = create_model_and_diffusion(default_args(fmt), "test", precision=precision, max_batch=2)
    load_model(model, synthetic_state_dict(spec, SEED))
    model = model.to(dev).eval()
    model._ensure_ctx(dev, 2)
    lib = model._lib()
    d, H = spec.latent_dim, spec.num_heads
    g = torch.Generator().manual_seed(11)
    tol = 1.0e-3 if precision == "fp16" else 6.0e-3

    def run(mode, qd, kd, vd, N, Tq, S):
        monkeypatch.setenv("A2P_ATTN3", mode)
        _lib.check(lib.a2p_reload_env(model._ctx), "a2p_reload_env")
        before = _debug_i64(model, b"attn3_launches")
        out = torch.empty(N, Tq, d, device=dev)
        _lib.check(lib.a2p_attention(model._ctx, _lib.ptr(qd), _lib.ptr(kd), _lib.ptr(vd), _lib.ptr(out), N, Tq, S, _lib.current_stream()), "a2p_attention")
        return out.cpu(), _debug_i64(model, b"attn3_launches") - before
    worst = 0.0
    for (N, Tq, S, kind) in [(2, 100, 77, "plain"), (3, 33, 20, "plain"), (2, 321, 128, "plain"), (1, 600, 800, "spike"), (2, 150, 150, "neg"),
                             (1, 600, 2000, "spike"), (2, 640, 254, "plain"), (1, 80, 1000, "spike"),
                             # round 6, v6 (the last tile is an ordinary step whose padding keys are masked in the ones fragment of the row-sum MFMAs): the reference has
                             # to move ON the last tile (its own vote behind the loop), on a partial and on a full last tile; and a spike on KEY 0 of a partial last tile --
                             # the key whose row is copied into the padding rows, so the padding scores are as large as the spike and only the mask keeps them out of l
                             (1, 600, 2000, "spike_last"), (2, 320, 256, "spike_last"), (1, 600, 600, "spike_last_key0"), (2, 100, 77, "spike_last_key0")]:
        q, k, v = (torch.randn(N, L, d, generator=g) for L in (Tq, S, S))
        if kind == "spike":
            k[0, S // 3] *= 6.0
            k[0, (2 * S) // 3] *= -5.0
        if kind == "spike_last":
            k[:, S - 1] *= 6.0
        if kind == "spike_last_key0":
            k[:, ((S - 1) // 64) * 64] *= 6.0
        if kind == "neg":
            q[:] = q.abs() * 1.5
            k[:] = -k.abs() * 1.5
        want = _attention_fp64(q, k, v, H)
        qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
        old, n_old = run("0", qd, kd, vd, N, Tq, S)
        new, n_new = run("2", qd, kd, vd, N, Tq, S)
        again, _ = run("2", qd, kd, vd, N, Tq, S)
        e_new, e_old, e_pair = rel_l2(new, want), rel_l2(old, want), rel_l2(new, old)
        worst = max(worst, e_new)
        record(f"attn3/{fmt}/{precision}/N{N}_T{Tq}_S{S}_{kind}", vs_fp64=e_new, attn_kernel_vs_fp64=e_old, vs_attn_kernel=e_pair)
        assert (n_old, n_new) == (0, 1), (n_old, n_new)
        assert torch.isfinite(new).all() and torch.equal(new, again), "attn3 must be deterministic"
        assert e_new < tol and e_new < 1.5 * e_old + 1e-4, (N, Tq, S, kind, e_new, e_old)
    monkeypatch.delenv("A2P_ATTN3", raising=False)
    _lib.check(lib.a2p_reload_env(model._ctx), "a2p_reload_env")
    model.release()


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_guided_forward_with_attn3_matches_attn_kernel_path(dev, precision, monkeypatch):
    """A whole guided forward of the face model (B = 4, T = 600: cached K/V slots, the shared unconditional slot, the two time tokens
    patched into the LAST, PARTIAL key tile) with attn3_kernel on (the default rule picks it for self and cross attention from 8 sequences
    on -- below, attn_kernel's 128-query workgroups are one round and faster, profiles/r06_attn3_small_batch.txt: 15 launches asserted at
    B = 4 -- layer 0's self attention runs on the 4 shared sequences --, none at B = 2) and off: the two paths must agree to the 16-bit operand rounding (they are not bit-identical, see above)."""
    spec = face_spec()
    B, T = 4, 600
    inp = synthetic_inputs(spec, B, T, SEED)
    model, _ = create_model_and_diffusion(default_args("face"), "test", precision=precision, max_batch=B)
    load_model(model, synthetic_state_dict(spec, SEED))
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0, device=dev)}
    t = torch.tensor([901, 33, 500, 7], device=dev)
    outs, launches = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("A2P_ATTN3", mode)
        cfg(inp["x_T"].to(dev), t, y)
        before = _debug_i64(model, b"attn3_launches")
        outs[mode] = cfg(inp["x_T"].to(dev), t, y).cpu()
        launches[mode] = _debug_i64(model, b"attn3_launches") - before
    monkeypatch.delenv("A2P_ATTN3", raising=False)
    model.check_finite()
    model.release()
    e = rel_l2(outs["1"], outs["0"])
    record(f"attn3/guided_forward/{precision}", vs_attn_kernel_path=e, launches=launches["1"])
    assert launches == {"0": 0, "1": 15}, launches      # 8 layers x (self + cross) minus layer 0's self attention: the shared-half trick runs it on 4 sequences (attn_kernel's round)
    assert e < (6e-4 if precision == "fp16" else 5e-3), e
    # the rule at B = 2 (4 sequences: attn_kernel's grid of 160 workgroups is one round): no attn3 launch by default
    model2, _ = create_model_and_diffusion(default_args("face"), "test", precision=precision, max_batch=2)
    load_model(model2, synthetic_state_dict(spec, SEED))
    cfg2 = ClassifierFreeSampleModel(model2.to(dev).eval())
    y2 = {"cond_embed": inp["cond_embed"][:2].to(dev), "scale": torch.full((2,), 10.0, device=dev)}
    cfg2(inp["x_T"][:2].to(dev), t[:2], y2)
    assert _debug_i64(model2, b"attn3_launches") == 0
    model2.release()


# ----------------------------------------------------------------------------- the FULL chains of configs[1] and configs[2] against the REFERENCE's own chains
# gates: fp32 ~10x the measured error; 16-bit min(1e-3, 2 x measured) (profiles/r06_chain_vs_oracle*.json); bf16 is recorded, its gate is 2 x measured (it misses 1e-3)
_CHAIN_GATES = {("face", "fp32"): 1e-4, ("face", "fp16"): 8.5e-4, ("face", "bf16"): 7e-3,
                ("body", "fp32"): 1e-4, ("body", "fp16"): 7.5e-4, ("body", "bf16"): 6e-3}   # measured: face 7.3e-6 / 4.1e-4 / 3.4e-3, body 1.2e-6 / 3.6e-4 / 3.0e-3


@pytest.mark.parametrize("workload,precision", sorted(_CHAIN_GATES))
def test_full_sampling_chain_vs_reference_states(dev, workload, precision):
    """The WHOLE chain of the benchmarked workloads -- face: 1000 DDPM steps (gaussian_diffusion.py:434-477, :525-607), body: ddim100 with keyframe conditioning
    (:667-779) -- at B=1, T=600, S=1998+2 through the product, against the chain states the REFERENCE ITSELF reached under identical weights / x_T / conditioning /
    per-step noise (tests/golden/golden_chain_<workload>_ref_v1.npz, produced by tests/golden/make_golden_chain.py from /root/reference: 25 CPU-minutes for the face
    chain) and, for the record, against the oracle's states (golden_chain_<workload>_v1.npz; tests/test_oracle_golden.py holds the two together).  Checked at every saved
    step, gated on the last (the loop's return value: the north_star's 1e-3 applies to fp16)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import chain_vs_oracle as CVO
    want = np.load(CVO.golden_path(workload).replace("_v1.npz", "_ref_v1.npz"))
    orc = np.load(CVO.golden_path(workload))
    states, seconds, ran_as = CVO.run_product(workload, precision, dev)
    assert ran_as == precision
    errs, errs_oracle = {}, {}
    for n, got in sorted(states.items()):
        assert torch.isfinite(got).all()
        errs[n] = rel_l2(got, torch.from_numpy(want[f"step{n}"]))
        errs_oracle[n] = rel_l2(got, torch.from_numpy(orc[f"step{n}"]))
    last = max(errs)
    record(f"chain_vs_reference/{workload}/{precision}", steps=last, gpu_seconds=round(seconds, 2), **{f"rel_l2_step{n}": e for n, e in errs.items()},
           **{f"vs_oracle_step{n}": e for n, e in errs_oracle.items()})
    assert last == CVO.WORKLOADS[workload]["steps"]
    assert errs[last] <= _CHAIN_GATES[(workload, precision)], (workload, precision, errs)
