"""Parity of the HIP path (through the C ABI) against the golden vectors produced by the
reference and against the pinned oracle.  Needs a real MI355X: `pytest -m gpu`.

Tolerances: fp32 mode <= 1e-3 relative (north_star; observed ~1e-6..1e-5); the 16-bit throughput modes ("fp16" = the
benchmarked one, "bf16") are gated at about 2x their measured errors, which the tests also write to gpurun_out/parity_tests.json.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from audio2photoreal_amd import _lib
from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
from audio2photoreal_amd.spec import face_spec, pose_spec
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict, synthetic_tensor
from conftest import record, rel_l2, rel_max

pytestmark = pytest.mark.gpu
SEED = 10


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


_MODELS = {}


def get_model(fmt, precision, dev, respacing="ddim10"):
    key = (fmt, precision)
    if key not in _MODELS:
        spec = face_spec() if fmt == "face" else pose_spec()
        args = default_args(fmt, timestep_respacing=respacing)
        model, _ = create_model_and_diffusion(args, "test", precision=precision, max_batch=4)
        load_model(model, synthetic_state_dict(spec, SEED))
        _MODELS[key] = (spec, model.to(dev).eval())
    return _MODELS[key]


def make_diffusion(fmt, respacing):
    from audio2photoreal_amd.model_util import create_gaussian_diffusion
    return create_gaussian_diffusion(default_args(fmt, timestep_respacing=respacing))


def y_for(spec, inp, dev, scale):
    B = inp["x_T"].shape[0]
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), scale, device=dev)}
    if spec.is_pose:
        y["keyframes"] = inp["keyframes"].clone().to(dev)
        y["mask"] = inp["mask"].clone().to(dev)
    return y


# ----------------------------------------------------------------------------- kernels
@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_gemm_kernel(dev, precision):
    spec, model = get_model("face", precision, dev)
    model._ensure_ctx(dev, 1)
    lib = model._lib()
    g = torch.Generator().manual_seed(1)
    for (M, N, K) in [(300, 512, 512), (129, 104, 256), (64, 1024, 2038), (1000, 256, 104)]:
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        ref = A.double() @ W.double().T + b.double()     # asymmetric operands: catches transposes
        Ad, Wd, bd = A.to(dev), W.to(dev), b.to(dev)
        out = torch.empty(M, N, device=dev)
        _lib.check(lib.a2p_gemm(model._ctx, _lib.ptr(Ad), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(out), M, N, K,
                                _lib.current_stream()), "a2p_gemm")
        err = rel_l2(out.cpu(), ref)
        record(f"gemm/{precision}/{M}x{N}x{K}", rel_l2=err)
        # fp16 gate = min(1e-3, 2x measured): 2.0e-4..2.3e-4 on MI355X (profiles/r04_parity_tests.json)
        assert err < {"fp32": 2e-6, "bf16": 1e-2, "fp16": 4.5e-4}[precision], (M, N, K, err)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("fmt", ["face", "pose"])
def test_attention_kernel(dev, precision, fmt):
    spec, model = get_model(fmt, precision, dev)
    model._ensure_ctx(dev, 1)
    lib = model._lib()
    d, H = spec.latent_dim, spec.num_heads
    g = torch.Generator().manual_seed(2)
    for (N, Tq, S) in [(2, 100, 77), (1, 240, 800), (3, 33, 20)]:
        q, k, v = (torch.randn(N, L, d, generator=g) for L in (Tq, S, S))
        k[0, 5] *= 6.0  # spike one key: forces the online-softmax rescale branch
        dh = d // H
        qh, kh, vh = (t.view(N, -1, H, dh).transpose(1, 2).double() for t in (q, k, v))
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) / dh ** 0.5, -1) @ vh).transpose(1, 2).reshape(N, Tq, d)
        out = torch.empty(N, Tq, d, device=dev)
        qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)   # keep alive: ptr() of a temporary dangles
        _lib.check(lib.a2p_attention(model._ctx, _lib.ptr(qd), _lib.ptr(kd), _lib.ptr(vd),
                                     _lib.ptr(out), N, Tq, S, _lib.current_stream()), "a2p_attention")
        err = rel_l2(out.cpu(), ref)
        record(f"attn/{fmt}/{precision}/{N}x{Tq}x{S}", rel_l2=err)
        # fp16 gate = min(1e-3, 2x measured): 4.3e-4..5.3e-4 measured -> the 1e-3 bar itself
        assert err < {"fp32": 5e-6, "bf16": 2e-2, "fp16": 1e-3}[precision], (N, Tq, S, err)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("fmt", ["face", "pose"])
def test_decoder_layer_vs_reference_golden(dev, golden, precision, fmt):
    spec, model = get_model(fmt, precision, dev)
    model._ensure_ctx(dev, 2)
    model._ensure_weights(model._lib(), dev)
    d = spec.latent_dim
    x = synthetic_tensor(SEED, "layer_x", (2, 48, d)).to(dev)
    mem = synthetic_tensor(SEED, "layer_mem", (2, 80, d)).to(dev)
    t = synthetic_tensor(SEED, "layer_t", (2, d)).to(dev)
    mem2 = synthetic_tensor(SEED, "layer_mem2", (2, 8, d)).to(dev) if spec.is_pose else None
    _lib.check(model._lib().a2p_decoder_layer_forward(model._ctx, 0, _lib.ptr(x), _lib.ptr(mem), _lib.ptr(t), _lib.ptr(mem2),
                                                     2, 48, 80, 8 if spec.is_pose else 0, _lib.current_stream()), "layer")
    err = rel_l2(x.cpu(), golden[f"{fmt}/layer0"])
    record(f"layer0/{fmt}/{precision}", rel_l2=err)
    # measured on MI355X (profiles/r02_parity_tests.json): fp32 4e-7, bf16 1.6e-3, fp16 2e-4
    assert err < {"fp32": 1e-4, "bf16": 3.2e-3, "fp16": 4e-4}[precision]


# ----------------------------------------------------------------------------- denoiser
@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("fmt", ["face", "pose"])
def test_forward_vs_reference_golden(dev, golden, precision, fmt):
    spec, model = get_model(fmt, precision, dev)
    inp = synthetic_inputs(spec, 2, 240, SEED)
    if spec.is_pose:
        inp["mask"][1, :, :, 90:] = False
    scale = 10.0 if fmt == "face" else 2.0
    y = y_for(spec, inp, dev, scale)
    times = torch.tensor([937, 12], device=dev)
    x = inp["x_T"].to(dev)
    c = model(x, times, y, cond_drop_prob=0.0)
    u = model(x, times, y, cond_drop_prob=1.0)
    g = ClassifierFreeSampleModel(model)(x, times, y)
    errs = (rel_l2(c.cpu(), golden[f"{fmt}/fwd_cond"]), rel_l2(u.cpu(), golden[f"{fmt}/fwd_uncond"]),
            rel_l2(g.cpu(), golden[f"{fmt}/fwd_cfg"]))
    record(f"fwd240/{fmt}/{precision}", cond=errs[0], uncond=errs[1], cfg=errs[2])
    # measured (profiles/r02_parity_tests.json): fp32 ~1e-6; bf16 cond/uncond 3.4-5.1e-3, guided 4.4e-3 (face) / 8.5e-3 (pose);
    # fp16 8x below bf16.  Gates = 2x the measured values (round 1 asserted 5e-2 / 0.25 here)
    # round 5: fp16 = min(1e-3, 2x measured) -- measured 2.1e-4..4.7e-4 (profiles/r04_parity_tests.json fwd240/*/fp16)
    tol, tol_cfg = {"fp32": (2e-4, 2e-4), "bf16": (1.1e-2, 1.8e-2), "fp16": (9.5e-4, 9.5e-4)}[precision]
    assert all(e < tol for e in errs[:2]) and errs[2] < tol_cfg, errs
    if spec.is_pose:   # the reference zeroes masked keyframes in y, in place (model/diffusion.py:320)
        assert float(y["keyframes"][1, 3:].abs().max()) == 0.0


@pytest.mark.parametrize("fmt", ["face", "pose"])
def test_ddim10_fp32_vs_reference_golden(dev, golden, fmt):
    """Config 0 (face B=1 T=240) and the pose B=2 T=600 variant, fused path, fp32 mode: <= 1e-3."""
    spec, model = get_model(fmt, "fp32", dev)
    B, frames = (1, 240) if fmt == "face" else (2, 600)
    inp = synthetic_inputs(spec, B, frames, SEED)
    y = y_for(spec, inp, dev, 10.0 if fmt == "face" else 2.0)
    diffusion = make_diffusion(fmt, "ddim10")
    res = diffusion.ddim_sample_loop(ClassifierFreeSampleModel(model), (B, spec.nfeats, 1, frames), clip_denoised=False,
                                     model_kwargs={"y": y}, noise=inp["x_T"].to(dev))
    e2, em = rel_l2(res.cpu(), golden[f"{fmt}/ddim10"]), rel_max(res.cpu(), golden[f"{fmt}/ddim10"])
    print(f"ddim10 {fmt} fp32: rel L2 {e2:.3e} max-norm {em:.3e}")
    record(f"ddim10_golden/{fmt}/fp32", rel_l2=e2, max_norm=em)
    assert e2 < 1e-3 and em < 1e-3


@pytest.mark.parametrize("fmt", ["face", "pose"])
def test_ddpm_restored_noise_fp32_vs_reference_golden(dev, golden, fmt):
    spec, model = get_model(fmt, "fp32", dev)
    cfg = ClassifierFreeSampleModel(model)
    inp = synthetic_inputs(spec, 1, 240, SEED, steps_of_noise=10)
    y = y_for(spec, inp, dev, 10.0 if fmt == "face" else 2.0)
    noise = inp["step_noise"].to(dev)
    res = make_diffusion(fmt, "ddim10").p_sample_loop(cfg, (1, spec.nfeats, 1, 240), clip_denoised=False, model_kwargs={"y": y},
                                                      noise=inp["x_T"].to(dev), step_noise=lambda n: noise[n])
    assert rel_l2(res.cpu(), golden[f"{fmt}/ddpm10"]) < 1e-3
    assert rel_max(res.cpu(), golden[f"{fmt}/ddpm10"]) < 1e-3
    # first 3 steps of the full 1000-step chain
    inp = synthetic_inputs(spec, 1, 240, SEED, steps_of_noise=3)
    y = y_for(spec, inp, dev, 10.0 if fmt == "face" else 2.0)
    noise = inp["step_noise"].to(dev)
    gen = make_diffusion(fmt, "").p_sample_loop_progressive(cfg, (1, spec.nfeats, 1, 240), clip_denoised=False,
                                                            model_kwargs={"y": y}, noise=inp["x_T"].to(dev),
                                                            step_noise=lambda n: noise[n])
    for _, out in zip(range(3), gen):
        last = out
    assert rel_l2(last["sample"].cpu(), golden[f"{fmt}/ddpm1000_first3"]) < 1e-3


@pytest.mark.parametrize("fmt,order", [("face", 2), ("face", 4), ("pose", 3)])
def test_plms_fp32_vs_reference_golden(dev, golden_plms, fmt, order):
    """plms_sample_loop (Euler start + Adams-Bashforth orders 1..4) against the reference's own output."""
    spec, model = get_model(fmt, "fp32", dev)
    B, frames = (1, 240) if fmt == "face" else (2, 240)
    inp = synthetic_inputs(spec, B, frames, SEED)
    y = y_for(spec, inp, dev, 10.0 if fmt == "face" else 2.0)
    res = make_diffusion(fmt, "ddim10").plms_sample_loop(ClassifierFreeSampleModel(model), (B, spec.nfeats, 1, frames),
                                                         clip_denoised=False, model_kwargs={"y": y}, noise=inp["x_T"].to(dev),
                                                         order=order)
    want = golden_plms[f"{fmt}/plms10_order{order}"]
    e2, em = rel_l2(res.cpu(), want), rel_max(res.cpu(), want)
    print(f"plms10 {fmt} order {order} fp32: rel L2 {e2:.3e} max-norm {em:.3e}")
    assert e2 < 1e-3 and em < 1e-3


@pytest.mark.parametrize("fmt", ["face", "pose"])
def test_plms_steps_and_ddim_reverse_fp32_vs_reference_golden(dev, golden_plms, fmt):
    spec, model = get_model(fmt, "fp32", dev)
    cfg = ClassifierFreeSampleModel(model)
    B, frames = (1, 240) if fmt == "face" else (2, 240)
    inp = synthetic_inputs(spec, B, frames, SEED)
    y = y_for(spec, inp, dev, 10.0 if fmt == "face" else 2.0)
    d = make_diffusion(fmt, "ddim10")
    gen = d.plms_sample_loop_progressive(cfg, (B, spec.nfeats, 1, frames), clip_denoised=False, model_kwargs={"y": y},
                                         noise=inp["x_T"].to(dev), order=2)
    for i, out in zip(range(2), gen):
        assert set(out) == {"sample", "pred_xstart", "old_eps"} and len(out["old_eps"]) == 1
        for k in ("sample", "pred_xstart"):
            assert rel_l2(out[k].cpu(), golden_plms[f"{fmt}/plms_step{i}/{k}"]) < 1e-3, (i, k)
    r = d.ddim_reverse_sample(cfg, inp["x_T"].to(dev), torch.tensor([5] * B, device=dev), clip_denoised=False, model_kwargs={"y": y})
    assert rel_l2(r["sample"].cpu(), golden_plms[f"{fmt}/ddim_reverse_t5"]) < 1e-3
    with pytest.raises(ValueError):
        d.plms_sample(cfg, inp["x_T"].to(dev), torch.tensor([5] * B, device=dev), model_kwargs={"y": y}, order=5)
    with pytest.raises(AssertionError):
        d.ddim_reverse_sample(cfg, inp["x_T"].to(dev), torch.tensor([5] * B, device=dev), model_kwargs={"y": y}, eta=0.5)


def test_generic_path_matches_fused(dev):
    """p_mean_variance through the model protocol (`model(x, ts, **kw)`, any callable) == fused step."""
    spec, model = get_model("face", "fp32", dev)
    cfg = ClassifierFreeSampleModel(model)
    inp = synthetic_inputs(spec, 2, 64, SEED)
    y = y_for(spec, inp, dev, 10.0)
    d = make_diffusion("face", "ddim10")
    x = inp["x_T"].to(dev)
    t = torch.tensor([7, 7], device=dev)
    fused = d.ddim_sample(cfg, x, t, clip_denoised=False, model_kwargs={"y": y})
    generic = d.ddim_sample(lambda xx, ts, **kw: cfg(xx, ts, **kw), x, t, clip_denoised=False, model_kwargs={"y": y})
    assert rel_l2(generic["sample"].cpu(), fused["sample"].cpu()) < 1e-6
    assert rel_l2(generic["pred_xstart"].cpu(), fused["pred_xstart"].cpu()) < 1e-6
    pf = d.p_sample(cfg, x, t, clip_denoised=True, model_kwargs={"y": y}, noise=torch.ones_like(x))
    pg = d.p_sample(lambda xx, ts, **kw: cfg(xx, ts, **kw), x, t, clip_denoised=True, model_kwargs={"y": y}, noise=torch.ones_like(x))
    assert rel_l2(pg["sample"].cpu(), pf["sample"].cpu()) < 1e-6
    assert float(pf["pred_xstart"].abs().max()) <= 1.0


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_config0_ddim10_16bit_vs_reference_golden(dev, golden, precision):
    """BASELINE configs[0] end to end (face, batch 1, 240 frames, ddim10 loop, guidance 10) in the 16-bit modes against the golden the
    REFERENCE ITSELF produced (tests/golden/make_golden.py).  480 rows: this is the small-forward path (csrc/kernels_small.h +
    attn_ksplit_kernel).  fp16 operands are the benchmarked mode and must meet the north-star bar (1e-3 rel-L2 on the loop's return
    value); bf16 operands are 8x coarser (measured 3.0e-3)."""
    spec, model = get_model("face", precision, dev)
    inp = synthetic_inputs(spec, 1, 240, SEED)
    y = y_for(spec, inp, dev, 10.0)
    res = make_diffusion("face", "ddim10").ddim_sample_loop(ClassifierFreeSampleModel(model), (1, spec.nfeats, 1, 240),
                                                            clip_denoised=False, model_kwargs={"y": y}, noise=inp["x_T"].to(dev))
    e = rel_l2(res.cpu(), golden["face/ddim10"])
    record(f"ddim10_240/face/{precision}", rel_l2=e)
    assert e < {"bf16": 6e-3, "fp16": 1e-3}[precision], e


@pytest.mark.parametrize("attn3", ["0", "2"])
def test_properties_full_size(dev, attn3, monkeypatch):
    """Size-independent properties at BASELINE config-1 size (face, B=8, T=600), bf16 mode:
    batch independence (sample b does not depend on its neighbours) and determinism.
    With the attention kernel pinned (A2P_ATTN3=0: attn_kernel everywhere, 2: attn3_kernel wherever it is legal): launch_attn's default rule sends a launch to one or the other
    by its SIZE (round 6: attn_kernel while its grid is one round -- the sub-batch's launches, and layer 0's shared-half self attention of a small guided batch), and the two
    are different 16-bit roundings of the same softmax (4e-4 in fp16, 2e-3 in bf16), not the same bits.  The property is about batch POSITION: same kernels on both sides."""
    monkeypatch.setenv("A2P_ATTN3", attn3)
    spec, _ = get_model("face", "bf16", dev)
    args = default_args("face", timestep_respacing="")
    model, _ = create_model_and_diffusion(args, "test", precision="bf16", max_batch=8)
    load_model(model, synthetic_state_dict(spec, SEED))
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    inp = synthetic_inputs(spec, 8, 600, SEED)
    y = y_for(spec, inp, dev, 10.0)
    x = inp["x_T"].to(dev)
    t = torch.full((8,), 500, device=dev)
    full = cfg(x, t, y)
    again = cfg(x, t, y)
    assert torch.equal(full, again)
    y2 = {"cond_embed": y["cond_embed"][2:4].contiguous(), "scale": y["scale"][2:4].contiguous()}
    part = cfg(x[2:4].contiguous(), t[2:4], y2)
    model.release()
    monkeypatch.delenv("A2P_ATTN3", raising=False)
    assert rel_l2(part.cpu(), full[2:4].cpu()) < 1e-6
    assert torch.isfinite(full).all()


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("fmt,mt", [("face", "2"), ("face", "3"), ("face", "4"), ("pose", "2"), ("pose", "3"), ("pose", "5"), ("pose", "6")])
def test_chain_kernels_match_the_per_op_kernels_16bit(dev, golden, fmt, mt, precision, monkeypatch):
    """The 16-bit modes run the decoder layers as fused row-panel chain kernels (csrc/kernels_chain.h); the per-op
    kernels (GEMM / LayerNorm launches) must give the same answer to operand rounding, for every panel height,
    including a ragged last panel (2*2*240 = 960 rows is not a multiple of 64 / 80 / 96) and a panel that
    straddles two sequences.  Both builds (bfloat16 and IEEE-half operands)."""
    spec, model = get_model(fmt, precision, dev)
    inp = synthetic_inputs(spec, 2, 240, SEED)
    if spec.is_pose:
        inp["mask"][1, :, :, 90:] = False
    scale = 10.0 if fmt == "face" else 2.0
    y = y_for(spec, inp, dev, scale)
    times = torch.tensor([937, 12], device=dev)
    x = inp["x_T"].to(dev)
    cfg = ClassifierFreeSampleModel(model)
    monkeypatch.setenv("A2P_CHAIN_MT", mt)
    monkeypatch.delenv("A2P_NO_CHAIN", raising=False)
    chained = cfg(x, times, y).cpu()
    monkeypatch.setenv("A2P_NO_CHAIN", "1")
    monkeypatch.setenv("A2P_NO_SMALL", "1")            # the per-op kernels are the counterpart (960 rows would take the small-forward GEMMs)
    per_op = cfg(x, times, y).cpu()
    monkeypatch.delenv("A2P_NO_CHAIN")
    monkeypatch.delenv("A2P_NO_SMALL")
    ref = golden[f"{fmt}/fwd_cfg"]
    e_pair, e_gold, e_old = rel_l2(chained, per_op), rel_l2(chained, ref), rel_l2(per_op, ref)
    record(f"chain_vs_perop/{precision}/{fmt}/MT{mt}", pair=e_pair, chain_vs_golden=e_gold, perop_vs_golden=e_old)
    # bf16 measured in round 2: chain vs per-op 3.0e-3 (face) / 8.4e-3 (pose), both 4.4e-3 / 8.5e-3 from the fp32 reference;
    # IEEE-half operands (the default throughput mode) are held to north_star's bar itself: 1e-3 against the fp32 reference
    # (measured 4.2e-4 / 4.8e-4; the two kernel families 2.6e-4 / 2.2e-4 apart)
    if precision == "fp16":
        assert e_pair < 1e-3 and e_gold < 1e-3 and e_gold < 1.2 * e_old + 2e-4
    else:
        assert e_pair < (6.5e-3 if fmt == "face" else 1.7e-2) and e_gold < (9e-3 if fmt == "face" else 1.7e-2) \
            and e_gold < 1.2 * e_old + 1e-3


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("fmt,B,frames", [("face", 4, 150), ("pose", 5, 210)])
def test_chain_kernels_with_frame_counts_that_are_not_a_multiple_of_4(dev, fmt, B, frames, precision, monkeypatch):
    """The reference derives the frame count from the audio length (demo/demo.py: int(len/sr) * 30), so T = 150, 210, ... occur:
    4-row groups of the transposed V^T store then straddle sequences.  Chain path == per-op path, and both close to the oracle."""
    from oracle import a2p_oracle as O
    spec, model = get_model(fmt, precision, dev)
    inp = synthetic_inputs(spec, B, frames, SEED)
    scale = 10.0 if fmt == "face" else 2.0
    y = y_for(spec, inp, dev, scale)
    times = torch.tensor([999, 500, 250, 3, 77][:B], device=dev)
    x = inp["x_T"].to(dev)
    cfg = ClassifierFreeSampleModel(model)
    monkeypatch.delenv("A2P_NO_CHAIN", raising=False)
    monkeypatch.setenv("A2P_CHAIN_ROWS", "1")          # 1200 / 2100 rows: below the default chain threshold for the face model
    chained = cfg(x, times, y).cpu()
    monkeypatch.setenv("A2P_NO_CHAIN", "1")
    monkeypatch.setenv("A2P_NO_SMALL", "1")            # ... and the per-op kernels, not the small-forward GEMMs, as the counterpart
    per_op = cfg(x, times, y).cpu()
    monkeypatch.delenv("A2P_NO_CHAIN")
    monkeypatch.delenv("A2P_NO_SMALL")
    monkeypatch.delenv("A2P_CHAIN_ROWS")
    den = O.OracleDenoiser(synthetic_state_dict(spec, SEED), fmt, spec.num_layers, spec.num_heads, torch.float32)
    ref = den.forward_cfg(inp["x_T"][:2], times[:2].cpu(), inp["cond_embed"][:2], torch.full((2,), scale),
                          inp.get("keyframes", [None])[:2] if spec.is_pose else None, inp["mask"][:2] if spec.is_pose else None)
    e_pair, e_ref = rel_l2(chained, per_op), rel_l2(chained[:2], ref)
    record(f"chain_ragged/{precision}/{fmt}/T{frames}", pair=e_pair, chain_vs_oracle=e_ref, perop_vs_oracle=rel_l2(per_op[:2], ref))
    if precision == "fp16":                          # the 1e-3 bar itself (measured 2.6e-4 / 2.3e-4 and 4.2e-4 / 4.9e-4)
        assert e_pair < 1e-3 and e_ref < 1e-3
    else:                                            # bf16 measured 3.1e-3 / 8.8e-3, 4.3e-3 / 7.5e-3
        assert e_pair < (6.5e-3 if fmt == "face" else 1.8e-2) and e_ref < (9e-3 if fmt == "face" else 1.5e-2)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("fmt,B,frames", [("face", 4, 240), ("pose", 4, 240)])
def test_chain_workgroup_shapes_are_bit_identical(dev, fmt, B, frames, precision, monkeypatch):
    """The library picks the 4- or the 8-wave chain workgroup shape per box from in-situ timings (chain_pick_nw), so the two
    must agree to the last bit in BOTH 16-bit builds and for both model widths: same GEMM accumulation order per output, one
    shared 8-partial LayerNorm reduction tree, no floating-point contraction (kernels_chain.h).  The 7 un-forced forwards
    afterwards walk through the calibration (alternating shapes) and the sticky choice."""
    spec, model = get_model(fmt, precision, dev)
    inp = synthetic_inputs(spec, B, frames, SEED)
    y = y_for(spec, inp, dev, 10.0 if fmt == "face" else 2.0)
    times = torch.tensor([901, 417, 33, 0][:B], device=dev)
    x = inp["x_T"].to(dev)
    cfg = ClassifierFreeSampleModel(model)
    outs = {}
    for nw in ("4", "8"):
        monkeypatch.setenv("A2P_CHAIN_NW", nw)
        outs[nw] = cfg(x, times, y).clone()
    monkeypatch.delenv("A2P_CHAIN_NW")
    auto = [cfg(x, times, y).clone() for _ in range(7)]          # calibration forwards alternate the shapes, then one sticks
    assert torch.equal(outs["4"], outs["8"])
    assert all(torch.equal(a, outs["4"]) for a in auto)
    record(f"nw_identity/{precision}/{fmt}_B{B}_T{frames}", max_abs_diff=0.0)


# ----------------------------------------------------------------------------- edge shapes / sampler API surface
@pytest.mark.parametrize("fmt,B,frames", [("face", 1, 100), ("pose", 3, 64), ("face", 2, 4), ("face", 2, 30), ("pose", 2, 90)])
def test_edge_shapes_fp32_vs_oracle(dev, fmt, B, frames):
    """Ragged sizes the reference accepts: frames not a multiple of the 64/128 tile sizes, token counts not a multiple of
    64, odd batch, partially masked keyframes, the minimum frame count.  fp32 mode vs the pinned oracle, <= 1e-3."""
    from oracle import a2p_oracle as O
    from audio2photoreal_amd.synthetic import cond_tokens_for_frames
    spec, model = get_model(fmt, "fp32", dev)
    n_tok = max(cond_tokens_for_frames(frames), 3)
    x = synthetic_tensor(SEED, "edge_x", (B, spec.nfeats, 1, frames))
    ce = synthetic_tensor(SEED, "edge_ce", (B, n_tok, spec.cond_feature_dim))
    scale = torch.full((B,), 10.0 if fmt == "face" else 2.0)
    y = {"cond_embed": ce.to(dev), "scale": scale.to(dev)}
    kf = mk = None
    if spec.is_pose:
        nk = len(range(frames)[:: spec.keyframe_step])
        kf = synthetic_tensor(SEED, "edge_kf", (B, nk, spec.keyframe_dim))
        mk = torch.ones(B, 1, 1, frames, dtype=torch.bool)
        mk[0, :, :, 30:] = False
        y["keyframes"], y["mask"] = kf.clone().to(dev), mk.clone().to(dev)
    times = torch.tensor([999, 0, 500][:B])
    got = ClassifierFreeSampleModel(model)(x.to(dev), times.to(dev), y).cpu()
    den = O.OracleDenoiser(synthetic_state_dict(spec, SEED), fmt, spec.num_layers, spec.num_heads)
    want = den.forward_cfg(x, times, ce, scale, kf, mk)
    e2, em = rel_l2(got, want), rel_max(got, want)
    print(f"edge {fmt} B={B} T={frames} S={n_tok}: rel L2 {e2:.3e} max-norm {em:.3e}")
    assert e2 < 1e-3 and em < 1e-3


def test_sampler_api_surface(dev):
    """dump_steps / const_noise / init_image + skip_timesteps / progressive generators behave like the reference's loops
    (gaussian_diffusion.py:525-665, :815-936), all arithmetic on the GPU."""
    spec, model = get_model("face", "fp32", dev)
    cfg = ClassifierFreeSampleModel(model)
    d = make_diffusion("face", "ddim10")
    inp = synthetic_inputs(spec, 2, 64, SEED, steps_of_noise=10)
    y = y_for(spec, inp, dev, 10.0)
    shape, x_T, nz = (2, spec.nfeats, 1, 64), inp["x_T"].to(dev), inp["step_noise"].to(dev)
    kw = dict(clip_denoised=False, model_kwargs={"y": y}, noise=x_T, step_noise=lambda n: nz[n])
    prog = [o["sample"].clone() for o in d.p_sample_loop_progressive(cfg, shape, **kw)]
    assert len(prog) == 10
    final = d.p_sample_loop(cfg, shape, **kw)
    assert torch.equal(final, prog[-1])
    dumped = d.p_sample_loop(cfg, shape, dump_steps=[0, 4, 9], **kw)
    assert len(dumped) == 3 and all(torch.equal(a, prog[i]) for a, i in zip(dumped, (0, 4, 9)))
    # const_noise: every sample of the batch receives sample 0's noise
    c1 = d.p_sample(cfg, x_T, torch.tensor([5, 5], device=dev), clip_denoised=False, model_kwargs={"y": y}, noise=nz[0], const_noise=True)
    c2 = d.p_sample(cfg, x_T, torch.tensor([5, 5], device=dev), clip_denoised=False, model_kwargs={"y": y},
                    noise=nz[0][[0]].repeat(2, 1, 1, 1))
    assert torch.equal(c1["sample"], c2["sample"])
    # ddim loop returns pred_xstart and refuses dump_steps / const_noise like the reference (:839-842)
    with pytest.raises(NotImplementedError):
        d.ddim_sample_loop(cfg, shape, dump_steps=[1], clip_denoised=False, model_kwargs={"y": y}, noise=x_T)
    with pytest.raises(NotImplementedError):
        d.ddim_sample_loop(cfg, shape, const_noise=True, clip_denoised=False, model_kwargs={"y": y}, noise=x_T)
    # init_image + skip_timesteps: the chain starts from q_sample(init_image, t_start, noise)   (:625-632)
    init = synthetic_tensor(SEED, "init_image", shape).to(dev)
    t_start = torch.full((2,), 10 - 4 - 1, device=dev)
    x_start = d.q_sample(init, t_start, x_T)
    tab = d._tables(dev)
    want = tab[8][4 + 1] * init + tab[9][4 + 1] * x_T          # sqrt(abar) x0 + sqrt(1 - abar) noise at step index 5
    assert rel_l2(x_start.cpu(), want.cpu()) < 1e-6
    outs = list(d.ddim_sample_loop_progressive(cfg, shape, clip_denoised=False, model_kwargs={"y": y}, noise=x_T,
                                               init_image=init, skip_timesteps=4))
    assert len(outs) == 6
    first = d.ddim_sample(cfg, x_start, t_start, clip_denoised=False, model_kwargs={"y": y})
    assert rel_l2(outs[0]["sample"].cpu(), first["sample"].cpu()) < 1e-6


def test_sample_parallel_world_size_1_and_generate_surface(dev):
    """`_generate_sequences` / `_run_single_diffusion` (sample/generate.py:74-152) over the accelerated path:
    results dict keys, un-normalisation, identical to calling the sampler directly."""
    import argparse
    from audio2photoreal_amd.sample.generate import _generate_sequences, make_inv_transform
    spec, model = get_model("face", "fp32", dev)
    cfg = ClassifierFreeSampleModel(model)
    d = make_diffusion("face", "ddim10")
    inp = synthetic_inputs(spec, 2, 64, SEED)
    stats = {"code_mean": np.full(256, 0.5), "code_std": np.full(256, 2.0), "pose_mean": np.zeros(104), "pose_std": np.ones(104),
             "audio_mean": np.zeros(2, np.float32), "audio_std": np.ones(2, np.float32), "audio_std_flat": np.ones(1, np.float32)}
    args = argparse.Namespace(batch_size=2, curr_seq_length=64, data_format="face", num_repetitions=2, guidance_param=10.0, device=dev)
    torch.manual_seed(0)
    res = _generate_sequences(args, {"y": {"cond_embed": inp["cond_embed"].to(dev), "lengths": torch.tensor([64, 64])}}, d, cfg,
                              make_inv_transform(stats))
    assert set(res) == {"motions", "audio", "gt", "lengths", "keyframes"}
    assert res["motions"].shape == (4, 256, 1, 64) and res["lengths"].shape == (4,) and res["audio"] is None
    torch.manual_seed(0)
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((2,), 10.0, device=dev)}
    direct = d.ddim_sample_loop(cfg, (2, 256, 1, 64), clip_denoised=False, model_kwargs={"y": y}).cpu().numpy() * 2.0 + 0.5
    assert np.allclose(res["motions"][:2], direct, rtol=1e-5, atol=1e-5)


def test_weight_updates_are_picked_up(dev):
    """The device copy of the weights follows the module's parameters like a plain nn.Module would: in-place updates under
    no_grad / load_state_dict (version counters), `.to()` re-seating (`_apply`), and the explicit `invalidate_weights()`."""
    spec = face_spec(num_layers=1)
    model, _ = create_model_and_diffusion(default_args("face", layers=1), "test", precision="fp32", max_batch=1)
    sd = synthetic_state_dict(spec, 7)
    load_model(model, sd)
    model = model.to(dev).eval()
    inp = synthetic_inputs(spec, 1, 64, SEED)
    y = y_for(spec, inp, dev, 10.0)
    x, t = inp["x_T"].to(dev), torch.tensor([100], device=dev)
    base = model(x, t, y).clone()
    assert torch.equal(model(x, t, y), base)
    with torch.no_grad():
        model.final_layer.bias.add_(1.0)
    assert torch.allclose(model(x, t, y), base + 1.0, atol=1e-5)
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["final_layer.bias"] = sd["final_layer.bias"] - 2.0
    load_model(model, sd2)
    assert torch.allclose(model(x, t, y), base - 2.0, atol=1e-5)
    model.final_layer.bias.data.copy_(sd["final_layer.bias"].to(dev))     # bypasses the version counter ...
    model.invalidate_weights()                                             # ... so the caller says so
    assert torch.allclose(model(x, t, y), base, atol=1e-5)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_chain_path_is_bitwise_reproducible_under_repetition(dev, precision):
    """Race screen for the LDS-DMA weight rings / panel hand-offs of the chain kernels: 24 repeated guided forwards at
    the full bench size (B=8, T=600, both 16-bit builds) must be bit-identical (a late DMA landing or an early fragment read
    shows up as run-to-run differences), and a second context with the same weights must agree too."""
    spec = face_spec()
    sd = synthetic_state_dict(spec, SEED)
    outs = []
    for rep in range(2):
        model, _ = create_model_and_diffusion(default_args("face", timestep_respacing=""), "test", precision=precision, max_batch=8)
        load_model(model, sd)
        cfg = ClassifierFreeSampleModel(model.to(dev).eval())
        inp = synthetic_inputs(spec, 8, 600, SEED)
        y = y_for(spec, inp, dev, 10.0)
        x = inp["x_T"].to(dev)
        t = torch.tensor([999, 750, 500, 250, 100, 10, 1, 0], device=dev)
        first = cfg(x, t, y).clone()
        for i in range(11):
            again = cfg(x, t, y)
            assert torch.equal(again, first), f"context {rep}, repeat {i}: max |diff| = {float((again - first).abs().max()):.3e}"
        outs.append(first)
    assert torch.equal(outs[0], outs[1]), f"two contexts disagree: max |diff| = {float((outs[0] - outs[1]).abs().max()):.3e}"
