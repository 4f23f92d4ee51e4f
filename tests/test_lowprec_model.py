"""oracle/lowprec_model.py (the CPU model of the GPU's 16-bit rounding sites behind tests/tools/error_budget.py) against the
oracle: with every site in fp32 it IS the oracle, and with IEEE-half sites it reproduces the orders of magnitude the GPU
measures (profiles/r03_error_budget.json holds the full-size run)."""
import torch

from audio2photoreal_amd.spec import face_spec, pose_spec
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict
from oracle import a2p_oracle as O
from oracle import lowprec_model as LP


def _run(fmt, rounding, T=48, layers=2):
    spec = (face_spec if fmt == "face" else pose_spec)(num_layers=layers)
    sd = synthetic_state_dict(spec, 10)
    inp = synthetic_inputs(spec, 2, T, 10)
    kf, mk = (inp["keyframes"], inp["mask"]) if spec.is_pose else (None, None)
    scale = torch.full((2,), 10.0 if fmt == "face" else 2.0)
    t = torch.tensor([700, 12])
    with torch.no_grad():
        want = O.OracleDenoiser(sd, fmt, layers, spec.num_heads).forward_cfg(inp["x_T"], t, inp["cond_embed"], scale, kf, mk)
        got = LP.LowPrecDenoiser(sd, fmt, layers, spec.num_heads, rounding).forward_cfg(inp["x_T"], t, inp["cond_embed"], scale, kf, mk)
    return float((got - want).norm() / want.norm())


def test_all_sites_fp32_is_the_oracle():
    for fmt in ("face", "pose"):
        assert _run(fmt, LP.Rounding("fp32")) < 5e-6


def test_half_sites_are_ordered_like_the_formats():
    e16, eb16 = _run("face", LP.Rounding("fp16")), _run("face", LP.Rounding("bf16"))
    assert 1e-5 < e16 < 3e-3 and 4.0 * e16 < eb16 < 16.0 * e16          # 3 mantissa bits apart
    exact_tail = {s: "fp32" for s in ("fin.a", "fin.w", "in.a", "in.w")}
    assert _run("face", LP.Rounding("fp16", exact_tail)) < e16
    assert _run("face", LP.Rounding("fp16", {"fin.a": "fp16x2"})) < e16  # the hi + lo pair is as good as exact for that site
