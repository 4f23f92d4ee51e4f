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
        before = _debug_i64(model, b"chain4_launches")
        got = cfg(x, t, y).cpu()
        launched = _debug_i64(model, b"chain4_launches") - before
        model.check_finite()
        model.release()
        e = {"rel_l2": rel_l2(got, want), "max_norm": rel_max(got, want), "worst_sample_rel_l2": _worst_sample(got, want)}
        record(f"oracle_at_bench_batch/face_B{B}_{rows}row/{precision}", tall_launches=launched, **e)
        if precision != "fp32":                               # fp32 parity mode runs the per-op exact-fp32 kernels, not the chain kernels
            assert launched == 16, launched
        assert e["rel_l2"] < tol and e["worst_sample_rel_l2"] < tol * (1.0 if precision == "fp32" else 1.5), e


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
