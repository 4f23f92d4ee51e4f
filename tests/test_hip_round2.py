"""Round-2 GPU tests (`pytest -m gpu`): parity of BOTH precisions at the benchmarked shape against the pinned oracle,
the conditioning cache under address reuse, the sample-parallel path on the HIP sampler with two ranks sharing the GPU, and
bit-identity of the chain kernels' workgroup shapes for the body model.

Measured errors are written to gpurun_out/parity_tests.json (pytest -q swallows prints) so a run leaves its numbers behind.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
from audio2photoreal_amd.model_util import create_gaussian_diffusion, create_model_and_diffusion, default_args, load_model
from audio2photoreal_amd.spec import face_spec, pose_spec
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict, synthetic_tensor
from conftest import ROOT, record, rel_l2, rel_max

pytestmark = pytest.mark.gpu
SEED = 10

@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def build(fmt, precision, dev, max_batch=2, layers=None, respacing="ddim10"):
    spec = (face_spec if fmt == "face" else pose_spec)(**({} if layers is None else {"num_layers": layers}))
    model, diffusion = create_model_and_diffusion(default_args(fmt, layers=layers, timestep_respacing=respacing), "test",
                                                  precision=precision, max_batch=max_batch)
    sd = synthetic_state_dict(spec, SEED)
    load_model(model, sd)
    return spec, sd, model.to(dev).eval(), diffusion


# ----------------------------------------------------------------------------- parity at the benchmarked shape
# Round 3: input_projection, final_layer and the pose conv tail are exact-fp32 islands in the 16-bit modes (a2p_ctx::tail32; the
# error budget in profiles/r03_error_budget*.json found final_layer's operand rows to carry 3.05e-3 of the face model's 3.11e-3
# loop error).  fp16 -- the benchmarked mode -- is now gated at the north_star bar itself, 1e-3, on the forward AND on the loop's
# return value (oracle/lowprec_model.py predicts 4.1e-4 / 3.6e-4 face, 3.1e-4 / 3.1e-4 body).
# bf16 (8 mantissa bits) cannot reach it: gates = 2x the model's prediction (face 3.4e-3 / 3.1e-3, body 2.5e-3 / 2.5e-3).
BF16_FWD_TOL = {"face": 7.0e-3, "pose": 5.0e-3}
BF16_DDIM10_TOL = {"face": 6.5e-3, "pose": 5.0e-3}
FP16_FWD_TOL = {"face": 1.0e-3, "pose": 1.0e-3}
FP16_DDIM10_TOL = {"face": 1.0e-3, "pose": 1.0e-3}


@pytest.mark.parametrize("fmt", ["face", "pose"])
def test_T600_forward_and_ddim10_vs_oracle_both_precisions(dev, fmt):
    """The shape the throughput is quoted on (T=600, S=1998+2): guided forward and 10 DDIM steps against the oracle.
    fp32 mode <= 1e-3 (rel-L2 and max-norm); fp16 -- the benchmarked mode -- <= 1e-3 rel-L2 on both; bf16 at 2x its error."""
    from oracle import a2p_oracle as O
    B, T = 1, 600
    spec = face_spec() if fmt == "face" else pose_spec()
    sd = synthetic_state_dict(spec, SEED)
    inp = synthetic_inputs(spec, B, T, SEED)
    scale = 10.0 if fmt == "face" else 2.0
    den = O.OracleDenoiser(sd, fmt, spec.num_layers, spec.num_heads)
    kf, mk = (inp["keyframes"], inp["mask"]) if spec.is_pose else (None, None)
    times = torch.tensor([700])
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    with torch.no_grad():
        want_fwd = den.forward_cfg(inp["x_T"], times, inp["cond_embed"], torch.full((B,), scale), kf, mk)
        fn = lambda x, ts: den.forward_cfg(x, ts, inp["cond_embed"], torch.full((B,), scale), kf, mk)
        want_x0, _ = O.OracleSampler("ddim10").ddim_sample_loop(fn, inp["x_T"])
    for precision in ("fp32", "bf16", "fp16"):
        _, _, model, diffusion = build(fmt, precision, dev, max_batch=1)
        cfg = ClassifierFreeSampleModel(model)
        y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), scale, device=dev)}
        if spec.is_pose:
            y["keyframes"], y["mask"] = inp["keyframes"].clone().to(dev), inp["mask"].clone().to(dev)
        got_fwd = cfg(inp["x_T"].to(dev), times.to(dev), y).cpu()
        got_x0 = diffusion.ddim_sample_loop(cfg, (B, spec.nfeats, 1, T), clip_denoised=False, model_kwargs={"y": y},
                                            noise=inp["x_T"].to(dev)).cpu()
        e = {"fwd_rel_l2": rel_l2(got_fwd, want_fwd), "fwd_max_norm": rel_max(got_fwd, want_fwd),
             "ddim10_rel_l2": rel_l2(got_x0, want_x0), "ddim10_max_norm": rel_max(got_x0, want_x0)}
        record(f"T600/{fmt}/{precision}", **e)
        if precision == "fp32":
            assert max(e.values()) < 1e-3, e
        elif precision == "bf16":
            assert e["fwd_rel_l2"] < BF16_FWD_TOL[fmt] and e["ddim10_rel_l2"] < BF16_DDIM10_TOL[fmt], e
        else:
            assert e["fwd_rel_l2"] < FP16_FWD_TOL[fmt] and e["ddim10_rel_l2"] < FP16_DDIM10_TOL[fmt], e
        model.release()


# ----------------------------------------------------------------------------- conditioning cache
def test_conditioning_cache_survives_address_reuse(dev):
    """Round-1 hazard (VERDICT 'stale-conditioning'): clip 1's cond_embed is freed, clip 2's lands on the same address with
    `_version` 0 -> the (data_ptr, version, shape) key matched and clip 2 was denoised against clip 1's K/V.  The module now pins
    the keyed tensors, so either the address is not recycled or the key differs; the outputs must follow the inputs."""
    spec, sd, model, _ = build("face", "fp32", dev, max_batch=1, layers=2)
    T = 64
    inp = synthetic_inputs(spec, 1, T, SEED)
    x, t = inp["x_T"].to(dev), torch.tensor([300], device=dev)
    scale = torch.full((1,), 10.0, device=dev)

    def run(clip):   # the pattern from ADVICE.md: a fresh y per clip, nothing kept by the caller
        ce = (inp["cond_embed"] * (1.0 if clip == 0 else -0.5) + clip).to(dev)
        ptr = ce.data_ptr()
        out = model(x, t, {"cond_embed": ce, "scale": scale}).clone()
        return out, ptr

    torch.cuda.empty_cache()
    out0, p0 = run(0)
    out1, p1 = run(1)
    model.invalidate_cond()
    fresh1, _ = run(1)
    assert not torch.equal(out0, out1), "clip 2 was denoised with clip 1's conditioning"
    assert torch.equal(out1, fresh1)
    record("cond_cache/address_reused", reused=bool(p0 == p1))
    # in-place edits of the SAME tensor are seen through the version counter
    ce = inp["cond_embed"].to(dev)
    y = {"cond_embed": ce, "scale": scale}
    a = model(x, t, y).clone()
    ce.mul_(0.25)
    b = model(x, t, y).clone()
    assert not torch.equal(a, b)
    # and an unchanged y is a cache hit: no second a2p_prepare_cond (same object, same version)
    key = model._cond_key
    model(x, t, y)
    assert model._cond_key == key
    model.release()


def test_guide_cache_survives_address_reuse(dev):
    from audio2photoreal_amd.model.guide import GuideTransformer
    from audio2photoreal_amd.spec import GuideSpec
    from audio2photoreal_amd.synthetic import synthetic_guide_state_dict
    gs = GuideSpec()
    g = GuideTransformer(tokens=gs.tokens, num_layers=gs.num_layers, dim=gs.dim, emb_len=gs.emb_len, num_audio_layers=gs.num_audio_layers,
                         max_batch=2, max_positions=16)
    g.load_state_dict(synthetic_guide_state_dict(gs, SEED), strict=False)
    g = g.to(dev).eval()
    toks = torch.zeros(1, 4, dtype=torch.int64, device=dev)

    def run(clip):
        cond = (synthetic_tensor(SEED, "guide_cond", (1, 798, 1024)) * (1.0 if clip == 0 else -1.0)).to(dev)
        return g(toks, cond).clone()
    a, b = run(0), run(1)
    assert not torch.equal(a, b)
    g.invalidate_cond()
    assert torch.equal(run(1), b)


# ----------------------------------------------------------------------------- N > 1 on the GPU that exists
@pytest.mark.parametrize("precision", ["bf16", "fp16", "fp32"])
def test_two_ranks_sharing_the_gpu_match_single_rank_bit_for_bit(dev, tmp_path, precision):
    """SURVEY §8e on the HIP sampler: 2 processes (gloo collectives, both on cuda:0) run sample_parallel(ddim_sample_loop)
    over a global batch of 4; the gathered samples must equal the single-process result bit for bit (noise indexed by
    global sample id, no op mixes batch elements, one all_gather at the end)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), A2P_DIST_OUT=str(tmp_path), A2P_DIST_PRECISION=precision)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_gpu.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker_gpu as W
    want = W.run_sampler(precision, dev, world=1, rank=0)          # no process group: sample_parallel degenerates to one shard
    got0, got1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(got0, got1), "all ranks must hold all samples after the single all_gather"
    diff = float((got0 - want.cpu()).abs().max())
    record(f"dist2/{precision}", max_abs_diff=diff)
    assert torch.equal(got0, want.cpu()), f"sharded != single-rank, max |diff| = {diff:.3e}"


# ----------------------------------------------------------------------------- chain workgroup shapes, body model
@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("B,T", [(3, 448), (4, 450), (3, 600), (16, 600)])
def test_chain_workgroup_shapes_are_bit_identical_pose(dev, B, T, precision, monkeypatch):
    """Round-1 open item: the 4- and 8-wave chain shapes disagreed at bf16-rounding level for d=256 on small forwards.
    Both 16-bit builds: liba2p_hip_f16.so is a second compile of every instantiation (the contraction bug was specific to one
    instantiation of one compile) and it is the build whose shape is picked at run time in the benchmark."""
    spec, sd, model, _ = build("pose", precision, dev, max_batch=B)
    cfg = ClassifierFreeSampleModel(model)
    inp = synthetic_inputs(spec, B, T, SEED)
    y = {"cond_embed": inp["cond_embed"].to(dev), "keyframes": inp["keyframes"].to(dev), "mask": inp["mask"].to(dev),
         "scale": torch.full((B,), 2.0, device=dev)}
    x = inp["x_T"].to(dev)
    t = torch.tensor(([901, 417, 33, 0] * 4)[:B], device=dev)
    outs = {}
    for nw in ("4", "8"):
        monkeypatch.setenv("A2P_CHAIN_NW", nw)
        outs[nw] = cfg(x, t, y).clone()
    monkeypatch.delenv("A2P_CHAIN_NW")
    d = float((outs["4"] - outs["8"]).abs().max())
    record(f"pose_nw/{precision}/B{B}_T{T}", max_abs_diff=d)
    assert torch.equal(outs["4"], outs["8"]), f"max |diff| = {d:.3e}"
    model.release()


# ----------------------------------------------------------------------------- layer-0 work shared by the two guidance halves
@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("fmt,B,T", [("face", 4, 240), ("face", 8, 600), ("pose", 16, 600)])
def test_layer0_shared_half_is_bit_identical_to_the_duplicated_path(dev, fmt, B, T, precision, monkeypatch):
    """Under classifier-free guidance both halves of the 2B sequences enter layer 0 with the same x, so norm1 / Q,K,V / the
    first self attention run once (csrc/a2p_lib_run.h `shared_half`).  Same bits as running them twice -- including grids of
    several rounds, where a second-half workgroup starts after the first-half one has stored its rows (the source rows live in
    a separate buffer for exactly that reason)."""
    spec, sd, model, _ = build(fmt, precision, dev, max_batch=B)
    cfg = ClassifierFreeSampleModel(model)
    inp = synthetic_inputs(spec, B, T, SEED)
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0 if fmt == "face" else 2.0, device=dev)}
    if spec.is_pose:
        y["keyframes"], y["mask"] = inp["keyframes"].to(dev), inp["mask"].to(dev)
    x = inp["x_T"].to(dev)
    t = torch.tensor(([901, 417, 33, 0] * 4)[:B], device=dev)
    outs = {}
    for nw in ("4", "8"):
        monkeypatch.setenv("A2P_CHAIN_NW", nw)
        monkeypatch.delenv("A2P_NO_SHARED_HALF", raising=False)
        shared = cfg(x, t, y).clone()
        monkeypatch.setenv("A2P_NO_SHARED_HALF", "1")
        dup = cfg(x, t, y).clone()
        assert torch.equal(shared, dup), f"NW={nw}: max |diff| = {float((shared - dup).abs().max()):.3e}"
        outs[nw] = shared
    assert torch.equal(outs["4"], outs["8"])
    record(f"layer0_shared/{precision}/{fmt}_B{B}_T{T}", max_abs_diff=0.0)
    model.release()


# ----------------------------------------------------------------------------- two panel heights in one chain launch
@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("B,T", [(11, 600), (32, 600), (13, 592)])
def test_mixed_panel_heights_are_bit_identical_to_the_uniform_launch(dev, B, T, precision, monkeypatch):
    """Forwards of more than 256 48-row panels (face, d=512) launch the chain kernels with 64-row panels for the first workgroups
    so that the grid fills whole rounds of the 256 CUs (csrc/kernels_chain.h `chain_kernel_mix`, a2p_lib_run.h `launch_chain`).
    Same bits as the uniform 48-row launch for both workgroup shapes, ragged tails included."""
    spec, sd, model, _ = build("face", precision, dev, max_batch=B)
    cfg = ClassifierFreeSampleModel(model)
    inp = synthetic_inputs(spec, B, T, SEED)
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0, device=dev)}
    x = inp["x_T"].to(dev)
    t = torch.tensor(([901, 417, 33, 0] * 8)[:B], device=dev)
    assert 2 * B * T > 256 * 48 and (2 * B * T + 47) // 48 % 256 != 0, "shape does not reach the mixed launch"
    outs = {}
    for nw in ("4", "8"):   # the mixed launch exists for the 8-wave shape; the 4-wave runs are the uniform reference
        monkeypatch.setenv("A2P_CHAIN_NW", nw)
        monkeypatch.delenv("A2P_CHAIN_NO_MIX", raising=False)
        mixed = cfg(x, t, y).clone()
        monkeypatch.setenv("A2P_CHAIN_NO_MIX", "1")
        uniform = cfg(x, t, y).clone()
        assert torch.isfinite(mixed).all()
        assert torch.equal(mixed, uniform), f"NW={nw}: max |diff| = {float((mixed - uniform).abs().max()):.3e}"
        outs[nw] = mixed
    assert torch.equal(outs["4"], outs["8"])
    record(f"mixed_panels/{precision}/B{B}_T{T}", max_abs_diff=0.0)
    model.release()
