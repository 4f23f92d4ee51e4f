"""Round 5 GPU tests: the tall chain kernels (csrc/kernels_chain4.h)."""
import pytest
import torch

from audio2photoreal_amd import _lib
from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
from audio2photoreal_amd.spec import face_spec
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict
from conftest import record

pytestmark = pytest.mark.gpu
SEED = 10


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("B,T,mt", [(2, 600, 5), (3, 208, 4), (2, 600, 3), (1, 88, 5), (16, 600, 0), (5, 320, 0)])
def test_tall_chain_kernels_are_bit_identical_to_the_48_row_kernels(dev, B, T, mt, precision, monkeypatch):
    """csrc/kernels_chain4.h (64 / 80-row panels, weights straight from L2 into a register ring of half stages, epilogue operands from
    LDS, residual rows parked around the feed-forward block, stored GEMMs in k-chunk-major tile pairs) against kernels_chain.h: the same
    column ownership, accumulation order, LayerNorm tree and epilogue arithmetic, hence the SAME BITS for the guided forward.
    Forced panel heights (A2P_CHAIN_MT) on small batches cover 80 / 64 / 48 rows with ragged last panels, panels that straddle two
    sequences (T = 208, 88, 320) and a clip barely longer than a panel (T = 88 >= 80); mt = 0 lets the library choose (B = 16: 80 rows,
    its own rule; B = 5 x 320: A2P_CHAIN_V=4 with the library's height)."""
    spec = face_spec()
    inp = synthetic_inputs(spec, B, T, SEED)
    model, _ = create_model_and_diffusion(default_args("face"), "test", precision=precision, max_batch=B)
    load_model(model, synthetic_state_dict(spec, SEED))
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    y = {"cond_embed": inp["cond_embed"].to(dev), "scale": torch.full((B,), 10.0, device=dev)}
    t = torch.tensor(([901, 417, 33, 650] * 4)[:B], device=dev)
    import ctypes as C

    def tall_launches():
        n = C.c_int64(0)
        _lib.check(model._lib().a2p_debug_read(model._ctx, b"chain4_launches", C.byref(n), 8), "a2p_debug_read")
        return int(n.value)
    outs, launched = {}, {}
    for name, v in (("gen1", "1"), ("tall", "4")):
        monkeypatch.setenv("A2P_CHAIN_V", v)
        if mt:
            monkeypatch.setenv("A2P_CHAIN_MT", str(mt))
        else:
            monkeypatch.delenv("A2P_CHAIN_MT", raising=False)
            if 2 * B * T < 1100:
                monkeypatch.setenv("A2P_CHAIN_ROWS", "1")
        before = tall_launches() if model._ctx is not None else 0
        outs[name] = cfg(inp["x_T"].to(dev), t, y).cpu()
        launched[name] = tall_launches() - before
    for k in ("A2P_CHAIN_V", "A2P_CHAIN_MT", "A2P_CHAIN_ROWS"):
        monkeypatch.delenv(k, raising=False)
    model.check_finite()
    model.release()
    diff = float((outs["tall"] - outs["gen1"]).abs().max())
    record(f"tall_chain/{precision}/B{B}_T{T}_mt{mt}", max_abs_diff=diff)
    # the two runs really were different kernels: 8 MID + 8 POST launches of the tall family per guided forward (the last layer's POST
    # without the next layer's projections), none under A2P_CHAIN_V=1
    # (round 6: + the input projection / PRE kernel of layer 0 where the clip is long enough for it: >= 80 frames)
    assert launched == {"gen1": 0, "tall": 17 if T >= 80 else 16}, launched
    assert torch.isfinite(outs["tall"]).all()
    assert torch.equal(outs["tall"], outs["gen1"]), diff
