import sys, torch
sys.path.insert(0, '.')
from audio2photoreal_amd import _lib
from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args
dev = torch.device('cuda:0')
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
model, _ = create_model_and_diffusion(default_args('face', layers=1), 'test', precision='fp32', max_batch=2)
model = model.to(dev); model._ensure_ctx(dev, 1); lib = _lib.load()
d, H, dh = 512, 8, 64
N, Tq, S = 1, 16, 16
def run(q, k, v):
    out = torch.empty(N, Tq, d, device=dev)
    _lib.check(lib.a2p_attention(model._ctx, _lib.ptr(q.to(dev)), _lib.ptr(k.to(dev)), _lib.ptr(v.to(dev)), _lib.ptr(out), N, Tq, S, _lib.current_stream()), "attn")
    return out.cpu()
q = torch.zeros(N, Tq, d); k = torch.zeros(N, S, d)
v = torch.arange(S).float()[None, :, None].expand(N, S, d).contiguous()
o = run(q, k, v); print("V=key index (expect 7.5):", o[0, :4, :8], o[0, :, 70].tolist())
v = torch.arange(d).float()[None, None, :].expand(N, S, d).contiguous()
o = run(q, k, v); print("V=col index (expect col):", o[0, 0, :20].tolist(), o[0, 5, 60:70].tolist())
# one-hot keys: q = e_a, k_s = big*e_a for s = 3 -> picks key 3
q = torch.zeros(N, Tq, d); q[..., 0] = 10.0; k = torch.zeros(N, S, d); k[0, 3, 0] = 10.0
v = torch.arange(S).float()[None, :, None].expand(N, S, d).contiguous()
o = run(q, k, v); print("head0 should be ~3, others 7.5:", o[0, :3, 0].tolist(), o[0, :3, 64].tolist())
