import sys, torch
sys.path.insert(0, '.')
from audio2photoreal_amd import _lib
from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args
dev = torch.device('cuda:0')
for prec in ('fp32', 'bf16'):
    model, _ = create_model_and_diffusion(default_args('face', layers=1), 'test', precision=prec, max_batch=2)
    model = model.to(dev)
    model._ensure_ctx(dev, 1)
    lib = _lib.load()
    d, H = 512, 8
    g = torch.Generator().manual_seed(2)
    for (N, Tq, S, mode) in [(1, 16, 16, 'rand'), (1, 16, 16, 'q0'), (1, 128, 64, 'rand'), (1, 16, 64, 'rand'), (1, 16, 77, 'rand'), (2, 100, 77, 'rand'), (1, 240, 800, 'rand')]:
        q, k, v = (torch.randn(N, L, d, generator=g) for L in (Tq, S, S))
        if mode == 'q0': q.zero_()
        dh = d // H
        qh, kh, vh = (t.view(N, -1, H, dh).transpose(1, 2).double() for t in (q, k, v))
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) / dh ** 0.5, -1) @ vh).transpose(1, 2).reshape(N, Tq, d)
        out = torch.empty(N, Tq, d, device=dev)
        _lib.check(lib.a2p_attention(model._ctx, _lib.ptr(q.to(dev)), _lib.ptr(k.to(dev)), _lib.ptr(v.to(dev)), _lib.ptr(out), N, Tq, S, _lib.current_stream()), "attn")
        o = out.cpu().double()
        err = float((o - ref).norm() / ref.norm())
        perhead = [(float((o[..., h*dh:(h+1)*dh] - ref[..., h*dh:(h+1)*dh]).norm() / ref[..., h*dh:(h+1)*dh].norm())) for h in range(H)]
        perq = [float((o[:, i] - ref[:, i]).norm() / ref[:, i].norm()) for i in range(min(Tq, 20))]
        print(prec, N, Tq, S, mode, 'err %.3e' % err, 'heads', ['%.1e' % e for e in perhead[:4]], 'q', ['%.1e' % e for e in perq[:6]], flush=True)
