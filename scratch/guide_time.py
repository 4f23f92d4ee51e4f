"""Timing of the guide path (hoisted prepare + one-launch generate) at the 600-frame geometry."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio2photoreal_amd.model.guide import GuideTransformer
from audio2photoreal_amd.spec import GuideSpec
from audio2photoreal_amd.synthetic import synthetic_guide_state_dict, synthetic_tensor
dev = torch.device("cuda:0")
gs = GuideSpec()
for B in (tuple(int(v) for v in os.environ.get("GUIDE_B", "1,8,32").split(","))):
    g = GuideTransformer(tokens=gs.tokens, num_layers=gs.num_layers, dim=gs.dim, emb_len=gs.emb_len, num_audio_layers=gs.num_audio_layers,
                         max_batch=B, max_positions=96)
    g.load_state_dict(synthetic_guide_state_dict(gs, 10), strict=False)
    g = g.to(dev).eval()
    cond = synthetic_tensor(10, "guide_cond_full", (B, 1998, gs.cond_feature_dim)).to(dev)
    u = torch.rand(80, B, device=dev)
    g.generate(cond, 20, 4, n_sequences=B, max_key_len=20, max_seq_len=600, uniforms=u)
    torch.cuda.synchronize()
    ts = []
    for i in range(3):
        cond2 = cond + i  # new tensor -> prepare runs again
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g._prepare(cond2, 0.0); torch.cuda.synchronize(); t1 = time.perf_counter()
        g.generate(cond2, 20, 4, n_sequences=B, max_key_len=20, max_seq_len=600, uniforms=u); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    p, a = min(t[0] for t in ts), min(t[1] for t in ts)
    gf = B * (12 * 1998 * 1024 * 1024 * 3 * 2 + 1998 * 1024 * 1024 * 2) / 1e9
    print(f"B={B}: prepare {p*1e3:.2f} ms ({gf/p/1e3:.1f} TFLOP/s fp32 on the conv stack), generate (80 steps, one launch) {a*1e3:.2f} ms = {a/80*1e6:.0f} us/step", flush=True)
    del g
