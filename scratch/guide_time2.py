"""Where does the AR kernel's time go: generate at full (1998) vs tiny (100) audio-token counts, B=8."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio2photoreal_amd.model.guide import GuideTransformer
from audio2photoreal_amd.spec import GuideSpec
from audio2photoreal_amd.synthetic import synthetic_guide_state_dict, synthetic_tensor
dev = torch.device("cuda:0")
gs = GuideSpec()
B = 8
g = GuideTransformer(tokens=gs.tokens, num_layers=gs.num_layers, dim=gs.dim, emb_len=gs.emb_len, num_audio_layers=gs.num_audio_layers, max_batch=B, max_positions=96)
g.load_state_dict(synthetic_guide_state_dict(gs, 10), strict=False)
g = g.to(dev).eval()
u = torch.rand(80, B, device=dev)
for S in (1998, 100):
    cond = synthetic_tensor(10, "guide_cond_full", (B, S, gs.cond_feature_dim)).to(dev)
    g.generate(cond, 20, 4, n_sequences=B, max_key_len=20, max_seq_len=600, uniforms=u); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g.generate(cond, 20, 4, n_sequences=B, max_key_len=20, max_seq_len=600, uniforms=u)
    torch.cuda.synchronize()
    print(f"S={S}: generate {((time.perf_counter()-t0)/3)*1e3:.2f} ms", flush=True)
