"""CPU baseline variant (A) of BASELINE.md section 3 -- the REFERENCE ITSELF, as is -- timed beside the oracle port that bench.py's
`cpu_baseline` leg times on the GPU box (the reference tree does not travel there).  Runs only in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/tools/cpu_reference_vs_port.py [--frames 600] [--steps 4]

Same synthetic weights and inputs for both (audio2photoreal_amd.synthetic), the bench workload's shape for ONE sample (face model,
T frames, 1998+2 conditioning tokens at T=600, classifier-free guidance = 2 passes), `p_sample` steps at the head of the 1000-step
chain, the same torch thread count.  The reference runs through tests/golden/ref_import.py (fairseq / torchaudio stubbed, audio
features fed in: decoder-only conditioning, the variant (B) path of BASELINE.md) -- what differs from the port is the reference's own
module code (nn.MultiheadAttention, einops rearranges, per-step recomputation of everything), not the algorithm.
Writes profiles/r03_cpu_reference_vs_port.json; the outputs of both are compared too (they are the parity pin of the oracle).

Round 4 (`--variant-a`): BASELINE.md section 3's variant (A) proper -- what a user of the reference gets: `y["audio"]` goes in and
the reference's UNMODIFIED `encode_audio` / `encode_lip` (model/diffusion.py:285-313, the reference's own Audio2LipRegressionTransformer)
re-encode it in every forward of every step (fairseq / torchaudio: the conv + ReLU / x[::3] stubs of ref_import.py, synthetic
weights).  Timed beside the decoder-only variant (B) above; writes profiles/r04_cpu_reference_variants.json."""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)

import ref_import as ri  # noqa: E402
from audio2photoreal_amd.spec import face_spec  # noqa: E402
from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict  # noqa: E402
from oracle import a2p_oracle as O  # noqa: E402

SEED = 10


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--variant-a", action="store_true", help="also time the reference with its own audio front end in every forward")
    a = ap.parse_args()
    torch.manual_seed(SEED)
    torch.set_num_threads(a.threads)
    spec = face_spec()
    sd = synthetic_state_dict(spec, SEED)
    inp = synthetic_inputs(spec, 1, a.frames, SEED, steps_of_noise=a.steps + 1)
    scale = 10.0
    ts = [999 - i for i in range(a.steps)]

    # ---- the reference, as is ----
    ns = ri.import_reference()
    with ri.cpu_cuda():
        model, diff = ri.build_reference_model(ns, "face", spec.num_layers, spec.num_heads, "")
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        cfg_model = ns.cfg.ClassifierFreeSampleModel(model)
        model._a2p_cond_embed = inp["cond_embed"]
        y = {"audio": torch.zeros(1, 1, 2), "scale": torch.full((1,), scale)}

        def ref_steps(n_warm):
            x = inp["x_T"].clone()
            outs = []
            with torch.no_grad():
                for k in range(n_warm):
                    diff._a2p_step_noise = [inp["step_noise"][a.steps].clone()]
                    diff.p_sample(cfg_model, x, torch.tensor([999]), clip_denoised=False, model_kwargs={"y": y})
                t0 = time.perf_counter()
                for k, t in enumerate(ts):
                    diff._a2p_step_noise = [inp["step_noise"][k].clone()]
                    out = diff.p_sample(cfg_model, x, torch.tensor([t]), clip_denoised=False, model_kwargs={"y": y})
                    x = out["sample"]
                    outs.append(x)
                return time.perf_counter() - t0, outs
        ref_dt, ref_out = ref_steps(1)
        var_a = None
        if a.variant_a:
            sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
            from make_golden_frontend import to_reference_keys
            from audio2photoreal_amd.synthetic import synthetic_audio, synthetic_frontend_state_dict
            model.lip_model = ns.md.Audio2LipRegressionTransformer().eval()
            miss, unexp = model.load_state_dict(to_reference_keys(synthetic_frontend_state_dict(SEED, lip=True)), strict=False)
            assert not unexp, unexp
            enc_b = (type(model).encode_audio, type(model).encode_lip)
            type(model).encode_audio, type(model).encode_lip = type(model)._ref_encode_audio, type(model)._ref_encode_lip
            y_keep = y
            y = {"audio": synthetic_audio(SEED, 1, a.frames), "scale": torch.full((1,), scale)}
            try:
                n_a = min(a.steps, 2)
                ts_keep, ts = ts, ts[:n_a]
                a_dt, _ = ref_steps(1)
                var_a = {"s_per_step": round(a_dt / n_a, 3), "steps_timed": n_a}
                ts = ts_keep
            finally:
                type(model).encode_audio, type(model).encode_lip = enc_b
                y = y_keep

    # ---- the oracle port ----
    den = O.OracleDenoiser(sd, "face", spec.num_layers, spec.num_heads)
    smp = O.OracleSampler("")
    fn = lambda xx, tt: den.forward_cfg(xx, tt, inp["cond_embed"], torch.full((1,), scale))
    with torch.no_grad():
        smp.p_sample(fn, inp["x_T"].clone(), torch.tensor([999]), inp["step_noise"][a.steps])
        x = inp["x_T"].clone()
        port_out = []
        t0 = time.perf_counter()
        for k, t in enumerate(ts):
            x = smp.p_sample(fn, x, torch.tensor([t]), inp["step_noise"][k])["sample"]
            port_out.append(x)
        port_dt = time.perf_counter() - t0
    rel = max(float((p - r).norm() / r.norm()) for p, r in zip(port_out, ref_out))
    rec = {"what": "one sample of the bench workload on the build container's CPU: the reference as is vs the oracle port",
           "shape": f"face, T={a.frames}, S={inp['cond_embed'].shape[1]}+2, guidance (2 passes), p_sample at t={ts}",
           "threads": a.threads, "host_cpus": os.cpu_count(),
           "reference_s_per_step": round(ref_dt / a.steps, 3), "port_s_per_step": round(port_dt / a.steps, 3),
           "reference_steps_per_s_batch1": round(a.steps / ref_dt, 4), "port_steps_per_s_batch1": round(a.steps / port_dt, 4),
           "port_over_reference_speed": round(ref_dt / port_dt, 3),
           "port_vs_reference_rel_l2_after_steps": rel,
           "note": "bench.py's cpu_baseline times the port on the GPU box's host (kind 'port'); this ratio relates it to variant (A)"}
    if var_a:
        rec["variant_a_reference_with_its_audio_front_end"] = {
            **var_a, "steps_per_s_batch1": round(1.0 / var_a["s_per_step"], 4),
            "slowdown_vs_decoder_only_reference": round(var_a["s_per_step"] / (ref_dt / a.steps), 2),
            "note": "the reference's unmodified encode_audio / encode_lip in every forward (2 per step); stub conv stack + x[::3] resampler, the "
                    "reference's own lip regressor; same threads"}
    print(json.dumps(rec, indent=1))
    with open(os.path.join(ROOT, "profiles", "r04_cpu_reference_variants.json" if var_a else "r03_cpu_reference_vs_port.json"), "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
