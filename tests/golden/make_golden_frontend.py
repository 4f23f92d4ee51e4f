"""Golden vectors for the audio front end, produced by the REFERENCE itself: the unmodified
`FiLMTransformer.encode_audio` / `encode_lip` (model/diffusion.py:285-313) with `Audio2LipRegressionTransformer`
(:37-79, built by the reference's own constructor) on the synthetic weights of audio2photoreal_amd.synthetic.
fairseq / torchaudio are the stubs of ref_import.py (conv + ReLU stack with the vq-wav2vec geometry,
x[::3] resampler) -- those two pieces are therefore NOT pinned by these fixtures, everything downstream of them is.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_frontend.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_import as ri  # noqa: E402
from audio2photoreal_amd.synthetic import synthetic_audio, synthetic_frontend_state_dict  # noqa: E402

SEED, B, FRAMES = 10, 2, 240          # 240 frames = 2 lip chunks of 120; 798 audio tokens


def to_reference_keys(sd):
    """product / fairseq naming `...feature_extractor.conv_layers.{i}.0.weight` -> the stub's `...net.{2i}.weight`."""
    out = {}
    for k, v in sd.items():
        if "feature_extractor.conv_layers." in k:
            head, tail = k.split("feature_extractor.conv_layers.")
            out[f"{head}net.{2 * int(tail.split('.')[0])}.weight"] = v
        else:
            out[k] = v
    return out


def main():
    torch.manual_seed(SEED)
    torch.set_num_threads(8)
    ns = ri.import_reference()
    with ri.cpu_cuda(), torch.no_grad():
        model, _ = ri.build_reference_model(ns, "face", 1, 8, "ddim10")
        model.lip_model = ns.md.Audio2LipRegressionTransformer().eval()      # the reference's constructor (the checkpoint file is absent)
        sd = to_reference_keys(synthetic_frontend_state_dict(SEED, lip=True))
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert not [k for k in missing if k.startswith(("audio_model.", "lip_model."))], missing
        audio = synthetic_audio(SEED, B, FRAMES)
        emb = model._ref_encode_audio(audio)                                   # [B, 798, 1024]
        full = model._ref_encode_lip(audio, emb)                               # [B, 798, 2038]
        lip = torch.zeros(B, FRAMES, 338, 3)
        reshaped = audio.reshape((B, -1, 1600, 2))[..., 0]
        for i in range(0, FRAMES, 120):
            lip[:, i:i + 120] = model.lip_model(reshaped[:, i:i + 120])
    out = {"emb_rows16": emb[:, ::16].contiguous().numpy(), "emb_norm": np.array(float(emb.norm())),
           "lip_frames4": lip[:, ::4].reshape(B, -1, 1014).contiguous().numpy(),
           "full_rows16": full[:, ::16].contiguous().numpy(), "full_norm": np.array(float(full.norm())),
           "shape": np.array(full.shape)}
    path = os.path.join(HERE, "golden_frontend_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()}, os.path.getsize(path))


if __name__ == "__main__":
    main()
