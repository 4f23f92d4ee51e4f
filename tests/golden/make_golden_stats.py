"""Fixture for the un-normalisation: the statistics the reference's `Social._load_std` reads from a
subject's data_stats.pth (data_loaders/data.py:100-110) and the result of `Social.inv_transform` (:71-91) on seeded inputs.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stats.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import as ri  # noqa: E402


def main():
    ri.install_stubs()
    sys.dont_write_bytecode = True
    sys.path.insert(0, ri.REF)
    import data_loaders.data as rd   # the reference's own inv_transform / _load_std
    stats = torch.load(os.path.join(ri.REF, "dataset", "PXB184", "data_stats.pth"), weights_only=False)
    keep = ("pose_mean", "pose_std", "code_mean", "code_std", "audio_mean", "audio_std", "audio_std_flat")
    out = {f"stats/{k}": np.asarray(stats[k]) for k in keep}
    obj = types.SimpleNamespace(data_root=os.path.join(ri.REF, "dataset", "PXB184"))
    orig_load = torch.load                                    # the reference predates torch's weights_only=True default
    torch.load = lambda *a, **k: orig_load(*a, **{**k, "weights_only": False})
    rd.Social._load_std(obj)                                  # fills mean/std/face_*/audio_* exactly as the dataset does
    torch.load = orig_load
    g = torch.Generator().manual_seed(7)
    pose = torch.randn(2, 5, 1, 104, generator=g)             # [B, T, 1, C] as _run_single_diffusion passes it
    face = torch.randn(2, 5, 1, 256, generator=g)
    audio = torch.randn(2, 8000, 2, generator=g).numpy()
    out["in/pose"], out["in/face"], out["in/audio"] = pose.numpy(), face.numpy(), audio
    out["out/pose"] = rd.Social.inv_transform(obj, pose, "pose").numpy()
    out["out/face"] = rd.Social.inv_transform(obj, face, "face").numpy()
    out["out/audio"] = rd.Social.inv_transform(obj, audio, "audio")
    np.savez(os.path.join(HERE, "golden_stats_v1.npz"), **out)
    print({k: (v.dtype, v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
