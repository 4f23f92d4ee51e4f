"""Orchestration of sampling with the reference's function surface
(reference sample/generate.py:74-152: `_run_single_diffusion`, `_generate_sequences`,
the results dict keys `motions/audio/gt/lengths/keyframes`).

Dataset loading, the guide transformer (`_replace_keyframes`) and rendering are outside the
accelerated path; callers pass `model_kwargs` directly
(y: cond_embed|audio, keyframes, mask, lengths ...) and an `inv_transform` callable
(data_loaders/data.py:71-110 semantics: x * std + mean).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch

from ..sample_parallel import sample_parallel


def make_inv_transform(stats: Dict[str, np.ndarray]) -> Callable:
    """`Social.inv_transform` (data_loaders/data.py:71-98) from a data_stats.pth dict."""
    def inv(data, data_type: str):
        if data_type == "pose":
            std, mean = stats["pose_std"], stats["pose_mean"]
        elif data_type == "face":
            std, mean = stats["code_std"], stats["code_mean"]
        elif data_type == "audio":
            std, mean = stats["audio_std"], stats["audio_mean"]
        else:
            raise ValueError(f"unknown data type {data_type}")
        if torch.is_tensor(data):
            return data * torch.as_tensor(std, dtype=data.dtype, device=data.device) + torch.as_tensor(mean, dtype=data.dtype, device=data.device)
        return data * std + mean
    return inv


def _run_single_diffusion(args, model_kwargs, diffusion, model, inv_transform: Callable, gt: Optional[torch.Tensor],
                          noise: Optional[torch.Tensor] = None):
    """One ddim_sample_loop over the (possibly rank-sharded) batch + un-normalisation (reference :74-107)."""
    shape = (args.batch_size, model.nfeats, 1, args.curr_seq_length)
    with torch.no_grad():
        sample = sample_parallel(diffusion.ddim_sample_loop, model, shape, model_kwargs, noise=noise,
                                 clip_denoised=False, init_image=None, progress=False, dump_steps=None, const_noise=False)
    sample = inv_transform(sample.cpu().permute(0, 2, 3, 1), args.data_format).permute(0, 3, 1, 2)
    y = model_kwargs["y"]
    curr_audio = inv_transform(y["audio"].cpu().numpy(), "audio") if "audio" in y else None
    keyframes = inv_transform(y["keyframes"].cpu(), args.data_format) if "keyframes" in y else None
    gt_seq = None if gt is None else inv_transform(gt.cpu().permute(0, 2, 3, 1), args.data_format).permute(0, 3, 1, 2)
    return sample, curr_audio, keyframes, gt_seq


def _generate_sequences(args, model_kwargs, diffusion, model, inv_transform: Callable, gt: Optional[torch.Tensor] = None):
    """Repetition loop + results dict (reference :110-152)."""
    motions, lengths, audio, gts, kfs = [], [], [], [], []
    for rep_i in range(args.num_repetitions):
        if args.guidance_param != 1:
            model_kwargs["y"]["scale"] = torch.ones(args.batch_size, device=args.device) * args.guidance_param
        model_kwargs["y"] = {k: v.to(args.device) if torch.is_tensor(v) else v for k, v in model_kwargs["y"].items()}
        sample, curr_audio, keyframes, gt_seq = _run_single_diffusion(args, model_kwargs, diffusion, model, inv_transform, gt)
        motions.append(sample.cpu().numpy())
        if curr_audio is not None:
            audio.append(curr_audio)
        if keyframes is not None:
            kfs.append(keyframes.cpu().numpy())
        if gt_seq is not None:
            gts.append(gt_seq.cpu().numpy())
        if "lengths" in model_kwargs["y"]:
            lengths.append(model_kwargs["y"]["lengths"].cpu().numpy())
    cat = lambda xs: np.concatenate(xs, axis=0) if xs else None
    return {"motions": cat(motions), "audio": cat(audio), "gt": cat(gts), "lengths": cat(lengths), "keyframes": cat(kfs)}
