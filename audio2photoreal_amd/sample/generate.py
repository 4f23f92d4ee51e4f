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

import itertools

from ..sample_parallel import derive_seed, per_sample_noise, sample_parallel, shared_base_seed

_CALLS = itertools.count()   # sampling calls of this process: every rank makes the same calls in the same order


def fixseed(seed: int) -> None:
    """utils/misc.py:138-142."""
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def load_data_stats(path: str) -> Dict[str, np.ndarray]:
    """A subject's `data_stats.pth` (numpy arrays `pose_mean/std[104]`, `code_mean/std[256]`, `audio_mean[2]`,
    `audio_std_flat[1]`, ...; data_loaders/data.py:100-110)."""
    return torch.load(path, map_location="cpu", weights_only=False)


def make_inv_transform(stats: Dict[str, np.ndarray]) -> Callable:
    """`Social.inv_transform` (data_loaders/data.py:71-91) over the statistics `Social._load_std` picks (:100-110): pose and
    face use the per-channel mean / std, audio the per-channel mean and the FLAT std.  Like the reference, tensors are scaled
    by `torch.tensor(std)` in the statistics' own dtype, so fp32 pose / face data comes back as float64."""
    table = {"pose": (np.asarray(stats["pose_std"]).reshape(-1), np.asarray(stats["pose_mean"]).reshape(-1)),
             "face": (np.asarray(stats["code_std"]), np.asarray(stats["code_mean"])),
             "audio": (np.asarray(stats["audio_std_flat"]), np.asarray(stats["audio_mean"]))}

    def inv(data, data_type: str):
        assert data_type in table, f"datatype not defined: {data_type}"
        std, mean = table[data_type]
        if torch.is_tensor(data):
            return data * torch.tensor(std, device=data.device, requires_grad=False) \
                + torch.tensor(mean, device=data.device, requires_grad=False)
        return data * std + mean
    return inv


def _setup_model(args, state_dict, guide=None, **model_kwargs):
    """Model + diffusion for sampling (reference sample/generate.py:165-198, with the checkpoint passed in instead of read
    from args.model_path): build, load, wrap for classifier-free guidance, move to args.device, eval.
    `guide=(transformer, tokenizer)`: the guide transformer and VQ tokenizer of the body model -- the reference's constructor
    loads them from `args.resume_trans` (model/diffusion.py:244-271); here the built modules are passed in."""
    from ..model.cfg_sampler import ClassifierFreeSampleModel
    from ..model_util import create_model_and_diffusion, load_model
    model, diffusion = create_model_and_diffusion(args, "test", **model_kwargs)
    load_model(model, state_dict)
    if guide is not None:
        model.setup_guide_predictor(*guide, resume_trans=getattr(args, "resume_trans", None) or "<in-memory>")
    if not getattr(args, "unconstrained", False):
        assert args.guidance_param != 1
    if args.guidance_param != 1:
        model = ClassifierFreeSampleModel(model)
    model.to(args.device)
    model.eval()
    return model, diffusion


def save_results(output_dir: str, data_block: Dict[str, np.ndarray]) -> str:
    """`results.npy` exactly as the reference writes it (sample/generate.py:282-285): a pickled dict."""
    import os
    os.makedirs(output_dir, exist_ok=True)
    npy_path = os.path.join(output_dir, "results.npy")
    np.save(npy_path, data_block)
    return npy_path


def load_results(npy_path: str) -> Dict[str, np.ndarray]:
    """reference sample/generate.py:288."""
    return np.load(npy_path, allow_pickle=True).item()


def _replace_keyframes(model_kwargs, model, uniforms: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Keyframes predicted by the guide transformer instead of ground truth (reference sample/generate.py:51-71):
    `model.transformer.generate` -> `[B, T, residual_depth]` tokens -> `model.tokenizer.decode`.  The condition is
    `y["cond_embed"]` (audio features) when present, else `y["audio"]` through the transformer's `audio_frontend`."""
    y = model_kwargs["y"]
    B, T = y["keyframes"].shape[0], y["keyframes"].shape[1]
    cond = y["cond_embed"] if "cond_embed" in y else y["audio"]
    with torch.no_grad():
        tokens = model.transformer.generate(cond, T, layers=model.tokenizer.residual_depth, n_sequences=B, max_key_len=T,
                                            max_seq_len=30 * T, uniforms=uniforms)
    tokens = tokens.reshape((B, -1, model.tokenizer.residual_depth))
    pred = model.tokenizer.decode(tokens).detach().cpu()
    assert y["keyframes"].shape == pred.shape, f"{y['keyframes'].shape} vs {pred.shape}"
    return pred


def _run_single_diffusion(args, model_kwargs, diffusion, model, inv_transform: Callable, gt: Optional[torch.Tensor],
                          noise: Optional[torch.Tensor] = None, rep_i: Optional[int] = None, call_i: Optional[int] = None):
    """One ddim_sample_loop over the (possibly rank-sharded) batch + un-normalisation (reference :74-107).

    Under torch.distributed every random draw is a function of (shared base seed, repetition, GLOBAL sample id): the initial
    noise via `per_sample_noise`, the guide transformer's uniforms from one generator all ranks seed alike -- so the gathered
    result does not depend on the world size, the keyframes rank 0 saves are the ones every rank conditioned on, and every
    repetition of every sampling call draws fresh noise like the reference's `randn` does: seed = derive_seed(base, call,
    repetition, global id), where `call` counts the sampling calls of this process (`call_i`; `_generate_sequences` takes one
    number for all its repetitions; every rank makes the same calls in the same order)."""
    import torch.distributed as dist
    sharded = dist.is_available() and dist.is_initialized()
    base = shared_base_seed() if sharded else None
    call = next(_CALLS) if call_i is None else int(call_i)
    rep = 0 if rep_i is None else int(rep_i)
    has_guide = getattr(args, "resume_trans", None) is not None or getattr(model, "resume_trans", None) is not None
    if args.data_format == "pose" and has_guide:   # reference :82-83
        y = model_kwargs["y"]
        uniforms = None
        if sharded:
            n = y["keyframes"].shape[1] * model.tokenizer.residual_depth
            uniforms = torch.rand(n, y["keyframes"].shape[0], generator=torch.Generator().manual_seed(derive_seed(base, call, rep, 0x6775696465)))
        y["keyframes"] = _replace_keyframes(model_kwargs, model, uniforms).to(y["keyframes"].device)
    shape = (args.batch_size, model.nfeats, 1, args.curr_seq_length)
    if noise is None and sharded:
        noise = per_sample_noise(shape, [derive_seed(base, call, rep, g) for g in range(shape[0])])
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
    call_i = next(_CALLS)   # two calls of one process (another clip, or the same clip again) must not reuse each other's noise
    for rep_i in range(args.num_repetitions):
        if args.guidance_param != 1:
            model_kwargs["y"]["scale"] = torch.ones(args.batch_size, device=args.device) * args.guidance_param
        model_kwargs["y"] = {k: v.to(args.device) if torch.is_tensor(v) else v for k, v in model_kwargs["y"].items()}
        sample, curr_audio, keyframes, gt_seq = _run_single_diffusion(args, model_kwargs, diffusion, model, inv_transform, gt,
                                                                      rep_i=rep_i, call_i=call_i)
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
