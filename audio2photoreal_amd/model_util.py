"""Model / diffusion factory with the reference's construction contract
(reference utils/model_util.py:30-114): same function names, same `args` fields."""
import torch
from torch.nn import functional as F

from .diffusion import gaussian_diffusion as gd
from .diffusion.respace import SpacedDiffusion, space_timesteps
from .model.diffusion import FiLMTransformer

_IGNORED_PREFIXES = ("audio_model.", "lip_model.")  # conditioning producers (the audio front end)


def load_model(model, state_dict):
    """Non-strict load with the reference's checks (utils/model_util.py:30-38): no unexpected keys, and only
    `transformer.` / `tokenizer.` keys may be missing.

    Front-end tensors (`audio_model.*`, `lip_model.*`): a model built WITHOUT the native front end is fed `y["cond_embed"]` and
    skips them.  A model built with audio_frontend="native" owns those sub-models and computes the conditioning from them, so a
    tensor that sits ON that path but is not implemented (fairseq's GroupNorm affine terms `conv_layers.{i}.2.*`, the lip
    encoder's `feature_aggregator.*`) must not be dropped silently -- the features would differ from the reference's without
    any error: `model.audio_frontend.check_keys` raises for those."""
    own = set(model.state_dict().keys())
    fe = getattr(model, "audio_frontend", None)
    if fe is not None and hasattr(fe, "check_keys"):
        fe.check_keys(k for k in state_dict if k.startswith(_IGNORED_PREFIXES) and k not in own)
    state_dict = {k: v for k, v in state_dict.items() if not k.startswith(_IGNORED_PREFIXES) or k in own}
    missing_keys, unexpected_keys = model.load_state_dict(state_dict, strict=False)
    assert len(unexpected_keys) == 0, unexpected_keys
    assert all(k.startswith(("transformer.", "tokenizer.")) or k.endswith("rotary.freqs") or
               (k.startswith(_IGNORED_PREFIXES) and k.endswith(".pe")) for k in missing_keys), missing_keys


def create_model_and_diffusion(args, split_type, **model_overrides):
    model = FiLMTransformer(**{**get_model_args(args, split_type=split_type), **model_overrides}).to(torch.float32)
    diffusion = create_gaussian_diffusion(args)
    return model, diffusion


def get_model_args(args, split_type):
    if args.data_format == "face":
        nfeat, lfeat = 256, 512
    elif args.data_format == "pose":
        nfeat, lfeat = 104, 256
    else:
        raise ValueError(args.data_format)
    if not hasattr(args, "num_audio_layers"):
        args.num_audio_layers = 3
    return {
        "args": args, "nfeats": nfeat, "latent_dim": lfeat, "ff_size": 1024, "num_layers": args.layers,
        "num_heads": args.heads, "dropout": 0.1, "cond_feature_dim": 512 * 2, "activation": F.gelu,
        "use_rotary": not args.not_rotary, "cond_mode": "uncond" if args.unconstrained else "audio",
        "split_type": split_type, "num_audio_layers": args.num_audio_layers, "device": args.device,
    }


def create_gaussian_diffusion(args):
    """cosine schedule, 1000 steps, START_X, FIXED_SMALL when sigma_small (reference :79-114)."""
    steps = 1000
    betas = gd.get_named_beta_schedule(args.noise_schedule, steps, 1.0)
    respacing = args.timestep_respacing if args.timestep_respacing else [steps]
    return SpacedDiffusion(
        use_timesteps=space_timesteps(steps, respacing), betas=betas,
        model_mean_type=gd.ModelMeanType.START_X,
        model_var_type=gd.ModelVarType.FIXED_SMALL if args.sigma_small else gd.ModelVarType.FIXED_LARGE,
        data_format=args.data_format, loss_type=gd.LossType.MSE, rescale_timesteps=False,
        lambda_vel=args.lambda_vel, model_path=getattr(args, "save_dir", getattr(args, "model_path", None)))


def default_args(data_format, layers=None, heads=8, timestep_respacing="", max_seq_length=600, device="cuda"):
    """argparse-free equivalent of the fields the factory reads (utils/diff_parser_utils.py)."""
    import argparse
    if layers is None:
        layers = 8 if data_format == "face" else 6
    return argparse.Namespace(
        data_format=data_format, layers=layers, heads=heads, add_frame_cond=1 if data_format == "pose" else None,
        max_seq_length=max_seq_length, not_rotary=False, unconstrained=False, device=device,
        timestep_respacing=timestep_respacing, noise_schedule="cosine", sigma_small=True, lambda_vel=0.0,
        model_path="synthetic", resume_trans=None)
