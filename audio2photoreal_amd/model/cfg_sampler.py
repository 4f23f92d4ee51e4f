"""Classifier-free-guidance wrapper for SAMPLING (contract of reference model/cfg_sampler.py:17-33).

Same constructor / attribute / `forward(x, timesteps, y)` surface.  The two denoiser passes the reference runs
back to back (`cond_drop_prob` 0 and 1) are batched here as 2B sequences through one launch sequence of the HIP
library; the guidance lerp `uncond + y["scale"] * (cond - uncond)` is fused into the kernel that also transposes
the result (csrc/kernels_misc.h: step_tail_kernel).
"""
import torch.nn as nn

from .. import _lib

SAMPLER_DDIM, SAMPLER_DDPM = _lib.SAMPLER_DDIM, _lib.SAMPLER_DDPM

# attributes callers read from the wrapper instead of the wrapped denoiser:
# diffusion/respace.py:133-135 (add_frame_cond, step), sample/generate.py:60-67,89 (nfeats, transformer, tokenizer)
_ALWAYS = ("nfeats", "cond_mode", "add_frame_cond")
_WITH_GUIDE = ("transformer", "tokenizer")


class ClassifierFreeSampleModel(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model
        for name in _ALWAYS:
            setattr(self, name, getattr(model, name))
        if self.add_frame_cond is None:
            return
        self.step = model.step
        if model.resume_trans is not None:          # guide transformer + VQ tokenizer ride along (pose model only)
            for name in _WITH_GUIDE:
                setattr(self, name, getattr(model, name))

    def forward(self, x, timesteps, y=None):
        return self.model.forward_cfg(x, timesteps, y)

    def a2p_sample_step(self, sampler, x, t_idx, timestep_map, tables, y, noise, eta, clip_denoised):
        """p_mean_variance + ddim_sample / p_sample in one library call; SpacedDiffusion's loops use it when present."""
        return self.model.sample_step(sampler, x, t_idx, timestep_map, tables, y, noise, eta, clip_denoised)
