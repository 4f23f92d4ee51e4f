"""Classifier-free-guidance wrapper for SAMPLING (reference model/cfg_sampler.py:17-33).

Same constructor/attribute/forward contract; the two denoiser passes the reference
runs back to back (cond_drop_prob 0 and 1) are batched as 2B sequences in one
launch sequence of the HIP library, followed by the guidance lerp kernel.
"""
import torch.nn as nn

from .. import _lib


class ClassifierFreeSampleModel(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model
        self.nfeats = self.model.nfeats
        self.cond_mode = self.model.cond_mode
        self.add_frame_cond = self.model.add_frame_cond
        if self.add_frame_cond is not None:
            if self.model.resume_trans is not None:
                self.transformer = self.model.transformer
                self.tokenizer = self.model.tokenizer
            self.step = self.model.step

    def forward(self, x, timesteps, y=None):
        # out_uncond + y["scale"].view(-1,1,1) * (out - out_uncond)
        return self.model.forward_cfg(x, timesteps, y)

    # fused p_mean_variance + posterior update, used by SpacedDiffusion's loops when available
    def a2p_sample_step(self, sampler, x, t_idx, timestep_map, tables, y, noise, eta, clip_denoised):
        return self.model.sample_step(sampler, x, t_idx, timestep_map, tables, y, noise, eta, clip_denoised)


SAMPLER_DDIM, SAMPLER_DDPM = _lib.SAMPLER_DDIM, _lib.SAMPLER_DDPM
