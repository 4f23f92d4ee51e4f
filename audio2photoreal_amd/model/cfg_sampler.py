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
        # The reference copies model.transformer / model.tokenizer here when resume_trans is set (cfg_sampler.py:25-28), because
        # its FiLMTransformer builds them from files in its constructor.  Here they are attached with
        # `setup_guide_predictor` -- possibly AFTER wrapping -- so the wrapper resolves them lazily (see __getattr__).

    def __getattr__(self, name):
        # nn.Module.__getattr__ covers parameters / buffers / sub-modules; the guide modules are looked up on the wrapped
        # denoiser at access time, so `_setup_model(...)` followed by `model.model.setup_guide_predictor(...)` works too
        if name in _WITH_GUIDE or name == "resume_trans":
            inner = super().__getattr__("model")
            if name == "resume_trans":
                return getattr(inner, "resume_trans", None)
            try:
                return nn.Module.__getattr__(inner, name)
            except AttributeError:
                raise AttributeError(f"the wrapped denoiser has no '{name}': attach the guide transformer / tokenizer with "
                                     "FiLMTransformer.setup_guide_predictor(transformer, tokenizer) (pose model)") from None
        return super().__getattr__(name)

    def forward(self, x, timesteps, y=None):
        return self.model.forward_cfg(x, timesteps, y)

    def a2p_check_finite(self):
        """Raise A2PError when a denoiser evaluation since the last check produced inf / nan (FiLMTransformer.check_finite)."""
        return self.model.check_finite()      # "escalated" when the model just moved itself to fp32 (the loops re-run the call)

    def a2p_wants_early_check(self) -> bool:
        """True when the wrapped denoiser runs on 16-bit operands and may still escalate: the loops then ask after the first step."""
        return self.model.wants_early_check()

    def a2p_sample_step(self, sampler, x, t_idx, timestep_map, tables, y, noise, eta, clip_denoised):
        """p_mean_variance + ddim_sample / p_sample in one library call; SpacedDiffusion's loops use it when present."""
        return self.model.sample_step(sampler, x, t_idx, timestep_map, tables, y, noise, eta, clip_denoised)
