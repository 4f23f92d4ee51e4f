"""Timestep respacing (reference diffusion/respace.py): `space_timesteps`,
`SpacedDiffusion`, `_WrappedModel` with the same signatures."""
from __future__ import annotations

import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """Timesteps of the original chain to keep (reference respace.py:21-74).

    "ddimN" -> the first integer stride that yields exactly N steps; otherwise a list /
    comma string of per-section counts spread evenly inside equal sections."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                kept = range(0, num_timesteps, stride)
                if len(kept) == want:
                    return set(kept)
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(tok) for tok in section_counts.split(",")]
    n_sec = len(section_counts)
    base, extra = divmod(num_timesteps, n_sec)
    kept, first = [], 0
    for sec, count in enumerate(section_counts):
        width = base + (1 if sec < extra else 0)
        if width < count:
            raise ValueError(f"cannot divide section of {width} steps into {count}")
        stride = 1 if count <= 1 else (width - 1) / (count - 1)
        kept.extend(first + round(pos) for pos in _accumulate(stride, count))
        first += width
    return set(kept)


def _accumulate(stride, count):
    # repeated addition (not j*stride) so rounding matches the reference's running sum
    pos = 0.0
    for _ in range(count):
        yield pos
        pos += stride


class SpacedDiffusion(GaussianDiffusion):
    """Diffusion over a subset of the base chain's timesteps (reference respace.py:77-127)."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        base_acp = np.cumprod(1.0 - np.array(kwargs["betas"], dtype=np.float64), axis=0)
        self.timestep_map, betas, prev = [], [], 1.0
        for i, acp in enumerate(base_acp):
            if i in self.use_timesteps:
                betas.append(1 - acp / prev)     # beta' = 1 - abar_i / abar_prev over kept steps
                prev = acp
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(betas)
        super().__init__(**kwargs)

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)

    def _scale_timesteps(self, t):
        return t  # done by the wrapped model

    def _timestep_map_tensor(self, device):
        key = ("tmap", str(device))
        if key not in self._dev_cache:
            self._dev_cache[key] = th.tensor(self.timestep_map, device=device, dtype=th.int64)
        return self._dev_cache[key]


class _WrappedModel:
    """Maps step index -> original timestep before calling the model (reference respace.py:130-145)."""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model = model
        if hasattr(model, "step"):
            self.step = model.step
        self.add_frame_cond = getattr(model, "add_frame_cond", None)  # plain callables are accepted too
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps
        self._map_cache = {}

    def __call__(self, x, ts, **kwargs):
        key = (str(ts.device), ts.dtype)
        if key not in self._map_cache:   # the reference rebuilds this tensor every call (an H2D per step)
            self._map_cache[key] = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        new_ts = self._map_cache[key][ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)
