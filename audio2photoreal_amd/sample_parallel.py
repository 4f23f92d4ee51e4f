"""Sample-parallel sampling over the GPUs of one node.

The reference has no inference parallelism (sample/generate.py:124-144 loops repetitions
serially on one device).  The path shards embarrassingly over samples: no op mixes batch
elements, so each rank denoises a contiguous block of the global batch with its own replica
of the weights and NO per-step communication; one `all_gather` (RCCL over xGMI with the
"nccl" backend, gloo in the CPU tests) returns the samples at the end.

Noise is indexed by GLOBAL sample id (the caller passes full-batch `noise` / `step_noise`,
or a per-sample seed list), so 1/2/4/8-GPU runs produce identical samples.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

_BATCH_KEYS = ("audio", "cond_embed", "keyframes", "mask", "scale", "lengths", "missing", "alengths", "klengths")


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `total` samples for `rank`; the first total % world ranks get one extra."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_model_kwargs(model_kwargs: Dict, lo: int, hi: int) -> Dict:
    """Slice every per-sample tensor of y (data_loaders/tensors.py:57-67 layout) to [lo, hi)."""
    y = model_kwargs["y"]
    out = {}
    for k, v in y.items():
        if torch.is_tensor(v) and v.dim() >= 1 and k in _BATCH_KEYS:
            out[k] = v[lo:hi].contiguous()
        else:
            out[k] = v
    return {**model_kwargs, "y": out}


def per_sample_noise(shape: Sequence[int], seeds: Sequence[int], device="cpu") -> torch.Tensor:
    """N(0,1) noise [len(seeds), *shape[1:]] where row i depends only on seeds[i] (host generator)."""
    rows = []
    for s in seeds:
        g = torch.Generator().manual_seed(int(s))
        rows.append(torch.randn(tuple(shape[1:]), generator=g))
    return torch.stack(rows).to(device)


def shared_base_seed(group=None) -> int:
    """One integer all ranks agree on (rank 0's torch seed): the root of every per-sample random stream of a sharded run.
    Control plane only (8 bytes, once per sampling call); the data path still has exactly one collective."""
    seed = [int(torch.initial_seed()) & 0x7FFFFFFFFFFF]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast_object_list(seed, src=0, group=group)
    return seed[0]


def derive_seed(base: int, *ids: int) -> int:
    """A 63-bit seed from (base, ids...), each component passed through the splitmix64 finaliser before it is folded in:
    distinct (repetition, sample) pairs of one run AND the same pair under neighbouring base seeds get unrelated streams (a
    plain `base + g` makes run s / sample 1 equal run s+1 / sample 0)."""
    m = (1 << 64) - 1

    def mix(x: int) -> int:
        x = (x + 0x9E3779B97F4A7C15) & m
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
        return x ^ (x >> 31)
    h = mix(int(base) & m)
    for v in ids:
        h = mix(h ^ mix(int(v) & m))
    return h & 0x7FFFFFFFFFFFFFFF


def gather_blocks(local: torch.Tensor, sizes: Sequence[int], group=None) -> torch.Tensor:
    """One all_gather of per-rank blocks of DIFFERENT lengths (`sizes[r]` rows on rank r, zero allowed): every rank pads its
    block to the longest, the result is the concatenation of the true blocks in rank order.  Works for host tensors (gloo)
    and device tensors (backend "nccl" = RCCL over xGMI); a rank without rows still takes part."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    assert len(sizes) == world and local.shape[0] == sizes[dist.get_rank(group)], (sizes, tuple(local.shape))
    maxn = max(max(sizes), 1)
    pad = torch.zeros((maxn,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][: sizes[r]] for r in range(world)], dim=0)


def gather_samples(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """The single end-of-run collective: all ranks receive all samples, in global order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [hi - lo for lo, hi in (shard_bounds(total, world, r) for r in range(world))]
    return gather_blocks(local, sizes, group)


def _set_batch_hint(model, total: int):
    """Tell the denoiser(s) behind `model` (a FiLMTransformer, possibly inside ClassifierFreeSampleModel-style wrappers) how large the
    unsharded batch is: the HIP library picks its kernel family from that, not from the shard (include/a2p_hip.h a2p_set_batch_hint)."""
    seen, out, m = set(), [], model
    while m is not None and id(m) not in seen:
        seen.add(id(m))
        if hasattr(m, "global_batch_hint"):
            m.global_batch_hint = total
            out.append(m)
        m = getattr(m, "model", None)
    return out


def sample_parallel(sample_fn: Callable, model, shape: Sequence[int], model_kwargs: Dict,
                    noise: Optional[torch.Tensor] = None, step_noise=None, group=None, **kw) -> torch.Tensor:
    """Run `sample_fn` (e.g. diffusion.ddim_sample_loop / p_sample_loop) on this rank's block of the
    global batch `shape[0]` and gather.  `noise` / `step_noise[i]` are full-batch tensors (or None)."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    total = shape[0]
    lo, hi = shard_bounds(total, world, rank)
    kwargs = shard_model_kwargs(model_kwargs, lo, hi)
    local_shape = (hi - lo,) + tuple(shape[1:])
    local_noise = None if noise is None else noise[lo:hi].contiguous()
    local_step = None
    if step_noise is not None:
        if callable(step_noise):
            local_step = lambda n: step_noise(n)[lo:hi].contiguous()
        else:
            local_step = [s[lo:hi].contiguous() for s in step_noise]
    if hi > lo:
        extra = {} if local_step is None else {"step_noise": local_step}
        hinted = _set_batch_hint(model, total if world > 1 else 0)   # every shard takes the kernel family of the unsharded run
        try:
            local = sample_fn(model, local_shape, noise=local_noise, model_kwargs=kwargs, **extra, **kw)
        finally:
            for m in hinted:
                m.global_batch_hint = 0
    else:
        ref = noise if noise is not None else torch.zeros(1)
        local = torch.zeros((0,) + tuple(shape[1:]), dtype=torch.float32, device=ref.device)
    return gather_samples(local, total, group)
