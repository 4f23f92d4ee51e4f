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

import os
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


def gather_blocks(local: torch.Tensor, sizes: Sequence[int], group=None, mode: Optional[str] = None) -> torch.Tensor:
    """The end-of-run exchange of per-rank blocks of DIFFERENT lengths (`sizes[r]` rows on rank r, zero allowed); the result is the
    concatenation of the true blocks in rank order on every rank.  Works for host tensors (gloo) and device tensors (backend
    "nccl" = RCCL over xGMI); a rank without rows still takes part.  Two transports (`mode`, default: env A2P_GATHER, else "ring"):

      "ring"  ONE `all_gather` of blocks padded to the longest (RCCL's ring: N-1 hops of the padded block over one link at a time);
      "p2p"   all pairs at once -- every rank posts its TRUE block to each peer and receives each peer's block straight into place
              (`batch_isend_irecv` = one grouped ncclSend/ncclRecv: on xGMI's point-to-point mesh every transfer has its own
              link, one hop, no padding).  The payload here is ~2 MB per rank and latency-bound, which is
              the regime where N-1 serial ring hops cost more than N-1 parallel direct copies.

    Both give identical bytes (tests/test_host_cpu.py: ring vs p2p with 2 and 3 gloo ranks, an empty rank included)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert len(sizes) == world and local.shape[0] == sizes[rank], (sizes, tuple(local.shape))
    mode = mode or os.environ.get("A2P_GATHER", "ring")
    if mode == "p2p":
        local = local.contiguous()
        out = torch.empty((sum(sizes),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        offs = [sum(sizes[:r]) for r in range(world)]
        out[offs[rank]: offs[rank] + sizes[rank]] = local
        ops = []
        for k in range(1, world):       # peer order staggered by rank: no two ranks start on the same destination
            dst, src = (rank + k) % world, (rank - k) % world
            gdst = dist.get_global_rank(group, dst) if group is not None else dst
            gsrc = dist.get_global_rank(group, src) if group is not None else src
            if sizes[rank] > 0:
                ops.append(dist.P2POp(dist.isend, local, gdst, group))
            if sizes[src] > 0:
                ops.append(dist.P2POp(dist.irecv, out[offs[src]: offs[src] + sizes[src]], gsrc, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return out
    assert mode == "ring", f"A2P_GATHER / mode must be 'ring' or 'p2p', not {mode!r}"
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


def agree_flags(values: Sequence[int], group=None, device=None) -> List[List[int]]:
    """Control plane, len(values) ints per rank: every rank's flags, in rank order.  ONE small all_gather in front of the data-path
    collective.  `device`: where the flags live under backend "nccl" -- the device of THIS rank's communicator (the shard's device;
    callers pass it explicitly: `torch.cuda.current_device()` is only a last resort, a rank that failed before it touched the GPU
    may never have called `torch.cuda.set_device`)."""
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    if backend == "nccl":
        dev = device if (device is not None and torch.device(device).type == "cuda") else torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    flag = torch.tensor([int(v) for v in values], dtype=torch.int32, device=dev)
    flags = [torch.zeros_like(flag) for _ in range(world)]
    dist.all_gather(flags, flag, group=group)
    return [[int(v) for v in flags[r].tolist()] for r in range(world)]


def agree_on_failure(failed: bool, group=None, device=None) -> List[int]:
    """The ranks that report `failed` (empty list = nobody): an exception on one rank becomes an exception on all of them instead of
    a collective timeout."""
    return [r for r, f in enumerate(agree_flags([1 if failed else 0], group, device)) if f[0] != 0]


def _denoisers(model) -> list:
    """The FiLMTransformer(s) behind `model` (possibly inside ClassifierFreeSampleModel-style wrappers)."""
    seen, out, m = set(), [], model
    while m is not None and id(m) not in seen:
        seen.add(id(m))
        if hasattr(m, "global_batch_hint"):
            out.append(m)
        m = getattr(m, "model", None)
    return out


def _set_batch_hint(model, total: int):
    """Tell the denoiser(s) behind `model` how large the unsharded batch is: the HIP library picks its kernel family from that, not
    from the shard (include/a2p_hip.h a2p_set_batch_hint)."""
    out = _denoisers(model)
    for m in out:
        m.global_batch_hint = total
    return out


def sample_parallel(sample_fn: Callable, model, shape: Sequence[int], model_kwargs: Dict,
                    noise: Optional[torch.Tensor] = None, step_noise=None, group=None, **kw) -> torch.Tensor:
    """Run `sample_fn` (e.g. diffusion.ddim_sample_loop / p_sample_loop) on this rank's block of the
    global batch `shape[0]` and gather.  `noise` / `step_noise[i]` are full-batch tensors (or None); a callable `step_noise` must be a
    pure function of the step index (it is called again when a shard is repeated).

    Precision is a property of the BATCH, not of a shard: a 1-GPU run whose 16-bit logits leave the validated envelope escalates the
    whole batch to fp32 (FiLMTransformer.check_finite), so when ANY shard escalated, every rank switches its replica to fp32 and
    repeats its shard under the same random draws -- the 1/2/4/8-GPU results stay identical.  The verdicts travel in the same 8-byte
    control-plane all_gather as the failure flags."""
    from .diffusion.gaussian_diffusion import _rng_restore, _rng_snapshot
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
    dens = _denoisers(model)
    dev = noise.device if noise is not None else None
    if dev is None:
        try:
            dev = next(model.parameters()).device
        except (StopIteration, AttributeError, TypeError):
            dev = None
    rng = _rng_snapshot(dev)

    def run_shard():
        if hi <= lo:
            ref = noise if noise is not None else torch.zeros(1)
            return torch.zeros((0,) + tuple(shape[1:]), dtype=torch.float32, device=ref.device)
        extra = {} if local_step is None else {"step_noise": local_step}
        hinted = _set_batch_hint(model, total if world > 1 else 0)   # every shard takes the kernel family of the unsharded run
        try:
            return sample_fn(model, local_shape, noise=local_noise, model_kwargs=kwargs, **extra, **kw)
        finally:
            for m in hinted:
                m.global_batch_hint = 0

    # A failure on one rank (A2PError from the non-finite check, a bad input, ...) must not leave the others waiting in the final
    # collective until it times out: every rank reports into one all_gather first and all of them raise together.
    err: Optional[BaseException] = None
    local = None
    before = [getattr(m, "precision", None) for m in dens]
    try:
        local = run_shard()
    except Exception as e:   # noqa: BLE001 -- re-raised below, on every rank
        if world == 1:
            raise
        err = e
    if world > 1:
        cdev = local.device if local is not None else dev
        escalated = any(b != "fp32" and getattr(m, "precision", None) == "fp32" for m, b in zip(dens, before))
        flags = agree_flags([1 if err is not None else 0, 1 if escalated else 0], group, device=cdev)
        failed = [r for r, f in enumerate(flags) if f[0]]
        if err is not None:
            raise err
        if failed:
            raise RuntimeError(f"sample_parallel: rank(s) {failed} failed inside the sampling loop; rank {rank} stops before the gather")
        if any(f[1] for f in flags) and not escalated and any(getattr(m, "precision", "fp32") != "fp32" for m in dens):
            import warnings
            from . import _lib
            for m in dens:
                if getattr(m, "precision", "fp32") != "fp32":
                    m.escalated_from = getattr(m, "escalated_from", None) or m.precision
                    m.set_precision("fp32")
            warnings.warn(f"sample_parallel: rank(s) {[r for r, f in enumerate(flags) if f[1]]} escalated their shard to precision=\"fp32\"; "
                          f"rank {rank} repeats its shard in fp32 so that the sharded batch equals the unsharded one", _lib.A2PPrecisionWarning,
                          stacklevel=2)
            try:
                _rng_restore(dev, rng)
                local = run_shard()
            except Exception as e:   # noqa: BLE001
                err = e
            again = agree_on_failure(err is not None, group, device=cdev)
            if err is not None:
                raise err
            if again:
                raise RuntimeError(f"sample_parallel: rank(s) {again} failed while repeating their shard in fp32")
        elif any(f[1] for f in flags):
            # this rank escalated itself (its loop already repeated the call): it still joins the second agreement of the others
            again = agree_on_failure(False, group, device=cdev)
            if again:
                raise RuntimeError(f"sample_parallel: rank(s) {again} failed while repeating their shard in fp32")
    return gather_samples(local, total, group)
