"""Data-parallel policy evaluation: one process per GPU, batch sharded, one all-gather of the action logits.

The reference has no distributed code at all. Every sample (episode) of a batch is independent
end-to-end, so the batch shards with no activation exchange: each rank holds a full weight replica (~0.7 GB bf16),
runs the policy on its contiguous slice of the batch and the only collective is ONE all-gather of the
[B_local, 700] fp32 raw logits (RCCL over xGMI when the backend is "nccl"; "gloo" for the CPU tests) --
5.7 MB at global batch 2048, latency-bound, so the plain ring/direct all-gather RCCL picks is fine.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous balanced shards: the first n % world ranks get one extra sample."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch_dim(x, dim: int, rank: int, world: int):
    """Slice tensors (or nested dict/list/tuple of tensors) along `dim`."""
    if isinstance(x, dict):
        return type(x)({k: shard_batch_dim(v, dim, rank, world) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return type(x)(shard_batch_dim(v, dim, rank, world) for v in x)
    if torch.is_tensor(x):
        lo, hi = shard_bounds(x.shape[dim], rank, world)
        return x.narrow(dim, lo, hi - lo)
    return x


def all_gather_logits(local: torch.Tensor, group=None, global_batch: int | None = None) -> torch.Tensor:
    """[B_local, W] -> [B_global, W] on every rank. Uneven shards are padded to the largest shard for the collective."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    if global_batch is None:
        sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
        counts = [int(s.item()) for s in sizes]
    else:
        counts = [shard_bounds(global_batch, r, world)[1] - shard_bounds(global_batch, r, world)[0] for r in range(world)]
    mx = max(counts)
    padded = local
    if local.shape[0] < mx:
        padded = torch.cat([local, local.new_zeros(mx - local.shape[0], *local.shape[1:])], dim=0)
    out = local.new_empty(world * mx, *local.shape[1:])
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(c == mx for c in counts):
        return out
    return torch.cat([out[r * mx: r * mx + c] for r, c in enumerate(counts)], dim=0)


def data_parallel_logits(step_fn, global_batch: int, group=None) -> torch.Tensor:
    """Run `step_fn(lo, hi) -> [hi-lo, W] logits` on this rank's shard of `global_batch` samples and return the
    gathered [global_batch, W] logits (identical on every rank)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_bounds(global_batch, rank, world)
    return all_gather_logits(step_fn(lo, hi), group=group, global_batch=global_batch)
