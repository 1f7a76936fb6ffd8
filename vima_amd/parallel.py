"""Data-parallel policy evaluation: one process per GPU, batch sharded, one all-gather of the action logits.

The reference has no distributed code at all. Every sample (episode) of a batch is independent
end-to-end, so the batch shards with no activation exchange: each rank holds a full weight replica (~0.7 GB bf16),
runs the policy on its contiguous slice of the batch and the only collective is ONE all-gather of the
[B_local, 700] fp32 raw logits (RCCL over xGMI when the backend is "nccl"; "gloo" for the CPU tests) --
5.7 MB at global batch 2048, latency-bound, so the plain ring/direct all-gather RCCL picks is fine.

On the GPU the collective goes through the C ABI (`vima_allgather_logits` in include/vima_hip.h, RCCL bound inside
libvima_hip.so): `LogitsComm` creates the communicator -- rank 0 draws the RCCL unique id (`vima_comm_unique_id`), the id
travels to the other ranks through the already initialised torch.distributed group (plumbing), every rank joins with
`vima_comm_create`. Without a `LogitsComm` (CPU tests, gloo) the same function falls back to torch.distributed.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.distributed as dist


class StubCommBackend:
    """Stand-in for the four vima_comm_* entry points of the C ABI on machines without a GPU: same call ORDER and the same contract
    (rank 0 draws an id, every rank joins with it, the collective is an all-gather of equal row blocks, destroy) over the already
    initialised torch.distributed group. It exists so that the whole N > 1 host path of bench.py -- communicator creation order, the id
    broadcast, the agreement all-reduce, max-over-ranks timing, teardown -- can be EXECUTED at world size 8 in the CPU suite
    (`bench.py --dry-ranks 8`, tests/test_parallel.py); it is never a measurement and never the product path."""
    ID_BYTES = 128

    def __init__(self, group=None):
        self.group = group
        self.joined = None
        self.calls = []

    def unique_id(self) -> bytes:
        self.calls.append("unique_id")
        return (b"vima-stub-comm-id:" + os.urandom(16).hex().encode()).ljust(self.ID_BYTES, b"\0")

    def create(self, unique_id: bytes, world: int, rank: int, device_index: int):
        self.calls.append("create")
        assert len(unique_id) == self.ID_BYTES and unique_id.startswith(b"vima-stub-comm-id:")
        # every rank must hold the SAME id (the real ncclCommInitRank would hang or fail otherwise): checked collectively
        ids = [None] * world
        dist.all_gather_object(ids, unique_id, group=self.group)
        if any(i != ids[0] for i in ids):
            raise RuntimeError("LogitsComm (stub): the ranks joined with different communicator ids")
        self.joined = (world, rank)

    def all_gather(self, local: torch.Tensor, rows_per_rank: int, world: int) -> torch.Tensor:
        self.calls.append("all_gather")
        out = local.new_empty(world * rows_per_rank, local.shape[1])
        dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def destroy(self):
        self.calls.append("destroy")
        self.joined = None


class LogitsComm:
    """RCCL communicator of the data-parallel policy path, owned by libvima_hip.so (C ABI: vima_comm_*). `backend` replaces the four
    entry points (StubCommBackend: the CPU dry run of the N > 1 host path); everything else -- who draws the id, how it travels, the
    order of the calls -- is this class and is the same in both."""

    def __init__(self, device, group=None, rank: int | None = None, world: int | None = None, unique_id: bytes | None = None, backend=None):
        device = torch.device(device)
        self._backend = backend
        if backend is None:
            from . import _lib
            self._lib = _lib.load()
            self._check = _lib.check
            id_bytes = _lib.COMM_ID_BYTES
            if device.type != "cuda":
                raise RuntimeError("LogitsComm needs a GPU device (RCCL); the CPU tests use torch.distributed/gloo or a StubCommBackend instead")
        else:
            id_bytes = backend.ID_BYTES
        self.device = device
        if rank is None or world is None:
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.rank, self.world = rank, world
        if unique_id is None:
            box = [None]
            if rank == 0:
                if backend is None:
                    buf = ctypes.create_string_buffer(id_bytes)
                    self._check(self._lib.vima_comm_unique_id(buf))
                    box[0] = buf.raw
                else:
                    box[0] = backend.unique_id()
            if world > 1:
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            unique_id = box[0]
        assert len(unique_id) == id_bytes
        self._h = None
        if backend is None:
            h = ctypes.c_void_p()
            idx = device.index if device.index is not None else torch.cuda.current_device()
            self._check(self._lib.vima_comm_create(unique_id, world, rank, idx, ctypes.byref(h)))
            self._h = h
        else:
            backend.create(unique_id, world, rank, device.index or 0)
            self._h = backend

    def all_gather(self, local: torch.Tensor, rows_per_rank: int) -> torch.Tensor:
        """local f32 [rows_per_rank, W] (contiguous, on this rank's GPU) -> [world * rows_per_rank, W]; enqueued on the
        current stream, no host synchronisation."""
        assert local.dtype == torch.float32 and local.is_contiguous() and local.shape[0] == rows_per_rank
        if self._backend is not None:
            return self._backend.all_gather(local, rows_per_rank, self.world)
        assert local.is_cuda
        out = local.new_empty(self.world * rows_per_rank, local.shape[1])
        stream = ctypes.c_void_p(torch.cuda.current_stream(local.device).cuda_stream)
        self._check(self._lib.vima_allgather_logits(self._h, ctypes.c_void_p(local.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                    rows_per_rank, local.shape[1], stream))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None:
            if self._backend is not None:
                self._backend.destroy()
            else:
                self._lib.vima_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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


def all_gather_logits(local: torch.Tensor, group=None, global_batch: int | None = None,
                      comm: LogitsComm | None = None) -> torch.Tensor:
    """[B_local, W] -> [B_global, W] on every rank. Uneven shards are padded to the largest shard for the collective.
    With `comm` the exchange is `vima_allgather_logits` (RCCL inside the C ABI); without it torch.distributed."""
    if comm is not None:
        world = comm.world
        if world == 1:
            return local
        if global_batch is None:
            raise ValueError("global_batch is required with a LogitsComm (shard sizes must be known without a collective)")
        counts = [shard_bounds(global_batch, r, world)[1] - shard_bounds(global_batch, r, world)[0] for r in range(world)]
        mx = max(counts)
        padded = local.contiguous()
        if local.shape[0] < mx:
            padded = torch.cat([local, local.new_zeros(mx - local.shape[0], *local.shape[1:])], dim=0)
        out = comm.all_gather(padded, mx)
        if all(c == mx for c in counts):
            return out
        return torch.cat([out[r * mx: r * mx + c] for r, c in enumerate(counts)], dim=0)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    if local.is_cuda and dist.get_backend(group) == "gloo":   # 2 ranks sharing one GPU in the tests: exchange on the host
        return all_gather_logits(local.cpu(), group=group, global_batch=global_batch).to(local.device)
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


def data_parallel_logits(step_fn, global_batch: int, group=None, comm: LogitsComm | None = None) -> torch.Tensor:
    """Run `step_fn(lo, hi) -> [hi-lo, W] logits` on this rank's shard of `global_batch` samples and return the
    gathered [global_batch, W] logits (identical on every rank)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_bounds(global_batch, rank, world)
    return all_gather_logits(step_fn(lo, hi), group=group, global_batch=global_batch, comm=comm)
