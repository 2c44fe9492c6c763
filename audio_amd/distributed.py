"""Batch sharding of the DSP hot path across the GPUs of one node: one process per GPU,
``torch.distributed`` over RCCL (backend ``"nccl"`` on ROCm) / xGMI.

Every op on the path is independent per (clip, channel) sequence (the reference only
*reshapes* the batch: functional.py:119-120, :1421, filtering.py:1089-1090), so a batch shards by
contiguous row ranges with constants (window, fb, dct, sinc table, coefficients, RIR) replicated.
There is NO data-path collective except:

  * ``scatter_batch`` / ``gather_batch`` -- moving a batch that was born on one rank (point-to-point
    sends under the hood: xGMI is a link mesh, not a switch, so the root's 7 links work in parallel);
  * MFCC on a ``(batch, time)`` input: ``amplitude_to_DB(top_db=80)`` takes its cut-off from the
    maximum over the WHOLE batch (functional.py:393-402), i.e. one fp32 ``all_reduce(MAX)`` between
    the mel/dB kernel and the clamp + DCT kernel.  ``ShardedTransform`` installs it through
    ``MFCC.group_max_hook``.

The same code runs on the ``gloo`` backend with CPU tensors (tests/test_distributed.py, world size 2).
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

__all__ = ["init_from_env", "shard_range", "scatter_batch", "gather_batch", "all_gather_batch",
           "allreduce_group_max", "ShardedTransform"]


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, torch.device]:
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as set by
    ``python -m torch.distributed.run``); returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
        backend = backend or ("nccl" if use_gpu else "gloo")
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    return rank, world, device


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of ``n_rows`` rows: the first ``n_rows % world`` ranks get one extra row."""
    base, extra = divmod(n_rows, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


_PRIMED_GROUPS: set = set()


def _prime_group(device, group=None) -> None:
    """The first operation of a process group on RCCL must be entered by EVERY rank of the group (the communicator is built
    inside it); a ragged split leaves ranks with an empty shard out of the batched point-to-point exchange below, which is
    undefined as a group's first operation.  One 4-byte all-reduce per group, once, builds the communicator with
    everybody present; gloo needs nothing."""
    if dist.get_backend(group) != "nccl":
        return
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    key = (id(pg), pg.size(), pg.rank())         # (a re-initialised default group is another object)
    if key in _PRIMED_GROUPS:
        return
    dist.all_reduce(torch.zeros(1, dtype=torch.float32, device=device), group=group)
    _PRIMED_GROUPS.add(key)


def scatter_batch(batch: Optional[Tensor], shape: Tuple[int, ...], device, root: int = 0, group=None,
                  dtype=torch.float32) -> Tensor:
    """Rank ``root`` holds ``batch`` of ``shape`` (leading dim = rows to shard); every rank returns its
    contiguous shard.  Ragged splits are sent as point-to-point messages (no padding traffic)."""
    rank, world = _world(group)
    lo, hi = shard_range(shape[0], world, rank)
    if world == 1:
        return batch.to(device)
    _prime_group(device, group)
    local = torch.empty((hi - lo,) + tuple(shape[1:]), dtype=dtype, device=device)
    # ONE batch of point-to-point operations per rank (dist.batch_isend_irecv = one ncclGroupStart / End on RCCL): the root's
    # sends to its world - 1 peers leave over their own xGMI links concurrently instead of one isend after the other on the
    # root's stream; row slices of a contiguous batch are contiguous, so nothing is copied first
    ops = []
    if rank == root:
        for r in range(world):
            a, b = shard_range(shape[0], world, r)
            if r == root:
                local.copy_(batch[a:b])
            elif b > a:
                ops.append(dist.P2POp(dist.isend, batch[a:b].contiguous(), r, group))
    elif hi > lo:
        ops.append(dist.P2POp(dist.irecv, local, root, group))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return local


def gather_batch(local: Tensor, n_rows: int, root: int = 0, group=None) -> Optional[Tensor]:
    """Inverse of ``scatter_batch``: rank ``root`` returns the (n_rows, ...) concatenation."""
    rank, world = _world(group)
    if world == 1:
        return local
    local = local.contiguous()
    _prime_group(local.device, group)
    if rank == root:
        out = torch.empty((n_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        ops = []
        for r in range(world):
            a, b = shard_range(n_rows, world, r)
            if r == root:
                out[a:b].copy_(local)
            elif b > a:
                ops.append(dist.P2POp(dist.irecv, out[a:b], r, group))
        if ops:                                   # one batch: the world - 1 receives are posted together (see scatter_batch)
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        return out
    lo, hi = shard_range(n_rows, world, rank)
    if hi > lo:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, root, group)]):
            q.wait()
    return None


def all_gather_batch(local: Tensor, n_rows: int, group=None) -> Tensor:
    """Every rank ends up with the full (n_rows, ...) result (equal or ragged shards)."""
    rank, world = _world(group)
    if world == 1:
        return local
    parts = []
    for r in range(world):
        a, b = shard_range(n_rows, world, r)
        parts.append(torch.empty((b - a,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device))
    dist.all_gather(parts, local.contiguous(), group=group) if len({p.shape for p in parts}) == 1 else \
        _ragged_all_gather(parts, local.contiguous(), rank, world, group)
    return torch.cat(parts, 0)


def _ragged_all_gather(parts, local, rank, world, group):
    parts[rank].copy_(local)
    for r in range(world):
        if parts[r].numel():
            dist.broadcast(parts[r], src=r, group=group)


def allreduce_group_max(group_max: Tensor, group=None) -> None:
    """In-place MAX over ranks of the per-group dB maxima (the batch-global top_db cut-off)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(group_max, op=dist.ReduceOp.MAX, group=group)


class ShardedTransform:
    """Apply ``transform`` to this rank's shard of a batch.

    ``transform`` is any of the drop-in modules.  If it exposes ``group_max_hook`` (MFCC) and the
    input is 2-D ``(batch, time)`` -- the reference then uses ONE cut-off for the whole batch -- the
    hook is set to the MAX all-reduce so the sharded result equals the unsharded one bit for bit in
    the clamp decision.  Inputs with a channel dim have per-item cut-offs and need no exchange.
    """

    def __init__(self, transform: Callable[[Tensor], Tensor], group=None):
        self.transform = transform
        self.group = group

    def __call__(self, local: Tensor) -> Tensor:
        hooked = hasattr(self.transform, "group_max_hook") and local.dim() == 2
        if hooked:
            prev = self.transform.group_max_hook
            self.transform.group_max_hook = lambda gmax: allreduce_group_max(gmax, self.group)
        try:
            return self.transform(local)
        finally:
            if hooked:
                self.transform.group_max_hook = prev

    def run_from_root(self, batch: Optional[Tensor], shape: Tuple[int, ...], device, root: int = 0) -> Optional[Tensor]:
        """scatter -> transform -> gather; returns the full result on ``root`` (None elsewhere)."""
        local = scatter_batch(batch, shape, device, root, self.group)
        out = self(local)
        return gather_batch(out, shape[0], root, self.group)
