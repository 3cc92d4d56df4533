"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI) for the single
exchange step of the path -- the all-gather of per-rank kNN candidate lists before the mutual-pair intersection.

Every rank holds all batches (8 x 100k x 50 doubles = 320 MB, nothing next to 288 GB of HBM); each kNN search is
split by query rows (bmx_shard_range), so the per-merge distance block is spread evenly over all ranks whatever the
merge tree looks like, and the only traffic is the gathered index / distance lists.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


def shard_range(n, rank, world):
    """Row range [begin, end) of `n` query rows owned by `rank` (same arithmetic the engine uses)."""
    b, e = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.lib().bmx_shard_range(ctypes.c_int64(int(n)), int(rank), int(world), ctypes.byref(b), ctypes.byref(e))
    return b.value, e.value


def shard_gather_bytes(n, world, bytes_per_row):
    """Bytes per rank of the in-place all-gather of a per-query list (bmx_shard_gather_bytes: padded equal slices)."""
    f = _lib.lib().bmx_shard_gather_bytes
    f.restype = ctypes.c_int64
    return int(f(ctypes.c_int64(int(n)), int(world), ctypes.c_int64(int(bytes_per_row))))


def sharded_rows(fn, nq, row_shape, dtype, exchange, rank, world):
    """The engine's sharded search on host arrays: `fn(begin, end)` returns this rank's rows [begin, end) of a per-query
    list (bmx_shard_range); they are written into their slice of the padded buffer and completed by ONE in-place
    all-gather through `exchange` (a TorchExchange); every rank returns the full [nq, ...] list.  The layout arithmetic is
    the library's (the engine sizes its device-side exchanges with the same two functions)."""
    import numpy as np
    import torch
    width = int(np.prod(row_shape)) if row_shape else 1
    item = np.dtype(dtype).itemsize
    per_bytes = shard_gather_bytes(nq, world, width * item)
    per_rows = per_bytes // (width * item) if width * item else 0
    buf = np.zeros((per_rows * world,) + tuple(row_shape), dtype=dtype)
    b, e = shard_range(nq, rank, world)
    if e > b:
        buf[b:e] = fn(b, e)
    exchange.allgather_tensor_(torch.from_numpy(buf.reshape(-1).view(np.uint8)), per_bytes)
    return buf[:nq]


class _RawDeviceBuffer:
    """Exposes a raw device pointer to torch through the CUDA array interface (no copy)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2}


class TorchExchange:
    """In-place all-gather of a device buffer over an initialised torch.distributed process group.

    nccl (RCCL): the buffer is aliased as a torch tensor and gathered in place on the GPU.
    gloo: staged through host memory (used for tests on boxes without several GPUs).
    """

    def __init__(self, device_index=0, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device = torch.device("cuda", device_index)
        self.calls = 0
        self.bytes = 0

    def allgather_tensor_(self, full, per):
        """full: 1-D uint8 tensor of world * per bytes whose slice `rank` is valid; gathers in place."""
        mine = full[self.rank * per:(self.rank + 1) * per]
        if self.backend == "nccl":
            # a private copy of the (few-MB) slice keeps send and receive buffers disjoint
            self.dist.all_gather_into_tensor(full, mine.clone(), group=self.group)
        else:
            host = mine.cpu() if full.is_cuda else mine.clone()
            parts = [self.torch.empty_like(host) for _ in range(self.world)]
            self.dist.all_gather(parts, host, group=self.group)
            full.copy_(self.torch.cat(parts).to(full.device))
        self.calls += 1
        self.bytes += per * self.world

    def __call__(self, ptr, bytes_per_rank):
        torch = self.torch
        per = int(bytes_per_rank)
        full = torch.as_tensor(_RawDeviceBuffer(ptr, per * self.world), device=self.device)
        self.allgather_tensor_(full, per)
        torch.cuda.synchronize(self.device)


def rccl_library_path():
    """The RCCL this process already uses: the one next to torch's HIP libraries (one RCCL per process)."""
    import os
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand if os.path.exists(cand) else "librccl.so"


def init_engine_rccl(engine, group=None):
    """Gives `engine` its own RCCL communicator over the ranks of an initialised torch.distributed group: the
    all-gathers of the path then run inside the engine, in place on its stream (bmx_engine_init_rccl).  The 128-byte
    id is made on rank 0 and handed round with broadcast_object_list (any backend).  Collective."""
    import torch.distributed as dist
    lib = _lib.lib()
    import torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    # loading RCCL is local and can fail on one rank only: agree on it before the collective set-up, so that either
    # every rank goes on or every rank raises (and the caller falls back together)
    err = None
    try:
        _lib.check(lib.bmx_rccl_load(rccl_library_path().encode()))
    except Exception as exc:  # noqa: BLE001
        err = exc
    if world > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()) and err is None:
            err = RuntimeError("RCCL could not be loaded on another rank")
    if err is not None:
        raise err
    buf = ctypes.create_string_buffer(128)
    box = [None]
    if rank == 0:
        # a failure here must reach every rank: the others are about to wait in the broadcast
        try:
            _lib.check(lib.bmx_rccl_unique_id(buf, 128))
            box = [buf.raw]
        except Exception as exc:  # noqa: BLE001
            box = [("error", str(exc))]
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if not isinstance(box[0], (bytes, bytearray)):
        raise RuntimeError("ncclGetUniqueId failed on rank 0: " + (box[0][1] if box[0] else "no id"))
    uid = ctypes.create_string_buffer(box[0], 128)
    try:
        _lib.check(lib.bmx_engine_init_rccl(engine._h, int(rank), int(world), uid, 128))
    except Exception as exc:  # noqa: BLE001
        err = exc
    if world > 1:
        # the same agreement after the communicator set-up: a rank that alone fell back to the host-supplied
        # all-gather would wait for the others in a different collective for ever
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()) and err is None:
            err = RuntimeError("the RCCL communicator could not be made on another rank")
    if err is not None:
        raise err
    return rank, world
