"""Multi-GPU sharding of a batch of diagrams (BASELINE configs[3]): one process per GPU, contiguous shards by
image index, no data-path collective; the only exchange is ONE all-gather of the 384-byte board records.

* `BoardGather` -- the product path: an RCCL communicator and a device-resident gather buffer behind the C ABI
  (`i2s_comm_create` / `i2s_set_board_sink` / `i2s_allgather_boards`, include/i2s.h).  Detect calls leave their records
  in this rank's shard of the buffer on the device; `ncclAllGather` runs in place on the context's stream (xGMI inside
  a node); one device-to-host copy hands the complete table to the SGF writer.
* `allgather_boards_host` -- the same exchange over any `torch.distributed` backend with host tensors (gloo): used where
  RCCL cannot run (CPU tests; two ranks sharing ONE GPU, which RCCL refuses).

Shard sizes are a pure function of (total, rank, world) -- `shard_range` -- so no counts are exchanged: every rank sends
ceil(total / world) records, the unused tail zeroed.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import I2sBoard, I2sError

BOARD_BYTES = C.sizeof(I2sBoard)   # 384


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` items for `rank` (first total % world ranks get one extra)."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def records_per_rank(total, world):
    """Records every rank contributes to the all-gather: the largest shard."""
    return max(1, -(-total // world))


def boards_to_numpy(boards):
    """(n,) I2sBoard ctypes array -> (n, 384) uint8 view."""
    n = len(boards)
    return np.frombuffer(boards, dtype=np.uint8, count=n * BOARD_BYTES).reshape(n, BOARD_BYTES)


def compact(padded, total, world):
    """(world * cap, 384) gathered table with per-rank padding -> (total, 384) in image order."""
    cap = records_per_rank(total, world)
    padded = padded.reshape(world, cap, BOARD_BYTES)
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(padded[r, :hi - lo])
    return np.concatenate(parts, axis=0)


def allgather_boards_host(boards, total, rank, world):
    """All ranks contribute their shard's (n_r, 384) board records (host memory) and receive the (total, 384) table in
    image order, over the initialised torch.distributed backend with CPU tensors (gloo).  world == 1: a copy."""
    mine = boards_to_numpy(boards)
    lo, hi = shard_range(total, rank, world)
    assert mine.shape[0] == hi - lo, "rank %d holds %d records, its shard is [%d, %d)" % (rank, mine.shape[0], lo, hi)
    if world == 1:
        return mine.copy()
    import torch
    import torch.distributed as dist
    cap = records_per_rank(total, world)
    send = torch.zeros((cap, BOARD_BYTES), dtype=torch.uint8)
    send[:mine.shape[0]] = torch.from_numpy(mine.copy())
    recv = torch.empty((world * cap, BOARD_BYTES), dtype=torch.uint8)
    dist.all_gather_into_tensor(recv, send)
    return compact(recv.numpy(), total, world)


class BoardGather:
    """RCCL communicator + device gather buffer of one rank (C ABI: i2s_comm_*).  `uid` = the 128 bytes rank 0 obtained
    from `BoardGather.unique_id()` and handed to every rank (any host channel).  Creation is collective."""

    def __init__(self, device, world, rank, total, uid, lib=None):
        self.lib = lib if lib is not None else _lib.load()
        self.world, self.rank, self.total, self.device = world, rank, total, device
        self.cap = records_per_rank(total, world)
        self.lo, self.hi = shard_range(total, rank, world)
        assert len(uid) == _lib.COMM_ID_BYTES
        self._comm = C.c_void_p()
        idb = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(bytes(uid))
        rc = self.lib.dll.i2s_comm_create(C.byref(self._comm), device, idb, world, rank, self.cap)
        if rc != 0:
            raise I2sError("i2s_comm_create failed: %s: %s" % (self.lib.dll.i2s_strerror(rc).decode(),
                                                              self.lib.dll.i2s_comm_last_error(None).decode()))
        self.shard_ptr = int(self.lib.dll.i2s_comm_shard(self._comm))     # device address of this rank's records
        self.all_ptr = int(self.lib.dll.i2s_comm_all(self._comm))
        self._host = np.zeros((world * self.cap, BOARD_BYTES), np.uint8)

    @staticmethod
    def unique_id(lib=None):
        lib = lib if lib is not None else _lib.load()
        idb = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        rc = lib.dll.i2s_comm_unique_id(idb)
        if rc != 0:
            raise I2sError("i2s_comm_unique_id failed: %s: %s" % (lib.dll.i2s_strerror(rc).decode(), lib.dll.i2s_comm_last_error(None).decode()))
        return bytes(idb)

    def in_place(self):
        """True when the all-gather of the shard is the in-place form NCCL/RCCL documents: sendbuff == recvbuff + rank * count."""
        return self.shard_ptr == self.all_ptr + self.rank * self.cap * BOARD_BYTES

    def sink(self, first=0):
        """Device address where the record of this rank's image `first` belongs (for Detector.set_board_sink)."""
        return self.shard_ptr + first * BOARD_BYTES

    def allgather(self, detector, to_host=True):
        """ncclAllGather of the shards in place on `detector`'s stream; returns the (total, 384) host table in image
        order (to_host) or None (the table stays in the device buffer at all_ptr, rank r at r * cap records)."""
        host = self._host.ctypes.data_as(C.c_void_p) if to_host else None
        rc = self.lib.dll.i2s_allgather_boards(detector._ctx, self._comm, None, self.hi - self.lo, None, host)
        if rc != 0:
            raise I2sError("i2s_allgather_boards: %s: %s" % (self.lib.dll.i2s_strerror(rc).decode(),
                                                            self.lib.dll.i2s_last_error(detector._ctx).decode()))
        return compact(self._host, self.total, self.world) if to_host else None

    def close(self):
        if self._comm:
            self.lib.dll.i2s_comm_destroy(self._comm)
            self._comm = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
