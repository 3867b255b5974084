"""Multi-GPU sharding of a batch of diagrams: one process per GPU, contiguous shards, no data-path collective; the
only exchange is one all-gather of the 384-byte board records (RCCL over xGMI when backend == nccl)."""
import ctypes as C

import numpy as np

from ._lib import I2sBoard

BOARD_BYTES = C.sizeof(I2sBoard)   # 384


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` items for `rank` (first total % world ranks get one extra)."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def boards_to_numpy(boards):
    """(n,) I2sBoard ctypes array -> (n, 384) uint8 view."""
    n = len(boards)
    return np.frombuffer(boards, dtype=np.uint8, count=n * BOARD_BYTES).reshape(n, BOARD_BYTES)


def allgather_boards(boards, world=1, device_index=None):
    """All ranks contribute their (n_r, 384) board records and receive the concatenation in rank order.
    Shards may differ by one record (shard_range), so records are padded to the largest shard for the collective."""
    mine = boards_to_numpy(boards)
    import torch.distributed as dist
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return mine.copy()
    import torch
    backend = dist.get_backend()
    dev = torch.device("cuda", device_index) if backend == "nccl" else torch.device("cpu")
    counts = torch.tensor([mine.shape[0]], dtype=torch.int64, device=dev)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    ns = [int(c.item()) for c in all_counts]
    nmax = max(ns)
    send = torch.zeros((nmax, BOARD_BYTES), dtype=torch.uint8, device=dev)
    send[:mine.shape[0]] = torch.from_numpy(mine.copy()).to(dev)
    recv = torch.empty((world * nmax, BOARD_BYTES), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send)
    out = recv.cpu().numpy().reshape(world, nmax, BOARD_BYTES)
    return np.concatenate([out[r, :ns[r]] for r in range(world)], axis=0)
