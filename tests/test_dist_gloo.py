"""N > 1 path on CPU: world_size 2 over gloo.  Shards are disjoint seed ranges, the only collective is the board
all-gather; every rank must end up with the same (total, 384) array, equal to a single-process run."""
import os
import subprocess
import sys

import numpy as np
import pytest

import emu_util
from helpers import free_port
from img2sgf_amd import dist as i2s_dist, synth
from img2sgf_amd.pipeline import Detector

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_range_partitions():
    for total in (0, 1, 7, 8, 4096, 32768):
        for world in (1, 2, 3, 8):
            r = [i2s_dist.shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_compact_and_records_per_rank():
    for total, world in ((3, 2), (8, 8), (7, 3), (1, 4), (4096 * 8, 8)):
        cap = i2s_dist.records_per_rank(total, world)
        assert cap * world >= total and cap >= 1
        padded = np.zeros((world, cap, 384), np.uint8)
        for r in range(world):
            lo, hi = i2s_dist.shard_range(total, r, world)
            for k in range(lo, hi):
                padded[r, k - lo] = k % 251 + 1
        out = i2s_dist.compact(padded.reshape(-1, 384), total, world)
        assert out.shape == (total, 384)
        assert (out[:, 0] == np.arange(total) % 251 + 1).all()


def _ranks(tmp_path, total, mode, port, world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(total), str(tmp_path), mode],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    a = np.load(tmp_path / "rank0.npy")
    for r in range(1, world):
        np.testing.assert_array_equal(a, np.load(tmp_path / ("rank%d.npy" % r)))
    return a


def _two_ranks(tmp_path, total, mode, port):
    return _ranks(tmp_path, total, mode, port, 2)


def test_two_rank_allgather(tmp_path):
    total = 3          # uneven shards: 2 + 1
    emu_util.emu_library()     # build once before the ranks race for it
    a = _two_ranks(tmp_path, total, "emu", free_port())
    det = Detector(0, 2, 300, 260, lib=emu_util.emu_library())
    imgs = [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in range(total)]
    single = i2s_dist.boards_to_numpy(det.detect_batch(imgs, full=False))
    np.testing.assert_array_equal(a, single)
    for s in range(total):
        assert (a[s, :361].reshape(19, 19)[:9, :8] == synth.occupancy(s, 9, 8)).all()


@pytest.mark.parametrize("total", [11, 5])
def test_eight_rank_allgather_uneven_shards(tmp_path, total):
    """BASELINE configs[3]'s shape -- 8 ranks -- on CPU over gloo with the emulated kernels: total % 8 != 0, so the shards
    differ in size (11 -> 2,2,2,1,1,1,1,1; 5 -> three ranks own nothing), every rank sends records_per_rank records with a zeroed
    tail, compact() must drop the padding, and all eight tables must equal the single-process run."""
    emu_util.emu_library()
    a = _ranks(tmp_path, total, "emu", free_port(), 8)
    assert a.shape == (total, 384)
    det = Detector(0, 2, 300, 260, lib=emu_util.emu_library())
    imgs = [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in range(total)]
    single = i2s_dist.boards_to_numpy(det.detect_batch(imgs, full=False))
    np.testing.assert_array_equal(a, single)


class _FakeCommDll:
    """Stand-in for the i2s_comm_* entry points of libi2s_hip.so: eight "ranks" in one process, each with its own
    [world][cap] gather buffer in host memory; i2s_allgather_boards copies every rank's shard into the caller's buffer the way
    ncclAllGather does (rank r's count records at recvbuff + r * count).  Checks dist.BoardGather's address arithmetic
    without a GPU: nothing here computes."""
    def __init__(self, world, cap):
        import ctypes as C
        self.C, self.world, self.cap = C, world, cap
        self.bufs = [np.zeros((world, cap, 384), np.uint8) for _ in range(world)]
        self.handles = {}

    def i2s_comm_create(self, out, device, idb, world, rank, cap):
        assert (world, cap) == (self.world, self.cap) and 0 <= rank < world
        h = 1000 + rank
        self.handles[h] = rank
        out._obj.value = h
        return 0

    def _rank(self, comm):
        return self.handles[comm.value if hasattr(comm, "value") else int(comm)]

    def i2s_comm_all(self, comm):
        return self.bufs[self._rank(comm)].ctypes.data

    def i2s_comm_shard(self, comm):
        r = self._rank(comm)
        return self.bufs[r].ctypes.data + r * self.cap * 384

    def i2s_allgather_boards(self, ctx, comm, d_boards, n_local, d_all, h_all):
        assert d_boards is None and d_all is None                  # the in-place form
        me = self._rank(comm)
        assert 0 <= n_local <= self.cap
        self.bufs[me][me, n_local:] = 0                             # the library zeroes the unused tail of the own shard
        for r in range(self.world):
            self.bufs[me][r] = self.bufs[r][r]                      # ncclAllGather: rank r's records at r * count
        if h_all is not None:
            self.C.memmove(h_all, self.bufs[me].ctypes.data, self.bufs[me].nbytes)
        return 0

    def i2s_comm_destroy(self, comm):
        return None


def test_board_gather_offsets_world8_with_fake_comm():
    """BoardGather with world = 8 and uneven shards against a fake i2s_comm: the sink address of image k of rank r, the
    records_per_rank every rank sends, the in-place relation sendbuff == recvbuff + rank * count, and the image-order table
    that comes out -- so that the first real `python bench.py --gpus 8` cannot fail on indexing."""
    import ctypes as C
    total, world = 29, 8                                            # shards 4,4,4,4,4,3,3,3
    cap = i2s_dist.records_per_rank(total, world)
    assert cap == 4
    dll = _FakeCommDll(world, cap)
    lib = type("FakeLib", (), {"dll": dll})()
    gathers = [i2s_dist.BoardGather(0, world, r, total, bytes(128), lib=lib) for r in range(world)]
    for r, g in enumerate(gathers):
        lo, hi = i2s_dist.shard_range(total, r, world)
        assert (g.lo, g.hi, g.cap) == (lo, hi, cap) and g.in_place()
        assert g.shard_ptr - g.all_ptr == r * cap * 384
        # a stale record beyond the shard's valid part must not survive the gather
        C.memset(g.sink(0), 0xEE, cap * 384)
        for k in range(lo, hi):                                     # what Detector.detect_device(..., sink=g.sink(first)) does on the device
            rec = np.full(384, 0, np.uint8)
            rec[0], rec[1], rec[383] = k % 251 + 1, r + 1, 0x5A
            C.memmove(g.sink(k - lo), rec.ctypes.data, 384)
    det = type("FakeDetector", (), {"_ctx": None})()
    tables = [g.allgather(det) for g in gathers]
    for t in tables:
        assert t.shape == (total, 384)
        np.testing.assert_array_equal(t, tables[0])
    t = tables[3]                                                   # rank 3's view
    assert (t[:, 0] == np.arange(total) % 251 + 1).all() and (t[:, 383] == 0x5A).all()
    owner = np.concatenate([np.full(i2s_dist.shard_range(total, r, world)[1] - i2s_dist.shard_range(total, r, world)[0], r + 1)
                            for r in range(world)])
    assert (t[:, 1] == owner).all()
    for g in gathers:
        g.close()


@pytest.mark.gpu
def test_two_ranks_real_hip_path_on_one_gpu(tmp_path):
    """Two ranks of the REAL library share the one leased GPU (so the exchange runs over gloo: RCCL refuses duplicate
    devices); rank r owns the contiguous seed shard shard_range(total, r, 2); both must end with the table a
    single-process run produces."""
    total = 7          # uneven shards: 4 + 3
    a = _two_ranks(tmp_path, total, "hip", free_port())
    det = Detector(0, 4, 300, 260)
    imgs = [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in range(total)]
    single = i2s_dist.boards_to_numpy(det.detect_batch(imgs, full=False))
    det.close()
    np.testing.assert_array_equal(a, single)
    for s in range(total):
        assert (a[s, :361].reshape(19, 19)[:9, :8] == synth.occupancy(s, 9, 8)).all()


@pytest.mark.gpu
def test_rccl_allgather_through_c_abi_world1():
    """The device-resident path of bench.py on one GPU: an RCCL communicator of one rank created through the C ABI, two
    contexts deposit their slices of a batch into the shard on the device (i2s_set_board_sink), ncclAllGather in place
    on the context's stream, one D2H -- the table must equal the host-returned records."""
    import torch
    total = 6
    imgs, _ = synth.synth_batch(range(total), geom=synth.GEOM_SMALL)
    dev = torch.from_numpy(np.ascontiguousarray(imgs)).cuda()
    g = i2s_dist.BoardGather(0, 1, 0, total, i2s_dist.BoardGather.unique_id())
    d0, d1 = Detector(0, 2, 300, 260), Detector(0, 2, 300, 260)
    b0 = d0.detect_device(dev[:4], None, sink=g.sink(0))
    b1 = d1.detect_device(dev[4:], None, sink=g.sink(4))
    table = g.allgather(d0)
    assert table.shape == (total, 384)
    np.testing.assert_array_equal(table[:4], i2s_dist.boards_to_numpy(b0))
    np.testing.assert_array_equal(table[4:], i2s_dist.boards_to_numpy(b1))
    # the sink is per call: a later call without one must not write into the shard
    d0.detect_device(dev[4:], None)
    np.testing.assert_array_equal(g.allgather(d0), table)
    g.close(); d0.close(); d1.close()
