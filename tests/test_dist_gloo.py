"""N > 1 path on CPU: world_size 2 over gloo.  Shards are disjoint seed ranges, the only collective is the board
all-gather; every rank must end up with the same (total, 384) array, equal to a single-process run."""
import os
import subprocess
import sys

import numpy as np
import pytest

import emu_util
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


def _two_ranks(tmp_path, total, mode, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(total), str(tmp_path), mode],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    np.testing.assert_array_equal(a, b)
    return a


def test_two_rank_allgather(tmp_path):
    total = 3          # uneven shards: 2 + 1
    emu_util.emu_library()     # build once before the ranks race for it
    a = _two_ranks(tmp_path, total, "emu", 29611)
    det = Detector(0, 2, 300, 260, lib=emu_util.emu_library())
    imgs = [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in range(total)]
    single = i2s_dist.boards_to_numpy(det.detect_batch(imgs, full=False))
    np.testing.assert_array_equal(a, single)
    for s in range(total):
        assert (a[s, :361].reshape(19, 19)[:9, :8] == synth.occupancy(s, 9, 8)).all()


@pytest.mark.gpu
def test_two_ranks_real_hip_path_on_one_gpu(tmp_path):
    """Two ranks of the REAL library share the one leased GPU (so the exchange runs over gloo: RCCL refuses duplicate
    devices); rank r owns the contiguous seed shard shard_range(total, r, 2); both must end with the table a
    single-process run produces."""
    total = 7          # uneven shards: 4 + 3
    a = _two_ranks(tmp_path, total, "hip", 29613)
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
