"""N > 1 path on CPU: world_size 2 over gloo.  Shards are disjoint seed ranges, the only collective is the board
all-gather; every rank must end up with the same (total, 384) array, equal to a single-process run."""
import os
import subprocess
import sys

import numpy as np

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


def test_two_rank_allgather(tmp_path):
    total = 3          # uneven shards: 2 + 1
    emu_util.emu_library()     # build once before the ranks race for it
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(total), str(tmp_path)],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    np.testing.assert_array_equal(a, b)
    det = Detector(0, 2, 300, 260, lib=emu_util.emu_library())
    imgs = [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in range(total)]
    single = i2s_dist.boards_to_numpy(det.detect_batch(imgs, full=False))
    np.testing.assert_array_equal(a, single)
    for s in range(total):
        assert (a[s, :361].reshape(19, 19)[:9, :8] == synth.occupancy(s, 9, 8)).all()
