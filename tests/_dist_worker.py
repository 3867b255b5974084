"""Worker of tests/test_dist_gloo.py: one rank of an N-process gloo job (N = WORLD_SIZE: 2 or 8).  Each rank detects its shard of a small
synthetic batch and all-gathers the boards over gloo (host tensors).  argv[3] == "emu": the emulated build of the
product sources (no GPU in the build container); "hip": the real library on GPU 0 -- both ranks share the one leased
GPU, which is why this job cannot use RCCL (it refuses two ranks on one device)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import torch.distributed as dist
    import emu_util
    from img2sgf_amd import dist as i2s_dist, synth
    from img2sgf_amd.pipeline import Detector
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    total = int(sys.argv[1])
    out = sys.argv[2]
    dist.init_process_group("gloo")
    lo, hi = i2s_dist.shard_range(total, rank, world)
    imgs = [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in range(lo, hi)]
    lib = emu_util.emu_library() if sys.argv[3] == "emu" else None
    det = Detector(0, 2, 300, 260, lib=lib)
    from img2sgf_amd._lib import I2sBoard
    boards = det.detect_batch(imgs, full=False) if imgs else (I2sBoard * 0)()      # a rank may own no image (total < world)
    allb = i2s_dist.allgather_boards_host(boards, total, rank, world)
    assert allb.shape == (total, 384)
    np.save(os.path.join(out, "rank%d.npy" % rank), allb)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
