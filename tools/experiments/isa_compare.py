"""Code-generation facts of an experiment next to the product's (no GPU needed): per kernel VGPR / LDS / scratch / waves per SIMD where they
differ, and the row walks of k_sobel_nms_rows (vector instructions per 4-pixel row, tools/isa_budget.py).
    python tools/experiments/isa_compare.py NAME [NAME ...]        (applies the patches into build/exp/NAME first)"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_budget as ib  # noqa: E402
from img2sgf_amd import build  # noqa: E402


def kernels_of(csrc):
    saved = build.CSRC, build.FLAGS
    build.FLAGS = [f if f != os.path.join(saved[0], "isa") else os.path.join(csrc, "isa") for f in build.FLAGS]
    build.CSRC = csrc
    try:
        with tempfile.TemporaryDirectory() as d:
            return ib.kernels(ib.compile_asm(d))
    finally:
        build.CSRC, build.FLAGS = saved


def main():
    base = kernels_of(build.CSRC)
    for name in sys.argv[1:]:
        subprocess.check_call([sys.executable, os.path.join(HERE, "apply.py"), name], stdout=subprocess.DEVNULL)
        ks = kernels_of(os.path.join(ROOT, "build", "exp", name, "pkg", "csrc"))
        print("== %s" % name)
        for k in sorted(set(base) | set(ks)):
            a, b = base.get(k), ks.get(k)
            fa = None if a is None else (a["vgpr"], a["lds"], a["scratch"], ib.waves_per_simd(a["vgpr"]), len(ib.instructions(a["body"])))
            fb = None if b is None else (b["vgpr"], b["lds"], b["scratch"], ib.waves_per_simd(b["vgpr"]), len(ib.instructions(b["body"])))
            if fa != fb:
                print("  %-36s (VGPR, LDS, scratch, waves/SIMD, static instructions): %s -> %s" % (k, fa, fb))
            if k.startswith("k_sobel_nms_rows") and a and b:
                wa, wb = ib.row_walk_costs(a["body"]), ib.row_walk_costs(b["body"])
                for kind in sorted(set(wa) & set(wb)):
                    if wa[kind] != wb[kind]:
                        print("    %-6s walk, per 4-pixel row: vector %.1f -> %.1f (%+.1f %%), of which half-rate %.1f -> %.1f, scalar %.1f -> %.1f"
                              % (kind, wa[kind][0], wb[kind][0], 100.0 * (wb[kind][0] / wa[kind][0] - 1), wa[kind][1], wb[kind][1], wa[kind][2], wb[kind][2]))


if __name__ == "__main__":
    main()
