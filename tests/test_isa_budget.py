"""ISA-budget regression test.  The product is compiled here with one HIP release and runs on the GPU boxes under
another; DESIGN.md's performance arguments rest on code-generation facts that no parity test re-checks: register budgets (waves per
SIMD), LDS per workgroup (workgroups per CU), no scratch in the hot kernels, no SLP-packed f32 pairs, the vote walk's 8 vector + 3
scalar instructions per step, the blur kernels' deep `vmcnt` waits.  One `hipcc -S` of the product sources with the product's flags
(~10 s, no GPU), then assertions; `python tools/isa_budget.py -o profiles/r06_isa_budget.txt` writes the table this was set from.
A second compile with -fslp-vectorize must FAIL the same checks: the guard is known to bite."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_budget as ib  # noqa: E402

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")

# kernel -> (VGPRs at most, LDS bytes exactly, scratch bytes at most).  VGPR ceilings are the next occupancy step's edge where
# that matters (k_blur<true>: 4 waves / SIMD up to 128; k_median57: 3 up to 168; the row kernels: 6 up to 80), today's figure + a few
# registers elsewhere.  LDS is exact: a changed figure is a changed kernel and the residency arguments (two k_vote_centres
# workgroups per CU at 78 404 B; four k_circles_final at 61 520 B) must be looked at again.
BUDGET = {
    "k_vote_centres<30>": (40, 78404, 0),
    "k_vote_centres<0>": (40, 78404, 0),
    "k_blur<true>": (128, 0, 0),
    "k_blur<false>": (144, 0, 0),
    "k_sobel_nms_rows<0, true>": (80, 0, 0),
    "k_sobel_nms_rows<1, true>": (80, 0, 0),
    "k_sobel_nms_rows<2, true>": (80, 0, 0),
    "k_sobel_nms_rows<3, false>": (128, 0, 0),
    "k_median57": (168, 17696, 8),
    "k_median57_bin": (40, 0, 0),
    "k_hysteresis": (80, 0, 0),
    "k_hysteresis_tail": (96, 4, 0),
    "k_edge_bins": (40, 792, 0),
    "k_radius": (64, 6144, 0),
    "k_circles_final<4096, 2048, true>": (40, 61520, 0),
    "k_circles_final<16384, 8192, false>": (40, 147536, 0),
    "k_erase_lines": (72, 18580, 20),
    "k_line_peaks": (32, 8200, 0),
    "k_grid": (64, 62448, 0),
    "k_grey": (40, 0, 0),
    "k_split_rgb": (24, 0, 0),
    "k_concat_circles": (32, 44, 0),
}
# waves per SIMD the design counts on
WAVES = {"k_blur<true>": 4, "k_blur<false>": 3, "k_median57": 3, "k_sobel_nms_rows<2, true>": 6, "k_sobel_nms_rows<0, true>": 6,
         "k_vote_centres<30>": 8, "k_radius": 8}


# vector instructions per 4-pixel row at most (today's figure + 2): profiles/r06_isa_budget.txt
ROW_WALK = {"k_sobel_nms_rows<0, true>": {"bytes": 105.9, "packed": 174.9}, "k_sobel_nms_rows<1, true>": {"bytes": 109.9, "packed": 178.9},
            "k_sobel_nms_rows<2, true>": {"bytes": 123.3, "packed": 209.8}, "k_sobel_nms_rows<3, false>": {"packed": 276.8}}


def violations(asm):
    ks = ib.kernels(asm)
    bad = []
    for name, (vg, lds, scr) in BUDGET.items():
        if name not in ks:
            bad.append("%s: kernel missing from the code object" % name)
            continue
        k = ks[name]
        if k["vgpr"] > vg:
            bad.append("%s: %d VGPRs > %d" % (name, k["vgpr"], vg))
        if k["lds"] != lds:
            bad.append("%s: %d B LDS != %d" % (name, k["lds"], lds))
        if k["scratch"] > scr:
            bad.append("%s: %d B scratch > %d" % (name, k["scratch"], scr))
    for name, w in WAVES.items():
        if name in ks and ib.waves_per_simd(ks[name]["vgpr"]) < w:
            bad.append("%s: %d waves / SIMD < %d" % (name, ib.waves_per_simd(ks[name]["vgpr"]), w))
    # scratch anywhere else on the detection path (JPEG's serial Huffman kernel spills by design: 152 B)
    for name, k in ks.items():
        if k["scratch"] and name not in BUDGET and name != "k_jpeg_huffman":
            bad.append("%s: %d B scratch in a kernel that had none" % (name, k["scratch"]))
    if ib.packed_f32(asm):
        bad.append("%d packed f32 instructions (v_pk_*_f32): SLP vectorisation is on (build.py: -fno-slp-vectorize)" % ib.packed_f32(asm))
    if "k_vote_centres<30>" in ks:
        n, steps = ib.vote_step_costs(ks["k_vote_centres<30>"]["body"])
        if n != 60:
            bad.append("k_vote_centres<30>: %d ds_add_u32, expected 2 unrolled walks of 30" % n)
        inner = {c: m for c, m in steps.items() if c[0] < 40}            # the one long gap is the code between the two walks
        if sum(inner.values()) != 58 or any(v > 8 for v, _ in inner) or any(s > 5 for _, s in inner):
            bad.append("k_vote_centres<30>: walk step costs (vector, scalar) %s, expected 58 steps of <= 8 vector + <= 5 scalar" % dict(steps))
        if steps.get((8, 3), 0) < 29:
            bad.append("k_vote_centres<30>: no walk with 8 vector + 3 scalar instructions per step: %s" % dict(steps))
    # the Canny row walks: vector instructions per 4-pixel row of the unrolled row loops -- HoughCircles' seven Cannys
    # are the one stage that saturates vector issue, so a compiler that adds ten instructions per row costs 6 % of it unseen
    for name, limits in ROW_WALK.items():
        if name in ks:
            got = ib.row_walk_costs(ks[name]["body"])
            for kind, most in limits.items():
                if kind not in got:
                    bad.append("%s: no %s row walk found (6 rows per trip)" % (name, kind))
                elif got[kind][0] > most:
                    bad.append("%s: %s walk %.1f vector instructions per 4-pixel row > %.1f" % (name, kind, got[kind][0], most))
    # k_blur: the wait for a prefetched row must leave the younger rows' loads AND stores in flight (vmcnt is in order): the row loops
    # wait at vmcnt(40) / vmcnt(35); round 4 found the compiler merging them into vmcnt(10) = a wait for nearly all stores (1.94 -> 1.5 us)
    for name, deep in (("k_blur<true>", 40), ("k_blur<false>", 35)):
        if name in ks:
            v = ib.vmcnt_values(ks[name]["body"])
            if v.get(deep, 0) < 2:
                bad.append("%s: expected two s_waitcnt vmcnt(%d) (row loop, both unrolled halves), got %s" % (name, deep, dict(sorted(v.items()))))
    return bad


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    return ib.compile_asm(str(tmp_path_factory.mktemp("isa")))


def test_kernel_budgets_and_loop_facts(asm):
    bad = violations(asm)
    assert not bad, "\n".join(bad)


def test_committed_table_is_current(asm):
    """profiles/r06_isa_budget.txt is what tools/isa_budget.py prints for today's sources (compiler banner aside)."""
    path = os.path.join(ROOT, "profiles", "r06_isa_budget.txt")
    with open(path) as f:
        committed = f.read().split("\n\n", 1)[1].strip()
    assert committed == ib.report(asm).strip(), "re-run: python tools/isa_budget.py -o profiles/r06_isa_budget.txt"


def test_guard_bites_on_an_slp_build(tmp_path):
    """The same checks on a deliberately mis-flagged build (-fslp-vectorize instead of -fno-slp-vectorize) must fail."""
    slp = ib.compile_asm(str(tmp_path), replace={"-fno-slp-vectorize": "-fslp-vectorize"})
    bad = violations(slp)
    assert any("packed f32" in b for b in bad), bad
