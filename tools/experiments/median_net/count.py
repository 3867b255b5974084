"""hipcc -S of the prototype (median_net.hip) and the instruction counts of its loop body, per output pixel, next to k_median57's.
usage: python tools/experiments/median_net/count.py   (runs gen.py first)"""
import collections
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
HALF_RATE = re.compile(r"v_pk_|v_perm_b32|v_and_or_b32|v_lshl_or_b32|v_or3_b32|v_bfe_u32|v_bfi_b32|v_alignb|v_lshlrev_b32|v_mad_|v_mul_|v_cndmask")   # measured classes: profiles/r02_a_valu_rate_8waves.txt


def main():
    subprocess.check_call([sys.executable, os.path.join(HERE, "gen.py")])
    s_path = "/tmp/median_net.s"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-o", s_path, os.path.join(HERE, "median_net.hip")],
                          stderr=subprocess.DEVNULL)
    lines = [l.strip() for l in open(s_path)]
    body = [l for l in lines if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
    idx = [i for i, l in enumerate(body) if l.startswith("v_pk_m")]
    first, last = idx[0], idx[-1]
    # the loop body: from the first packed min/max back to the preceding label's start is loads/packing; take the enclosing region
    # between the last s_cbranch before `first` and the first s_cbranch after `last`
    lo = max([i for i, l in enumerate(body[:first]) if l.startswith("s_cbranch")] + [0])
    hi = min([i for i, l in enumerate(body) if i > last and l.startswith("s_cbranch")] + [len(body) - 1])
    loop = body[lo + 1:hi + 1]
    kinds = collections.Counter()
    for l in loop:
        op = l.split()[0]
        if op.startswith("v_pk_m"):
            kinds["v_pk_min/max_u16"] += 1
        elif op.startswith("v_accvgpr") or op.startswith("v_mov"):
            kinds["v_mov / v_accvgpr moves"] += 1
        elif op.startswith("v_"):
            kinds["other VALU (pack, unpack, addresses)"] += 1
        elif op.startswith("ds_"):
            kinds["LDS"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            kinds["VMEM" + (" (scratch)" if op.startswith("scratch_") else "")] += 1
        elif op.startswith("s_"):
            kinds["SALU / waits"] += 1
    px = 16
    meta = {k: re.search(k + r":\s*(\d+)", "\n".join(lines)).group(1) for k in ("NumVgprs", "TotalNumVgprs", "ScratchSize", "Occupancy")}
    print("loop body of k_median57_net (gfx950, -O3): %d instructions for %d output pixels of both medians" % (len(loop), px))
    for k, v in sorted(kinds.items(), key=lambda kv: -kv[1]):
        print("  %-40s %5d   %.2f per pixel" % (k, v, v / px))
    valu = sum(v for k, v in kinds.items() if k.startswith(("v_", "other VALU")))
    print("  VALU total: %.2f per pixel;  registers: %s" % (valu / px, meta))


if __name__ == "__main__":
    main()
