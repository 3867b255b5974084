"""Code-generation facts of the product's kernels, read off `hipcc -S` (no GPU needed): per kernel the VGPR / SGPR / LDS / scratch
figures of the code object's metadata, and the loop-level facts DESIGN.md's performance arguments rest on.

    python tools/isa_budget.py                 # compile with the product's flags, print the table
    python tools/isa_budget.py -o profiles/r06_isa_budget.txt

tests/test_isa_budget.py asserts the budget below on every CPU run: the product is compiled in the build container with one HIP
release and runs under another, and a compiler that starts packing f32 pairs, spills the median kernel or merges the blur kernel's
waits would cost performance that no GPU-less test otherwise sees."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def compile_asm(out_dir, extra=(), replace=None):
    """Device assembly of csrc/i2s_api.hip with the product's own flags (img2sgf_amd/build.py FLAGS)."""
    from img2sgf_amd import build
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    if replace:
        flags = [replace.get(f, f) for f in flags]
    out = os.path.join(out_dir, "i2s_api.s")
    cmd = ["hipcc"] + flags + list(extra) + ["-S", "--cuda-device-only", "-o", out, os.path.join(build.CSRC, "i2s_api.hip")]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    with open(out) as f:
        return f.read()


def kernels(asm):
    """{demangled name without arguments: dict(vgpr, sgpr, lds, scratch, body)}"""
    md = asm[asm.index("amdhsa.kernels:"):]
    rows = re.findall(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?"
                      r"\.sgpr_count:\s*(\d+).*?\.vgpr_count:\s*(\d+)", md, re.S)
    names = [r[1] for r in rows]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    out = {}
    for (lds, mangled, scr, sg, vg), d in zip(rows, dem):
        d = re.sub(r"^void ", "", re.sub(r"\(.*", "", d)).replace("i2s::", "")
        m = re.search(r"^%s:[^\n]*\n(.*?)^\.Lfunc_end\d+:" % re.escape(mangled), asm, re.S | re.M)
        out[d] = dict(vgpr=int(vg), sgpr=int(sg), lds=int(lds), scratch=int(scr), body=m.group(1) if m else "")
    return out


def instructions(body):
    return [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((";", "."))]


def vote_step_costs(body):
    """(vector, scalar) instruction counts between consecutive ds_add_u32 of the unrolled radius walk, as a Counter."""
    ins = instructions(body)
    idx = [i for i, l in enumerate(ins) if l.startswith("ds_add_u32")]
    c = collections.Counter()
    for a, b in zip(idx, idx[1:]):
        seg = ins[a + 1:b]
        c[(sum(1 for l in seg if l.startswith("v_")), sum(1 for l in seg if l.startswith("s_")))] += 1
    return len(idx), c


def loops(body):
    """(instructions, [(first, last)]) -- the natural loops of a kernel body: a branch to a label at or above itself closes one."""
    ins, labels = [], {}
    for l in body.splitlines():
        l = l.strip()
        if not l or l.startswith(";"):
            continue
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = len(ins)
        elif not l.startswith("."):
            ins.append(l.split(";")[0].strip())
    head = {}
    for i, l in enumerate(ins):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", l)
        if m and labels.get(m.group(1), len(ins)) <= i:
            head[labels[m.group(1)]] = max(head.get(labels[m.group(1)], 0), i)
    return ins, sorted(head.items())


# instruction classes that issue at half rate on gfx950 (profiles/r02_a_valu_rate_8waves.txt: 4.1 - 4.6 cycles against 2.3 - 3.0)
HALF_RATE = re.compile(r"v_pk_|v_perm_b32|v_and_or_b32|v_lshl_or_b32|v_or3_b32|v_bfe_|v_bfi_b32|v_alignb|v_lshlrev_b32|v_mad_|v_mul_|v_cndmask|v_dot|v_min|v_max|"
                       r"v_med3|v_add3|v_lshl_add|v_add_lshl|v_cmp|v_xad|v_sad|v_bcnt|v_cvt")


def row_walk_costs(body):
    """The row kernels (k_sobel_nms_rows) walk a band six rows per trip of an unrolled loop: {"bytes" | "packed": (vector instructions,
    of which half-rate classes, scalar instructions) per 4-pixel row} for the two-valued byte walk and the 16-bit packed walk."""
    ins, lp = loops(body)
    out = {}
    for a, b in lp:
        seg = ins[a:b + 1]
        v = [x for x in seg if x.startswith("v_")]
        if len(v) < 300:
            continue
        kind = "packed" if any(x.startswith("v_pk_") for x in seg) else "bytes"
        if kind not in out or len(v) < out[kind][0] * 6:            # the innermost of the loops that share a body
            out[kind] = (len(v) / 6.0, sum(1 for x in v if HALF_RATE.match(x)) / 6.0, sum(1 for x in seg if x.startswith("s_")) / 6.0)
    return out


def vmcnt_values(body):
    return collections.Counter(int(v) for v in re.findall(r"s_waitcnt[^\n]*?vmcnt\((\d+)\)", body))


def packed_f32(asm):
    return len(re.findall(r"\bv_pk_\w+_f32\b", asm))


def waves_per_simd(vgpr):
    """gfx950: 512 VGPRs per SIMD lane, allocated in blocks of 8, at most 8 wavefronts."""
    return min(8, 512 // (((vgpr + 7) // 8) * 8))


def report(asm):
    ks = kernels(asm)
    lines = ["kernel                                   VGPR  SGPR     LDS  scratch  waves/SIMD"]
    for name in sorted(ks):
        k = ks[name]
        lines.append("%-40s %4d  %4d  %6d  %7d  %d" % (name, k["vgpr"], k["sgpr"], k["lds"], k["scratch"], waves_per_simd(k["vgpr"])))
    n, c = vote_step_costs(ks["k_vote_centres<30>"]["body"])
    lines += ["", "k_vote_centres<30>: %d ds_add_u32; (vector, scalar) instructions between consecutive ones: %s" % (n, dict(c)),
              "k_blur<true>  s_waitcnt vmcnt values: %s" % dict(sorted(vmcnt_values(ks["k_blur<true>"]["body"]).items())),
              "k_blur<false> s_waitcnt vmcnt values: %s" % dict(sorted(vmcnt_values(ks["k_blur<false>"]["body"]).items())),
              "packed f32 instructions (v_pk_*_f32) in the whole code object: %d" % packed_f32(asm), "",
              "row walks of k_sobel_nms_rows (unrolled by 6): instructions per 4-pixel row -- vector (of which half-rate classes), scalar"]
    for name in sorted(ks):
        if name.startswith("k_sobel_nms_rows"):
            for kind, (v, hr, sc) in sorted(row_walk_costs(ks[name]["body"]).items()):
                lines.append("  %-28s %-6s walk: %6.1f vector (%5.1f half rate), %5.1f scalar  = %.1f vector per pixel" % (name, kind, v, hr, sc, v / 4))
    return "\n".join(lines)


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        asm = compile_asm(d)
    ver = subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout.strip().split("\n")
    text = "hipcc: %s\n%s\n\n%s\n" % (ver[0], " | ".join(v.strip() for v in ver[1:3]), report(asm))
    if len(sys.argv) > 2 and sys.argv[1] == "-o":
        with open(sys.argv[2], "w") as f:
            f.write(text)
    print(text)
