"""Lands an experiment in the product AFTER tools/experiments/ab.sh has timed it on a GPU box and tests/test_gpu_parity.py passed on its
build (the rule of rounds 5 / 6: nothing lands in img2sgf_amd/csrc on a timing argument without a timing).

    python tools/experiments/merge.py NAME[+NAME...]  [--dry-run]

1. applies the patches to a copy of csrc (apply.py), 2. copies the patched sources over img2sgf_amd/csrc, 3. rebuilds the product and the
emulated library, 4. prints the code-generation facts that changed (tools/experiments/isa_compare.py) -- the LDS / VGPR budgets of
tests/test_isa_budget.py are EXACT for LDS and have to be edited by hand where a kernel's figure changed, on purpose --, 5. regenerates
profiles/rNN_isa_budget.txt if --budget-file is given.  Then: pytest -m "not gpu", and on the GPU box the suite and the bench."""
import argparse
import filecmp
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--budget-file")
    a = ap.parse_args()
    subprocess.check_call([sys.executable, os.path.join(HERE, "isa_compare.py"), a.name])          # applies the patches as well
    src = os.path.join(ROOT, "build", "exp", a.name, "pkg", "csrc")
    dst = os.path.join(ROOT, "img2sgf_amd", "csrc")
    changed = []
    for dirpath, _, files in os.walk(src):
        for f in files:
            s = os.path.join(dirpath, f)
            d = os.path.join(dst, os.path.relpath(s, src))
            if f.endswith((".orig", ".rej")):
                continue
            if not os.path.exists(d) or not filecmp.cmp(s, d, shallow=False):
                changed.append(os.path.relpath(s, src))
                if not a.dry_run:
                    shutil.copy(s, d)
    print("%s: %s" % ("would change" if a.dry_run else "changed", ", ".join(sorted(changed)) or "nothing"))
    if a.dry_run or not changed:
        return
    sys.path.insert(0, ROOT)
    from img2sgf_amd import build
    build.build(force=True)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "emu", "build_emu.py")], stdout=subprocess.DEVNULL)
    if a.budget_file:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "isa_budget.py"), "-o", a.budget_file], stdout=subprocess.DEVNULL)
    print("merged.  Next: edit the changed LDS / VGPR figures in tests/test_isa_budget.py (see above), `python -m pytest tests -m 'not gpu'`, and on the GPU "
          "box `python -m pytest tests -m gpu -x -q; python bench.py --steps 20 --warmup 5`.  The patches of the merged experiments no longer apply: delete them.")


if __name__ == "__main__":
    main()
