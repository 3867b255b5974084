#!/bin/bash
# On a GPU box: the product library against the experiments of tools/experiments/*.patch, kernel durations on the benchmark workload
# (tools/kernel_times.py, one stream, 256 diagrams per pass), three interleaved runs each.   usage: tools/experiments/ab.sh NAME [NAME ...]
# Output: gpurun_out/r05/ab_<NAME>.txt  (one JSON line per run: "lib", "boards_ok", us per image and kernel group).
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${I2S_ROUND:-r06}; mkdir -p $O
for N in "$@"; do
  python tools/experiments/apply.py $N --hip > $O/ab_${N}_build.log 2>&1 || { echo "$N: build failed"; tail -5 $O/ab_${N}_build.log; continue; }
  : > $O/ab_$N.txt
  for R in 1 2 3; do
    python tools/kernel_times.py --images 256 --pass-size 256 --reps 3 | tail -1 >> $O/ab_$N.txt
    python tools/kernel_times.py --images 256 --pass-size 256 --reps 3 --lib build/exp/$N/libi2s_hip.so | tail -1 >> $O/ab_$N.txt
  done
  # parity of the experimental build on the GPU: the parity module of the suite, through the same C ABI (I2S_LIBRARY: img2sgf_amd/_lib.py)
  I2S_EXPERIMENT=1 I2S_LIBRARY=$PWD/build/exp/$N/libi2s_hip.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "synthetic or reference_image or tiny or ragged or large or record_indices or phone or capacity" > $O/ab_${N}_parity.log 2>&1; echo "$N parity: $(tail -1 $O/ab_${N}_parity.log)"
  # the Canny row kernels at every column-group boundary, on the hardware build (tests/stress/canny_widths.py takes any build of the library)
  case "$N" in *canny_lean*) timeout 900 python tests/stress/canny_widths.py $PWD/build/exp/$N/libi2s_hip.so > $O/ab_${N}_canny_widths.log 2>&1; echo "$N canny widths: $(tail -1 $O/ab_${N}_canny_widths.log)";; esac
  python - "$O/ab_$N.txt" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
for r in rows:
    u = r["us_per_image"]
    print("%-40s boards_ok=%s  vote %.2f  edge_bins %.2f  radius %.2f  total %.2f" % (r["lib"][-40:], r["boards_ok"], u.get("k_vote_centres", 0), u.get("k_edge_bins", 0), u.get("k_radius", 0), u["TOTAL"]))
PY
done
