#!/bin/bash
# Profiles of one round, run on the GPU box from the repo root; everything lands in gpurun_out/<tag>_* (copy what is to be judged to profiles/).
#   usage: tools/profile_round.sh r04
# SEPARATE runs per workload: kernel traces of the clean and of the noisy workload (tools/kernel_times.py, one
# stream, 256 diagrams per pass, 3 passes), the two PMC passes for HBM traffic (clean, 256 diagrams) and the SQ passes (clean / noisy).
# The rocpd databases stay in gpurun_out/<tag>_db/ (scratch, not merged back beyond the size limit): only the CSV / JSON summaries matter.
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
KT="python tools/kernel_times.py --images 256 --pass-size 256"
DB=/tmp/i2s_profile_db
mkdir -p gpurun_out "$DB"
for W in clean noisy; do
  EXTRA=""; [ $W = noisy ] && EXTRA="--noisy"
  rocprofv3 --kernel-trace --stats -d "$DB/${TAG}_${W}_trace" -o r -- $KT --reps 3 $EXTRA > gpurun_out/${TAG}_${W}_trace.log 2>&1
  python tools/rocpd_stats.py "$DB/${TAG}_${W}_trace/r_results.db" gpurun_out/${TAG}_${W}_kernel_stats.csv > /dev/null
  python tools/rocpd_sequence.py "$DB/${TAG}_${W}_trace/r_results.db" gpurun_out/${TAG}_${W}_dispatches.csv
  grep '^{' gpurun_out/${TAG}_${W}_trace.log | tail -1 > gpurun_out/${TAG}_${W}_kernel_times.json
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$DB/${TAG}_pmc_fetch" -o r -- $KT --reps 1 > gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$DB/${TAG}_pmc_write" -o r -- $KT --reps 1 > gpurun_out/${TAG}_pmc_write.log 2>&1
python tools/rocpd_pmc.py "$DB/${TAG}_pmc_fetch/r_results.db" gpurun_out/${TAG}_pmc_fetch_size_pass256.csv > /dev/null
python tools/rocpd_pmc.py "$DB/${TAG}_pmc_write/r_results.db" gpurun_out/${TAG}_pmc_write_size_pass256.csv > /dev/null
python tools/pmc_traffic.py "$DB/${TAG}_pmc_fetch/r_results.db" "$DB/${TAG}_pmc_write/r_results.db" 256 gpurun_out/${TAG}_traffic.json > /dev/null
for W in clean noisy; do
  EXTRA=""; [ $W = noisy ] && EXTRA="--noisy"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d "$DB/${TAG}_pmc_sq_$W" -o r -- $KT --reps 1 $EXTRA > gpurun_out/${TAG}_pmc_sq_$W.log 2>&1
  python tools/rocpd_pmc.py "$DB/${TAG}_pmc_sq_$W/r_results.db" gpurun_out/${TAG}_pmc_sq_${W}_pass256.csv > /dev/null
done
ls -la gpurun_out/${TAG}_*
