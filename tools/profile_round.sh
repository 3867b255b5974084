#!/bin/bash
# Profiles of one round, run on the GPU box from the repo root: kernel trace (per-kernel durations) of the roofline command,
# the two PMC passes for HBM traffic, and the SQ pass.  usage: tools/profile_round.sh r02   (writes gpurun_out/<tag>_*)
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python tools/kernel_times.py --images 128 --reps 1"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace -o r -- python bench.py --batch 256 --pass-size 128 --streams 1 --steps 1 --warmup 1 --roofline-images 256 --no-cpu > gpurun_out/${TAG}_trace.log 2>&1
python tools/rocpd_stats.py gpurun_out/${TAG}_trace/r_results.db gpurun_out/${TAG}_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${TAG}_pmc_fetch -o r -- $CMD > gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${TAG}_pmc_write -o r -- $CMD > gpurun_out/${TAG}_pmc_write.log 2>&1
python tools/rocpd_pmc.py gpurun_out/${TAG}_pmc_fetch/r_results.db gpurun_out/${TAG}_pmc_fetch.csv > /dev/null
python tools/rocpd_pmc.py gpurun_out/${TAG}_pmc_write/r_results.db gpurun_out/${TAG}_pmc_write.csv > /dev/null
python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_fetch/r_results.db gpurun_out/${TAG}_pmc_write/r_results.db 128 gpurun_out/${TAG}_traffic.json > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d gpurun_out/${TAG}_pmc_sq -o r -- $CMD > gpurun_out/${TAG}_pmc_sq.log 2>&1
python tools/rocpd_pmc.py gpurun_out/${TAG}_pmc_sq/r_results.db gpurun_out/${TAG}_pmc_sq.csv > /dev/null
ls -la gpurun_out/${TAG}_*.csv gpurun_out/${TAG}_traffic.json
head -12 gpurun_out/${TAG}_kernel_stats.csv
cat gpurun_out/${TAG}_traffic.json | head -40
