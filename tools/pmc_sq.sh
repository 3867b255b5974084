#!/bin/bash
# SQ counters of one device pass of the benchmark workload (two rocprofv3 --pmc passes, 8 SQ slots each).
# usage: tools/pmc_sq.sh OUTDIR [kernel_times.py args]      -> OUTDIR/pmc_sq.csv, OUTDIR/pmc_sq2.csv
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/pmc_sq -o r -- python tools/kernel_times.py --images 128 --reps 1 "$@" > $out/pmc_sq.log 2>&1
python tools/rocpd_pmc.py $out/pmc_sq/r_results.db $out/pmc_sq.csv > /dev/null
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU -d $out/pmc_sq2 -o r -- python tools/kernel_times.py --images 128 --reps 1 "$@" > $out/pmc_sq2.log 2>&1
python tools/rocpd_pmc.py $out/pmc_sq2/r_results.db $out/pmc_sq2.csv > /dev/null
rm -rf $out/pmc_sq $out/pmc_sq2
