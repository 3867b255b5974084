#!/bin/bash
# Round 6's FIRST call on a GPU box after GPU use reopens, everything in ONE gpurun call because a reopening may
# be short:   gpurun --timeout 2400 -- 'bash tools/r06_gpu_session.sh'
# 1. the suite the driver runs, exactly as it runs it; 2. the bench exactly as the driver runs it (20 / 5); 3. smoke();
# 4. the round's profiles (tools/profile_round.sh r06).  Logs under gpurun_out/r06/ ; the csrc hash ties every log to its sources.
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
# a live cv2 is worth more than any kernel timing
python -c "import cv2; print('cv2', cv2.__version__)" > $O/cv2_probe.txt 2>&1; cat $O/cv2_probe.txt
if grep -q "^cv2 " $O/cv2_probe.txt; then python -m oracle.cv2_harness --digests > $O/cv2_digests.txt 2>&1; python -m pytest tests/test_cv2_crosscheck.py -q > $O/cv2_crosscheck.log 2>&1; python tools/cpu_baseline.py > $O/cpu_baseline_cv2.txt 2>&1; fi
rocminfo | grep -m3 -E "gfx|Marketing" ; nproc
cat img2sgf_amd/csrc/*.h img2sgf_amd/csrc/*.hip img2sgf_amd/csrc/isa/* include/* | sha256sum | cut -d' ' -f1 > $O/csrc_sha256.txt
echo "csrc sha256: $(cat $O/csrc_sha256.txt)"
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1; tail -4 $O/gputests.log
# a failure under -x hides everything behind it: the whole list then, once
grep -q " failed" $O/gputests.log && { timeout 1500 python -m pytest tests -m gpu -q -n 4 > $O/gputests_all.log 2>&1; tail -15 $O/gputests_all.log; }
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; head -c 400 $O/bench_20_5.json; echo
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
# 4b. which instruction classes SQ_INSTS_VALU counts (profiles/r06_valu_model.md: k_vote_centres' model is 10.8 % above the counter)
( cd /tmp && export TMPDIR=/tmp && hipcc --offload-arch=gfx950 -O3 -w -o /tmp/sq_probe "$GRAFT_REPO_ROOT/tools/micro/sq_valu_count_probe.hip" && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d /tmp/sq_probe_db -o r -- /tmp/sq_probe ) > $O/sq_valu_count_probe.log 2>&1
python tools/rocpd_pmc.py /tmp/sq_probe_db/r_results.db $O/sq_valu_count_probe.csv > /dev/null 2>&1; head -20 $O/sq_valu_count_probe.csv
# 5. the experiments that were proven bit-exact on the emulated kernels while the GPU was closed (tools/experiments/): A/B timings
timeout 3000 bash tools/experiments/ab.sh stride133 tile128+cull_fast tile128 canny_lean edge_lut radius_staged tile128+vastr133 tile128+vastr133+cull_fast+edge_lut+canny_lean+radius_staged tile128+cull_fast+edge_lut+canny_lean+radius_staged radius_pre2 tile128+cull_fast+vote_setup > $O/ab.log 2>&1; tail -8 $O/ab.log
# 6. the whole path with everything together: the bench line of the candidate build (its "lib" field names the file), beside the product's
ALL=tile128+vastr133+cull_fast+edge_lut+canny_lean+radius_staged
[ -f build/exp/$ALL/libi2s_hip.so ] && I2S_LIBRARY=$PWD/build/exp/$ALL/libi2s_hip.so timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $O/bench_candidate_20_5.json 2> $O/bench_candidate_20_5.err; head -c 300 $O/bench_candidate_20_5.json; echo
