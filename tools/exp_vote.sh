cd $GRAFT_REPO_ROOT
for F in NOATOMIC NOWALK NOCENTRE; do
  I2S_EXTRA_FLAGS="-DI2S_EXP_$F" python -c "from img2sgf_amd import build; build.build(force=True)" > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace -d gpurun_out/exp_$F -o r -- python bench.py --batch 128 --pass-size 64 --steps 1 --warmup 1 --no-cpu < /dev/null > gpurun_out/exp_$F.log 2>&1
done
