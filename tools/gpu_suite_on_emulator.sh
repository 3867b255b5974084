#!/bin/bash
# The whole `-m gpu` suite on the EMULATED kernels (no GPU): the emulated build of the product sources -- or, with --exp NAME, of an
# experiment of tools/experiments/ (its patches applied to a copy of csrc) -- under the product library's name, handed to the suite
# through I2S_LIBRARY (img2sgf_amd/_lib.py).  Every GPU test body runs as it would on a GPU box; what cannot work without the hardware
# is listed in profiles/r05_c_emulated_runs.txt (torch.cuda tensors, RCCL, two round counts of the JPEG entropy iteration that depend on
# how workgroups are scheduled) -- plus, since round 6, test_native_library_loaded: the device says "emulated", so a run of this
# script can never read as a run on an MI355X.
#   usage: tools/gpu_suite_on_emulator.sh [--exp NAME] [pytest args]        (~35 min on 7 processes)
set -u
cd "$(dirname "$0")/.."
if [ "${1:-}" = "--exp" ]; then
  NAME=$2; shift 2
  python tools/experiments/apply.py "$NAME" --emu --seeds 1 > /dev/null 2>&1 || { echo "$NAME: does not apply / not exact"; exit 1; }
  mkdir -p "build/emu_$NAME" && cp "build/exp/$NAME/libi2s_emu.so" "build/emu_$NAME/libi2s_hip.so"
  LIB=$PWD/build/emu_$NAME/libi2s_hip.so
else
  python tests/emu/build_emu.py > /dev/null 2>&1 || exit 1
  mkdir -p build/emu && cp tests/emu/libi2s_emu.so build/emu/libi2s_hip.so
  LIB=$PWD/build/emu/libi2s_hip.so
fi
# I2S_EXPERIMENT=1: "not the product file" is declared; test_native_library_loaded still FAILS, by design
I2S_EXPERIMENT=1 I2S_LIBRARY=$LIB python -m pytest tests -m gpu -q -n 7 -p no:cacheprovider --timeout 1200 "$@"
