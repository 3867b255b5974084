#!/bin/bash
# The whole `-m gpu` suite on the EMULATED kernels (no GPU): the emulated build of the product sources under the product library's name,
# handed to the suite through I2S_LIBRARY (img2sgf_amd/_lib.py).  Every GPU test body runs as it would on a GPU box; what cannot work
# without the hardware is listed in profiles/r05_c_emulated_runs.txt (torch.cuda tensors, RCCL, two round counts of the JPEG entropy iteration that depend on how workgroups are scheduled).
#   usage: tools/gpu_suite_on_emulator.sh [pytest args]        (~35 min on 7 processes)
set -u
cd "$(dirname "$0")/.."
python tests/emu/build_emu.py > /dev/null 2>&1 || exit 1
mkdir -p build/emu && cp tests/emu/libi2s_emu.so build/emu/libi2s_hip.so
# I2S_EXPERIMENT=1: "not the product file" is declared; test_native_library_loaded still FAILS, by design (the device says "emulated"):
# a run of this script can never read as a run on an MI355X
I2S_EXPERIMENT=1 I2S_LIBRARY=$PWD/build/emu/libi2s_hip.so python -m pytest tests -m gpu -q -n 7 -p no:cacheprovider --timeout 1200 "$@"
