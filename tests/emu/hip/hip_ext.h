// tests/emu: hipExtLaunchKernelGGL (kernel launch with start / stop events) -- the emulated build launches and ignores the events
#pragma once
#include <hip/hip_runtime.h>
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, start_event, stop_event, flags, ...) \
    do { (void)(start_event); (void)(stop_event); hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__); } while (0)
