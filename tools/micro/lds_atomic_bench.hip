// Micro-benchmark: LDS atomic (ds_add_u32, no return) cost per wave instruction on one MI355X under different address
// patterns and active-lane counts.  The loop body is 2 VALU + the atomic, so the LDS is the limiter.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o lds_atomic_bench lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int CELLS = 8192;

// MODE: address pattern; ACTIVE: lanes with (lane % (64 / ACTIVE)) == 0 issue the atomic
template <int MODE, int ACTIVE>
__global__ __launch_bounds__(512) void k(unsigned* out, int iters)
{
    __shared__ unsigned s[CELLS];
    for (int i = threadIdx.x; i < CELLS; i += 512) s[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned h = (threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u) >> 7;
    unsigned idx, step;
    if (MODE == 0) { idx = lane; step = 64; }                         // consecutive dwords, all banks evenly
    else if (MODE == 1) { idx = h; step = 2 * (h >> 3) + 1; }          // random cells, random odd stride
    else if (MODE == 2) { idx = (lane >> 1) * 33; step = 64; }        // 2 lanes per cell
    else if (MODE == 3) { idx = (lane >> 2) * 33; step = 64; }        // 4 lanes per cell
    else if (MODE == 4) { idx = (lane >> 3) * 33; step = 64; }        // 8 lanes per cell
    else if (MODE == 5) { idx = (lane >> 5) * 33; step = 64; }        // 32 lanes per cell
    else if (MODE == 6) { idx = lane * 32; step = 32; }               // one bank, distinct cells
    else { idx = lane * 129; step = 1; }                              // column walk, stride 129
    const bool act = (lane % (64 / ACTIVE)) == 0;
    for (int it = 0; it < iters; it++) {
        if (act) atomicAdd(&s[idx & (CELLS - 1)], 1u);
        idx += step;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[100] + s[0];
}

template <int MODE, int ACTIVE>
static void run(const char* name, unsigned* d_out)
{
    const int blocks = 1024, iters = 8192;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, ACTIVE>), dim3(blocks), dim3(512), 0, 0, d_out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, ACTIVE>), dim3(blocks), dim3(512), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_cu = (double)blocks * 8 * iters / 256;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-34s active %2d  %8.3f ms  %6.2f cycles per wave atomic  %6.2f active lanes per CU-cycle\n", name, ACTIVE, ms,
           cycles / wave_instr_per_cu, wave_instr_per_cu * ACTIVE / cycles);
}

int main()
{
    unsigned* d_out;
    hipMalloc(&d_out, 4096 * sizeof(unsigned));
    run<0, 64>("consecutive dwords", d_out);
    run<0, 32>("consecutive dwords", d_out);
    run<0, 16>("consecutive dwords", d_out);
    run<0, 4>("consecutive dwords", d_out);
    run<1, 64>("random cells", d_out);
    run<1, 32>("random cells", d_out);
    run<1, 16>("random cells", d_out);
    run<1, 4>("random cells", d_out);
    run<2, 64>("2 lanes per cell", d_out);
    run<3, 64>("4 lanes per cell", d_out);
    run<4, 64>("8 lanes per cell", d_out);
    run<5, 64>("32 lanes per cell", d_out);
    run<6, 64>("one bank, distinct cells", d_out);
    run<7, 64>("column walk, stride 129", d_out);
    hipFree(d_out);
    return 0;
}
