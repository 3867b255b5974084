// Streaming-store ceilings on gfx950 in the access patterns of the row kernels (k_blur, k_sobel_nms_rows, k_median57_bin):
// a wavefront walks down R rows of a 256-pixel column group; per row it reads 256 B of one plane and writes 256 B (dword per lane) to
// each of P output planes.  Variants: stores plain / nt; dword per lane vs dwordx4 per lane (a wavefront then covers 1024 B of a row).
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../img2sgf_amd/csrc/isa/gfx950_ops.h"
using namespace i2s;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static inline __device__ int imin(int a, int b) { return a < b ? a : b; }

template <int P, bool NT, bool READ>
__global__ __launch_bounds__(256) void k_rows_dword(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int pitch_w, int rows_per, int h, size_t plane_w)
{
    // grid: (w / 1024) x (h / rows_per) x images; block 256 = 4 wavefronts side by side
    const int lane_x = blockIdx.x * 256 + threadIdx.x;            // dword column
    const int y0 = blockIdx.y * rows_per;
    const size_t img = (size_t)blockIdx.z * plane_w;
    unsigned acc = threadIdx.x;
    for (int r = 0; r < rows_per && y0 + r < h; r++) {
        const size_t o = img + (size_t)(y0 + r) * pitch_w + lane_x;
        if (READ) acc += src[o];
#pragma unroll
        for (int p = 0; p < P; p++) {
            unsigned* d = dst + (size_t)p * plane_w * gridDim.z + o;
            if (NT) __builtin_nontemporal_store(acc + p, d); else *d = acc + p;
        }
    }
}

template <int P, bool NT, bool READ>
__global__ __launch_bounds__(256) void k_rows_x4(const uint4* __restrict__ src, uint4* __restrict__ dst, int pitch_q, int rows_per, int h, size_t plane_q)
{
    // a lane owns 16 bytes of a row: block 256 = 4096 B of a row; grid (w / 4096) x (h / rows_per) x images
    const int lane_x = blockIdx.x * 256 + threadIdx.x;
    const int y0 = blockIdx.y * rows_per;
    const size_t img = (size_t)blockIdx.z * plane_q;
    uint4 acc = make_uint4(threadIdx.x, 1, 2, 3);
    for (int r = 0; r < rows_per && y0 + r < h; r++) {
        const size_t o = img + (size_t)(y0 + r) * pitch_q + lane_x;
        if (READ) { const uint4 v = src[o]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
        for (int p = 0; p < P; p++) {
            uint4* d = dst + (size_t)p * plane_q * gridDim.z + o;
            uint4 v = acc; v.x += p;
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u vv = {v.x, v.y, v.z, v.w};
            if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<v4u*>(d)); else *d = v;
        }
    }
}


// the row kernels' own load / store scheme: buffer descriptors, the edge dword of lanes 0 / 63, DEPTH rows of loads in flight, the wait
// for the next row placed before this row's stores (BL_CONSUME), P planes of non-temporal dword stores; ALU = a few XORs
template <int P, int DEPTH, bool EDGE, bool FENCE, int ALU = 0>
__global__ __launch_bounds__(256) void k_rows_scheme(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int pitch, int rows_per, int h, size_t plane)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = ((blockIdx.x * 4 + wave) * 64 + lane) * 4;
    const int y0 = blockIdx.y * rows_per;
    const BlBuf sbuf = bl_buf(src + (size_t)blockIdx.z * plane);
    BlBuf ob[P];
#pragma unroll
    for (int p = 0; p < P; p++) ob[p] = bl_buf(dst + ((size_t)p * gridDim.z + blockIdx.z) * plane);
    const unsigned xm = x0, xe = lane == 0 ? (x0 >= 4 ? x0 - 4 : 0) : (lane == 63 && x0 + 4 < pitch ? x0 + 4 : 0);
    unsigned qM[DEPTH], qE[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) { const int ro = imin(y0 + d, h - 1) * pitch; qM[d] = bl_bload(sbuf, ro, xm); qE[d] = EDGE ? bl_bload(sbuf, ro, xe) : 0u; }
    unsigned acc = 0;
    for (int r = 0; r < rows_per; r++) {
        const unsigned M = qM[0], E = qE[0];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; d++) { qM[d] = qM[d + 1]; qE[d] = qE[d + 1]; }
        { const int ro = imin(y0 + r + DEPTH, h - 1) * pitch; qM[DEPTH - 1] = bl_bload(sbuf, ro, xm); qE[DEPTH - 1] = EDGE ? bl_bload(sbuf, ro, xe) : 0u; }
        const unsigned L = bl_from_prev_lane(M, E), R = bl_from_next_lane(M, E);
        acc ^= L ^ M ^ R;
        // ALU: dependent half-rate vector instructions per row (v_perm_b32), the row kernels' kind of work
#pragma unroll
        for (int a = 0; a < ALU; a++) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(acc) : "v"(M), "v"(0x06050403u));
        if (FENCE) { BL_SCHED_FENCE(); BL_CONSUME(qM[0], qE[0]); BL_SCHED_FENCE(); }
        const int off = (y0 + r) * pitch;
#pragma unroll
        for (int p = 0; p < P; p++) bl_bstore(ob[p], off, xm, acc + p);
    }
}

template <class F> static double time_ms(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const int W = 1024, H = 1024, NIMG = 256, PMAX = 6;
    const size_t plane = (size_t)W * H;                        // bytes per image plane
    unsigned *src, *dst;
    CK(hipMalloc(&src, plane * NIMG));
    CK(hipMalloc(&dst, plane * NIMG * PMAX));
    CK(hipMemset(src, 1, plane * NIMG));
    printf("%-44s %8s %8s\n", "variant (256 images of 1024 x 1024 bytes)", "ms", "TB/s");
#define RUN_D(P, NT, RD, ROWS) do { \
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows_dword<P, NT, RD>), dim3(W / 1024, H / ROWS, NIMG), dim3(256), 0, 0, src, dst, W / 4, ROWS, H, plane / 4); }, 5); \
        const double bytes = (double)plane * NIMG * (P + (RD ? 1 : 0)); \
        printf("dword/lane  planes=%d %-5s read=%d rows/wave=%-3d     %8.3f %8.2f\n", P, NT ? "nt" : "plain", RD ? 1 : 0, ROWS, ms, bytes / ms / 1e9); } while (0)
#define RUN_Q(P, NT, RD, ROWS) do { \
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows_x4<P, NT, RD>), dim3(1, H / ROWS, NIMG), dim3(64), 0, 0, (const uint4*)src, (uint4*)dst, W / 16, ROWS, H, plane / 16); }, 5); \
        const double bytes = (double)plane * NIMG * (P + (RD ? 1 : 0)); \
        printf("dwordx4/lane planes=%d %-5s read=%d rows/wave=%-3d    %8.3f %8.2f\n", P, NT ? "nt" : "plain", RD ? 1 : 0, ROWS, ms, bytes / ms / 1e9); } while (0)
    RUN_D(0, false, true, 64);
    RUN_D(1, false, false, 64); RUN_D(1, true, false, 64);
    RUN_D(2, true, true, 64); RUN_D(2, false, true, 64);
    RUN_D(4, true, true, 64); RUN_D(4, false, true, 64); RUN_D(4, true, true, 32); RUN_D(4, true, true, 16);
    RUN_D(6, true, true, 64);
    RUN_Q(0, false, true, 64);
    RUN_Q(1, true, false, 64); RUN_Q(1, false, false, 64);
    RUN_Q(2, true, true, 64);
    RUN_Q(4, true, true, 64); RUN_Q(4, false, true, 64); RUN_Q(4, true, true, 16);
    RUN_Q(6, true, true, 64);
#define RUN_A(P, DEPTH, ALU, ROWS) do { \
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows_scheme<P, DEPTH, true, true, ALU>), dim3(W / 1024, H / ROWS, NIMG), dim3(256), 0, 0, (const uint8_t*)src, (uint8_t*)dst, W, ROWS, H, plane); }, 5); \
        const double bytes = (double)plane * NIMG * (P + 1); \
        printf("row scheme  planes=%d depth=%d + %3d v_perm per row              %8.3f %8.2f\n", P, DEPTH, ALU, ms, bytes / ms / 1e9); } while (0)
#define RUN_S(P, DEPTH, EDGE, FENCE, ROWS) do { \
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows_scheme<P, DEPTH, EDGE, FENCE>), dim3(W / 1024, H / ROWS, NIMG), dim3(256), 0, 0, (const uint8_t*)src, (uint8_t*)dst, W, ROWS, H, plane); }, 5); \
        const double bytes = (double)plane * NIMG * (P + 1); \
        printf("row scheme  planes=%d depth=%d edge=%d fence=%d rows/wave=%-3d  %8.3f %8.2f\n", P, DEPTH, EDGE ? 1 : 0, FENCE ? 1 : 0, ROWS, ms, bytes / ms / 1e9); } while (0)
    RUN_S(2, 2, true, true, 64); RUN_S(2, 2, false, true, 64); RUN_S(2, 2, true, false, 64); RUN_S(2, 4, true, true, 64); RUN_S(2, 8, true, true, 64);
    RUN_A(2, 2, 25, 64); RUN_A(2, 2, 50, 64); RUN_A(2, 2, 100, 64); RUN_A(2, 2, 200, 64); RUN_A(2, 4, 50, 64); RUN_A(2, 4, 100, 64);
    RUN_A(4, 4, 100, 64); RUN_A(4, 4, 200, 64);
    RUN_S(4, 2, true, true, 64); RUN_S(4, 4, true, true, 64); RUN_S(4, 4, false, false, 64); RUN_S(6, 4, true, true, 64);
    // Round 4: what ONE launch costs in a pipeline, where it is neither repeated nor preceded by itself: single launches of the
    // 6-plane scheme, timed one by one -- (a) back to back, (b) each after the GPU sat idle for 3 ms, (c) idle and into a region
    // of memory that was not written for three launches.  (The table above averages five back-to-back launches after a warm-up.)
    {
        unsigned* big;
        CK(hipMalloc(&big, plane * NIMG * PMAX * 4));
        hipEvent_t ev[9];
        for (int i = 0; i < 9; i++) CK(hipEventCreate(&ev[i]));
        auto launch = [&](unsigned* d) { hipLaunchKernelGGL((k_rows_scheme<6, 4, true, true>), dim3(W / 1024, H / 64, NIMG), dim3(256), 0, 0, (const uint8_t*)src, (uint8_t*)d, W, 64, H, plane); };
        for (int mode = 0; mode < 3; mode++) {
            CK(hipDeviceSynchronize());
            usleep(5000);
            float ms[8];
            if (mode == 0) {
                for (int i = 0; i < 8; i++) { CK(hipEventRecord(ev[i])); launch(big); }
                CK(hipEventRecord(ev[8])); CK(hipEventSynchronize(ev[8]));
                for (int i = 0; i < 8; i++) CK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
            } else {
                for (int i = 0; i < 8; i++) {
                    usleep(3000);
                    CK(hipEventRecord(ev[0])); launch(big + (mode == 2 ? (size_t)(i % 4) * plane * NIMG * PMAX / 4 : 0)); CK(hipEventRecord(ev[1]));
                    CK(hipEventSynchronize(ev[1])); CK(hipEventElapsedTime(&ms[i], ev[0], ev[1]));
                }
            }
            printf("%-52s", mode == 0 ? "single launches, back to back (ms):" : (mode == 1 ? "single launches, 3 ms idle before each:" : "... idle and a region not written for 3 launches:"));
            for (int i = 0; i < 8; i++) printf(" %.3f", ms[i]);
            printf("\n");
        }
    }
    return 0;
}
