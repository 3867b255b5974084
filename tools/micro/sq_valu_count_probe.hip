// Which instruction classes does SQ_INSTS_VALU count?  (profiles/r06_valu_model.md: the modelled vector-instruction count of
// k_vote_centres is 10.8 % above round 4's counter while four other kernels match to 0.6 - 5.4 %; k_vote_centres is the one rich in
// v_cmp -> SGPR pair, v_readlane, DPP moves and EXEC-masked regions.)  Every kernel below issues a KNOWN number of one instruction class
// per wavefront: 64 per loop trip, `iters` trips, one wavefront per workgroup.  Run under
//     rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d DIR -o r -- ./sq_valu_count_probe
// and compare each kernel's SQ_INSTS_VALU / SQ_WAVES with 64 * iters (+ the handful of set-up instructions): tools/rocpd_pmc.py prints it.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o sq_valu_count_probe tools/micro/sq_valu_count_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(S) S S S S S S S S
#define REP64(S) REP8(REP8(S))
#define PROBE(NAME, ASM, CLOBBER...)                                                                  \
    __global__ __launch_bounds__(64) void probe_##NAME(unsigned* out, int iters)                     \
    {                                                                                                 \
        unsigned a = threadIdx.x * 3 + 1, b = threadIdx.x * 7 + 5;                                    \
        for (int it = 0; it < iters; it++) asm volatile(REP64(ASM) : "+v"(a) : "v"(b) : CLOBBER);     \
        out[blockIdx.x * 64 + threadIdx.x] = a;                                                       \
    }

PROBE(add, "v_add_u32 %0, %0, %1\n", "memory")
PROBE(cmp_vcc, "v_cmp_gt_u32 vcc, %0, %1\n", "vcc")
PROBE(cmp_sgpr, "v_cmp_gt_u32 s[20:21], %0, %1\n", "s20", "s21")
PROBE(cmpx, "v_cmp_le_u32 vcc, 0, %1\n s_and_b64 s[20:21], vcc, exec\n", "vcc", "s20", "s21")
PROBE(readlane, "v_readlane_b32 s20, %0, 3\n", "s20")
PROBE(readfirstlane, "v_readfirstlane_b32 s20, %0\n", "s20")
PROBE(dpp, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n", "memory")
PROBE(cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n", "memory")
PROBE(bitop3, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96\n", "memory")
PROBE(mbcnt, "v_mbcnt_lo_u32_b32 %0, -1, %0\n", "memory")
// the same v_add under an EXEC mask with 1 lane, and with NO lane (an instruction issued with EXEC = 0)
__global__ __launch_bounds__(64) void probe_add_one_lane(unsigned* out, int iters)
{
    unsigned a = threadIdx.x * 3 + 1, b = threadIdx.x * 7 + 5;
    if (threadIdx.x == 5)
        for (int it = 0; it < iters; it++) asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
__global__ __launch_bounds__(64) void probe_add_exec_zero(unsigned* out, int iters)
{
    unsigned a = threadIdx.x * 3 + 1, b = threadIdx.x * 7 + 5;
    for (int it = 0; it < iters; it++)
        asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0\n" REP64("v_add_u32 %0, %0, %1\n") "s_mov_b64 exec, s[20:21]\n" : "+v"(a) : "v"(b) : "s20", "s21");
    out[blockIdx.x * 64 + threadIdx.x] = a;
}

int main()
{
    const int blocks = 1024, iters = 256;
    unsigned* out;
    if (hipMalloc(&out, blocks * 64 * sizeof(unsigned)) != hipSuccess) { printf("no device\n"); return 1; }
#define RUN(NAME) hipLaunchKernelGGL(probe_##NAME, dim3(blocks), dim3(64), 0, 0, out, iters); if (hipDeviceSynchronize() != hipSuccess) { printf(#NAME " failed\n"); return 1; }
    RUN(add) RUN(cmp_vcc) RUN(cmp_sgpr) RUN(cmpx) RUN(readlane) RUN(readfirstlane) RUN(dpp) RUN(cndmask) RUN(bitop3) RUN(mbcnt) RUN(add_one_lane) RUN(add_exec_zero)
    printf("expected per wavefront: %d probe instructions (+ set-up); %d wavefronts per kernel\n", 64 * iters, blocks);
    return 0;
}
