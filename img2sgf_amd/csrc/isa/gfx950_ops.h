// Instruction-level wrappers of the gfx950 kernels: everything that names a machine instruction, a DPP control word, a
// buffer descriptor or an inline-asm scheduling trick lives here, so that the kernel sources themselves contain no
// conditional compilation.  The GPU-less test build (tests/emu) puts its own header of the same name -- plain C stand-ins
// with the same semantics -- in front of this one on the include path; the product build never sees that file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace i2s {

// three-operand min / med / max: one instruction each (the compiler forms v_med3 but leaves min3 / max3 as two instructions)
__device__ __forceinline__ int imin3(int a, int b, int c) { int r; asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int imax3(int a, int b, int c) { int r; asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int imed3(int a, int b, int c) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Two things the compiler must not "improve" (both measured, profiles/r02_a_valu_rate_*.txt):
//  * a multiply-add whose tap sits in a scalar register issues at HALF rate (v_fmac_f32 with an SGPR operand 4.4 cycles,
//    with vector operands 2.9), and uniform kernel arguments land in SGPRs -- the taps are moved to vector registers once;
//  * (float)a + (float)b of two bytes becomes an SDWA integer add + v_cvt_f32_u32 (two half-rate instructions per sum);
//    converting every byte once with v_cvt_f32_ubyteN and adding floats (full rate) is cheaper.
__device__ __forceinline__ float bl_vgpr(float x) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(x)); return r; }
__device__ __forceinline__ unsigned bl_vgpr_u(unsigned x) { unsigned r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(x)); return r; }
template <int BYTE> __device__ __forceinline__ float bl_fb(unsigned v)
{
    float r;
    if (BYTE == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(v));
    else if (BYTE == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(v));
    else if (BYTE == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(v));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// Neighbour lanes' dwords: DPP whole-wave shifts (one VALU move each) instead of a round trip through the LDS crossbar.
// The lane without a source keeps the `old` operand (bound_ctrl off), i.e. `fill`, at no extra cost.
__device__ __forceinline__ unsigned bl_from_prev_lane(unsigned v, unsigned fill) { return (unsigned)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false); }   // wave_shr:1
__device__ __forceinline__ unsigned bl_from_next_lane(unsigned v, unsigned fill) { return (unsigned)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false); }   // wave_shl:1

// Loads and stores go through a buffer descriptor: plane base in four scalar registers, the row's byte offset in a scalar register, the
// lane's column offset in one vector register -- `buffer_load_dword v, v_off, s[rsrc], s_row offen` needs no address
// arithmetic on the vector unit at all, where a global load costs a 64-bit vector add (v_lshl_add_u64) per access
// (k_sobel_nms_rows 9.06 -> 8.84 us per diagram).
struct BlBuf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ BlBuf bl_buf(const void* p)
{
    return BlBuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000)};     // raw, no stride, 2 GB window
}
__device__ __forceinline__ unsigned bl_bload(BlBuf b, int row_off, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b32(b.r, (int)off, row_off, 0); }
// the planes are streamed out once and read by LATER kernels: non-temporal stores (aux bit 1) measured 3-7 % faster on the
// row kernels (k_blur 1.78 -> 1.73, k_median57_bin 0.76 -> 0.73, main Canny 1.32 -> 1.23 us per diagram)
#ifndef BL_STORE_AUX
#define BL_STORE_AUX 2
#endif
// A lane offset beyond the descriptor's 2 GB window (BL_NO_STORE) makes the hardware drop that lane's store: lanes and whole rows are
// switched off this way, WITHOUT a branch around the instruction -- the compiler's s_waitcnt pass counts only the memory operations
// issued on EVERY path, so stores behind a branch do not count and its waits for the prefetched rows become waits for nearly all
// outstanding stores as well (k_blur: vmcnt(10) instead of vmcnt(40), 3.4 instead of ~6 TB/s of plane writes)
constexpr unsigned BL_NO_STORE = 0xffffffffu;
__device__ __forceinline__ void bl_bstore(BlBuf b, int row_off, unsigned off, unsigned v) { __builtin_amdgcn_raw_buffer_store_b32(v, b.r, (int)off, row_off, BL_STORE_AUX); }

// "The prefetched registers are needed HERE": an empty statement that consumes them, so that the wait for the loads is placed
// at this point.  On gfx9 loads and stores retire in order through one counter and the compiler, after the branches around
// the predicated stores, must assume none of them is pending: a wait for last row's load placed AFTER this row's stores
// therefore drains the stores as well (measured: 41 % of all wave cycles in s_waitcnt).  Waiting for the prefetch first and
// storing afterwards leaves the stores a whole row of arithmetic to complete.
#define BL_CONSUME(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define BL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Placed inside a wave-uniform `if`: keeps it a BRANCH.  Left alone the compiler if-converts small bodies into "compute both, select"
// (three v_perm + three v_cndmask per row in every wavefront for a border fix only the image's edge columns need).
#define BL_KEEP_BRANCH() asm volatile("; uniform branch kept")
// a value the optimiser cannot see through (keeps sign-mask arithmetic from being folded back into compare + select)
__device__ __forceinline__ int opaque_vgpr(int d) { asm("" : "+v"(d)); return d; }

// v_bitop3_b32: any boolean function of three operands in one full-rate instruction.  TT = truth table with the operands
// enumerated as a = 0xF0, b = 0xCC, c = 0xAA (e.g. a ^ b ^ c = 0x96, majority = 0xE8, a ? b : c = 0xCA).
template <int TT> __device__ __forceinline__ unsigned bitop3(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, TT); }

// v_alignbyte_b32: bytes n .. n + 3 of the 8-byte value {hi:lo}
__device__ __forceinline__ unsigned alignbyte(unsigned hi, unsigned lo, unsigned n) { return __builtin_amdgcn_alignbyte(hi, lo, n); }

// 0xff / 0x00 per byte from the byte's top bit: v_perm_b32 selectors 8 .. 11 replicate the sign of bytes 1, 3, 5, 7 of {hi:lo};
// with hi = t << 8 those are bytes 1, 3 (lo) and 0, 2 (hi) of t.
__device__ __forceinline__ unsigned bytes_from_sign(unsigned t) { return __builtin_amdgcn_perm(t << 8, t, 0x090b080au); }

// Two unsigned 16-bit values per register (VOP3P): the two-valued mode of k_blur keeps pixels (x0, x0 + 2) and (x0 + 1, x0 + 3) of a
// lane's dword in the halves of two registers.  pk_mad_u16_sat saturates each half at 65535 (the instruction's clamp bit).
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2u16 pk_v(unsigned u) { v2u16 r; __builtin_memcpy(&r, &u, 4); return r; }
__device__ __forceinline__ unsigned pk_u(v2u16 v) { unsigned u; __builtin_memcpy(&u, &v, 4); return u; }
__device__ __forceinline__ unsigned pk_add_u16(unsigned a, unsigned b) { return pk_u(pk_v(a) + pk_v(b)); }
__device__ __forceinline__ unsigned pk_mul_u16(unsigned a, unsigned b) { return pk_u(pk_v(a) * pk_v(b)); }
__device__ __forceinline__ unsigned pk_mad_u16(unsigned a, unsigned b, unsigned c) { return pk_u(pk_v(a) * pk_v(b) + pk_v(c)); }
__device__ __forceinline__ unsigned pk_mad_u16_sat(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_pk_mad_u16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// a.lo * b.lo + a.hi * b.hi + c in 32 bits (v_dot2_u32_u16)
__device__ __forceinline__ unsigned udot2_u16(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_udot2(pk_v(a), pk_v(b), c, false); }

// ---- grid-wide barrier of a small persistent grid (k_hysteresis_tail) ----------------------------------------------------------
// HY_TAIL_BLOCKS workgroups of 256 threads -- one per CU, an eighth of what the chip keeps resident, so every workgroup of the grid
// is running whatever else shares the GPU.  One monotonic counter in global memory; barrier number n is passed when it reaches
// (n + 1) * gridDim.x.  Per MI355X_MICROARCH.md (inter-workgroup visibility): EVERY wavefront releases at agent scope before the
// workgroup barrier in front of the arrival (its own map stores become visible in L2 / memory -- round 3 left that to lane 0 and the
// barrier's cumulativity), lane 0 arrives and polls with relaxed agent-scope loads, and every wavefront acquires at agent
// scope behind the second barrier (its CU's vector L1 is invalidated for ITS later loads); the inline s_waitcnt keeps the compiler
// from dropping the wait behind the release.  Every spin is bounded: on a timeout the caller gives up and the host falls back to
// plain launches -- for the rest of the context's life (i2s_ctx::hy_no_tail).
constexpr int HY_TAIL_BLOCKS = 256;
__device__ __forceinline__ bool grid_barrier(int* counter, int& target, int* s_ok)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        target += (int)gridDim.x;
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        for (long spins = 0; __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; spins++) {
            __builtin_amdgcn_s_sleep(8);
            if (spins > (1l << 22)) { ok = 0; break; }
        }
        *s_ok = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return *s_ok != 0;
}
// a value another workgroup wrote before the last grid barrier
__device__ __forceinline__ int load_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace i2s
