// tests/emu/gfx950_ops.h -- TEST INFRASTRUCTURE ONLY.  Plain C stand-ins, with the same semantics, for the instruction-level
// wrappers of img2sgf_amd/csrc/isa/gfx950_ops.h; the emulated build (tests/emu/build_emu.py) finds this header first.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace i2s {

static inline int imin3(int a, int b, int c) { return std::min(std::min(a, b), c); }
static inline int imax3(int a, int b, int c) { return std::max(std::max(a, b), c); }
static inline int imed3(int a, int b, int c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
static inline float bl_vgpr(float x) { return x; }
static inline unsigned bl_vgpr_u(unsigned x) { return x; }
template <int BYTE> static inline float bl_fb(unsigned v) { return (float)((v >> (8 * BYTE)) & 0xffu); }
// value of lane - 1 / lane + 1; lane 0 / lane 63, which have no such neighbour, get `fill`
static inline unsigned bl_from_prev_lane(unsigned v, unsigned fill) { const unsigned u = (unsigned)__shfl_up((int)v, 1); return __lane_id() == 0 ? fill : u; }
static inline unsigned bl_from_next_lane(unsigned v, unsigned fill) { const unsigned u = (unsigned)__shfl_down((int)v, 1); return __lane_id() == 63 ? fill : u; }
constexpr unsigned BL_NO_STORE = 0xffffffffu;
struct BlBuf { uint8_t* p; };
static inline BlBuf bl_buf(const void* p) { return BlBuf{const_cast<uint8_t*>(static_cast<const uint8_t*>(p))}; }
static inline unsigned bl_bload(BlBuf b, int row_off, unsigned off) { unsigned v; memcpy(&v, b.p + row_off + off, 4); return v; }
// (the descriptor's range check: a lane offset beyond the 2 GB window drops the store -- how the row kernels switch lanes / rows off)
static inline void bl_bstore(BlBuf b, int row_off, unsigned off, unsigned v) { if (off >= 0x7fffffffu) return; memcpy(b.p + row_off + off, &v, 4); }
#define BL_KEEP_BRANCH() do {} while (0)
#define BL_CONSUME(a, b) do { } while (0)
#define BL_SCHED_FENCE() do { } while (0)
static inline int opaque_vgpr(int d) { return d; }
template <int TT> static inline unsigned bitop3(unsigned a, unsigned b, unsigned c)
{
    unsigned r = 0;
    for (int i = 0; i < 8; i++)          // minterm i: a = bit 2, b = bit 1, c = bit 0 of i
        if ((TT >> i) & 1) r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
    return r;
}
static inline unsigned alignbyte(unsigned hi, unsigned lo, unsigned n) { return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * (n & 3))); }
static inline unsigned pk_add_u16(unsigned a, unsigned b) { return ((a + b) & 0xffffu) | ((((a >> 16) + (b >> 16)) & 0xffffu) << 16); }
static inline unsigned pk_mul_u16(unsigned a, unsigned b) { return (((a & 0xffffu) * (b & 0xffffu)) & 0xffffu) | ((((a >> 16) * (b >> 16)) & 0xffffu) << 16); }
static inline unsigned pk_mad_u16(unsigned a, unsigned b, unsigned c) { return pk_add_u16(pk_mul_u16(a, b), c); }
static inline unsigned pk_mad_u16_sat(unsigned a, unsigned b, unsigned c)
{
    const unsigned lo = (a & 0xffffu) * (b & 0xffffu) + (c & 0xffffu), hi = (a >> 16) * (b >> 16) + (c >> 16);
    return (lo > 0xffffu ? 0xffffu : lo) | ((hi > 0xffffu ? 0xffffu : hi) << 16);
}
static inline unsigned udot2_u16(unsigned a, unsigned b, unsigned c) { return (a & 0xffffu) * (b & 0xffffu) + (a >> 16) * (b >> 16) + c; }
static inline unsigned bytes_from_sign(unsigned t) { return ((t >> 7) & 0x01010101u) * 0xffu; }

// the emulator runs the workgroups of a grid one after the other: the persistent tail kernel is launched with ONE workgroup, for
// which the grid barrier is a workgroup barrier
constexpr int HY_TAIL_BLOCKS = 1;
static inline bool grid_barrier(int* counter, int& target, int* s_ok) { (void)counter; (void)target; (void)s_ok; __syncthreads(); return true; }
static inline int load_agent(const int* p) { return *p; }

}  // namespace i2s
