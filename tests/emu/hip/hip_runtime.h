// tests/emu/hip/hip_runtime.h  --  TEST INFRASTRUCTURE ONLY (never shipped, never loaded by the product).
//
// A minimal emulation (one kernel at a time, host threads welcome) of the slice of the HIP programming model that
// img2sgf_amd/csrc uses, so that the *unmodified* kernel and host sources can be compiled with g++
// (-I tests/emu puts this file in front of the real <hip/hip_runtime.h>) and exercised against the
// oracle in the GPU-less build container.  Every GPU thread of a workgroup is a fiber (own stack, user-space switch);
// __syncthreads() and the wave-level primitives yield to a scheduler that advances 64-lane waves in
// lockstep.  hipMalloc'd memory is filled with 0xCD so reads of uninitialised device memory show up.
// It catches indexing / logic errors; it does not model caches, memory ordering or performance.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct short2 { short x, y; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline int2 make_int2(int a, int b) { return int2{a, b}; }

typedef int hipError_t;
typedef struct hipemu_stream* hipStream_t;
typedef struct hipemu_event* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

namespace hipemu {
enum YieldKind { Y_NONE = 0, Y_BLOCK = 1, Y_WAVE = 2, Y_DONE = 3 };
struct State {
    uint3 tIdx, bIdx;
    dim3 bDim, gDim;
    int lane, wave, linear;
    // wave-op scratch: [parity][lane]
    unsigned long long scratch[2][64];
    unsigned long long stamp[2][64];
};
extern State* cur;          // state of the running fiber
void yield(YieldKind k);
void run_grid(dim3 grid, dim3 block, const std::function<void()>& body);
unsigned long long wave_exchange(unsigned long long v, unsigned long long* all, unsigned long long* active);
}  // namespace hipemu

#define threadIdx (hipemu::cur->tIdx)
#define blockIdx (hipemu::cur->bIdx)
#define blockDim (hipemu::cur->bDim)
#define gridDim (hipemu::cur->gDim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::run_grid(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::yield(hipemu::Y_BLOCK); }

// ---- wave primitives (all live lanes of the wave must call them together) ----
static inline unsigned long long __ballot(int pred)
{
    unsigned long long all[64], active;
    hipemu::wave_exchange(pred ? 1ull : 0ull, all, &active);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) if (((active >> i) & 1) && all[i]) m |= 1ull << i;
    return m;
}
template <class T> static inline T __shfl(T v, int src)
{
    static_assert(sizeof(T) <= 8, "shfl type");
    unsigned long long all[64], active, raw = 0;
    memcpy(&raw, &v, sizeof(T));
    hipemu::wave_exchange(raw, all, &active);
    src &= 63;
    if (!((active >> src) & 1)) return v;
    T out;
    memcpy(&out, &all[src], sizeof(T));
    return out;
}
template <class T> static inline T __shfl_down(T v, unsigned d) { return __shfl(v, hipemu::cur->lane + (int)d < 64 ? hipemu::cur->lane + (int)d : hipemu::cur->lane); }
template <class T> static inline T __shfl_up(T v, unsigned d) { return __shfl(v, hipemu::cur->lane - (int)d >= 0 ? hipemu::cur->lane - (int)d : hipemu::cur->lane); }
static inline int __any(int pred) { return __ballot(pred) != 0ull; }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
template <class T> static inline T __shfl_xor(T v, int m) { return __shfl(v, hipemu::cur->lane ^ m); }
// wave-uniform lane select (v_readlane_b32): all live lanes call it with the same lane index
static inline int __builtin_amdgcn_readlane(int v, int lane) { return __shfl(v, lane); }
// value of the first live lane; the callers pass wave-uniform values, for which this is the identity (and safe after lanes left)
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
// wave-level barrier: on hardware only a scheduling barrier (a wave runs in lockstep); here all live lanes rendezvous
static inline void __builtin_amdgcn_wave_barrier() { unsigned long long a[64], m; hipemu::wave_exchange(0, a, &m); }
static inline int __lane_id() { return hipemu::cur->lane; }
// v_mbcnt_lo / _hi: bits of the mask below this lane (per 32-bit half), plus the running count
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned add)
{
    const int l = hipemu::cur->lane;
    return add + (unsigned)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u)));
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned add)
{
    const int l = hipemu::cur->lane;
    return add + (l > 32 ? (unsigned)__builtin_popcount(mask & ((1u << (l - 32)) - 1u)) : 0u);
}

static inline unsigned long long __brevll(unsigned long long v)
{
    unsigned long long r = 0;
    for (int i = 0; i < 64; i++) r |= ((v >> i) & 1ull) << (63 - i);
    return r;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
// v_perm_b32: result byte i = byte sel[i] of the 8-byte value {hi:lo} (0-3 = lo, 4-7 = hi), 0x0c = 0x00
static inline unsigned __builtin_amdgcn_perm(unsigned hi, unsigned lo, unsigned sel)   // 8 .. 11: sign of byte 1, 3, 5, 7 replicated
{
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    unsigned r = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned s = (sel >> (8 * i)) & 0xff;
        unsigned b = 0;
        if (s < 8) b = (unsigned)(v >> (8 * s)) & 0xff;
        else if (s == 0x0c) b = 0;
        else if (s >= 0x0d) b = 0xff;
        else if (s <= 0x0b) b = ((v >> (8 * (2 * (s - 8) + 1) + 7)) & 1) ? 0xff : 0;
        else { fprintf(stderr, "hipemu: unsupported v_perm selector %#x\n", s); abort(); }
        r |= b << (8 * i);
    }
    return r;
}
// v_alignbit_b32: ({hi:lo} >> shift) & 0xffffffff
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned shift)
{
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (shift & 31));
}
static inline int __float_as_int(float f) { int i; __builtin_memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; __builtin_memcpy(&f, &i, 4); return f; }
// v_sad_u8: sum of absolute differences of the four bytes of a and b, plus c
static inline unsigned __builtin_amdgcn_sad_u8(unsigned a, unsigned b, unsigned c)
{
    for (int i = 0; i < 4; i++) {
        const int x = (int)((a >> (8 * i)) & 0xffu), y = (int)((b >> (8 * i)) & 0xffu);
        c += (unsigned)(x > y ? x - y : y - x);
    }
    return c;
}
// v_dot4_u32_u8: byte-wise dot product of a and b, plus c
static inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned c, bool)
{
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xffu) * ((b >> (8 * i)) & 0xffu);
    return c;
}
// v_dot4_i32_i8: the same with signed bytes
static inline int __builtin_amdgcn_sdot4(int a, int b, int c, bool)
{
    for (int i = 0; i < 4; i++) c += (int)(signed char)((unsigned)a >> (8 * i)) * (int)(signed char)((unsigned)b >> (8 * i));
    return c;
}
static inline int __mul24(int a, int b) { return a * b; }
static inline unsigned __umul24(unsigned a, unsigned b) { return a * b; }
static inline int __float2int_rn(float v) { return (int)lrintf(v); }
static inline int __double2int_rn(double v) { return (int)lrint(v); }

// ---- atomics (single OS thread: plain ops) ----
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, int v) { unsigned o = *p; *p = o + (unsigned)v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

// ---- host runtime ----
static inline const char* hipGetErrorString(hipError_t e) { return e == 0 ? "hipSuccess (emu)" : "hipError (emu)"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
struct hipDeviceProp_t { char gcnArchName[256]; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { snprintf(p->gcnArchName, sizeof p->gcnArchName, "emulated"); return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); if (!*p) return 2; memset(*p, 0xCD, n); return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t wbytes, size_t h, hipMemcpyKind, hipStream_t = 0)
{
    for (size_t y = 0; y < h; y++) memcpy((char*)d + y * dp, (const char*)s + y * sp, wbytes);
    return hipSuccess;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = 0; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = 0) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
