// Shared device/host types of the board-detection library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/i2s.h"

namespace i2s {

// Distinct HoughCircles inputs ("variants") in blur-bank order (img2sgf.py:171-175):
// 0 grey, 1 edges, 2 median3, 3 gauss3, 4 median5, 5 gauss5, 6 median7, 7 gauss7.
// The bank's k=1 median and Gaussian are copies of grey, so slots 0, 2, 3 share variant 0.
constexpr int NVAR = 8;
constexpr int NSLOT = I2S_NSLOTS;
constexpr int NMAP = 9;   // Canny maps: 0 = main Canny (50/200 on the source), 1+v = HoughCircles' internal Canny of variant v
__device__ __host__ inline int slot_variant(int s)
{
    return s == 0 ? 0 : s == 1 ? 1 : s == 2 ? 0 : s == 3 ? 0 : s - 2;
}

// Capacities of the HoughCircles lists per (image, variant).  The reference's lists are unbounded (img2sgf.py:179-186); here
// they are sized at i2s_create from the largest image the context accepts and carried in Geo: one accumulator maximum per 8
// pixels; EST_UNIT / VCIRC_UNIT estimates / circles per started megapixel, up to CAP_SCALE_MAX units (the estimates of one
// HoughCircles call are sorted in LDS) -- so a 2048 x 2048 page scan has four times the room of a 1024 x 1024 diagram.
// An overflow is still reported (I2S_ST_CAPACITY), never truncated.
constexpr int CENT_UNIT = 8192;   // accumulator local maxima: lower bound of the per-area capacity
constexpr int EST_UNIT = 4096;    // supported circle estimates (k_circles_final sorts them in LDS: 8 bytes each)
constexpr int VCIRC_UNIT = 2048;  // circles kept after the min-dist pass
constexpr int CAP_SCALE_MAX = 4;

// Per-image descriptor (device array, one per image of the current pass).
struct ImgDesc {
    const uint8_t* src;   // source pixels (device), channels interleaved
    int w, h;
    int sstride;          // bytes per source row
    int cn;               // 1 or 3
    int line_thr;         // Hough-lines threshold for this image
    int gpitch;           // bytes per row of the grey plane
    const uint8_t* grey;  // grey plane (variant 0) of this image: its slot in the plane array, or -- for a single-channel source
                          // whose rows are dword-aligned -- the source itself (cvtColor of a grey image is the identity)
};

// Plane addressing: plane p of image b starts at base + (p * nb + b) * slot; rows are `pitch` bytes.
struct Geo {
    int pitch;            // bytes per plane row (multiple of 64)
    int hmax;             // rows per plane slot
    int nb;               // images in this pass
    int wmax;
    int bw, bins;         // edge bins per row / per plane (32x32-pixel cells)
    int tw, tiles;        // 64x32 Canny tiles per row / per plane (hysteresis work flags)
    int cent_cap, est_cap, vcirc_cap;   // HoughCircles list capacities per (image, variant), see CENT_UNIT
    long long slot;       // pitch * hmax
};

struct Taps { int k[8]; };

// Tile kernels are launched as 1-D grids of T = tiles_x * tiles_y * planes workgroups.  Workgroup L is observed to run
// on XCD L % 8 (each XCD has its own L2), so L is remapped such that every XCD walks ONE contiguous range of the
// (plane, row-major tile) sequence: neighbouring tiles, which share apron rows/columns, then share an L2 instead of
// each fetching the lines from HBM.  Pure speed: any placement gives the same result.  The map is a bijection on [0, T).
struct TileId { int tx, ty, z; };
__device__ __forceinline__ TileId tile_of_block(unsigned gx, unsigned gy)
{
    const unsigned T = gridDim.x, L = blockIdx.x;
    const unsigned q = T >> 3, r = T & 7u, x = L & 7u, i = L >> 3;
    const unsigned t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    const unsigned per = gx * gy;
    TileId id;
    id.z = (int)(t / per);
    const unsigned rem = t - (unsigned)id.z * per;
    id.ty = (int)(rem / gx);
    id.tx = (int)(rem - (unsigned)id.ty * gx);
    return id;
}

// the same for kernels whose workgroups take `per` consecutive tiles each: first tile of this workgroup's chunk (the chunks of
// one XCD are contiguous), and tile index -> (tx, ty, z)
__device__ __forceinline__ unsigned tile_chunk_of_block(unsigned per, unsigned ntiles)
{
    const unsigned T = gridDim.x, L = blockIdx.x;
    const unsigned q = T >> 3, r = T & 7u, x = L & 7u, i = L >> 3;
    (void)ntiles;
    return ((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i) * per;
}
__device__ __forceinline__ TileId tile_of_index(unsigned t, unsigned gx, unsigned gy)
{
    const unsigned per = gx * gy;
    TileId id;
    id.z = (int)(t / per);
    const unsigned rem = t - (unsigned)id.z * per;
    id.ty = (int)(rem / gx);
    id.tx = (int)(rem - (unsigned)id.ty * gx);
    return id;
}

struct HoughTrig {        // tables of the three HoughLines calls of find_lines (img2sgf.py:236-244)
    int n[3];             // number of angles: [0] horizontal, [1] vertical near 0, [2] vertical near pi
    float sin_[3][4];
    float cos_[3][4];
};

// two 16-bit lanes per register: arithmetic on these compiles to the packed v_pk_* instructions
// byte offset of row y inside a plane slot.  Row indices and pitches are < 2^23 and a slot is < 2^31 bytes, so the 24-bit
// multiply (full rate; a 64-bit (size_t) y * pitch compiles to the quarter-rate v_mad_i64_i32) is exact.
__device__ __forceinline__ int rowoff(int y, int pitch) { return __mul24(y, pitch); }

typedef short v2s __attribute__((vector_size(4)));
__device__ __forceinline__ v2s pk_from(unsigned u) { v2s r; __builtin_memcpy(&r, &u, 4); return r; }
__device__ __forceinline__ unsigned pk_bits(v2s v) { unsigned u; __builtin_memcpy(&u, &v, 4); return u; }
__device__ __forceinline__ v2s pk_abs(v2s a) { const v2s n = -a; return a > n ? a : n; }
typedef unsigned short v2u __attribute__((vector_size(4)));
__device__ __forceinline__ v2u pku_from(unsigned u) { v2u r; __builtin_memcpy(&r, &u, 4); return r; }
__device__ __forceinline__ unsigned pku_bits(v2u v) { unsigned u; __builtin_memcpy(&u, &v, 4); return u; }
// per-half mask (0xffff / 0) of a > b for packed values below 32768 (signed 16-bit subtract, sign spread)
__device__ __forceinline__ unsigned pk_gt(unsigned a, unsigned b) { return pk_bits((pk_from(b) - pk_from(a)) >> 15); }
// bitwise select: m ? a : b
__device__ __forceinline__ unsigned bsel(unsigned m, unsigned a, unsigned b) { return ((a ^ b) & m) ^ b; }

__device__ __host__ inline int imin(int a, int b) { return a < b ? a : b; }
__device__ __host__ inline int imax(int a, int b) { return a > b ? a : b; }
__device__ __host__ inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __host__ inline int iabs_(int v) { return v < 0 ? -v : v; }

// BORDER_REFLECT_101 for any p (gfedcb|abcdefgh|gfedcba)
__device__ __host__ inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
    return p;
}

}  // namespace i2s
