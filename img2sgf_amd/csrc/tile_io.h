// LDS tile staging with dword-wide global loads.  Planes are 4-byte aligned (pitch is a multiple of 64), so a tile
// whose x origin is a multiple of 4 is fetched as whole dwords; only dwords that straddle the image border are
// assembled bytewise with the border rule.
#pragma once
#include "i2s_types.h"

namespace i2s {

enum { BORDER_REPL = 0, BORDER_R101 = 1, BORDER_ONE = 2 };   // BORDER_ONE: bytes outside the image read as 1

template <int MODE>
__device__ __forceinline__ int border_idx(int p, int n)
{
    if (MODE == BORDER_R101) return reflect101(p, n);
    return iclamp(p, 0, n - 1);
}

// dword holding plane bytes (x .. x + 3, y) with border handling; x must be a multiple of 4 (it may be negative)
template <int MODE>
__device__ __forceinline__ unsigned tile_word(const uint8_t* __restrict__ plane, int pitch, int w, int h, int x, int y)
{
    if (MODE == BORDER_ONE) {
        unsigned v1 = 0x01010101u;
        if (y >= 0 && y < h) {
            const uint8_t* row1 = plane + rowoff(y, pitch);
            if (x >= 0 && x + 3 < w) v1 = *reinterpret_cast<const unsigned*>(row1 + x);
            else {
                v1 = 0;
                for (int q = 0; q < 4; q++) v1 |= (unsigned)((x + q >= 0 && x + q < w) ? row1[x + q] : 1) << (8 * q);
            }
        }
        return v1;
    }
    const int gy = border_idx<MODE>(y, h);
    const uint8_t* row = plane + rowoff(gy, pitch);
    if (x >= 0 && x + 3 < w) return *reinterpret_cast<const unsigned*>(row + x);
    return (unsigned)row[border_idx<MODE>(x, w)] | ((unsigned)row[border_idx<MODE>(x + 1, w)] << 8) |
           ((unsigned)row[border_idx<MODE>(x + 2, w)] << 16) | ((unsigned)row[border_idx<MODE>(x + 3, w)] << 24);
}

// dst[r * DSTRIDE + c] = dword holding plane bytes (xa + 4c .. xa + 4c + 3, ya + r) with border handling.
// xa must be a multiple of 4 (it may be negative).  All NT threads of the block call this.
template <int ROWS, int WORDS, int DSTRIDE, int NT, int MODE>
__device__ __forceinline__ void load_tile_words(unsigned* __restrict__ dst, const uint8_t* __restrict__ plane, int pitch,
                                                int w, int h, int xa, int ya, int tid)
{
    for (int i = tid; i < ROWS * WORDS; i += NT) {
        const int r = i / WORDS, c = i - r * WORDS;
        dst[r * DSTRIDE + c] = tile_word<MODE>(plane, pitch, w, h, xa + 4 * c, ya + r);
    }
}

// bytes -1 .. 4 around the dword b (a = previous dword, c = next dword)
__device__ __forceinline__ void unpack6(unsigned a, unsigned b, unsigned c, int* p /* p[0] = byte -1 ... p[5] = byte 4 */)
{
    p[0] = (int)(a >> 24);
    p[1] = (int)(b & 0xffu); p[2] = (int)((b >> 8) & 0xffu); p[3] = (int)((b >> 16) & 0xffu); p[4] = (int)(b >> 24);
    p[5] = (int)(c & 0xffu);
}

}  // namespace i2s
