// Batch move generation / terminal detection / 3-plane encoding for Breakthrough
// on gfx950.  HBM-bound integer work: 17 B in, 25 B out per position for the
// packed outputs; 192 B more for the byte mask and 768 B more for float planes.
//
// Work split: the packed outputs are one position per lane (three shifts and
// masks on two u64 words).  The byte mask and the float planes are written by
// a whole wavefront per position group so that every store instruction covers
// contiguous bytes: a wave walks its 64 positions and writes each one's 192
// mask bytes as 48 lanes x 4 B and its 768 plane bytes as 48 lanes x 16 B.
#include "az_common.h"

namespace az {

// Reference semantics: src/game.py:109-150 (legal moves), :175-194 (winner /
// terminal), :280-307 (planes), :309-315 (mask).
template <bool MASK, bool PLANES>
__global__ __launch_bounds__(256) void movegen_kernel(
    int64_t n, const uint64_t* __restrict__ white, const uint64_t* __restrict__ black,
    const int8_t* __restrict__ turn, uint64_t* __restrict__ legal3, uint8_t* __restrict__ mask192,
    float* __restrict__ planes, int8_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
        const int64_t i = base + lane;
        uint64_t mine = 0, theirs = 0;
        Legal3 l = {0, 0, 0};
        if (i < n) {
            const uint64_t w = white[i], b = black[i];
            const int t = turn[i];
            // canonical form: Black sees the board rotated by 180 degrees
            mine = (t > 0) ? w : bitrev64(b);
            theirs = (t > 0) ? b : bitrev64(w);
            l = legal_sources(mine, theirs);
            if (legal3) {
                legal3[3 * i + 0] = l.fwd;
                legal3[3 * i + 1] = l.dl;
                legal3[3 * i + 2] = l.dr;
            }
            if (status) {
                int s = 0;
                if (w & ROW_7) s = 1;
                else if (b & ROW_0) s = -1;
                else if (b == 0) s = 1;
                else if (w == 0) s = -1;
                else if ((l.fwd | l.dl | l.dr) == 0) s = 2;
                status[i] = (int8_t)s;
            }
        }
        if (MASK || PLANES) {
            const int cnt = (n - base < 64) ? (int)(n - base) : 64;
            for (int j = 0; j < cnt; ++j) {
                // position j of this group, broadcast from lane j (all lanes take part)
                const uint64_t m = __shfl(mine, j), th = __shfl(theirs, j);
                if (MASK) {
                    const uint64_t f = __shfl(l.fwd, j), dl = __shfl(l.dl, j), dr = __shfl(l.dr, j);
                    if (lane < 48) {
                        // bytes 4*lane .. 4*lane+3 of the 192-byte row: action a = sq*3+dir
                        uint32_t word = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int a = 4 * lane + q;
                            const int sq = a / 3, d = a - 3 * sq;
                            const uint64_t src = (d == 0) ? f : ((d == 1) ? dl : dr);
                            word |= (uint32_t)((src >> sq) & 1ull) << (8 * q);
                        }
                        reinterpret_cast<uint32_t*>(mask192 + 192 * (base + j))[lane] = word;
                    }
                }
                if (PLANES && lane < 48) {
                    // floats 4*lane .. 4*lane+3 of the 192-float image [3][8][8]
                    const int p = lane >> 4;              // plane
                    const int s0 = (4 * lane) & 63;       // first square
                    const uint64_t src = (p == 0) ? m : ((p == 1) ? th : ~0ull);
                    float4 v;
                    v.x = (float)((src >> (s0 + 0)) & 1ull);
                    v.y = (float)((src >> (s0 + 1)) & 1ull);
                    v.z = (float)((src >> (s0 + 2)) & 1ull);
                    v.w = (float)((src >> (s0 + 3)) & 1ull);
                    reinterpret_cast<float4*>(planes + 192 * (base + j))[lane] = v;
                }
            }
        }
    }
}

}  // namespace az

extern "C" int az_movegen_encode(int64_t n, const uint64_t* white_dev, const uint64_t* black_dev,
                                 const int8_t* turn_dev, uint64_t* legal3_dev,
                                 uint8_t* mask192_dev, float* planes_dev, int8_t* status_dev,
                                 void* stream) {
    AZ_CHECK_ARG(n >= 0, "az_movegen_encode: n = %lld < 0", (long long)n);
    if (n == 0) return AZ_OK;
    AZ_CHECK_ARG(white_dev && black_dev && turn_dev, "az_movegen_encode: null input pointer");
    AZ_CHECK_ARG(((uintptr_t)planes_dev & 15) == 0 && ((uintptr_t)mask192_dev & 3) == 0,
                 "az_movegen_encode: planes must be 16-byte and mask 4-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t groups = (n + 63) / 64;  // one wave per 64 positions
    int64_t blocks = (groups + 3) / 4;
    const bool heavy = mask192_dev || planes_dev;
    const int64_t cap = heavy ? 256 * 32 : 256 * 8;
    if (blocks > cap) blocks = cap;
    dim3 grid((unsigned)blocks), block(256);
    if (mask192_dev && planes_dev)
        hipLaunchKernelGGL((az::movegen_kernel<true, true>), grid, block, 0, s, n, white_dev,
                           black_dev, turn_dev, legal3_dev, mask192_dev, planes_dev, status_dev);
    else if (mask192_dev)
        hipLaunchKernelGGL((az::movegen_kernel<true, false>), grid, block, 0, s, n, white_dev,
                           black_dev, turn_dev, legal3_dev, mask192_dev, planes_dev, status_dev);
    else if (planes_dev)
        hipLaunchKernelGGL((az::movegen_kernel<false, true>), grid, block, 0, s, n, white_dev,
                           black_dev, turn_dev, legal3_dev, mask192_dev, planes_dev, status_dev);
    else
        hipLaunchKernelGGL((az::movegen_kernel<false, false>), grid, block, 0, s, n, white_dev,
                           black_dev, turn_dev, legal3_dev, mask192_dev, planes_dev, status_dev);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}
