// Shared host/device helpers for the gfx950 self-play library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/azamd.h"

namespace az {

// ---- error plumbing: every C entry point returns int and records a message --
void set_error(const std::string& msg);

#define AZ_FAIL(code, ...)                                   \
    do {                                                     \
        char _b[512];                                        \
        snprintf(_b, sizeof _b, __VA_ARGS__);                \
        az::set_error(_b);                                   \
        return (code);                                       \
    } while (0)

#define AZ_HIP(expr)                                                          \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess)                                                 \
            AZ_FAIL(AZ_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                    hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

#define AZ_CHECK_ARG(cond, ...)                              \
    do {                                                     \
        if (!(cond)) AZ_FAIL(AZ_ERR_ARG, __VA_ARGS__);       \
    } while (0)

// ---- board geometry (bit = row*8+col; side to move always moves to higher rows
// in canonical form) -----------------------------------------------------------
constexpr uint64_t FILE_A = 0x0101010101010101ull;  // col 0
constexpr uint64_t FILE_H = 0x8080808080808080ull;  // col 7
constexpr uint64_t ROW_0 = 0x00000000000000FFull;
constexpr uint64_t ROW_7 = 0xFF00000000000000ull;

constexpr int NUM_ACTIONS = 192;
constexpr int MAX_CHILDREN = 48;

struct Legal3 {
    uint64_t fwd, dl, dr;  // source squares (canonical) with a legal forward / diag-left / diag-right move
};

// game.py:109-150 in canonical bitboard form: forward needs an empty target,
// a diagonal needs a target that is not one of the mover's own pieces.
__host__ __device__ inline Legal3 legal_sources(uint64_t mine, uint64_t theirs) {
    const uint64_t empty = ~(mine | theirs);
    Legal3 l;
    l.fwd = mine & (empty >> 8);
    l.dl = mine & ~FILE_A & (~mine >> 7);
    l.dr = mine & ~FILE_H & (~mine >> 9);
    return l;
}

__device__ inline uint64_t bitrev64(uint64_t x) { return __brevll(x); }

}  // namespace az
