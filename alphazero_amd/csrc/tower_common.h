// Device helpers shared by the forms of the tower kernel (net.hip, tower16.hip): the launch
// parameters, the LDS map, bf16 packing, DPP lane exchanges, the LDS-DMA copies of the weight
// stream and the row assignment of the fused search.
#pragma once
#include "az_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

namespace az {

constexpr int CH = 128;                        // filters
constexpr int ACT_PITCH = 272;                 // bytes per pixel row in LDS (256 + 16 pad)
constexpr int ACT_REGION = 64 * ACT_PITCH;     // 17408 B per position
constexpr int LDS_ZERO = 4 * ACT_REGION;       // 69632: a zeroed pixel row for out-of-board taps
constexpr int LDS_W = LDS_ZERO + 512;          // 70144: two 32 KB weight buffers
constexpr int TAP_BYTES = 8 * 4 * 64 * 16;     // 32768: [ks 8][nt 4][lane 64][16 B]
constexpr int LDS_BIAS = LDS_W + 2 * TAP_BYTES;   // 135680: folded-BN bias of the running conv, 2 x 512 B
constexpr int LDS_SE = LDS_BIAS + 1024;           // 136704: squeeze-excitation weights of the running block
// SE image (float32): W1 [16][132] (row pitch padded for conflict-free b128 reads),
// b1 [16], W2 transposed [16][256], b2 [256]; padded to a whole number of 1 KB DMA pieces
constexpr int SE_W1_PITCH = 132;
constexpr int SE_OFF_B1 = 16 * SE_W1_PITCH;            // floats
constexpr int SE_OFF_W2 = SE_OFF_B1 + 16;
constexpr int SE_OFF_B2 = SE_OFF_W2 + 16 * 256;
constexpr int SE_FLOATS = SE_OFF_B2 + 256;             // 6480 floats = 25920 B
constexpr int SE_PIECES = (SE_FLOATS * 4 + 1023) / 1024;  // 26
constexpr int SE_BYTES = SE_PIECES * 1024;             // 26624
constexpr int LDS_TOTAL = LDS_SE + SE_BYTES;           // 163328 <= 163840
constexpr int CONV_BYTES = 9 * TAP_BYTES;      // 294912
constexpr int VALUE_CH = 32;
constexpr int SE_HID = 16;
constexpr int POLICY_K = 64 * CH;              // 8192
constexpr int VALUE_K = 64 * VALUE_CH;         // 2048

struct TowerParams {
    const uint64_t* mine;       // [B] canonical bitboards
    const uint64_t* theirs;
    // Row assignment on the device (the fused search): when `need` is given, the rows of the batch
    // are the games whose flag is set, in game order; every workgroup finds its own rows by a
    // scan over the flags, writes row_of[game] for them, and leaves at once if it has none.  B is
    // then only the number of rows the grid was sized for (an upper bound).
    const uint8_t* need;        // [n_games] 0/1, readable (zeros) up to the next multiple of 16
    int n_games;
    int32_t* row_of;            // out [n_games]: row of each game that has one
    int32_t* n_rows;            // out: rows in use (written by the last workgroup)
    unsigned long long* rows_total;  // += rows in use (NULL: not counted)
    int B;
    int blocks;
    const uint8_t* stem_w;      // [ks 2][nt 4][lane 64][16 B]
    const float* stem_b;        // [128]
    const uint8_t* conv_w;      // [2*blocks+1][CONV_BYTES]; the last one is the policy conv
    const float* conv_b;        // [2*blocks+1][128]
    const uint8_t* se_img;      // [blocks][SE_BYTES] squeeze-excitation weights in their LDS image layout
    const uint8_t* value_w;     // [ks 8][lane 64][16 B]
    const float* value_b;       // [32]
    // head activations, tiled for the head GEMM's fragment loads: position p, element k (k = px * C + ch)
    // lives at ((p / 16 * (K / 32) + k / 32) * 16 + p % 16) * 32 + k % 32, so the 16 rows x 32 k of one
    // MFMA A-fragment are one contiguous 1 KB block
    __bf16* p_act;              // K = 64 px * 128 ch
    __bf16* v_act;              // K = 64 px * 32 ch
    int dbg;                    // diagnostic ablation flags (AZ_TOWER_DBG), 0 in production
    unsigned long long* stamps; // diagnostic: [workgroup][10] cycle sums {loop, conv1 epilogue, block epilogue, 100 MHz ticks, squeeze, fc1, fc2, final, first tick, last tick}; NULL in production
    // k_tower_any only (the other kernels are built for 128 filters and 16 hidden units): filters and
    // squeeze-excitation hidden units.  Its images: stem_w [nt C/16][lane 64][16 B], conv_w
    // [conv][tap 9][ks C/32][nt C/16][lane 64][16 B] with output channel nt * 16 + (lane & 15) and input
    // channel ks * 32 + 8 * (lane >> 4) + j, value_w [ks C/32][nt 2][lane 64][16 B], se_img float32
    // [block]{W1 [H][C], b1 [H], W2 [2C][H], b2 [2C]}
    int ch, se_hid;
};

// two floats -> packed bf16 (round to nearest even) in ONE v_cvt_pk_bf16_f32: the
// conversion has to be written on the 2-vector, element-wise casts cost two
// conversions plus a v_perm_b32 each
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const f32x2 f = {a, b};
    bf16x2 v = __builtin_convertvector(f, bf16x2);
    return *reinterpret_cast<uint32_t*>(&v);
}
// ReLU on two packed bf16 values: as signed 16-bit integers every negative float is
// negative, so one v_pk_max_i16 against zero clamps both (-0.0 becomes +0.0)
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {
    typedef __attribute__((ext_vector_type(2))) short short2_t;
    short2_t x = *reinterpret_cast<short2_t*>(&v);
    const short2_t z = {0, 0};
    x = __builtin_elementwise_max(x, z);
    return *reinterpret_cast<uint32_t*>(&x);
}
__device__ __forceinline__ float bf16_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t v) { return __uint_as_float(v & 0xFFFF0000u); }

// value of lane (lane ^ D) for D in {1, 2, 4, 8}: DPP lane exchange inside a row of 16
// (row_shr:n = lane i reads lane i-n, row_shl:n = lane i reads lane i+n; lanes whose source
// falls outside the row, or whose bank is masked off, keep `old`)
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_update(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xF, BANK, false));
}
__device__ __forceinline__ float lane_xor(float x, int d) {
    switch (d) {
        case 1: return dpp_update<0xB1, 0xF>(x, x);                                     // quad_perm [1,0,3,2]
        case 2: return dpp_update<0x4E, 0xF>(x, x);                                     // quad_perm [2,3,0,1]
        case 4: return dpp_update<0x114, 0xA>(dpp_update<0x104, 0x5>(x, x), x);          // shl:4 banks 0,2 | shr:4 banks 1,3
        default: return dpp_update<0x118, 0xF>(dpp_update<0x108, 0xF>(x, x), x);         // shl:8 | shr:8
    }
}

// orders this wave's LDS writes before its later LDS reads by other lanes
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bf16x8 lds_read_frag(const char* lds, uint32_t off) {
    return *reinterpret_cast<const bf16x8*>(lds + off);
}

// LDS-DMA copy of 64 x 16 B: global (per-lane address) -> LDS (wave-uniform base +
// lane * 16).  Issued from inline asm so that the compiler's wait-count pass does not
// see a FLAT instruction touching LDS (it would then drain lgkmcnt to 0 before every
// MFMA group); the matching s_waitcnt vmcnt(0) is placed by hand before the barrier
// that publishes the buffer (wait_weights()).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
// same, 64 x 4 B
__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
// 8 KB: eight 64 x 16 B copies, 1 KB apart in both address spaces.  The instruction's
// immediate offset is added to the global and to the LDS address alike, so one M0 value
// and one address register serve all eight; both bases are passed 4 KB in (the
// immediate is signed 13-bit).
__device__ __forceinline__ void glds16x8(const void* gsrc_plus_4k, uint32_t lds_dst_plus_4k) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off offset:-4096\n\t"
        "global_load_lds_dwordx4 %1, off offset:-3072\n\t"
        "global_load_lds_dwordx4 %1, off offset:-2048\n\t"
        "global_load_lds_dwordx4 %1, off offset:-1024\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\t"
        "global_load_lds_dwordx4 %1, off offset:2048\n\t"
        "global_load_lds_dwordx4 %1, off offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc_plus_4k), "s"(lds_dst_plus_4k)
        : "memory");
}
// the two halves of glds16x8 (same bases), for callers that spread the copy over an MFMA stream
__device__ __forceinline__ void glds16x4_lo(const void* gsrc_plus_4k, uint32_t lds_dst_plus_4k) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off offset:-4096\n\t"
        "global_load_lds_dwordx4 %1, off offset:-3072\n\t"
        "global_load_lds_dwordx4 %1, off offset:-2048\n\t"
        "global_load_lds_dwordx4 %1, off offset:-1024\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc_plus_4k), "s"(lds_dst_plus_4k)
        : "memory");
}
__device__ __forceinline__ void glds16x4_hi(const void* gsrc_plus_4k, uint32_t lds_dst_plus_4k) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\t"
        "global_load_lds_dwordx4 %1, off offset:2048\n\t"
        "global_load_lds_dwordx4 %1, off offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc_plus_4k), "s"(lds_dst_plus_4k)
        : "memory");
}
__device__ __forceinline__ void wait_weights() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Row assignment of the fused search (see TowerParams::need).  Returns the game behind local
// position lp of this workgroup (-1: none) and whether the workgroup has any row at all.
// Every workgroup scans the flags from game 0 until it has passed its own rows (the last
// workgroup scans them all and so knows the total); 16 flags per lane and pass.
template <int THREADS, int PPB>
__device__ __forceinline__ int assign_rows(const TowerParams& p, int tid, int lp, bool& has_rows) {
    __shared__ uint32_t s_wave[THREADS / 64];
    __shared__ int s_game[PPB];
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t row0 = blockIdx.x * PPB;
    const bool last = blockIdx.x == gridDim.x - 1;
    if (tid < PPB) s_game[tid] = -1;
    uint32_t carry = 0;
    const int padded = (p.n_games + 15) & ~15;
    for (int base = 0; base < padded; base += THREADS * 16) {
        const int g0 = base + tid * 16;
        uint4 f = make_uint4(0u, 0u, 0u, 0u);
        if (g0 < padded) f = *reinterpret_cast<const uint4*>(p.need + g0);
        const uint32_t c = __popc(f.x & 0x01010101u) + __popc(f.y & 0x01010101u) + __popc(f.z & 0x01010101u) +
                           __popc(f.w & 0x01010101u);
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        __syncthreads();   // s_game is initialised; the previous pass has read s_wave
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = carry, total = 0;
#pragma unroll
        for (int i = 0; i < THREADS / 64; ++i) {
            const uint32_t x = s_wave[i];
            total += x;
            if (i < wave) before += x;
        }
        const uint32_t excl = before + incl - c;
        if (c) {
#pragma unroll
            for (int j = 0; j < PPB; ++j) {
                const uint32_t r = row0 + j;
                if (r >= excl && r < excl + c) {
                    int k = (int)(r - excl), idx = -1;   // the k-th set flag of this lane's 16
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const uint32_t word = (q < 4) ? f.x : ((q < 8) ? f.y : ((q < 12) ? f.z : f.w));
                        if ((word >> ((q & 3) * 8)) & 1u) {
                            if (k == 0 && idx < 0) idx = q;
                            --k;
                        }
                    }
                    s_game[j] = g0 + idx;
                    p.row_of[g0 + idx] = (int32_t)r;
                }
            }
        }
        carry += total;
        if (!last && carry >= row0 + PPB) break;
    }
    if (last && tid == 0) {
        *p.n_rows = (int32_t)carry;
        if (p.rows_total) atomicAdd(p.rows_total, (unsigned long long)carry);
    }
    __syncthreads();
    has_rows = s_game[0] >= 0;
    return s_game[lp];
}

// k_tower16 (tower16.hip); tp's weight pointers must be the 16x16x32 images
void launch_tower16(const TowerParams& tp, int B, hipStream_t s);
// k_tower_any (tower_any.hip); tp's weight pointers must be that kernel's images, tp.ch / tp.se_hid set
void launch_tower_any(const TowerParams& tp, int B, hipStream_t s);

}  // namespace az
