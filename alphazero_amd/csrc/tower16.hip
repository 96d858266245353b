// k_tower16: the whole-position-per-wave tower (see k_tower in net.hip) on the other bf16 MFMA
// shape, v_mfma_f32_16x16x32_bf16, at the same output tile per wave (128 channels x 64 pixels).
// Reference semantics: AlphaZeroNet.forward (src/model.py:144-180), SEResidualBlock (:75-92),
// SEBlock (:37-53).
//
// What differs from k_tower:
//   * a wave's tile is 8 channel tiles x 4 pixel tiles of 16 x 16, a k-step is 32 input channels
//     (4 k-steps per tap, 32 MFMAs and 12 fragment reads each);
//   * an accumulator lane holds 4 channels of ONE pixel, so the squeeze is an all-reduce of 32
//     values over the 16 lanes of a DPP row (four v_add_f32 row_ror each), the gate / shift / bias
//     vectors a lane needs are 32 channels (8 reads of 16 B) and the epilogue stores 16 bytes
//     per instruction;
//   * activations in LDS: 256 bytes per pixel with no padding.  The 16-byte chunk of channels
//     [32 s + 8 q, +8) of pixel (y, x) sits at chunk position ((q & 1) * 8 + (q >> 1) * 4 + s) ^ x
//     of the pixel's row.  A B-fragment read (lane = pixel c of the tile, quarter q) then has,
//     inside each of the four lane groups ds_read_b128 is served in, eight lanes of quarter q
//     with eight different x (all low-three-bit values) and eight lanes of quarter q ^ 1 (the
//     other half of the row): 16 different chunk positions, every bank once, for every tap;
//     the epilogue's 16-byte stores (8 lanes a group, 8 different x) likewise.  k-step s of a
//     fragment is the address of k-step 0 XOR 16 s;
//   * output channels are numbered so that the two channel tiles 2t, 2t + 1 hold, per lane, the
//     8 consecutive channels of one chunk: tile nt, row 4 q + i  <->  channel
//     (nt >> 1) * 32 + q * 8 + (nt & 1) * 4 + i.  The weight images carry that order
//     (pack_conv3x3_16 in net.hip);
//   * the block loop runs the two convs of a block per pass, so a conv's parity is a literal (weight
//     buffer and bias row are immediates, the registers of the two epilogue kinds never meet in a
//     phi); the register form of the accumulators is chosen in build.sh.
#include "tower_common.h"

namespace az {

constexpr int A16_ROW = 256;
constexpr int A16_REGION = 64 * A16_ROW;            // 16384 B per position
// the weight buffers come first: every fragment read of a tap is then lane * 16 plus an immediate
// offset (16 bits), no address arithmetic in the loop
constexpr int L16_W = 0;                            // two 32 KB weight buffers, [ks 4][nt 8][lane 64][16 B]
constexpr int L16_ACT = 2 * TAP_BYTES;              // 65536: four activation regions
constexpr int L16_ZERO = L16_ACT + 4 * A16_REGION;  // 131072: 256 zero bytes for out-of-board taps
constexpr int L16_BIAS = L16_ZERO + 1024;           // 132096
constexpr int L16_SE = L16_BIAS + 1024;             // 133120
constexpr int L16_TOTAL = L16_SE + SE_BYTES;        // 159744

// one 1 KB piece of an LDS-DMA copy whose global and LDS bases are both passed 4 KB in (the
// immediate offset, -4096 ... 3072, is added to both).  M0 is not restored: the compiler keeps
// nothing in it (a reserved register, set before each of its own uses).
#ifndef AZ_T16_DIAG
#define AZ_T16_DIAG 0
#endif
template <int OFF, bool SET_M0 = false>
__device__ __forceinline__ void glds16_piece(const void* gsrc_plus_4k, uint32_t lds_dst_plus_4k) {
    if (AZ_T16_DIAG >= 2) return;
    if (AZ_T16_DIAG == 1 && __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) != 0) return;
    // The eight pieces of a tap share one M0 (the immediate offset moves both addresses): only the
    // first piece writes it.  Nothing else in this kernel writes M0 except the bias / SE copies, which
    // put the old value back; volatile asm statements keep their order.
    if (SET_M0)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2"
                     :
                     : "v"(gsrc_plus_4k), "s"(lds_dst_plus_4k), "n"(OFF)
                     : "memory");
    else
        asm volatile("global_load_lds_dwordx4 %0, off offset:%1" : : "v"(gsrc_plus_4k), "n"(OFF) : "memory");
}

struct Frags16 {
    bf16x8 x[4];  // activation fragments of the four pixel tiles
    bf16x8 w[8];  // weight fragments of the eight channel tiles
};

// k16[s] holds 16 * s in a register the compiler cannot see through: the XORed addresses are
// then not loop invariants it could hoist (and spill)
__device__ __forceinline__ void load_frags16(Frags16& f, const char* lds, const uint32_t (&a)[4], uint32_t wb,
                                             int ks, const uint32_t (&k16)[4]) {
    if (AZ_T16_DIAG >= 4 && ks != 0) { asm volatile("" : "+v"(f.x[0]), "+v"(f.x[1]), "+v"(f.x[2]), "+v"(f.x[3])); return; }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) f.x[pt] = lds_read_frag(lds, ks ? (a[pt] ^ k16[ks]) : a[pt]);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) f.w[nt] = lds_read_frag(lds, wb + (ks * 8 + nt) * 1024);
}

template <int NT0, int NT1>
__device__ __forceinline__ void mma16(f32x4 (&acc)[8][4], const Frags16& f) {
#pragma unroll
    for (int nt = NT0; nt < NT1; ++nt)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
            acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[nt], f.x[pt], acc[nt][pt], 0, 0, 0);
}

// one pipeline step: 32 MFMAs with the 12 fragment reads of the next step spread between them
__device__ __forceinline__ void interleave16() {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // two MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// acc (+ bias) -> optional ReLU -> bf16 -> the wave's LDS region (16 bytes per store: the 8
// channels of tiles 2t and 2t + 1), optionally kept packed in registers as the next residual.
// wa[t]: byte address of this lane's chunk t for pixel tile 0; pixel tile pt is 4096 B further.
template <bool RELU, bool KEEP, bool BIAS>
__device__ __forceinline__ void store_act16(char* lds, const uint32_t (&wa)[4], const f32x4 (&acc)[8][4],
                                            const float* bias, uint32_t (&keep)[8][4][2], int q) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        if (BIAS) {
            b0 = *reinterpret_cast<const f32x4*>(bias + t * 32 + q * 8);
            b1 = *reinterpret_cast<const f32x4*>(bias + t * 32 + q * 8 + 4);
        }
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            f32x4 u = acc[2 * t][pt], v = acc[2 * t + 1][pt];
            if (BIAS) { u += b0; v += b1; }      // (x + 0 is not x for x = -0: the compiler would keep the adds)
            uint4 o = make_uint4(pack_bf16x2(u[0], u[1]), pack_bf16x2(u[2], u[3]), pack_bf16x2(v[0], v[1]),
                                 pack_bf16x2(v[2], v[3]));
            if (RELU) { o.x = relu_bf16x2(o.x); o.y = relu_bf16x2(o.y); o.z = relu_bf16x2(o.z); o.w = relu_bf16x2(o.w); }
            if (KEEP) { keep[2 * t][pt][0] = o.x; keep[2 * t][pt][1] = o.y; keep[2 * t + 1][pt][0] = o.z; keep[2 * t + 1][pt][1] = o.w; }
            *reinterpret_cast<uint4*>(lds + wa[t] + pt * 4096) = o;
        }
    }
}

__global__ __launch_bounds__(256, 1) void k_tower16(TowerParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[L16_TOTAL];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, q = lane >> 4;   // pixel of the tile / k quarter (operands), channel quad (accumulators)
    const int pos = blockIdx.x * 4 + wave;
    bool live = pos < p.B;
    int assigned = -1;
    const int nconv = 2 * p.blocks + 1;
    const int total_taps = nconv * 9;

    if (tid < 64) reinterpret_cast<uint32_t*>(lds + L16_ZERO)[tid] = 0;

    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void*)lds;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto prefetch = [&](int T) {
        const uint8_t* src = p.conv_w + (size_t)T * TAP_BYTES + wave_u * 8192 + lane * 16;
        const uint32_t dst = lds_base + L16_W + (T & 1) * TAP_BYTES + wave_u * 8192;
        glds16x8(src + 4096, __builtin_amdgcn_readfirstlane(dst + 4096));
    };
    // (no branches round these copies: they sit in the tap loop, which is one basic block without them.
    // Waves 2, 3 copy the halves of the bias row that waves 0, 1 copy; a piece index past the image
    // copies the last piece again.)
    auto prefetch_bias = [&](int cv) {
        const int half = wave_u & 1;
        glds4(reinterpret_cast<const uint8_t*>(p.conv_b + (size_t)cv * CH) + half * 256 + lane * 4,
              __builtin_amdgcn_readfirstlane(lds_base + L16_BIAS + (cv & 1) * 512 + half * 256));
    };
    auto prefetch_se = [&](int blk, int piece) {
        const int pc = piece < SE_PIECES ? piece : SE_PIECES - 1;
        glds16(p.se_img + (size_t)blk * SE_BYTES + pc * 1024 + lane * 16,
               __builtin_amdgcn_readfirstlane(lds_base + L16_SE + pc * 1024));
    };
    prefetch(0);
    prefetch_bias(0);
    if (p.need) {
        bool has_rows;
        assigned = assign_rows<256, 4>(p, tid, wave, has_rows);
        if (!has_rows) { wait_weights(); return; }   // no copy into LDS may outlive the workgroup
        live = assigned >= 0;
    }

    // chunk position of this lane's quarter at x = 0, k-step 0
    const int qpos = (q & 1) * 8 + (q >> 1) * 4;
    // per-tap fragment addresses (bytes from the LDS base, k-step 0) of the four pixel tiles.  An
    // out-of-board pixel reads the zero area at the chunk position it would have had in its row.
    uint32_t aaddr[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int y = pt * 2 + (c >> 3) + dy, x = (c & 7) + dx;
            const bool ok = (y >= 0) && (y < 8) && (x >= 0) && (x < 8);
            const uint32_t chunk = (uint32_t)(qpos ^ (x & 7)) * 16;
            aaddr[tap][pt] = (ok ? (uint32_t)(L16_ACT + wave * A16_REGION + (y * 8 + x) * A16_ROW) : (uint32_t)L16_ZERO) + chunk;
        }
    }

    f32x4 acc[8][4];
    uint32_t resid[8][4][2];
    uint32_t k16[4];

    // ---------------- stem: 3x3 conv over the 3 input planes, K = 27 padded to 32 (one k-step)
    {
        const int src_pos = p.need ? assigned : pos;
        const uint64_t mine = live ? p.mine[src_pos] : 0ull, theirs = live ? p.theirs[src_pos] : 0ull;
        bf16x8 xb[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int px = pt * 16 + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = q * 8 + j;  // k = tap*3 + plane
                const int tap = k / 3, pl = k - tap * 3;
                const int y = (px >> 3) + tap / 3 - 1, x = (px & 7) + tap % 3 - 1;
                const bool ok = (k < 27) && (y >= 0) && (y < 8) && (x >= 0) && (x < 8);
                const int sq = (y * 8 + x) & 63;
                const uint64_t src = (pl == 0) ? mine : ((pl == 1) ? theirs : ~0ull);
                xb[pt][j] = (__bf16)((ok && ((src >> sq) & 1ull)) ? 1.0f : 0.0f);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const bf16x8 w = *reinterpret_cast<const bf16x8*>(p.stem_w + (nt * 64 + lane) * 16);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xb[pt], z, 0, 0, 0);
            }
        }
        uint32_t wa[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            wa[t] = (uint32_t)(L16_ACT + wave * A16_REGION + c * A16_ROW + ((qpos ^ (c & 7) ^ t) * 16));
        store_act16<true, true, true>(lds, wa, acc, p.stem_b, resid, q);
    }
    wait_weights();
    __syncthreads();  // tap 0 landed, zero area visible

    unsigned long long t_loop = 0, t_ep1 = 0, t_ep2 = 0, t_sq = 0, t_f1 = 0, t_f2 = 0, t_fin = 0;
    const unsigned long long rt0 = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // One conv's bias fill and tap loop.  `par` is the parity of conv as a literal at every call (the
    // block loop below runs two convs per pass), so the weight buffer of a tap and the bias row are
    // immediates, and the residual / address registers of the two epilogue kinds never meet in a phi.
    uint32_t oz;
    auto run_conv = [&](const int conv, const int par) __attribute__((always_inline)) {
        // opaque per-pass constants (see load_frags16) and an opaque zero for the epilogue addresses
        asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
        asm volatile("v_mov_b32 %0, 16" : "=v"(k16[1]));
        asm volatile("v_mov_b32 %0, 32" : "=v"(k16[2]));
        asm volatile("v_mov_b32 %0, 48" : "=v"(k16[3]));
        k16[0] = 0;
        // the accumulators start from the folded-BN bias of this conv (in LDS since the previous conv)
        {
            const float* b0 = reinterpret_cast<const float*>(lds + L16_BIAS + par * 512) + q * 8;
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(b0 + (nt >> 1) * 32 + (nt & 1) * 4);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[nt][pt] = b;
            }
        }
        // Software pipeline: the 12 fragments of k-step s + 1 are read from LDS while the 32 MFMAs
        // of k-step s run.  The workgroup barrier of a tap sits before its last k-step (whose
        // fragments are in registers by then), so the first fragments of the next tap are read
        // behind it, under that k-step's MFMAs.
        Frags16 f0, f1;
        load_frags16(f0, lds, aaddr[0], L16_W + par * TAP_BYTES + lane * 16, 0, k16);
        if (AZ_T16_DIAG >= 4) f1 = f0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int T = conv * 9 + tap;
            const int tpar = par ^ (tap & 1);   // T & 1
            const uint32_t wb = L16_W + tpar * TAP_BYTES + lane * 16;
            const uint32_t wb_next = L16_W + (tpar ^ 1) * TAP_BYTES + lane * 16;
            // k-step 0: also carries the copy of the next tap (its buffer was released by the barrier
            // inside the previous tap) and the bias / SE pieces, between groups of MFMAs
            // Eight groups of four MFMAs; behind each of them one 1 KB piece of the next tap's copy, so
            // that an MFMA runs under every copy instruction's issue, and the 12 fragment reads of
            // k-step 1.  Past the network's last tap the last tap is copied again (never read): the
            // loop then has no branch around the copies.
            {
                const int Tn = (T + 1 < total_taps) ? T + 1 : total_taps - 1;
                const uint8_t* src = p.conv_w + (size_t)Tn * TAP_BYTES + wave_u * 8192 + lane * 16 + 4096;
                const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + L16_W + (tpar ^ 1) * TAP_BYTES + wave_u * 8192 + 4096);
                auto group = [](int reads) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    if (reads > 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    if (reads > 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                if (AZ_T16_DIAG < 4) f1.x[0] = lds_read_frag(lds, aaddr[tap][0] ^ k16[1]);
                if (AZ_T16_DIAG < 4) f1.x[1] = lds_read_frag(lds, aaddr[tap][1] ^ k16[1]);
                mma16<0, 1>(acc, f0); group(2);
                glds16_piece<-4096, true>(src, dst);
                if (AZ_T16_DIAG < 4) f1.x[2] = lds_read_frag(lds, aaddr[tap][2] ^ k16[1]);
                if (AZ_T16_DIAG < 4) f1.x[3] = lds_read_frag(lds, aaddr[tap][3] ^ k16[1]);
                mma16<1, 2>(acc, f0); group(2);
                glds16_piece<-3072>(src, dst);
                if (AZ_T16_DIAG < 4) f1.w[0] = lds_read_frag(lds, wb + 8 * 1024);
                if (AZ_T16_DIAG < 4) f1.w[1] = lds_read_frag(lds, wb + 9 * 1024);
                mma16<2, 3>(acc, f0); group(2);
                glds16_piece<-2048>(src, dst);
                if (AZ_T16_DIAG < 4) f1.w[2] = lds_read_frag(lds, wb + 10 * 1024);
                if (AZ_T16_DIAG < 4) f1.w[3] = lds_read_frag(lds, wb + 11 * 1024);
                mma16<3, 4>(acc, f0); group(2);
                glds16_piece<-1024>(src, dst);
                if (AZ_T16_DIAG < 4) f1.w[4] = lds_read_frag(lds, wb + 12 * 1024);
                if (AZ_T16_DIAG < 4) f1.w[5] = lds_read_frag(lds, wb + 13 * 1024);
                mma16<4, 5>(acc, f0); group(2);
                glds16_piece<0>(src, dst);
                if (AZ_T16_DIAG < 4) f1.w[6] = lds_read_frag(lds, wb + 14 * 1024);
                if (AZ_T16_DIAG < 4) f1.w[7] = lds_read_frag(lds, wb + 15 * 1024);
                mma16<5, 6>(acc, f0); group(2);
                glds16_piece<1024>(src, dst);
                mma16<6, 7>(acc, f0); group(0);
                glds16_piece<2048>(src, dst);
                if (tap == 0) prefetch_bias(conv + 1);   // the bias array has a spare row past the last conv
                mma16<7, 8>(acc, f0); group(0);
                glds16_piece<3072>(src, dst);
                if (par && tap < 7) prefetch_se(conv >> 1, tap * 4 + wave_u);   // an odd conv is the second of a block
            }
            // k-step 1
            load_frags16(f0, lds, aaddr[tap], wb, 2, k16);
            mma16<0, 8>(acc, f1);
            interleave16();
            // k-step 2
            load_frags16(f1, lds, aaddr[tap], wb, 3, k16);
            mma16<0, 8>(acc, f0);
            interleave16();
            // k-step 3, behind the barrier
            if (AZ_T16_DIAG < 3) {
            wait_weights();
            __syncthreads();  // next tap landed; everyone is done reading this buffer
            }
            if (tap + 1 < 9) load_frags16(f0, lds, aaddr[tap + 1 < 9 ? tap + 1 : 0], wb_next, 0, k16);
            mma16<0, 8>(acc, f1);
            interleave16();
        }
    };
    // this lane's store addresses after a conv: pixel c of tile 0, chunk t of quarter q
    auto store_addresses = [&](uint32_t (&wa)[4]) __attribute__((always_inline)) {
        const uint32_t w0 = oz + (uint32_t)(L16_ACT + wave * A16_REGION + c * A16_ROW + ((qpos ^ (c & 7)) * 16));
#pragma unroll
        for (int t = 0; t < 4; ++t) wa[t] = w0 ^ k16[t];
    };
    auto stamp = [&]() __attribute__((always_inline)) { return p.stamps ? __builtin_amdgcn_s_memtime() : 0ull; };

    for (int blk = 0; blk < p.blocks; ++blk) {
        // ---- conv1 of the block: ReLU -> LDS
        {
            const unsigned long long ts0 = stamp();
            run_conv(2 * blk, 0);
            const unsigned long long ts1 = stamp();
            uint32_t wa[4];
            store_addresses(wa);
            store_act16<true, false, false>(lds, wa, acc, nullptr, resid, q);
            if (p.stamps) { const unsigned long long ts2 = stamp(); t_loop += ts1 - ts0; t_ep1 += ts2 - ts1; }
        }
        // ---- conv2 of the block: SE, + residual, ReLU -> LDS and registers
        const int conv = 2 * blk + 1;
        const unsigned long long ts0 = stamp();
        run_conv(conv, 1);
        const unsigned long long ts1 = stamp();
        uint32_t wa[4];
        store_addresses(wa);
        char* region = lds + oz + L16_ACT + wave * A16_REGION;
        {
            float* scratch = reinterpret_cast<float*>(region);  // the region is free now
            const float* se = reinterpret_cast<const float*>(lds + oz + L16_SE);
            if (p.dbg & 1) {   // diagnostic ablation: no squeeze-excitation
                store_act16<true, true, false>(lds, wa, acc, nullptr, resid, q);
                continue;
            }
            // Squeeze: per-channel mean of z (= acc, bias included) over the 64 pixels.  A lane sums
            // its four pixel tiles; the per-lane sums are then all-reduced over the 16 lanes of the
            // DPP row that holds the same channels: four rotate-and-add steps (row_ror 8, 4, 2, 1),
            // each ONE v_add_f32 with a DPP operand.  Two tiles (8 channels, 8 independent chains) at
            // a time.
            // (volatile asm keeps the step-major order inside a group: left to the compiler, one
            // chain's four dependent steps run back to back and the squeeze is bound by their
            // latency.  Seven independent adds lie between a write and the DPP read of it; the
            // s_nop covers the first step's operands.)
#define AZ_DPP_STEP(ROR)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                          \
        asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:" #ROR " row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
            // fc1 (128 -> 16) rides on the all-reduce: after it every lane of a row holds the row's 32
            // channel sums, so lane (row q, column c) takes hidden unit c over those 32 channels (its
            // weights come from LDS before the sums exist), and the four rows are added with the two
            // lane-swap instructions: no LDS round trip for the means, none for the partial sums.
            f32x4 w1v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                w1v[i] = *reinterpret_cast<const f32x4*>(se + c * SE_W1_PITCH + (i >> 1) * 32 + q * 8 + (i & 1) * 4);
            // fc2's weights too (this lane's gate and shift rows; row 16 of the image is the bias b2)
            const float* w2c = se + SE_OFF_W2 + lane;
            float wv[4][17];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int jj = 0; jj < 17; ++jj) {
                    wv[m][jj] = w2c[jj * 256 + 64 * m];
                    wv[2 + m][jj] = w2c[jj * 256 + CH + 64 * m];
                }
            __builtin_amdgcn_sched_barrier(0);
            float y = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int nt = g * 2 + h;
                    const f32x4 s4 = (acc[nt][0] + acc[nt][1]) + (acc[nt][2] + acc[nt][3]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[h * 4 + i] = s4[i];
                }
                asm volatile("s_nop 1");
                AZ_DPP_STEP(8)
                AZ_DPP_STEP(4)
                AZ_DPP_STEP(2)
                AZ_DPP_STEP(1)
#pragma unroll
                for (int i = 0; i < 8; ++i) y += v[i] * w1v[g * 2 + (i >> 2)][i & 3];   // channel g * 32 + q * 8 + i
            }
#undef AZ_DPP_STEP
            const unsigned long long tsa = p.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
            float hid[16];
            {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                // rows 0 + 1 and 2 + 3 (odd rows of one copy swapped with even rows of the other), then the two halves
                const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(y), __float_as_uint(y), false, false);
                y = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                const u32x2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
                y = __uint_as_float(t[0]) + __uint_as_float(t[1]);
                y = fmaxf(y * (1.0f / 64.0f) + se[SE_OFF_B1 + c], 0.f);   // hidden unit c, in every row
#pragma unroll
                for (int j = 0; j < 16; ++j) hid[j] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y), j));
            }
            const unsigned long long tsb = p.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
            // fc2 (16 -> 256) from the transposed weights: this lane computes the gate and the
            // shift of channels lane and lane + 64 (the hidden units are wave-uniform: scalar operands);
            // all 68 LDS reads are issued before the first multiply so that their latency is paid once
            {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float g = wv[m][16], sh = wv[2 + m][16];
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) {
                        g += wv[m][jj] * hid[jj];
                        sh += wv[2 + m][jj] * hid[jj];
                    }
                    g = __builtin_amdgcn_rcpf(1.0f + __expf(-g));  // sigmoid
                    scratch[160 + lane + 64 * m] = g;
                    scratch[160 + CH + lane + 64 * m] = sh;
                }
            }
            wave_lds_sync();
            const unsigned long long tsc = p.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
            // out = relu(acc * gate + shift + residual)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const int ch0 = (nt >> 1) * 32 + q * 8 + (nt & 1) * 4;
                const f32x4 gt = *reinterpret_cast<const f32x4*>(scratch + 160 + ch0);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(scratch + 160 + CH + ch0);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    const uint32_t rlo = resid[nt][pt][0], rhi = resid[nt][pt][1];
                    acc[nt][pt][0] = acc[nt][pt][0] * gt[0] + sh[0] + bf16_lo(rlo);
                    acc[nt][pt][1] = acc[nt][pt][1] * gt[1] + sh[1] + bf16_hi(rlo);
                    acc[nt][pt][2] = acc[nt][pt][2] * gt[2] + sh[2] + bf16_lo(rhi);
                    acc[nt][pt][3] = acc[nt][pt][3] * gt[3] + sh[3] + bf16_hi(rhi);
                }
            }
            store_act16<true, true, false>(lds, wa, acc, nullptr, resid, q);
            if (p.stamps) {
                const unsigned long long tsd = __builtin_amdgcn_s_memtime();
                t_sq += tsa - ts1; t_f1 += tsb - tsa; t_f2 += tsc - tsb; t_fin += tsd - tsc;
            }
            if (conv == nconv - 2) {
                // ---- value head conv 1x1 (128 -> 32) on the tower output, + bias, ReLU -> v_act
                f32x4 va[2][4];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt) va[nt][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    bf16x8 x[4];
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt) x[pt] = lds_read_frag(lds, aaddr[4][pt] ^ (uint32_t)(ks * 16));
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const bf16x8 w = *reinterpret_cast<const bf16x8*>(p.value_w + ((ks * 2 + nt) * 64 + lane) * 16);
#pragma unroll
                        for (int pt = 0; pt < 4; ++pt)
                            va[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x[pt], va[nt][pt], 0, 0, 0);
                    }
                }
                if (live) {
                    char* out = reinterpret_cast<char*>(p.v_act) + (size_t)(pos >> 4) * (VALUE_K / 32) * 1024 + (pos & 15) * 64;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int ch0 = nt * 16 + q * 4;
                        const f32x4 b = *reinterpret_cast<const f32x4*>(p.value_b + ch0);
#pragma unroll
                        for (int pt = 0; pt < 4; ++pt) {
                            const float v0 = fmaxf(va[nt][pt][0] + b[0], 0.f);
                            const float v1 = fmaxf(va[nt][pt][1] + b[1], 0.f);
                            const float v2 = fmaxf(va[nt][pt][2] + b[2], 0.f);
                            const float v3 = fmaxf(va[nt][pt][3] + b[3], 0.f);
                            *reinterpret_cast<uint2*>(out + (pt * 16 + c) * 1024 + ch0 * 2) =   // k-step = pixel
                                make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                        }
                    }
                }
            }
        }
        if (p.stamps) { const unsigned long long ts2 = stamp(); t_loop += ts1 - ts0; t_ep2 += ts2 - ts1; }
    }
    // ---- policy conv: ReLU -> LDS -> p_act (coalesced 16 B per lane)
    {
        const unsigned long long ts0 = stamp();
        run_conv(nconv - 1, 0);
        const unsigned long long ts1 = stamp();
        uint32_t wa[4];
        store_addresses(wa);
        char* region = lds + oz + L16_ACT + wave * A16_REGION;
        store_act16<true, false, false>(lds, wa, acc, nullptr, resid, q);
        if (live) {
            char* out = reinterpret_cast<char*>(p.p_act) + (size_t)(pos >> 4) * (POLICY_K / 32) * 1024 + (pos & 15) * 64;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int idx = i * 64 + lane;  // 16-byte granule: pixel idx / 16, channels [8 (idx % 16), +8)
                const int px = idx >> 4, lc = idx & 15;
                const int chunk = (((lc & 1) * 8 + ((lc >> 1) & 1) * 4 + (lc >> 2)) ^ (px & 7));
                const uint4 v = *reinterpret_cast<const uint4*>(region + px * A16_ROW + chunk * 16);
                *reinterpret_cast<uint4*>(out + (idx >> 2) * 1024 + (idx & 3) * 16) = v;
            }
        }
        if (p.stamps) { const unsigned long long ts2 = stamp(); t_loop += ts1 - ts0; t_ep1 += ts2 - ts1; }
    }
    if (p.stamps && tid == 0) {
        unsigned long long* o = p.stamps + blockIdx.x * 10;
        o[0] = t_loop; o[1] = t_ep1; o[2] = t_ep2; o[3] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz ticks over the conv sequence
        o[4] = t_sq; o[5] = t_f1; o[6] = t_f2; o[7] = t_fin;
        o[8] = rt0; o[9] = __builtin_amdgcn_s_memrealtime();
    }
}

void launch_tower16(const TowerParams& tp, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_tower16, dim3((B + 3) / 4), dim3(256), 0, s, tp);
}

}  // namespace az
