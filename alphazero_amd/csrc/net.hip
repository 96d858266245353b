// Batched SE-ResNet (N blocks x 128 filters) inference for gfx950, bf16 MFMA.
// Reference semantics: AlphaZeroNet.forward (src/model.py:144-180),
// SEResidualBlock (:75-92), SEBlock (:37-53), and the softmax / value recipe of
// MCTS._batch_evaluate (src/mcts.py:252-256).
//
// One forward = three launches:
//   k_tower       stem + every residual block + policy conv + value conv, fused.
//                 A workgroup of 4 waves owns 4 positions; each wave owns ONE
//                 position (64 pixels x 128 channels) for the whole network.
//                 Activations never leave the CU: the current layer input lives
//                 in the wave's private LDS region ([64 px][128 ch] bf16, 272-byte
//                 pixel pitch so that ds_read_b128 fragment reads are conflict
//                 free), the residual stays in registers, and only the folded
//                 weights stream in (L2 -> LDS by global_load_lds, one 32 KB tap
//                 at a time, double buffered, stored in MFMA fragment order so the
//                 LDS image is lane-linear).
//                 3x3 conv = 9 taps x 8 k-steps of v_mfma_f32_32x32x16_bf16 with
//                 D[channel][pixel] = W[channel][k] * X[k][pixel]; a tap is just a
//                 per-lane LDS address (out-of-board pixels read a zero row).
//                 Below 513 positions k_tower_split runs instead: the channels of a
//                 position are split over 4 or 2 waves so that a small batch still
//                 spreads over the chip (see its comment).
//   k_head_gemm   policy FC (8192->192) and value FC1 (2048->128) as split-K MFMA
//                 GEMMs over the bf16 head activations.
//   k_head_finish bias, ReLU, value FC2, both softmaxes, value = P(win) - P(loss).
//
// BatchNorm (eval mode, eps 1e-5) is folded into the conv weights and a bias at
// load time.  Arithmetic: bf16 operands, float32 accumulation, float32 epilogues.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "az_common.h"
#include "head_dev.h"
#include "net_internal.h"

#include "tower_common.h"


// Cycle stamps inside k_tower_split cost 5 % even when switched off at run time, so they are
// compiled in only on request (AZ_SPLIT_STAMPS=1 bash build.sh); k_tower always carries its own.
#ifndef AZ_SPLIT_STAMPS
#define AZ_SPLIT_STAMPS 0
#endif

namespace az {


struct Frags {
    bf16x8 x[2];  // activation fragments of the two pixel tiles
    bf16x8 w[4];  // weight fragments of the four channel tiles
};

__device__ __forceinline__ void load_frags(Frags& f, const char* lds, uint32_t a0, uint32_t a1,
                                           uint32_t wb, int ks) {
    f.x[0] = lds_read_frag(lds, a0 + ks * 32);
    f.x[1] = lds_read_frag(lds, a1 + ks * 32);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) f.w[nt] = lds_read_frag(lds, wb + (ks * 4 + nt) * 1024);
}

__device__ __forceinline__ void mma_step(f32x16 (&acc)[4][2], const Frags& f) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[nt], f.x[0], acc[nt][0], 0, 0, 0);
        acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[nt], f.x[1], acc[nt][1], 0, 0, 0);
    }
}

// Scheduling recipe for one pipeline step: MFMA, DS read, MFMA, DS read, ... then the
// remaining MFMAs; ends with a full scheduling barrier so steps stay in source order.
__device__ __forceinline__ void interleave_step() {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// acc[nt][pt][v] + bias -> optional ReLU -> bf16 -> the wave's LDS region, and
// (optionally) the packed copy kept in registers as the next residual.
template <bool RELU, bool KEEP, bool BIAS>
__device__ __forceinline__ void store_act(char* region, const f32x16 (&acc)[4][2],
                                          const float* bias, uint32_t (&keep)[4][2][8], int r,
                                          int h) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int ch0 = nt * 32 + gq * 8 + h * 4;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (BIAS) b = *reinterpret_cast<const f32x4*>(bias + ch0);
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                float v0 = acc[nt][pt][gq * 4 + 0] + b[0];
                float v1 = acc[nt][pt][gq * 4 + 1] + b[1];
                float v2 = acc[nt][pt][gq * 4 + 2] + b[2];
                float v3 = acc[nt][pt][gq * 4 + 3] + b[3];
                uint32_t lo = pack_bf16x2(v0, v1), hi = pack_bf16x2(v2, v3);
                if (RELU) { lo = relu_bf16x2(lo); hi = relu_bf16x2(hi); }
                if (KEEP) { keep[nt][pt][gq * 2 + 0] = lo; keep[nt][pt][gq * 2 + 1] = hi; }
                *reinterpret_cast<uint2*>(region + (pt * 32 + r) * ACT_PITCH + ch0 * 2) = make_uint2(lo, hi);
            }
        }
    }
}



__global__ __launch_bounds__(256, 1) void k_tower(TowerParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_TOTAL];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int pos = blockIdx.x * 4 + wave;
    bool live = pos < p.B;
    int assigned = -1;
    char* region0 = lds + wave * ACT_REGION;
    const int nconv = 2 * p.blocks + 1;
    const int total_taps = nconv * 9;

    // zero row for out-of-board taps
    for (int i = tid; i < 512 / 4; i += 256) reinterpret_cast<uint32_t*>(lds + LDS_ZERO)[i] = 0;

    // weight stream: tap T of the conv sequence -> buffer T&1; each wave copies 8 KB
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void*)lds;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto prefetch = [&](int T) {
        const uint8_t* src = p.conv_w + (size_t)T * TAP_BYTES + wave_u * 8192 + lane * 16;
        const uint32_t dst = lds_base + LDS_W + (T & 1) * TAP_BYTES + wave_u * 8192;
        glds16x8(src + 4096, __builtin_amdgcn_readfirstlane(dst + 4096));
    };
    auto prefetch_half = [&](int T, int half) {
        const uint8_t* src = p.conv_w + (size_t)T * TAP_BYTES + wave_u * 8192 + lane * 16;
        const uint32_t dst = lds_base + LDS_W + (T & 1) * TAP_BYTES + wave_u * 8192;
        if (half == 0) glds16x4_lo(src + 4096, __builtin_amdgcn_readfirstlane(dst + 4096));
        else glds16x4_hi(src + 4096, __builtin_amdgcn_readfirstlane(dst + 4096));
    };
    // folded-BN bias of conv c -> LDS bias buffer c&1 (waves 0 and 1, 256 B each)
    auto prefetch_bias = [&](int c) {
        if (wave_u < 2)
            glds4(reinterpret_cast<const uint8_t*>(p.conv_b + (size_t)c * CH) + wave_u * 256 + lane * 4,
                  __builtin_amdgcn_readfirstlane(lds_base + LDS_BIAS + (c & 1) * 512 + wave_u * 256));
    };
    // squeeze-excitation weights of block blk -> LDS, piece `piece` (1 KB) of SE_PIECES
    auto prefetch_se = [&](int blk, int piece) {
        if (piece < SE_PIECES)
            glds16(p.se_img + (size_t)blk * SE_BYTES + piece * 1024 + lane * 16,
                   __builtin_amdgcn_readfirstlane(lds_base + LDS_SE + piece * 1024));
    };
    prefetch(0);
    prefetch_bias(0);
    if (p.need) {
        bool has_rows;
        assigned = assign_rows<256, 4>(p, tid, wave, has_rows);
        if (!has_rows) { wait_weights(); return; }   // no copy into LDS may outlive the workgroup
        live = assigned >= 0;
    }

    // per-tap fragment addresses (bytes from the LDS base) for the two pixel tiles.  An
    // out-of-board pixel reads zeros from the 512-byte zero area, at the offset that falls on the
    // same LDS banks as the pixel's own row would (272 = 16 mod 256): the lanes of a fragment read
    // then still cover distinct banks
    uint32_t aaddr[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int px = pt * 32 + r;
            const int y = (px >> 3) + dy, x = (px & 7) + dx;
            const bool ok = (y >= 0) && (y < 8) && (x >= 0) && (x < 8);
            aaddr[tap][pt] = ok ? (uint32_t)(wave * ACT_REGION + (y * 8 + x) * ACT_PITCH + h * 16)
                                : (uint32_t)(LDS_ZERO + (((y * 8 + x) & 15) * 16 + h * 16));  // the banks the pixel would have had
        }
    }

    f32x16 acc[4][2];
    uint32_t resid[4][2][8];

    // ---------------- stem: 3x3 conv over the 3 input planes, K = 27 padded to 32
    {
        const int src_pos = p.need ? assigned : pos;
        const uint64_t mine = live ? p.mine[src_pos] : 0ull, theirs = live ? p.theirs[src_pos] : 0ull;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[nt][pt][v] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xb[2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int px = pt * 32 + r;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = ks * 16 + h * 8 + j;  // k = tap*3 + plane
                    const int tap = k / 3, pl = k - tap * 3;
                    const int y = (px >> 3) + tap / 3 - 1, x = (px & 7) + tap % 3 - 1;
                    const bool ok = (k < 27) && (y >= 0) && (y < 8) && (x >= 0) && (x < 8);
                    const int sq = (y * 8 + x) & 63;
                    const uint64_t src = (pl == 0) ? mine : ((pl == 1) ? theirs : ~0ull);
                    xb[pt][j] = (__bf16)((ok && ((src >> sq) & 1ull)) ? 1.0f : 0.0f);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const bf16x8 w = *reinterpret_cast<const bf16x8*>(p.stem_w + ((ks * 4 + nt) * 64 + lane) * 16);
                acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xb[0], acc[nt][0], 0, 0, 0);
                acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xb[1], acc[nt][1], 0, 0, 0);
            }
        }
        store_act<true, true, true>(region0, acc, p.stem_b, resid, r, h);
    }
    wait_weights();
    __syncthreads();  // tap 0 landed, zero row visible

    unsigned long long t_loop = 0, t_ep1 = 0, t_ep2 = 0, t_sq = 0, t_f1 = 0, t_f2 = 0, t_fin = 0;
    const unsigned long long rt0 = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    for (int conv = 0; conv < nconv; ++conv) {
        const unsigned long long ts0 = p.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
        // the accumulators start from the folded-BN bias of this conv (in LDS since the
        // previous conv), so no epilogue has to add it
        {
            const float* b0 = reinterpret_cast<const float*>(lds + LDS_BIAS + (conv & 1) * 512) + h * 4;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(b0 + nt * 32 + gq * 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { acc[nt][0][gq * 4 + q] = b[q]; acc[nt][1][gq * 4 + q] = b[q]; }
                }
        }
        // Software pipeline: the fragments of k-step s+1 are read from LDS while the
        // 8 MFMAs of k-step s run.  The workgroup barrier of a tap sits between its
        // k-steps 6 and 7 (k-step 7's fragments are in registers by then), so the
        // first fragments of the next tap are read behind it, under k-step 7's MFMAs.
        Frags f0, f1;
        load_frags(f0, lds, aaddr[0][0], aaddr[0][1], LDS_W + ((conv * 9) & 1) * TAP_BYTES + lane * 16, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int T = conv * 9 + tap;
            const uint32_t wb = LDS_W + (T & 1) * TAP_BYTES + lane * 16;
            const uint32_t wb_next = LDS_W + ((T + 1) & 1) * TAP_BYTES + lane * 16;
#pragma unroll
            for (int ks = 0; ks < 8; ks += 2) {
                // one pipeline step = 6 fragment reads (for the step after) + 8 MFMAs; the reads
                // are spread one per MFMA so that each issues in an MFMA's shadow
                load_frags(f1, lds, aaddr[tap][0], aaddr[tap][1], wb, ks + 1);
                if (ks == 0) {
                    // The first step of a tap also carries the copy of the next tap (its buffer was
                    // released by the barrier inside the previous tap) and the bias / SE pieces, in
                    // three groups between MFMA pairs: this wave is its SIMD's only source of MFMAs,
                    // whatever is issued outside the stream leaves the matrix pipe idle.
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0.w[nt], f0.x[0], acc[nt][0], 0, 0, 0);
                        acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0.w[nt], f0.x[1], acc[nt][1], 0, 0, 0);
                        if (nt < 3) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        } else {
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (nt < 2) {
                            if (T + 1 < total_taps) prefetch_half(T + 1, nt);
                        } else if (nt == 2) {
                            if (tap == 0 && conv + 1 < nconv) prefetch_bias(conv + 1);
                            if ((conv & 1) && tap < 7 && conv < nconv - 1) prefetch_se(conv >> 1, tap * 4 + wave_u);
                        }
                    }
                } else {
                    mma_step(acc, f0);
                    interleave_step();
                }
                if (ks + 2 < 8) {
                    load_frags(f0, lds, aaddr[tap][0], aaddr[tap][1], wb, ks + 2);
                    mma_step(acc, f1);
                    interleave_step();
                } else {
                    __builtin_amdgcn_sched_barrier(0);
                    wait_weights();
                    __syncthreads();  // next tap landed; everyone is done reading this buffer
                    if (tap + 1 < 9) load_frags(f0, lds, aaddr[tap + 1 < 9 ? tap + 1 : 0][0], aaddr[tap + 1 < 9 ? tap + 1 : 0][1], wb_next, 0);
                    mma_step(acc, f1);
                    interleave_step();
                }
            }
        }
        const unsigned long long ts1 = p.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
        // Epilogue addresses are rebuilt from an opaque zero on every pass: left
        // loop-invariant, the compiler hoists dozens of them out of the conv loop and
        // spills them to scratch, and every reload then sits on the critical path.
        uint32_t oz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
        char* lds_e = lds + oz;
        char* region = lds_e + wave * ACT_REGION;
        if (conv == nconv - 1) {
            // ---- policy conv: + bias, ReLU -> LDS -> p_act (coalesced 16 B per lane)
            store_act<true, false, false>(region, acc, nullptr, resid, r, h);
            if (live) {
                // 64-byte pieces (one k-step of this position) at a 1 KB stride
                char* out = reinterpret_cast<char*>(p.p_act) + (size_t)(pos >> 4) * (POLICY_K / 32) * 1024 + (pos & 15) * 64;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int idx = i * 64 + lane;  // 16-byte granule: pixel idx/16, part idx%16
                    const uint4 v = *reinterpret_cast<const uint4*>(region + (idx >> 4) * ACT_PITCH + (idx & 15) * 16);
                    *reinterpret_cast<uint4*>(out + (idx >> 2) * 1024 + (idx & 3) * 16) = v;
                }
            }
        } else if ((conv & 1) == 0) {
            // ---- conv1 of a block: + bias, ReLU -> LDS
            store_act<true, false, false>(region, acc, nullptr, resid, r, h);
        } else {
            // ---- conv2 of a block: + bias, SE, + residual, ReLU -> LDS and registers
            float* scratch = reinterpret_cast<float*>(region);  // the region is free now
            const float* se = reinterpret_cast<const float*>(lds_e + LDS_SE);
            if (p.dbg & 1) {
                store_act<true, true, false>(region, acc, nullptr, resid, r, h);
                continue;
            }
            // Squeeze: per-channel mean of z (= acc, bias included) over the 64 pixels.  Per
            // channel tile the 16 per-lane partial sums are reduce-scattered over the 16 lanes
            // of a row with DPP lane exchanges (VALU rate, no LDS round trips), leaving lane
            // (r & 15) with the row total of accumulator register r & 15; one cross-row
            // exchange then adds the two rows of the half-wave.
            {
                float rowsum[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x16 v = acc[nt][0] + acc[nt][1];  // a register vector: constant indices only
#pragma unroll
                    for (int step = 0; step < 4; ++step) {
                        const int half = 8 >> step;   // exchange distance in lanes == values kept
                        const bool up = (r & half) != 0;
#pragma unroll
                        for (int i = 0; i < half; ++i) {
                            float a = v[i], bq = v[i + half];
                            // opaque copy: otherwise the two selects below are rewritten as a
                            // dynamically indexed vector read (a 16-way v_cndmask chain each)
                            asm("" : "+v"(a));
                            const float send = up ? a : bq;
                            const float keepv = up ? bq : a;
                            v[i] = keepv + lane_xor(send, half);
                        }
                    }
                    rowsum[nt] = v[0];
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float tot = rowsum[nt] + __shfl_xor(rowsum[nt], 16);
                    const int idx = r & 15;  // accumulator register index this lane reduced
                    const int ch = nt * 32 + (idx >> 2) * 8 + h * 4 + (idx & 3);
                    if (r < 16) scratch[ch] = tot * (1.0f / 64.0f);
                }
            }
            wave_lds_sync();
            const unsigned long long tsa = p.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
            // fc1 (128 -> 16) + ReLU: lane = (j = lane&15, quarter = lane>>4)
            {
                const int j = lane & 15, q4 = lane >> 4;
                const float* w1 = se + j * SE_W1_PITCH + q4 * 32;
                float y = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(w1 + i * 4);
                    const f32x4 m = *reinterpret_cast<const f32x4*>(scratch + q4 * 32 + i * 4);
                    y += w[0] * m[0] + w[1] * m[1] + w[2] * m[2] + w[3] * m[3];
                }
                y += __shfl_xor(y, 16);
                y += __shfl_xor(y, 32);
                y = fmaxf(y + se[SE_OFF_B1 + j], 0.f);
                if (lane < 16) scratch[128 + lane] = y;  // hidden[16]
            }
            wave_lds_sync();
            const unsigned long long tsb = p.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
            // fc2 (16 -> 256) from the transposed weights: this lane computes the gate
            // and the shift of channels lane and lane + 64.  All 68 LDS reads are issued
            // before the first multiply so that their latency is paid once.
            {
                f32x4 hid[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) hid[i] = *reinterpret_cast<const f32x4*>(scratch + 128 + i * 4);
                const float* w2c = se + SE_OFF_W2 + lane;  // one base register, immediate offsets below
                float wv[4][17];                            // [gate0, gate1, shift0, shift1][16 weights + bias]
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int jj = 0; jj < 17; ++jj) {        // row 16 of the image is the bias b2
                        wv[m][jj] = w2c[jj * 256 + 64 * m];
                        wv[2 + m][jj] = w2c[jj * 256 + CH + 64 * m];
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float g = wv[m][16], sh = wv[2 + m][16];
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) {
                        g += wv[m][jj] * hid[jj >> 2][jj & 3];
                        sh += wv[2 + m][jj] * hid[jj >> 2][jj & 3];
                    }
                    g = __builtin_amdgcn_rcpf(1.0f + __expf(-g));  // sigmoid
                    scratch[160 + lane + 64 * m] = g;
                    scratch[160 + CH + lane + 64 * m] = sh;
                }
            }
            wave_lds_sync();
            const unsigned long long tsc = p.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
            // out = relu(acc * gate + shift' + residual)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int ch0 = nt * 32 + gq * 8 + h * 4;
                    const f32x4 gt = *reinterpret_cast<const f32x4*>(scratch + 160 + ch0);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(scratch + 160 + CH + ch0);
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) {
                        const uint32_t rlo = resid[nt][pt][gq * 2 + 0], rhi = resid[nt][pt][gq * 2 + 1];
                        acc[nt][pt][gq * 4 + 0] = acc[nt][pt][gq * 4 + 0] * gt[0] + sh[0] + bf16_lo(rlo);
                        acc[nt][pt][gq * 4 + 1] = acc[nt][pt][gq * 4 + 1] * gt[1] + sh[1] + bf16_hi(rlo);
                        acc[nt][pt][gq * 4 + 2] = acc[nt][pt][gq * 4 + 2] * gt[2] + sh[2] + bf16_lo(rhi);
                        acc[nt][pt][gq * 4 + 3] = acc[nt][pt][gq * 4 + 3] * gt[3] + sh[3] + bf16_hi(rhi);
                    }
                }
            }
            store_act<true, true, false>(region, acc, nullptr, resid, r, h);
            if (p.stamps) {
                const unsigned long long tsd = __builtin_amdgcn_s_memtime();
                t_sq += tsa - ts1; t_f1 += tsb - tsa; t_f2 += tsc - tsb; t_fin += tsd - tsc;
            }
            if (conv == nconv - 2) {
                // ---- value head conv 1x1 (128 -> 32) on the tower output, + bias, ReLU -> v_act
                f32x16 va[2];
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int v = 0; v < 16; ++v) va[pt][v] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const bf16x8 w = *reinterpret_cast<const bf16x8*>(p.value_w + (ks * 64 + lane) * 16);
                    const bf16x8 x0 = lds_read_frag(lds, aaddr[4][0] + ks * 32);
                    const bf16x8 x1 = lds_read_frag(lds, aaddr[4][1] + ks * 32);
                    va[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x0, va[0], 0, 0, 0);
                    va[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x1, va[1], 0, 0, 0);
                }
                if (live) {
                    char* out = reinterpret_cast<char*>(p.v_act) + (size_t)(pos >> 4) * (VALUE_K / 32) * 1024 + (pos & 15) * 64;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int ch0 = gq * 8 + h * 4;
                        const f32x4 b = *reinterpret_cast<const f32x4*>(p.value_b + ch0);
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt) {
                            const float v0 = fmaxf(va[pt][gq * 4 + 0] + b[0], 0.f);
                            const float v1 = fmaxf(va[pt][gq * 4 + 1] + b[1], 0.f);
                            const float v2 = fmaxf(va[pt][gq * 4 + 2] + b[2], 0.f);
                            const float v3 = fmaxf(va[pt][gq * 4 + 3] + b[3], 0.f);
                            *reinterpret_cast<uint2*>(out + (pt * 32 + r) * 1024 + ch0 * 2) =   // k-step = pixel
                                make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                        }
                    }
                }
            }
        }
        if (p.stamps) {
            const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
            t_loop += ts1 - ts0;
            if (conv & 1) t_ep2 += ts2 - ts1; else t_ep1 += ts2 - ts1;
        }
    }
    if (p.stamps && tid == 0) {
        unsigned long long* o = p.stamps + blockIdx.x * 10;
        o[0] = t_loop; o[1] = t_ep1; o[2] = t_ep2; o[3] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz ticks over the conv sequence
        o[4] = t_sq; o[5] = t_f1; o[6] = t_f2; o[7] = t_fin;
        o[8] = rt0; o[9] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------
// Small-batch form of the tower.  k_tower gives every wave a whole position, which
// fills the chip only from 1024 positions up and makes the latency of one forward the
// time one wave needs for all 41 convs.  Here the 128 output channels of a position are
// split over 4 / NT waves (NT channel tiles of 32 per wave), a workgroup holds NT
// positions, and a batch of B positions occupies B / NT compute units: NT = 1 up to 256
// positions, NT = 2 up to 512.  Same weight images, same LDS layout (the activation
// regions the workgroup does not need hold the squeeze-excitation scratch), same
// arithmetic per output element; a workgroup barrier closes every epilogue because a
// conv now reads channels written by other waves.
template <int NT, bool RELU, bool KEEP>
__device__ __forceinline__ void store_act_n(char* region, const f32x16 (&acc)[NT][2], uint32_t (&keep)[NT][2][8],
                                            int r, int h, int nt0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int ch0 = (nt0 + nt) * 32 + gq * 8 + h * 4;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                uint32_t lo = pack_bf16x2(acc[nt][pt][gq * 4 + 0], acc[nt][pt][gq * 4 + 1]);
                uint32_t hi = pack_bf16x2(acc[nt][pt][gq * 4 + 2], acc[nt][pt][gq * 4 + 3]);
                if (RELU) { lo = relu_bf16x2(lo); hi = relu_bf16x2(hi); }
                if (KEEP) { keep[nt][pt][gq * 2 + 0] = lo; keep[nt][pt][gq * 2 + 1] = hi; }
                *reinterpret_cast<uint2*>(region + (pt * 32 + r) * ACT_PITCH + ch0 * 2) = make_uint2(lo, hi);
            }
        }
    }
}

template <int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k_tower_split(TowerParams p) {
    constexpr int WPP = 4 / NT;        // waves per position
    constexpr int PPW = WAVES / WPP;   // positions per workgroup
    // WAVES = 8 (NT = 2, four positions): two waves per SIMD, one's LDS and barrier waits run
    // under the other's MFMAs
    static_assert((NT == 1 || NT == 2) && (WAVES == 4 || (WAVES == 8 && NT == 2)), "a whole position per wave is k_tower");
    __shared__ __attribute__((aligned(1024))) char lds[LDS_TOTAL];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = wave_u / WPP;            // position of this wave inside the workgroup
    const int cw = wave_u % WPP;            // its channel group
    const int nt0 = cw * NT;                // first channel tile of the group
    const int pos = blockIdx.x * PPW + lp;
    bool live = pos < p.B;
    int assigned = -1;
    char* region = lds + lp * ACT_REGION;
    // floats: [0,128) channel means, [128 + 16 cw, +16) hidden layer (per wave), [192,320) gates, [320,448) shifts.
    // With spare activation regions (WAVES = 4) it lives there; otherwise in the position's own
    // region, which nobody reads between a block's last tap and the block's output store.
    float* scratch = (WAVES == 4) ? reinterpret_cast<float*>(lds + 3 * ACT_REGION + lp * 2048)
                                  : reinterpret_cast<float*>(region);
    const int nconv = 2 * p.blocks + 1;
    const int total_taps = nconv * 9;

    if (tid < 512 / 4) reinterpret_cast<uint32_t*>(lds + LDS_ZERO)[tid] = 0;

    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void*)lds;
    // Weight taps of the forms with two channel tiles per wave: copied one tap ahead into two
    // buffers shared by the workgroup.  (NT = 1 keeps its weights in registers, see below.)
    constexpr int AHEAD = 1;
    auto tap_buffer = [&](int slot) -> uint32_t { return (uint32_t)(LDS_W + slot * TAP_BYTES); };  // slot = tap parity
    auto prefetch = [&](int T, int slot) {
        if (WAVES == 4) {
            const uint8_t* src = p.conv_w + (size_t)T * TAP_BYTES + wave_u * 8192 + lane * 16;
            const uint32_t dst = lds_base + tap_buffer(slot) + wave_u * 8192;
            glds16x8(src + 4096, __builtin_amdgcn_readfirstlane(dst + 4096));
        } else {
            const uint8_t* src = p.conv_w + (size_t)T * TAP_BYTES + wave_u * 4096 + lane * 16;
            const uint32_t dst = lds_base + tap_buffer(slot) + wave_u * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(src + i * 1024, __builtin_amdgcn_readfirstlane(dst + i * 1024));
        }
    };
    auto prefetch_bias = [&](int c) {
        if (wave_u < 2)
            glds4(reinterpret_cast<const uint8_t*>(p.conv_b + (size_t)c * CH) + wave_u * 256 + lane * 4,
                  __builtin_amdgcn_readfirstlane(lds_base + LDS_BIAS + (c & 1) * 512 + wave_u * 256));
    };
    auto prefetch_se = [&](int blk, int piece) {
        if (piece < SE_PIECES)
            glds16(p.se_img + (size_t)blk * SE_BYTES + piece * 1024 + lane * 16,
                   __builtin_amdgcn_readfirstlane(lds_base + LDS_SE + piece * 1024));
    };
    if constexpr (NT != 1) {
#pragma unroll
        for (int t = 0; t < AHEAD; ++t) prefetch(t, t);
    }
    prefetch_bias(0);
    if (p.need) {
        bool has_rows;
        assigned = assign_rows<WAVES * 64, PPW>(p, tid, lp, has_rows);
        if (!has_rows) { wait_weights(); return; }   // no copy into LDS may outlive the workgroup
        live = assigned >= 0;
    }

    uint32_t aaddr[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int px = pt * 32 + r;
            const int y = (px >> 3) + dy, x = (px & 7) + dx;
            const bool ok = (y >= 0) && (y < 8) && (x >= 0) && (x < 8);
            aaddr[tap][pt] = ok ? (uint32_t)(lp * ACT_REGION + (y * 8 + x) * ACT_PITCH + h * 16)
                                : (uint32_t)(LDS_ZERO + (((y * 8 + x) & 15) * 16 + h * 16));  // the banks the pixel would have had
        }
    }

    f32x16 acc[NT][2];
    uint32_t resid[NT][2][8];

    // ---------------- stem
    {
        const int src_pos = p.need ? assigned : pos;
        const uint64_t mine = live ? p.mine[src_pos] : 0ull, theirs = live ? p.theirs[src_pos] : 0ull;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.stem_b + (nt0 + nt) * 32 + gq * 8 + h * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[nt][0][gq * 4 + q] = b[q]; acc[nt][1][gq * 4 + q] = b[q]; }
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xb[2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int px = pt * 32 + r;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = ks * 16 + h * 8 + j;  // k = tap*3 + plane
                    const int tap = k / 3, pl = k - tap * 3;
                    const int y = (px >> 3) + tap / 3 - 1, x = (px & 7) + tap % 3 - 1;
                    const bool ok = (k < 27) && (y >= 0) && (y < 8) && (x >= 0) && (x < 8);
                    const int sq = (y * 8 + x) & 63;
                    const uint64_t src = (pl == 0) ? mine : ((pl == 1) ? theirs : ~0ull);
                    xb[pt][j] = (__bf16)((ok && ((src >> sq) & 1ull)) ? 1.0f : 0.0f);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x8 w = *reinterpret_cast<const bf16x8*>(p.stem_w + ((ks * 4 + nt0 + nt) * 64 + lane) * 16);
                acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xb[0], acc[nt][0], 0, 0, 0);
                acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xb[1], acc[nt][1], 0, 0, 0);
            }
        }
        store_act_n<NT, true, true>(region, acc, resid, r, h, nt0);
    }
    wait_weights();
    __syncthreads();  // the first taps landed, zero row and every channel group's stem output visible

    // NT = 1: a wave is the only reader of its weight fragments (channel tile nt0 of every k-step),
    // so they never pass through the LDS: three taps of them live in registers, loaded from L2 in
    // the fragment layout, each 1 KB piece replaced by the same piece three taps on right behind
    // the MFMAs that read it.  The LDS then carries the activation fragments only.
    bf16x8 wq[3][8];
    const uint8_t* wsrc = p.conv_w + nt0 * 1024 + lane * 16;
    if constexpr (NT == 1) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                wq[t][ks] = *reinterpret_cast<const bf16x8*>(wsrc + (size_t)t * TAP_BYTES + ks * 4096);
    }

    const bool stamps_on = AZ_SPLIT_STAMPS && p.stamps != nullptr;
    // diagnostic cycle sums (p.stamps): conv start, tap loops, wait at the end of the taps, plain and
    // squeeze-excitation epilogues
    unsigned long long t_pre = 0, t_taps = 0, t_bar = 0, t_ep1 = 0, t_ep2 = 0;
    const unsigned long long rt0 = stamps_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
    for (int conv = 0; conv < nconv; ++conv) {
        const unsigned long long ts0 = stamps_on ? __builtin_amdgcn_s_memtime() : 0ull;
        {
            const float* b0 = reinterpret_cast<const float*>(lds + LDS_BIAS + (conv & 1) * 512) + h * 4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(b0 + (nt0 + nt) * 32 + gq * 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { acc[nt][0][gq * 4 + q] = b[q]; acc[nt][1][gq * 4 + q] = b[q]; }
                }
        }
        // Activation fragments run one tap ahead of the MFMAs (they do not depend on the weight
        // copy); weight fragments are requested right behind the barrier that publishes their
        // tap.  A deeper software pipeline does not pay here: with 1.5 fragment reads per MFMA
        // (twice k_tower's) the LDS is as busy as the matrix pipe.
        bf16x8 xc[8][2];
        if constexpr (WAVES == 4) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                xc[ks][0] = lds_read_frag(lds, aaddr[0][0] + ks * 32);
                xc[ks][1] = lds_read_frag(lds, aaddr[0][1] + ks * 32);
            }
        }
        const unsigned long long ts1 = stamps_on ? __builtin_amdgcn_s_memtime() : 0ull;
        unsigned long long ts2 = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int T = conv * 9 + tap;
            const int slot = (conv + tap) & 1;  // conv * 9 + tap has the parity of conv + tap
            if constexpr (NT == 1) {
                // One wave per SIMD feeds the matrix pipe, so everything else is issued between MFMAs.
                // Per k-step: the two MFMAs; the load of the same weight piece of tap T + 3 into the
                // registers the MFMAs just read (past the end of the network the last tap is loaded
                // again and never used, so that every tap issues the same 8 loads and the counted wait
                // at the end of the conv holds everywhere); the activation fragments of k-step ks of
                // tap T + 1.
                bf16x8 xn1[8][2];
                const int T3 = (T + 3 < total_taps) ? T + 3 : total_taps - 1;
                const uint8_t* src3 = wsrc + (size_t)T3 * TAP_BYTES;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    // instructions issue in order: what follows an MFMA issues while it runs, what
                    // follows two MFMAs in a row waits for the first to leave the pipe
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[tap % 3][ks], xc[ks][0], acc[0][0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap + 1 < 9) {
                        xn1[ks][0] = lds_read_frag(lds, aaddr[tap + 1 < 9 ? tap + 1 : 0][0] + ks * 32);
                        xn1[ks][1] = lds_read_frag(lds, aaddr[tap + 1 < 9 ? tap + 1 : 0][1] + ks * 32);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[tap % 3][ks], xc[ks][1], acc[0][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    wq[tap % 3][ks] = *reinterpret_cast<const bf16x8*>(src3 + ks * 4096);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (tap == 0 && conv + 1 < nconv) prefetch_bias(conv + 1);
                if ((conv & 1) && tap < 7 && conv < nconv - 1) prefetch_se(conv >> 1, tap * WAVES + wave_u);
                if (tap + 1 < 9) {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) { xc[ks][0] = xn1[ks][0]; xc[ks][1] = xn1[ks][1]; }
                } else {
                    // End of the conv: the SE image and the next bias (copied by other waves too) must
                    // have landed, and every wave must be done reading the activations.  Only the 16
                    // weight loads of taps 7 and 8 are younger than those pieces.
                    if (stamps_on) ts2 = __builtin_amdgcn_s_memtime();
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    __syncthreads();
                }
                continue;
            }
            const uint32_t wb = tap_buffer(slot) + lane * 16 + nt0 * 1024;
            if constexpr (WAVES == 4 && NT == 2) {
                // One wave per SIMD: weight fragments run one k-step ahead of the MFMAs, the activation
                // fragments of tap T + 1 are fetched as this tap goes, and every read is issued right
                // behind an MFMA (instructions issue in order: what follows an MFMA issues while it
                // runs).  The first weight fragments are requested before the copy of the next tap is
                // set up, so that their latency runs under those instructions.
                bf16x8 fw[2][NT], xn[8][2];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fw[0][nt] = lds_read_frag(lds, wb + nt * 1024);
                if (T + AHEAD < total_taps) prefetch(T + AHEAD, slot ^ 1);
                if (tap == 0 && conv + 1 < nconv) prefetch_bias(conv + 1);
                if ((conv & 1) && tap < 7 && conv < nconv - 1) prefetch_se(conv >> 1, tap * WAVES + wave_u);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int c = ks & 1, n = c ^ 1;
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[c][0], xc[ks][0], acc[0][0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 1 < 8) fw[n][0] = lds_read_frag(lds, wb + ((ks + 1) * 4 + 0) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[c][0], xc[ks][1], acc[0][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 1 < 8) fw[n][1] = lds_read_frag(lds, wb + ((ks + 1) * 4 + 1) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[c][1], xc[ks][0], acc[1][0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap + 1 < 9) xn[ks][0] = lds_read_frag(lds, aaddr[tap + 1 < 9 ? tap + 1 : 0][0] + ks * 32);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[c][1], xc[ks][1], acc[1][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap + 1 < 9) xn[ks][1] = lds_read_frag(lds, aaddr[tap + 1 < 9 ? tap + 1 : 0][1] + ks * 32);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // next tap landed and visible, everyone done reading this buffer (and, behind the last
                // tap, the activations; the SE image and the bias are read next)
                wait_weights();
                __syncthreads();
                if (tap + 1 < 9) {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) { xc[ks][0] = xn[ks][0]; xc[ks][1] = xn[ks][1]; }
                }
                continue;
            }
            if (T + AHEAD < total_taps) prefetch(T + AHEAD, slot ^ 1);
            if (tap == 0 && conv + 1 < nconv) prefetch_bias(conv + 1);
            if ((conv & 1) && tap < 7 && conv < nconv - 1) prefetch_se(conv >> 1, tap * WAVES + wave_u);
            if constexpr (WAVES == 8) {
                // 256 registers per wave: fragments one k-step ahead only; the other wave of the
                // SIMD covers what this leaves exposed
                bf16x8 fw[2][NT], fx[2][2];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fw[0][nt] = lds_read_frag(lds, wb + nt * 1024);
                fx[0][0] = lds_read_frag(lds, aaddr[tap][0]);
                fx[0][1] = lds_read_frag(lds, aaddr[tap][1]);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int c = ks & 1, n = c ^ 1;
                    if (ks + 1 < 8) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) fw[n][nt] = lds_read_frag(lds, wb + ((ks + 1) * 4 + nt) * 1024);
                        fx[n][0] = lds_read_frag(lds, aaddr[tap][0] + (ks + 1) * 32);
                        fx[n][1] = lds_read_frag(lds, aaddr[tap][1] + (ks + 1) * 32);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[c][nt], fx[c][0], acc[nt][0], 0, 0, 0);
                        acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[c][nt], fx[c][1], acc[nt][1], 0, 0, 0);
                    }
                }
                wait_weights();
                __syncthreads();
                continue;
            }
        }
        const unsigned long long ts3 = stamps_on ? __builtin_amdgcn_s_memtime() : 0ull;
        if (NT != 1) ts2 = ts3;
        if (conv == nconv - 1) {
            // ---- policy conv: ReLU -> LDS -> p_act
            store_act_n<NT, true, false>(region, acc, resid, r, h, nt0);
            __syncthreads();
            if (live) {
                char* out = reinterpret_cast<char*>(p.p_act) + (size_t)(pos >> 4) * (POLICY_K / 32) * 1024 + (pos & 15) * 64;
                for (int i = cw; i < 16; i += WPP) {
                    const int idx = i * 64 + lane;
                    const uint4 v = *reinterpret_cast<const uint4*>(region + (idx >> 4) * ACT_PITCH + (idx & 15) * 16);
                    *reinterpret_cast<uint4*>(out + (idx >> 2) * 1024 + (idx & 3) * 16) = v;
                }
            }
        } else if ((conv & 1) == 0) {
            store_act_n<NT, true, false>(region, acc, resid, r, h, nt0);
            __syncthreads();
        } else {
            const float* se = reinterpret_cast<const float*>(lds + LDS_SE);
            // squeeze of this wave's channels
            {
                float rowsum[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x16 v = acc[nt][0] + acc[nt][1];
#pragma unroll
                    for (int step = 0; step < 4; ++step) {
                        const int half = 8 >> step;
                        const bool up = (r & half) != 0;
#pragma unroll
                        for (int i = 0; i < half; ++i) {
                            float a = v[i], bq = v[i + half];
                            asm("" : "+v"(a));
                            const float send = up ? a : bq;
                            const float keepv = up ? bq : a;
                            v[i] = keepv + lane_xor(send, half);
                        }
                    }
                    rowsum[nt] = v[0];
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float tot = rowsum[nt] + __shfl_xor(rowsum[nt], 16);
                    const int idx = r & 15;
                    const int ch = (nt0 + nt) * 32 + (idx >> 2) * 8 + h * 4 + (idx & 3);
                    if (r < 16) scratch[ch] = tot * (1.0f / 64.0f);
                }
            }
            // the SE weights this lane needs are requested before the barrier: only the means and
            // the hidden layer are then on the critical path behind it
            const int j1 = lane & 15, q4 = lane >> 4;
            f32x4 w1v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w1v[i] = *reinterpret_cast<const f32x4*>(se + j1 * SE_W1_PITCH + q4 * 32 + i * 4);
            const float b1v = se[SE_OFF_B1 + j1];
            float w2v[NT][17];
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int jj = 0; jj < 17; ++jj) w2v[m][jj] = se[SE_OFF_W2 + h * CH + (nt0 + m) * 32 + r + jj * 256];
            __syncthreads();  // the means of every channel group
            // fc1 (128 -> 16) + ReLU, by every wave of the position for itself
            {
                float y = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 m = *reinterpret_cast<const f32x4*>(scratch + q4 * 32 + i * 4);
                    y += w1v[i][0] * m[0] + w1v[i][1] * m[1] + w1v[i][2] * m[2] + w1v[i][3] * m[3];
                }
                y += __shfl_xor(y, 16);
                y += __shfl_xor(y, 32);
                y = fmaxf(y + b1v, 0.f);
                if (lane < 16) scratch[128 + cw * 16 + lane] = y;
            }
            wave_lds_sync();
            // fc2 for this wave's channels: lanes 0-31 the gates, lanes 32-63 the shifts
            {
                f32x4 hid[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) hid[i] = *reinterpret_cast<const f32x4*>(scratch + 128 + cw * 16 + i * 4);
#pragma unroll
                for (int m = 0; m < NT; ++m) {
                    const int col = h * CH + (nt0 + m) * 32 + r;
                    float g = w2v[m][16];
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) g += w2v[m][jj] * hid[jj >> 2][jj & 3];
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-g));
                    scratch[192 + col] = h ? g : sg;
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int ch0 = (nt0 + nt) * 32 + gq * 8 + h * 4;
                    const f32x4 gt = *reinterpret_cast<const f32x4*>(scratch + 192 + ch0);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(scratch + 192 + CH + ch0);
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) {
                        const uint32_t rlo = resid[nt][pt][gq * 2 + 0], rhi = resid[nt][pt][gq * 2 + 1];
                        acc[nt][pt][gq * 4 + 0] = acc[nt][pt][gq * 4 + 0] * gt[0] + sh[0] + bf16_lo(rlo);
                        acc[nt][pt][gq * 4 + 1] = acc[nt][pt][gq * 4 + 1] * gt[1] + sh[1] + bf16_hi(rlo);
                        acc[nt][pt][gq * 4 + 2] = acc[nt][pt][gq * 4 + 2] * gt[2] + sh[2] + bf16_lo(rhi);
                        acc[nt][pt][gq * 4 + 3] = acc[nt][pt][gq * 4 + 3] * gt[3] + sh[3] + bf16_hi(rhi);
                    }
                }
            }
            if (WAVES == 8) __syncthreads();  // every wave has read its gates: the scratch may be overwritten
            store_act_n<NT, true, true>(region, acc, resid, r, h, nt0);
            __syncthreads();
            if (conv == nconv - 2 && cw == 0) {
                // ---- value head conv 1x1 (128 -> 32) on the tower output, by the first wave of the position
                f32x16 va[2];
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int v = 0; v < 16; ++v) va[pt][v] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const bf16x8 w = *reinterpret_cast<const bf16x8*>(p.value_w + (ks * 64 + lane) * 16);
                    const bf16x8 x0 = lds_read_frag(lds, aaddr[4][0] + ks * 32);
                    const bf16x8 x1 = lds_read_frag(lds, aaddr[4][1] + ks * 32);
                    va[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x0, va[0], 0, 0, 0);
                    va[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x1, va[1], 0, 0, 0);
                }
                if (live) {
                    char* out = reinterpret_cast<char*>(p.v_act) + (size_t)(pos >> 4) * (VALUE_K / 32) * 1024 + (pos & 15) * 64;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int ch0 = gq * 8 + h * 4;
                        const f32x4 b = *reinterpret_cast<const f32x4*>(p.value_b + ch0);
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt) {
                            const float v0 = fmaxf(va[pt][gq * 4 + 0] + b[0], 0.f);
                            const float v1 = fmaxf(va[pt][gq * 4 + 1] + b[1], 0.f);
                            const float v2 = fmaxf(va[pt][gq * 4 + 2] + b[2], 0.f);
                            const float v3 = fmaxf(va[pt][gq * 4 + 3] + b[3], 0.f);
                            *reinterpret_cast<uint2*>(out + (pt * 32 + r) * 1024 + ch0 * 2) =
                                make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                        }
                    }
                }
            }
        }
        if (stamps_on) {
            const unsigned long long ts4 = __builtin_amdgcn_s_memtime();
            t_pre += ts1 - ts0; t_taps += ts2 - ts1; t_bar += ts3 - ts2;
            if (conv & 1) t_ep2 += ts4 - ts3; else t_ep1 += ts4 - ts3;
        }
    }
    if (stamps_on && tid == 0 && blockIdx.x < (unsigned)((p.B + 3) / 4)) {
        unsigned long long* o = p.stamps + blockIdx.x * 10;
        o[0] = t_taps; o[1] = t_ep1; o[2] = t_ep2; o[3] = __builtin_amdgcn_s_memrealtime() - rt0;
        o[4] = t_pre; o[5] = t_bar; o[6] = 0; o[7] = 0; o[8] = rt0; o[9] = __builtin_amdgcn_s_memrealtime();
    }
    wait_weights();  // no copy into LDS may outlive the workgroup
}

// ---------------------------------------------------------------------------
// Head GEMMs: out_part[chunk][pos][N] = act[pos][chunk K-range] . W[n][same K-range]
// v_mfma_f32_16x16x32_bf16, 16 positions per workgroup, wave w owns NTW n-tiles.
struct HeadParams {
    const __bf16* p_act;   // K = 8192, tiled as described at TowerParams
    const __bf16* v_act;   // K = 2048
    const uint8_t* wp;     // policy FC, [kstep 256][ntile 12][lane 64][16 B]
    const uint8_t* wv;     // value FC1, [kstep 64][ntile 8][lane 64][16 B]
    float* part_p;         // [8][B][192]
    float* part_v;         // [2][B][128]
    int B;
    const int32_t* n_rows; // rows in use when the tower assigned them itself (else NULL: B)
};

// A workgroup covers HEAD_MT x 16 positions and one K chunk; wave w owns NTW n-tiles.
// Each weight fragment is used for HEAD_MT position tiles, which divides the L2 -> CU
// weight traffic (the bound of this kernel) by HEAD_MT.
constexpr int HEAD_MT = 1;

template <int NTW, int NTILES>
__device__ __forceinline__ void head_gemm_body(const __bf16* act, int K, const uint8_t* w, float* part,
                                               int N, int B, int chunk, int tile, int wave, int lane) {
    const int row = lane & 15, quarter = lane >> 4;
    const __bf16* arow[HEAD_MT];
#pragma unroll
    for (int m = 0; m < HEAD_MT; ++m) {
        // rows past B in the last tile hold stale data: an MFMA output row depends on its own
        // input row only, and the stores below are masked
        arow[m] = act + ((size_t)(tile * HEAD_MT + m) * (K / 32) + chunk * (HEAD_KCHUNK / 32)) * 512 + row * 32 + quarter * 8;
    }
    f32x4 acc[HEAD_MT][NTW];
#pragma unroll
    for (int m = 0; m < HEAD_MT; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ks0 = chunk * (HEAD_KCHUNK / 32);
    const uint8_t* wbase = w + (((size_t)ks0 * NTILES + wave * NTW) * 64 + lane) * 16;
    // latency-bound: two register sets of HEAD_DEPTH k-steps each, the loads of the next set in
    // flight while the MFMAs consume the current one
    constexpr int HEAD_DEPTH = 4, GROUPS = HEAD_KCHUNK / 32 / HEAD_DEPTH;
    bf16x8 a[2][HEAD_DEPTH][HEAD_MT], bb[2][HEAD_DEPTH][NTW];
    auto load_group = [&](int g, int set) {
#pragma unroll
        for (int d = 0; d < HEAD_DEPTH; ++d) {
            const int ks = g * HEAD_DEPTH + d;
#pragma unroll
            for (int m = 0; m < HEAD_MT; ++m) a[set][d][m] = *reinterpret_cast<const bf16x8*>(arow[m] + ks * 512);
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                bb[set][d][t] = *reinterpret_cast<const bf16x8*>(wbase + ((size_t)ks * NTILES + t) * 1024);
        }
    };
    load_group(0, 0);
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        if (g + 1 < GROUPS) load_group(g + 1, (g + 1) & 1);
#pragma unroll
        for (int d = 0; d < HEAD_DEPTH; ++d)
#pragma unroll
            for (int m = 0; m < HEAD_MT; ++m)
#pragma unroll
                for (int t = 0; t < NTW; ++t)
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g & 1][d][m], bb[g & 1][d][t], acc[m][t], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < HEAD_MT; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int n = (wave * NTW + t) * 16 + (lane & 15);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int pp = (tile * HEAD_MT + m) * 16 + (lane >> 4) * 4 + v;
                if (pp < B) part[((size_t)chunk * B + pp) * N + n] = acc[m][t][v];
            }
        }
}

__global__ __launch_bounds__(256) void k_head_gemm(HeadParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x, c = blockIdx.y;
    if (p.n_rows && tile * 16 * HEAD_MT >= *p.n_rows) return;   // no position of this tile is in use
    if (c < POLICY_CHUNKS)
        head_gemm_body<3, 12>(p.p_act, POLICY_K, p.wp, p.part_p, NUM_ACTIONS, p.B, c, tile, wave, lane);
    else
        head_gemm_body<2, 8>(p.v_act, VALUE_K, p.wv, p.part_v, 128, p.B, c - POLICY_CHUNKS, tile, wave, lane);
}

// The same GEMMs for a policy FC of any K = 64 * num_filters (k_tower_any's networks): POLICY_CHUNKS
// chunks of K / POLICY_CHUNKS, so that the partial sums have the shape the finish code reads;
// plain k-step loop (these networks are small or off the tuned path).
template <int NTW, int NTILES>
__device__ __forceinline__ void head_gemm_any_body(const __bf16* act, int K, int kchunk, const uint8_t* w, float* part,
                                                   int N, int B, int chunk, int tile, int wave, int lane) {
    const int row = lane & 15, quarter = lane >> 4;
    const int ks0 = chunk * (kchunk / 32);
    const __bf16* arow = act + ((size_t)tile * (K / 32) + ks0) * 512 + row * 32 + quarter * 8;
    const uint8_t* wbase = w + (((size_t)ks0 * NTILES + wave * NTW) * 64 + lane) * 16;
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < kchunk / 32; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + (size_t)ks * 512);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(wbase + ((size_t)ks * NTILES + t) * 1024);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int n = (wave * NTW + t) * 16 + (lane & 15);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int pp = tile * 16 + (lane >> 4) * 4 + v;
            if (pp < B) part[((size_t)chunk * B + pp) * N + n] = acc[t][v];
        }
    }
}

__global__ __launch_bounds__(256) void k_head_gemm_any(HeadParams p, int policy_k) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x, c = blockIdx.y;
    if (p.n_rows && tile * 16 >= *p.n_rows) return;
    if (c < POLICY_CHUNKS)
        head_gemm_any_body<3, 12>(p.p_act, policy_k, policy_k / POLICY_CHUNKS, p.wp, p.part_p, NUM_ACTIONS, p.B, c, tile, wave, lane);
    else
        head_gemm_any_body<2, 8>(p.v_act, VALUE_K, HEAD_KCHUNK, p.wv, p.part_v, 128, p.B, c - POLICY_CHUNKS, tile, wave, lane);
}

// one wave per position
__global__ __launch_bounds__(256) void k_head_finish(FinishParams p) {
    const int lane = threadIdx.x & 63;
    const int pos = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pos >= p.B || (p.n_rows && pos >= *p.n_rows)) return;
    float pr[3];
    const float value = dev_head_finish(p, pos, lane, pr);
#pragma unroll
    for (int i = 0; i < 3; ++i) p.probs[(size_t)pos * NUM_ACTIONS + lane + 64 * i] = pr[i];
    if (lane == 0) p.value[pos] = value;
}

// planes [B][3][8][8] float32 (0/1 values, third plane ones) -> canonical bitboards
__global__ void k_planes_to_bits(const float* planes, int B, uint64_t* mine, uint64_t* theirs, int32_t* bad) {
    const int lane = threadIdx.x & 63;
    const int pos = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pos >= B) return;
    const float a = planes[(size_t)pos * 192 + lane], b = planes[(size_t)pos * 192 + 64 + lane],
                c = planes[(size_t)pos * 192 + 128 + lane];
    const bool ok = (a == 0.f || a == 1.f) && (b == 0.f || b == 1.f) && (c == 1.f);
    const unsigned long long m = __ballot(a == 1.f), t = __ballot(b == 1.f);
    if (__ballot(!ok) && lane == 0) atomicOr(bad, 1);
    if (lane == 0) { mine[pos] = m; theirs[pos] = t; }
}

}  // namespace az

using namespace az;

// ---------------------------------------------------------------------------
// Head buffers of one forward: tiled head activations and split-K partial sums.
struct HeadBuffers {
    __bf16 *p_act, *v_act;
    float *part_p, *part_v;
};

// az_net_forward_noalloc keeps those buffers in device memory that belongs to the loaded
// library instead of memory obtained from hipMalloc: room for AZ_NET_NOALLOC_MAX_BATCH
// positions (6.75 MiB; every GPU the library runs on holds its own copy until the process
// ends).  In host code the array's name is only a handle for hipGetSymbolAddress: the
// device address is looked up once per device, in az_net_create with that device
// current, and every launch is given that address.
constexpr size_t NOALLOC_B = AZ_NET_NOALLOC_MAX_BATCH;
static_assert(NOALLOC_B % 16 == 0, "whole head tiles");
constexpr size_t NOALLOC_OFF_VACT = NOALLOC_B * POLICY_K * sizeof(__bf16);
constexpr size_t NOALLOC_OFF_PARTP = NOALLOC_OFF_VACT + NOALLOC_B * VALUE_K * sizeof(__bf16);
constexpr size_t NOALLOC_OFF_PARTV = NOALLOC_OFF_PARTP + (size_t)POLICY_CHUNKS * NOALLOC_B * NUM_ACTIONS * sizeof(float);
constexpr size_t NOALLOC_BYTES = NOALLOC_OFF_PARTV + (size_t)VALUE_CHUNKS * NOALLOC_B * 128 * sizeof(float);
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

constexpr int MAX_DEVICES = 64;
static unsigned char* g_scratch_ptr[MAX_DEVICES];   // device address of `scratch` on each device, 0 = not asked yet
static std::mutex g_scratch_mutex;

// The calling thread's current device must be `device`.
static int scratch_address(int device, unsigned char** out) {
    AZ_CHECK_ARG(device >= 0 && device < MAX_DEVICES, "device ordinal %d out of range", device);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        AZ_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) AZ_FAIL(AZ_ERR_HIP, "hipGetSymbolAddress returned a null address for the static head buffers");
        g_scratch_ptr[device] = (unsigned char*)ptr;
    }
    *out = g_scratch_ptr[device];
    return AZ_OK;
}

struct az_net {
    int blocks = 0, max_batch = 0, device = 0;
    int ch = CH, se_hid = SE_HID;
    bool any_width = false;   // not 128 filters with 16 hidden units: k_tower_any + k_head_gemm_any, their own weight images
    bool loaded = false;
    std::vector<void*> allocs;
    uint8_t *stem_w = nullptr, *conv_w = nullptr, *value_w = nullptr, *wp = nullptr, *wv = nullptr;
    uint8_t* se_img = nullptr;
    uint8_t *stem_w16 = nullptr, *conv_w16 = nullptr, *value_w16 = nullptr;   // the same weights in k_tower16's fragment order
    float *stem_b = nullptr, *conv_b = nullptr, *value_b = nullptr, *bp = nullptr, *bv1 = nullptr, *wv2 = nullptr, *bv2 = nullptr;
    __bf16 *p_act = nullptr, *v_act = nullptr;
    float *part_p = nullptr, *part_v = nullptr;
    uint64_t *bits_mine = nullptr, *bits_theirs = nullptr;
    int32_t* bad = nullptr;
    unsigned long long* stamps = nullptr;  // diagnostic cycle stamps (AZ_TOWER_STAMPS=1)
    // optional HIP-event timing of the tower kernel (az_net_set_timing)
    bool timing = false;
    int tower_form = 0;   // 0: chosen by batch size; 1, 2, 4, 8, 16: forced (az_net_set_tower_form)
    int64_t launches = 0;
    int ev_used = 0;
    std::vector<hipEvent_t> ev0, ev1;
    HeadBuffers own() const { return HeadBuffers{p_act, v_act, part_p, part_v}; }
};
constexpr int TIMING_POOL = 512, TIMING_STRIDE = 8;

static uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x7FFFFFu)) return (uint16_t)((u >> 16) | 0x40);  // NaN
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

template <typename T>
static int dev_alloc(az_net* n, T*& ptr, size_t count) {
    void* p = nullptr;
    AZ_HIP(hipMalloc(&p, count * sizeof(T)));
    n->allocs.push_back(p);  // owned from here on: az_net_destroy frees it on every later failure
    AZ_HIP(hipMemset(p, 0, count * sizeof(T)));
    ptr = (T*)p;
    return AZ_OK;
}

// diagnostic: copies the [workgroups][10] cycle stamps of the last tower launch
extern "C" int az_net_debug_stamps(az_net* n, unsigned long long* out_host, int32_t workgroups) {
    AZ_CHECK_ARG(n && n->stamps && out_host, "az_net_debug_stamps: stamps are off (set AZ_TOWER_STAMPS=1 before az_net_create)");
    AZ_CHECK_ARG(workgroups >= 0 && workgroups <= (n->max_batch + 3) / 4,
                 "az_net_debug_stamps: %d workgroups asked, the network holds stamps for %d", workgroups, (n->max_batch + 3) / 4);
    AZ_HIP(hipDeviceSynchronize());
    AZ_HIP(hipMemcpy(out_host, n->stamps, (size_t)workgroups * 10 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return AZ_OK;
}

extern "C" int az_net_max_batch(az_net* n) { return n ? n->max_batch : 0; }

extern "C" int az_net_set_tower_form(az_net* n, int32_t form) {
    AZ_CHECK_ARG(n, "az_net_set_tower_form: null network");
    AZ_CHECK_ARG(form == 0 || form == 1 || form == 2 || form == 4 || form == 8 || form == 16, "az_net_set_tower_form: form %d is not one of 0, 1, 2, 4, 8, 16", form);
    AZ_CHECK_ARG(form == 0 || !n->any_width, "az_net_set_tower_form: a %d-filter network with %d hidden units has one tower kernel (form 0)", n->ch, n->se_hid);
    n->tower_form = form;
    return AZ_OK;
}

extern "C" int az_net_set_timing(az_net* n, int32_t enable) {
    AZ_CHECK_ARG(n, "az_net_set_timing: null network");
    if (enable && n->ev0.empty()) {
        n->ev0.resize(TIMING_POOL); n->ev1.resize(TIMING_POOL);
        for (int i = 0; i < TIMING_POOL; ++i) { AZ_HIP(hipEventCreate(&n->ev0[i])); AZ_HIP(hipEventCreate(&n->ev1[i])); }
    }
    n->timing = enable != 0;
    n->ev_used = 0;
    n->launches = 0;
    return AZ_OK;
}

extern "C" int az_net_tower_time(az_net* n, double* mean_ms, int64_t* samples) {
    AZ_CHECK_ARG(n && mean_ms && samples, "az_net_tower_time: null argument");
    double sum = 0;
    for (int i = 0; i < n->ev_used; ++i) {
        AZ_HIP(hipEventSynchronize(n->ev1[i]));
        float ms = 0;
        AZ_HIP(hipEventElapsedTime(&ms, n->ev0[i], n->ev1[i]));
        sum += ms;
    }
    *samples = n->ev_used;
    *mean_ms = n->ev_used ? sum / n->ev_used : 0.0;
    n->ev_used = 0;
    return AZ_OK;
}

extern "C" int az_net_destroy(az_net* n) {
    if (!n) return AZ_OK;
    for (hipEvent_t ev : n->ev0) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : n->ev1) (void)hipEventDestroy(ev);
    for (void* p : n->allocs) (void)hipFree(p);
    delete n;
    return AZ_OK;
}

extern "C" int az_net_create(az_net** out, const az_net_config* cfg) {
    AZ_CHECK_ARG(out && cfg, "az_net_create: null argument");
    AZ_CHECK_ARG(cfg->num_filters >= 32 && cfg->num_filters <= 256 && cfg->num_filters % 32 == 0,
                 "az_net_create: num_filters must be a multiple of 32 between 32 and 256 (got %d)", cfg->num_filters);
    AZ_CHECK_ARG(cfg->se_ratio >= 1 && cfg->se_ratio <= cfg->num_filters,
                 "az_net_create: se_ratio %d leaves no hidden unit for %d filters", cfg->se_ratio, cfg->num_filters);
    AZ_CHECK_ARG(cfg->num_blocks >= 1 && cfg->num_blocks <= 64, "az_net_create: num_blocks %d out of range", cfg->num_blocks);
    AZ_CHECK_ARG(cfg->max_batch >= 1 && cfg->max_batch <= (1 << 20), "az_net_create: max_batch %d out of range", cfg->max_batch);
    if (cfg->device >= 0) AZ_HIP(hipSetDevice(cfg->device));
    int device = 0;
    AZ_HIP(hipGetDevice(&device));
    az_net* n = new az_net();
    n->blocks = cfg->num_blocks;
    n->max_batch = cfg->max_batch;
    n->device = device;
    n->ch = cfg->num_filters;
    n->se_hid = cfg->num_filters / cfg->se_ratio;   // nn.Linear(channels, channels // se_ratio), src/model.py:33
    n->any_width = !(n->ch == CH && n->se_hid == SE_HID);
    const size_t nb = n->blocks, nconv = 2 * nb + 1, B = n->max_batch;
    const size_t Cw = n->ch, Hw = n->se_hid;
    unsigned char* static_buffers = nullptr;
    int rc = scratch_address(device, &static_buffers);   // once per device, while it is current
#define A(ptr, cnt) if (rc == AZ_OK) rc = dev_alloc(n, n->ptr, (cnt))
    if (!n->any_width) {
        A(stem_w, 2 * 4 * 64 * 16);
        A(conv_w, nconv * CONV_BYTES);
        A(se_img, nb * SE_BYTES);
        A(stem_w16, 8 * 64 * 16); A(conv_w16, nconv * CONV_BYTES); A(value_w16, 4 * 2 * 64 * 16);
        A(value_w, 8 * 64 * 16);
    } else {   // k_tower_any's images (TowerParams::ch)
        A(stem_w16, (Cw / 16) * 64 * 16);
        A(conv_w16, nconv * 9 * (Cw / 32) * (Cw / 16) * 1024);
        A(se_img, nb * (Hw * Cw + Hw + 2 * Cw * Hw + 2 * Cw) * sizeof(float));
        A(value_w16, (Cw / 32) * 2 * 64 * 16);
    }
    A(stem_b, Cw); A(conv_b, (nconv + 1) * Cw);  // + one zero row
    A(value_b, VALUE_CH);
    A(wp, (size_t)(64 * Cw / 32) * 12 * 64 * 16); A(bp, NUM_ACTIONS);
    A(wv, (size_t)(VALUE_K / 32) * 8 * 64 * 16); A(bv1, 128); A(wv2, 256); A(bv2, 2);
    A(p_act, (size_t)((B + 15) / 16) * 16 * 64 * Cw); A(v_act, (size_t)((B + 15) / 16) * 16 * VALUE_K);
    A(part_p, (size_t)POLICY_CHUNKS * B * NUM_ACTIONS); A(part_v, (size_t)VALUE_CHUNKS * B * 128);
    A(bits_mine, B); A(bits_theirs, B); A(bad, 1);
    if (getenv("AZ_TOWER_STAMPS")) A(stamps, (B + 3) / 4 * 10);
#undef A
    if (rc != AZ_OK) { az_net_destroy(n); return rc; }
    *out = n;
    return AZ_OK;
}

namespace {
struct Folded { std::vector<float> scale, bias; };
// eval-mode BatchNorm2d(eps=1e-5): y = (x - mean) / sqrt(var + eps) * gamma + beta
Folded fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, int n) {
    Folded f;
    f.scale.resize(n); f.bias.resize(n);
    for (int i = 0; i < n; ++i) {
        const float s = gamma[i] / std::sqrt(var[i] + 1e-5f);
        f.scale[i] = s;
        f.bias[i] = beta[i] - mean[i] * s;
    }
    return f;
}
// 3x3 conv weight [128][128][3][3] (scaled per output channel) -> [tap][ks][nt][lane][8] bf16
void pack_conv3x3(const float* w, const std::vector<float>& scale, uint16_t* out) {
    for (int tap = 0; tap < 9; ++tap)
        for (int ks = 0; ks < 8; ++ks)
            for (int nt = 0; nt < 4; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = nt * 32 + (lane & 31), c = ks * 16 + 8 * (lane >> 5) + j;
                        const float v = w[((size_t)n * CH + c) * 9 + tap] * scale[n];
                        out[((((size_t)tap * 8 + ks) * 4 + nt) * 64 + lane) * 8 + j] = f32_to_bf16(v);
                    }
}
// the same conv for k_tower16 (v_mfma_f32_16x16x32_bf16): [tap][ks 4][nt 8][lane][8] bf16, where tile nt,
// row rho = lane & 15 is output channel (nt >> 1) * 32 + (rho >> 2) * 8 + (nt & 1) * 4 + (rho & 3)
// (the two tiles 2t, 2t + 1 give a lane 8 consecutive channels) and k = ks * 32 + 8 * (lane >> 4) + j
inline int tower16_channel(int nt, int rho) { return (nt >> 1) * 32 + (rho >> 2) * 8 + (nt & 1) * 4 + (rho & 3); }
void pack_conv3x3_16(const float* w, const std::vector<float>& scale, uint16_t* out) {
    for (int tap = 0; tap < 9; ++tap)
        for (int ks = 0; ks < 4; ++ks)
            for (int nt = 0; nt < 8; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = tower16_channel(nt, lane & 15), c = ks * 32 + 8 * (lane >> 4) + j;
                        const float v = w[((size_t)n * CH + c) * 9 + tap] * scale[n];
                        out[((((size_t)tap * 4 + ks) * 8 + nt) * 64 + lane) * 8 + j] = f32_to_bf16(v);
                    }
}
template <typename T>
int upload(T* dst, const std::vector<T>& src) {
    AZ_HIP(hipMemcpy(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return AZ_OK;
}
}  // namespace

// The state of a network of any supported width -> k_tower_any's / k_head_gemm_any's images.
// Same tensor order as below (state_dict() order of src/model.py:116-142).
static int load_state_any(az_net* n, const float* s, int64_t count) {
    const int nb = n->blocks, C = n->ch, H = n->se_hid, KS = C / 32, NT = C / 16;
    const int64_t per_bn = 4 * C, conv_floats = (int64_t)C * C * 9, policy_k = 64ll * C;
    const int64_t per_block = 2 * (conv_floats + per_bn) + (int64_t)H * C + H + 2ll * C * H + 2 * C;
    const int64_t expect = (int64_t)C * 27 + per_bn + nb * per_block + (conv_floats + per_bn) +
                           (int64_t)NUM_ACTIONS * policy_k + NUM_ACTIONS + ((int64_t)VALUE_CH * C + 4 * VALUE_CH) +
                           128ll * VALUE_K + 128 + 256 + 2;
    AZ_CHECK_ARG(count == expect, "az_net_load_state: got %lld floats, a %d x %d network with %d hidden units has %lld",
                 (long long)count, nb, C, H, (long long)expect);
    AZ_HIP(hipSetDevice(n->device));
    AZ_HIP(hipDeviceSynchronize());   // see az_net_load_state
    const float* q = s;
    auto take = [&](int64_t k) { const float* r = q; q += k; return r; };
    int rc;
    const size_t conv_u16 = (size_t)9 * KS * NT * 512;   // bf16 elements per conv image
    // [tap][ks][nt][lane][8]: output channel nt * 16 + (lane & 15), input channel ks * 32 + 8 * (lane >> 4) + j
    auto pack_conv = [&](const float* w, const std::vector<float>& scale, uint16_t* out) {
        for (int tap = 0; tap < 9; ++tap)
            for (int ks = 0; ks < KS; ++ks)
                for (int nt = 0; nt < NT; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int o = nt * 16 + (lane & 15), c = ks * 32 + 8 * (lane >> 4) + j;
                            out[((((size_t)tap * KS + ks) * NT + nt) * 64 + lane) * 8 + j] =
                                f32_to_bf16(w[((size_t)o * C + c) * 9 + tap] * scale[o]);
                        }
    };
    {   // stem: one k-step of 32 (k = tap * 3 + plane, 27 used)
        const float* w = take((int64_t)C * 27);
        const float *g = take(C), *b = take(C), *m = take(C), *v = take(C);
        const Folded f = fold_bn(g, b, m, v, C);
        std::vector<uint16_t> pk((size_t)NT * 64 * 8, 0);
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int k = 8 * (lane >> 4) + j, o = nt * 16 + (lane & 15);
                    if (k < 27) pk[((size_t)nt * 64 + lane) * 8 + j] = f32_to_bf16(w[((size_t)o * 3 + k % 3) * 9 + k / 3] * f.scale[o]);
                }
        if ((rc = upload((uint16_t*)n->stem_w16, pk))) return rc;
        if ((rc = upload(n->stem_b, f.bias))) return rc;
    }
    std::vector<uint16_t> convs((size_t)(2 * nb + 1) * conv_u16);
    std::vector<float> conv_b((size_t)(2 * nb + 2) * C, 0.f);
    std::vector<float> se;
    for (int blk = 0; blk < nb; ++blk) {
        for (int half = 0; half < 2; ++half) {
            const float* w = take(conv_floats);
            const float *g = take(C), *b = take(C), *m = take(C), *v = take(C);
            const Folded f = fold_bn(g, b, m, v, C);
            const int ci = blk * 2 + half;
            pack_conv(w, f.scale, convs.data() + (size_t)ci * conv_u16);
            memcpy(&conv_b[(size_t)ci * C], f.bias.data(), C * sizeof(float));
        }
        // fc1.weight [H][C], fc1.bias [H], fc2.weight [2C][H], fc2.bias [2C]: the kernel reads them in this order
        const int64_t se_floats = (int64_t)H * C + H + 2ll * C * H + 2 * C;
        const float* blob = take(se_floats);
        se.insert(se.end(), blob, blob + se_floats);
    }
    {   // policy head
        const float* w = take(conv_floats);
        const float *g = take(C), *b = take(C), *m = take(C), *v = take(C);
        const Folded f = fold_bn(g, b, m, v, C);
        pack_conv(w, f.scale, convs.data() + (size_t)(2 * nb) * conv_u16);
        memcpy(&conv_b[(size_t)(2 * nb) * C], f.bias.data(), C * sizeof(float));
        const float* fc = take((int64_t)NUM_ACTIONS * policy_k);
        const float* fb = take(NUM_ACTIONS);
        // k' = px * C + c  <->  reference flatten index c * 64 + px (src/model.py:168)
        std::vector<uint16_t> pk((size_t)(policy_k / 32) * 12 * 64 * 8);
        for (int ks = 0; ks < policy_k / 32; ++ks)
            for (int nt = 0; nt < 12; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int o = nt * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4) + j;
                        const int px = k / C, c = k % C;
                        pk[(((size_t)ks * 12 + nt) * 64 + lane) * 8 + j] = f32_to_bf16(fc[(size_t)o * policy_k + c * 64 + px]);
                    }
        if ((rc = upload((uint16_t*)n->wp, pk))) return rc;
        if ((rc = upload(n->bp, std::vector<float>(fb, fb + NUM_ACTIONS)))) return rc;
    }
    {   // value head
        const float* w = take((int64_t)VALUE_CH * C);
        const float *g = take(VALUE_CH), *b = take(VALUE_CH), *m = take(VALUE_CH), *v = take(VALUE_CH);
        const Folded f = fold_bn(g, b, m, v, VALUE_CH);
        std::vector<uint16_t> pk((size_t)KS * 2 * 64 * 8);   // [ks][nt 2][lane][8], output channel nt * 16 + (lane & 15)
        for (int ks = 0; ks < KS; ++ks)
            for (int nt = 0; nt < 2; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int o = nt * 16 + (lane & 15), c = ks * 32 + 8 * (lane >> 4) + j;
                        pk[(((size_t)ks * 2 + nt) * 64 + lane) * 8 + j] = f32_to_bf16(w[(size_t)o * C + c] * f.scale[o]);
                    }
        if ((rc = upload((uint16_t*)n->value_w16, pk))) return rc;
        if ((rc = upload(n->value_b, f.bias))) return rc;
        const float* fc1 = take(128ll * VALUE_K);
        const float* fb1 = take(128);
        std::vector<uint16_t> pv((size_t)(VALUE_K / 32) * 8 * 64 * 8);
        for (int ks = 0; ks < VALUE_K / 32; ++ks)
            for (int nt = 0; nt < 8; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int o = nt * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4) + j;
                        const int px = k / VALUE_CH, c = k % VALUE_CH;
                        pv[(((size_t)ks * 8 + nt) * 64 + lane) * 8 + j] = f32_to_bf16(fc1[(size_t)o * VALUE_K + c * 64 + px]);
                    }
        if ((rc = upload((uint16_t*)n->wv, pv))) return rc;
        if ((rc = upload(n->bv1, std::vector<float>(fb1, fb1 + 128)))) return rc;
        const float* fc2 = take(256);
        const float* fb2 = take(2);
        if ((rc = upload(n->wv2, std::vector<float>(fc2, fc2 + 256)))) return rc;
        if ((rc = upload(n->bv2, std::vector<float>(fb2, fb2 + 2)))) return rc;
    }
    if (q - s != count) AZ_FAIL(AZ_ERR_ARG, "az_net_load_state: internal layout mismatch");
    if ((rc = upload((uint16_t*)n->conv_w16, convs))) return rc;
    if ((rc = upload(n->conv_b, conv_b))) return rc;
    if ((rc = upload(reinterpret_cast<float*>(n->se_img), se))) return rc;
    n->loaded = true;
    return AZ_OK;
}

extern "C" int az_net_load_state(az_net* n, const float* s, int64_t count) {
    AZ_CHECK_ARG(n && s, "az_net_load_state: null argument");
    if (n->any_width) return load_state_any(n, s, count);
    const int nb = n->blocks;
    const int64_t per_bn = 4 * CH;
    const int64_t per_block = 2 * ((int64_t)CH * CH * 9 + per_bn) + SE_HID * CH + SE_HID + 2 * CH * SE_HID + 2 * CH;
    const int64_t expect = (int64_t)CH * 3 * 9 + per_bn + nb * per_block + ((int64_t)CH * CH * 9 + per_bn) +
                           (int64_t)NUM_ACTIONS * POLICY_K + NUM_ACTIONS + ((int64_t)VALUE_CH * CH + 4 * VALUE_CH) +
                           128ll * VALUE_K + 128 + 256 + 2;
    AZ_CHECK_ARG(count == expect, "az_net_load_state: got %lld floats, a %d x 128 network has %lld", (long long)count,
                 nb, (long long)expect);
    AZ_HIP(hipSetDevice(n->device));
    // the uploads below are blocking copies on the null stream: forwards the caller queued on
    // other streams must have finished reading the old weights first
    AZ_HIP(hipDeviceSynchronize());
    const float* q = s;
    auto take = [&](int64_t k) { const float* r = q; q += k; return r; };
    int rc;
    // stem
    {
        const float* w = take((int64_t)CH * 27);
        const float *g = take(CH), *b = take(CH), *m = take(CH), *v = take(CH);
        const Folded f = fold_bn(g, b, m, v, CH);
        std::vector<uint16_t> pk(2 * 4 * 64 * 8, 0);
        for (int ks = 0; ks < 2; ++ks)
            for (int nt = 0; nt < 4; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int k = ks * 16 + 8 * (lane >> 5) + j, o = nt * 32 + (lane & 31);
                        float val = 0.f;
                        if (k < 27) { const int tap = k / 3, pl = k % 3; val = w[((size_t)o * 3 + pl) * 9 + tap] * f.scale[o]; }
                        pk[(((size_t)ks * 4 + nt) * 64 + lane) * 8 + j] = f32_to_bf16(val);
                    }
        if ((rc = upload((uint16_t*)n->stem_w, pk))) return rc;
        std::vector<uint16_t> pk16(8 * 64 * 8, 0);   // one k-step of 32 (27 used)
        for (int nt = 0; nt < 8; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int k = 8 * (lane >> 4) + j, o = tower16_channel(nt, lane & 15);
                    float val = 0.f;
                    if (k < 27) { const int tap = k / 3, pl = k % 3; val = w[((size_t)o * 3 + pl) * 9 + tap] * f.scale[o]; }
                    pk16[((size_t)nt * 64 + lane) * 8 + j] = f32_to_bf16(val);
                }
        if ((rc = upload((uint16_t*)n->stem_w16, pk16))) return rc;
        if ((rc = upload(n->stem_b, f.bias))) return rc;
    }
    // tower
    std::vector<uint16_t> convs((size_t)(2 * nb + 1) * CONV_BYTES / 2), convs16(convs.size());
    std::vector<float> conv_b((size_t)(2 * nb + 2) * CH, 0.f);
    std::vector<uint8_t> se_img((size_t)nb * SE_BYTES, 0);
    for (int blk = 0; blk < nb; ++blk) {
        for (int half = 0; half < 2; ++half) {
            const float* w = take((int64_t)CH * CH * 9);
            const float *g = take(CH), *b = take(CH), *m = take(CH), *v = take(CH);
            const Folded f = fold_bn(g, b, m, v, CH);
            const int ci = blk * 2 + half;
            pack_conv3x3(w, f.scale, convs.data() + (size_t)ci * CONV_BYTES / 2);
            pack_conv3x3_16(w, f.scale, convs16.data() + (size_t)ci * CONV_BYTES / 2);
            memcpy(&conv_b[(size_t)ci * CH], f.bias.data(), CH * sizeof(float));
        }
        {   // squeeze-excitation: fc1.weight [16][128], fc1.bias [16], fc2.weight [256][16], fc2.bias [256]
            float* img = reinterpret_cast<float*>(se_img.data() + (size_t)blk * SE_BYTES);
            const float* fw1 = take(SE_HID * CH);
            const float* fb1 = take(SE_HID);
            const float* fw2 = take(2 * CH * SE_HID);
            const float* fb2 = take(2 * CH);
            for (int j = 0; j < SE_HID; ++j)
                for (int c = 0; c < CH; ++c) img[j * SE_W1_PITCH + c] = fw1[j * CH + c];
            for (int j = 0; j < SE_HID; ++j) img[SE_OFF_B1 + j] = fb1[j];
            for (int o = 0; o < 2 * CH; ++o)
                for (int j = 0; j < SE_HID; ++j) img[SE_OFF_W2 + j * 256 + o] = fw2[o * SE_HID + j];
            for (int o = 0; o < 2 * CH; ++o) img[SE_OFF_B2 + o] = fb2[o];
        }
    }
    // policy head
    {
        const float* w = take((int64_t)CH * CH * 9);
        const float *g = take(CH), *b = take(CH), *m = take(CH), *v = take(CH);
        const Folded f = fold_bn(g, b, m, v, CH);
        pack_conv3x3(w, f.scale, convs.data() + (size_t)(2 * nb) * CONV_BYTES / 2);
        pack_conv3x3_16(w, f.scale, convs16.data() + (size_t)(2 * nb) * CONV_BYTES / 2);
        memcpy(&conv_b[(size_t)(2 * nb) * CH], f.bias.data(), CH * sizeof(float));
        const float* fc = take((int64_t)NUM_ACTIONS * POLICY_K);
        const float* fb = take(NUM_ACTIONS);
        // [kstep][ntile 12][lane][8]: n = nt*16 + (lane&15), k' = kstep*32 + 8*(lane>>4) + j,
        // k' = px*128 + c  <->  reference flatten index c*64 + px (src/model.py:168)
        std::vector<uint16_t> pk((size_t)(POLICY_K / 32) * 12 * 64 * 8);
        for (int ks = 0; ks < POLICY_K / 32; ++ks)
            for (int nt = 0; nt < 12; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int o = nt * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4) + j;
                        const int px = k / CH, c = k % CH;
                        pk[(((size_t)ks * 12 + nt) * 64 + lane) * 8 + j] = f32_to_bf16(fc[(size_t)o * POLICY_K + c * 64 + px]);
                    }
        if ((rc = upload((uint16_t*)n->wp, pk))) return rc;
        if ((rc = upload(n->bp, std::vector<float>(fb, fb + NUM_ACTIONS)))) return rc;
    }
    // value head
    {
        const float* w = take((int64_t)VALUE_CH * CH);
        const float *g = take(VALUE_CH), *b = take(VALUE_CH), *m = take(VALUE_CH), *v = take(VALUE_CH);
        const Folded f = fold_bn(g, b, m, v, VALUE_CH);
        std::vector<uint16_t> pk(8 * 64 * 8);
        for (int ks = 0; ks < 8; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int o = lane & 31, c = ks * 16 + 8 * (lane >> 5) + j;
                    pk[((size_t)ks * 64 + lane) * 8 + j] = f32_to_bf16(w[(size_t)o * CH + c] * f.scale[o]);
                }
        if ((rc = upload((uint16_t*)n->value_w, pk))) return rc;
        std::vector<uint16_t> pk16(4 * 2 * 64 * 8);   // [ks 4][nt 2][lane][8], output channel nt * 16 + (lane & 15)
        for (int ks = 0; ks < 4; ++ks)
            for (int nt = 0; nt < 2; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int o = nt * 16 + (lane & 15), c = ks * 32 + 8 * (lane >> 4) + j;
                        pk16[(((size_t)ks * 2 + nt) * 64 + lane) * 8 + j] = f32_to_bf16(w[(size_t)o * CH + c] * f.scale[o]);
                    }
        if ((rc = upload((uint16_t*)n->value_w16, pk16))) return rc;
        if ((rc = upload(n->value_b, f.bias))) return rc;
        const float* fc1 = take(128ll * VALUE_K);
        const float* fb1 = take(128);
        std::vector<uint16_t> pv((size_t)(VALUE_K / 32) * 8 * 64 * 8);
        for (int ks = 0; ks < VALUE_K / 32; ++ks)
            for (int nt = 0; nt < 8; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int o = nt * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4) + j;
                        const int px = k / VALUE_CH, c = k % VALUE_CH;
                        pv[(((size_t)ks * 8 + nt) * 64 + lane) * 8 + j] = f32_to_bf16(fc1[(size_t)o * VALUE_K + c * 64 + px]);
                    }
        if ((rc = upload((uint16_t*)n->wv, pv))) return rc;
        if ((rc = upload(n->bv1, std::vector<float>(fb1, fb1 + 128)))) return rc;
        const float* fc2 = take(256);
        const float* fb2 = take(2);
        if ((rc = upload(n->wv2, std::vector<float>(fc2, fc2 + 256)))) return rc;
        if ((rc = upload(n->bv2, std::vector<float>(fb2, fb2 + 2)))) return rc;
    }
    if (q - s != count) AZ_FAIL(AZ_ERR_ARG, "az_net_load_state: internal layout mismatch");
    if ((rc = upload((uint16_t*)n->conv_w, convs))) return rc;
    if ((rc = upload((uint16_t*)n->conv_w16, convs16))) return rc;
    if ((rc = upload(n->conv_b, conv_b))) return rc;
    if ((rc = upload(n->se_img, se_img))) return rc;
    n->loaded = true;
    return AZ_OK;
}

// tower + head GEMMs: leaves the split-K partial sums in hb
static int forward_partial(az_net* n, int32_t B, const uint64_t* mine_dev, const uint64_t* theirs_dev,
                           const RowAssign* rows, const HeadBuffers& hb, hipStream_t s) {
    AZ_CHECK_ARG(n && n->loaded, "az_net_forward: network has no weights (call az_net_load_state)");
    if (B == 0) return AZ_OK;
    AZ_CHECK_ARG(mine_dev && theirs_dev, "az_net_forward: null pointer");
    // kernels run on the calling thread's current device; the weights and head buffers (the
    // static ones included) live on the network's: launching across the two would fault
    int current = -1;
    AZ_HIP(hipGetDevice(&current));
    if (current != n->device)
        AZ_FAIL(AZ_ERR_STATE, "az_net_forward: the network lives on device %d but device %d is current", n->device, current);
    TowerParams tp;
    tp.mine = mine_dev; tp.theirs = theirs_dev; tp.B = B; tp.blocks = n->blocks;
    tp.need = nullptr; tp.n_games = 0; tp.row_of = nullptr; tp.n_rows = nullptr; tp.rows_total = nullptr;
    if (rows) {
        AZ_CHECK_ARG(rows->need && rows->row_of && rows->n_rows && rows->n_games >= B,
                     "az_net_forward: incomplete row assignment");
        tp.need = rows->need; tp.n_games = rows->n_games; tp.row_of = rows->row_of; tp.n_rows = rows->n_rows;
        tp.rows_total = rows->rows_total;
    }
    tp.stem_w = n->stem_w; tp.stem_b = n->stem_b; tp.conv_w = n->conv_w; tp.conv_b = n->conv_b;
    tp.se_img = n->se_img;
    tp.value_w = n->value_w; tp.value_b = n->value_b; tp.p_act = hb.p_act; tp.v_act = hb.v_act;
    static const int dbg = getenv("AZ_TOWER_DBG") ? atoi(getenv("AZ_TOWER_DBG")) : 0;
    tp.dbg = dbg;
    tp.stamps = B <= n->max_batch ? n->stamps : nullptr;   // the stamp buffer holds max_batch positions' workgroups (the static-buffer entry point takes larger batches)
    tp.ch = n->ch; tp.se_hid = n->se_hid;
    const bool stamp = n->timing && (n->launches++ % TIMING_STRIDE == 0) && n->ev_used < TIMING_POOL;
    if (stamp) AZ_HIP(hipEventRecord(n->ev0[n->ev_used], s));
    if (n->any_width) {
        tp.stem_w = n->stem_w16; tp.conv_w = n->conv_w16; tp.value_w = n->value_w16; tp.stamps = nullptr;
        launch_tower_any(tp, B, s);
        if (stamp) AZ_HIP(hipEventRecord(n->ev1[n->ev_used++], s));
        HeadParams hp{hb.p_act, hb.v_act, n->wp, n->wv, hb.part_p, hb.part_v, B, tp.n_rows};
        hipLaunchKernelGGL(k_head_gemm_any, dim3((B + 15) / 16, POLICY_CHUNKS + VALUE_CHUNKS), dim3(256), 0, s, hp, 64 * n->ch);
        AZ_HIP(hipGetLastError());
        return AZ_OK;
    }
    // form of the tower: whole positions per wave from 513 positions up (k_tower16; 4 = k_tower, the
    // same tiling on the 32x32x16 MFMA shape), channel-split waves
    // below (az_net_set_tower_form or AZ_TOWER_FORM force one; with stamps on the default is k_tower16 at every batch size: the whole-position kernels are the ones that record stamps)
    static const int env_form = getenv("AZ_TOWER_FORM") ? atoi(getenv("AZ_TOWER_FORM")) : 0;
    const int forced = n->tower_form ? n->tower_form : env_form;
    int form = forced ? forced : (B <= 256 ? 1 : (B <= 512 ? 2 : 16));
    if (tp.stamps && !forced) form = 16;
    if (dbg && !forced) form = 4;
    if (form == 1) hipLaunchKernelGGL((k_tower_split<1, 4>), dim3(B), dim3(256), 0, s, tp);
    else if (form == 2) hipLaunchKernelGGL((k_tower_split<2, 4>), dim3((B + 1) / 2), dim3(256), 0, s, tp);
    else if (form == 8) hipLaunchKernelGGL((k_tower_split<2, 8>), dim3((B + 3) / 4), dim3(512), 0, s, tp);
    else if (form == 16) {
        tp.stem_w = n->stem_w16; tp.conv_w = n->conv_w16; tp.value_w = n->value_w16;
        launch_tower16(tp, B, s);
    } else hipLaunchKernelGGL(k_tower, dim3((B + 3) / 4), dim3(256), 0, s, tp);
    if (stamp) AZ_HIP(hipEventRecord(n->ev1[n->ev_used++], s));
    HeadParams hp{hb.p_act, hb.v_act, n->wp, n->wv, hb.part_p, hb.part_v, B, tp.n_rows};
    hipLaunchKernelGGL(k_head_gemm, dim3((B + 16 * HEAD_MT - 1) / (16 * HEAD_MT), POLICY_CHUNKS + VALUE_CHUNKS), dim3(256), 0, s, hp);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

static FinishParams finish_params(az_net* n, int32_t B, const HeadBuffers& hb, const int32_t* n_rows) {
    return FinishParams{hb.part_p, hb.part_v, n->bp, n->bv1, n->wv2, n->bv2, nullptr, nullptr, nullptr, nullptr, B, n_rows};
}

static int forward_full(az_net* n, int32_t B, const uint64_t* mine_dev, const uint64_t* theirs_dev,
                        const RowAssign* rows, const HeadBuffers& hb, float* probs_dev, float* value_dev,
                        float* policy_logits_dev, float* wl_logits_dev, hipStream_t s) {
    const int rc = forward_partial(n, B, mine_dev, theirs_dev, rows, hb, s);
    if (rc != AZ_OK || B == 0) return rc;
    AZ_CHECK_ARG(probs_dev && value_dev, "az_net_forward: null pointer");
    FinishParams fp = finish_params(n, B, hb, rows ? rows->n_rows : nullptr);
    fp.probs = probs_dev; fp.value = value_dev; fp.policy_logits = policy_logits_dev; fp.wl_logits = wl_logits_dev;
    hipLaunchKernelGGL(k_head_finish, dim3((B + 3) / 4), dim3(256), 0, s, fp);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

int az::net_forward_partial(az_net* n, int32_t B, const uint64_t* mine_dev, const uint64_t* theirs_dev,
                            const RowAssign* rows, hipStream_t s) {
    AZ_CHECK_ARG(n, "az_net_forward: null network");
    AZ_CHECK_ARG(B >= 0 && B <= n->max_batch, "az_net_forward: batch %d exceeds max_batch %d", B, n->max_batch);
    return forward_partial(n, B, mine_dev, theirs_dev, rows, n->own(), s);
}

int az::net_device(const az_net* n) { return n->device; }

az::FinishParams az::net_finish_params(az_net* n, int32_t B, const int32_t* n_rows) {
    return finish_params(n, B, n->own(), n_rows);
}

int az::net_forward_full(az_net* n, int32_t B, const uint64_t* mine_dev, const uint64_t* theirs_dev,
                         const RowAssign* rows, float* probs_dev, float* value_dev, float* policy_logits_dev,
                         float* wl_logits_dev, hipStream_t s) {
    AZ_CHECK_ARG(n, "az_net_forward: null network");
    AZ_CHECK_ARG(B >= 0 && B <= n->max_batch, "az_net_forward: batch %d exceeds max_batch %d", B, n->max_batch);
    return forward_full(n, B, mine_dev, theirs_dev, rows, n->own(), probs_dev, value_dev, policy_logits_dev,
                        wl_logits_dev, s);
}

extern "C" int az_net_forward(az_net* n, int32_t B, const uint64_t* mine_dev, const uint64_t* theirs_dev,
                              float* probs_dev, float* value_dev, float* policy_logits_dev,
                              float* wl_logits_dev, void* stream) {
    return az::net_forward_full(n, B, mine_dev, theirs_dev, nullptr, probs_dev, value_dev, policy_logits_dev,
                                wl_logits_dev, (hipStream_t)stream);
}

// az_net_forward for callers that may not allocate.  The head activations and the split-K
// partial sums of this call live in the library's static device array, so the batch is
// bounded by AZ_NET_NOALLOC_MAX_BATCH whatever max_batch the network was created with.
// Nothing in that array has to start at zero: every element a kernel reads was written by
// the tower or the head GEMM of the same call (rows past B in the last 16-position tile
// feed only output rows that are never stored).
extern "C" int az_net_forward_noalloc(az_net* n, int32_t B, const uint64_t* mine_dev, const uint64_t* theirs_dev,
                                      float* probs_dev, float* value_dev, float* policy_logits_dev,
                                      float* wl_logits_dev, void* stream) {
    AZ_CHECK_ARG(n, "az_net_forward_noalloc: null network");
    AZ_CHECK_ARG(B >= 0, "az_net_forward_noalloc: negative batch %d", B);
    AZ_CHECK_ARG(B <= AZ_NET_NOALLOC_MAX_BATCH,
                 "az_net_forward_noalloc: batch %d exceeds the static buffers' %d positions (use az_net_forward)", B,
                 AZ_NET_NOALLOC_MAX_BATCH);
    AZ_CHECK_ARG(n->ch <= CH, "az_net_forward_noalloc: the static buffers hold the head activations of networks up to %d filters (this one has %d; use az_net_forward)", CH, n->ch);
    unsigned char* base = (n->device >= 0 && n->device < MAX_DEVICES) ? g_scratch_ptr[n->device] : nullptr;
    if (!base) AZ_FAIL(AZ_ERR_STATE, "az_net_forward_noalloc: the static buffers of device %d were never located", n->device);
    HeadBuffers hb;
    hb.p_act = reinterpret_cast<__bf16*>(base);
    hb.v_act = reinterpret_cast<__bf16*>(base + NOALLOC_OFF_VACT);
    hb.part_p = reinterpret_cast<float*>(base + NOALLOC_OFF_PARTP);
    hb.part_v = reinterpret_cast<float*>(base + NOALLOC_OFF_PARTV);
    return forward_full(n, B, mine_dev, theirs_dev, nullptr, hb, probs_dev, value_dev, policy_logits_dev, wl_logits_dev,
                        (hipStream_t)stream);
}

extern "C" int az_net_forward_planes(az_net* n, int32_t B, const float* planes_dev, float* probs_dev,
                                     float* value_dev, float* policy_logits_dev, float* wl_logits_dev,
                                     void* stream) {
    AZ_CHECK_ARG(n && n->loaded, "az_net_forward_planes: network has no weights");
    AZ_CHECK_ARG(B >= 0 && B <= n->max_batch, "az_net_forward_planes: batch %d exceeds max_batch %d", B, n->max_batch);
    if (B == 0) return AZ_OK;
    AZ_CHECK_ARG(planes_dev, "az_net_forward_planes: null pointer");
    hipStream_t s = (hipStream_t)stream;
    AZ_HIP(hipMemsetAsync(n->bad, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_planes_to_bits, dim3((B + 3) / 4), dim3(256), 0, s, planes_dev, B, n->bits_mine,
                       n->bits_theirs, n->bad);
    const int rc = az_net_forward(n, B, n->bits_mine, n->bits_theirs, probs_dev, value_dev, policy_logits_dev,
                                  wl_logits_dev, stream);
    if (rc != AZ_OK) return rc;
    int32_t bad = 0;
    AZ_HIP(hipMemcpyAsync(&bad, n->bad, sizeof bad, hipMemcpyDeviceToHost, s));
    AZ_HIP(hipStreamSynchronize(s));
    if (bad) AZ_FAIL(AZ_ERR_ARG, "az_net_forward_planes: input is not an encoded Breakthrough state "
                                 "(planes 0/1 must hold 0 or 1, plane 2 must be all ones)");
    return AZ_OK;
}
