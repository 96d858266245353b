// k_tower_any: the residual tower for the widths the tuned kernels do not cover (num_filters a
// multiple of 32 up to 256, any squeeze-excitation ratio that divides it).  Reference semantics:
// AlphaZeroNet.forward (src/model.py:144-180), SEResidualBlock (:75-92), SEBlock (:37-53); the
// reference's own tests build 2 x 32 networks (tests/test_mcts.py:54, tests/test_model.py:64).
//
// One workgroup of four waves per position; wave w owns the 16 pixels of board rows 2w, 2w + 1 and
// all C output channels of them (C / 16 accumulator tiles of v_mfma_f32_16x16x32_bf16).  The
// activations of the position sit in two LDS images [64 px][C] bf16 (pixel pitch 2C + 16 bytes);
// a conv reads one and writes the other, the block's output goes back over the block's input (the
// residual, which each lane reads at exactly the addresses it then writes).  Weights are read as
// MFMA fragments straight from global memory (L2), a few k-steps ahead of the MFMAs that use them
// (a register ring): with one position per workgroup nothing is reused inside it, which is why the 128-wide kernels (four positions per workgroup, weights
// staged through LDS) are the ones to use at that width.  Same arithmetic as those kernels:
// bf16 operands, float32 accumulation started from the folded-BN bias, float32 squeeze-excitation.
#include "tower_common.h"

namespace az {

// fragments of a conv's weight stream kept in flight ahead of the MFMAs: D k-steps of NT fragments
// (at most 36 fragments = 144 registers)
template <int NT>
struct AnyDepth { static constexpr int D = NT <= 2 ? 9 : (NT <= 4 ? 6 : (NT <= 6 ? 5 : (NT <= 8 ? 4 : (NT <= 12 ? 3 : 2)))); };

__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true));  // row_ror:8
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x124, 0xF, 0xF, true));  // row_ror:4
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x122, 0xF, 0xF, true));  // row_ror:2
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x121, 0xF, 0xF, true));  // row_ror:1
    return v;
}

// NT = C / 16 channel tiles (C = 32, 64, ..., 256)
template <int NT>
__global__ __launch_bounds__(256) void k_tower_any(TowerParams p) {
    constexpr int C = NT * 16, KS = NT / 2, P = 2 * C + 16, T = 9 * KS, D = AnyDepth<NT>::D;
    __shared__ __attribute__((aligned(16))) char act[2 * 64 * P];
    __shared__ __attribute__((aligned(16))) char zero_row[P];
    __shared__ float part[4][C];
    __shared__ float mean[C];
    __shared__ float hid[C];
    __shared__ __attribute__((aligned(16))) float gate[2 * C];   // gate[0, C), shift[C, 2C)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, q = lane >> 4;   // pixel of the wave's tile / k quarter (operands), channel quad (accumulators)
    const int H = p.se_hid;
    const int pos = blockIdx.x;
    int src = pos;
    if (p.need) {
        bool has_rows;
        src = assign_rows<256, 1>(p, tid, 0, has_rows);
        if (!has_rows) return;
    }
    for (int i = tid; i < P / 4; i += 256) reinterpret_cast<uint32_t*>(zero_row)[i] = 0;

    const int px = wave * 16 + c, py = px >> 3, pxx = px & 7;
    char* regA = act;
    char* regB = act + 64 * P;
    f32x4 acc[NT];

    // acc -> ReLU -> bf16 -> this lane's pixel in `out` (8 bytes per channel tile)
    auto store_relu = [&](char* out) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const uint2 o = make_uint2(relu_bf16x2(pack_bf16x2(acc[nt][0], acc[nt][1])),
                                       relu_bf16x2(pack_bf16x2(acc[nt][2], acc[nt][3])));
            *reinterpret_cast<uint2*>(out + px * P + (nt * 16 + q * 4) * 2) = o;
        }
    };
    // 3x3 conv number ci over the image `in`: acc = bias + sum over taps and k-steps.  The conv's
    // weights are one contiguous stream of T x NT fragments; D k-steps of them are in flight.
    auto conv3x3 = [&](int ci, const char* in) {
        const float* bias = p.conv_b + (size_t)ci * C + q * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = *reinterpret_cast<const f32x4*>(bias + nt * 16);
        const uint8_t* wconv = p.conv_w + (size_t)ci * T * NT * 1024 + lane * 16;
        bf16x8 ring[D][NT];
#pragma unroll
        for (int d = 0; d < D && d < T; ++d)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ring[d][nt] = *reinterpret_cast<const bf16x8*>(wconv + (size_t)(d * NT + nt) * 1024);
        const char* xrow = nullptr;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int tap = t / KS, ks = t % KS;
            if (ks == 0) {
                const int y = py + tap / 3 - 1, x = pxx + tap % 3 - 1;
                const bool ok = (y >= 0) && (y < 8) && (x >= 0) && (x < 8);
                xrow = (ok ? in + (y * 8 + x) * P : zero_row) + q * 16;
            }
            const bf16x8 xb = *reinterpret_cast<const bf16x8*>(xrow + ks * 64);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[t % D][nt], xb, acc[nt], 0, 0, 0);
            if (t + D < T) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    ring[t % D][nt] = *reinterpret_cast<const bf16x8*>(wconv + (size_t)((t + D) * NT + nt) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);   // or the scheduler sinks every load to just before its use
        }
    };

    // ---- stem: 3x3 conv over the 3 input planes, K = 27 padded to one k-step of 32
    {
        const uint64_t mine = p.mine[src], theirs = p.theirs[src];
        bf16x8 xb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = q * 8 + j;   // k = tap * 3 + plane
            const int tap = k / 3, pl = k - tap * 3;
            const int y = py + tap / 3 - 1, x = pxx + tap % 3 - 1;
            const bool ok = (k < 27) && (y >= 0) && (y < 8) && (x >= 0) && (x < 8);
            const int sq = (y * 8 + x) & 63;
            const uint64_t plane = (pl == 0) ? mine : ((pl == 1) ? theirs : ~0ull);
            xb[j] = (__bf16)((ok && ((plane >> sq) & 1ull)) ? 1.0f : 0.0f);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bf16x8 w = *reinterpret_cast<const bf16x8*>(p.stem_w + (nt * 64 + lane) * 16);
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.stem_b + nt * 16 + q * 4);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xb, b, 0, 0, 0);
        }
        store_relu(regA);
    }
    __syncthreads();   // image A and the zero row are complete

    const size_t se_stride = (size_t)H * C + H + (size_t)2 * C * H + 2 * C;   // floats per block: W1, b1, W2, b2
    for (int blk = 0; blk < p.blocks; ++blk) {
        conv3x3(2 * blk, regA);
        store_relu(regB);
        __syncthreads();
        conv3x3(2 * blk + 1, regB);
        // squeeze: channel sums over this wave's 16 pixels, then over the four waves
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s = row16_sum(acc[nt][i]);
                if (c == 0) part[wave][nt * 16 + q * 4 + i] = s;
            }
        __syncthreads();
        if (tid < C) mean[tid] = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) * (1.0f / 64.0f);
        __syncthreads();
        const float* se = reinterpret_cast<const float*>(p.se_img) + (size_t)blk * se_stride;
        if (tid < H) {   // fc1 + ReLU
            const float* w1 = se + (size_t)tid * C;
            float y = se[(size_t)H * C + tid];
            for (int k = 0; k < C; ++k) y += w1[k] * mean[k];
            hid[tid] = fmaxf(y, 0.f);
        }
        __syncthreads();
        {   // fc2: outputs [0, C) are the gate (sigmoid), [C, 2C) the shift
            const float* w2 = se + (size_t)H * C + H;
            const float* b2 = w2 + (size_t)2 * C * H;
            for (int o = tid; o < 2 * C; o += 256) {
                float s = b2[o];
                for (int j = 0; j < H; ++j) s += w2[(size_t)o * H + j] * hid[j];
                gate[o] = o < C ? __builtin_amdgcn_rcpf(1.0f + __expf(-s)) : s;
            }
        }
        __syncthreads();
        // out = relu(z * gate + shift + residual), over the block's input
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ch0 = nt * 16 + q * 4;
            uint2* cell = reinterpret_cast<uint2*>(regA + px * P + ch0 * 2);
            const uint2 r = *cell;
            const f32x4 g = *reinterpret_cast<const f32x4*>(gate + ch0);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(gate + C + ch0);
            const float o0 = acc[nt][0] * g[0] + sh[0] + bf16_lo(r.x);
            const float o1 = acc[nt][1] * g[1] + sh[1] + bf16_hi(r.x);
            const float o2 = acc[nt][2] * g[2] + sh[2] + bf16_lo(r.y);
            const float o3 = acc[nt][3] * g[3] + sh[3] + bf16_hi(r.y);
            *cell = make_uint2(relu_bf16x2(pack_bf16x2(o0, o1)), relu_bf16x2(pack_bf16x2(o2, o3)));
        }
        __syncthreads();
    }

    // ---- value head conv 1x1 (C -> 32) on the tower output, + bias, ReLU -> v_act (k = px * 32 + ch)
    {
        f32x4 va[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) va[nt] = *reinterpret_cast<const f32x4*>(p.value_b + nt * 16 + q * 4);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 xb = *reinterpret_cast<const bf16x8*>(regA + px * P + ks * 64 + q * 16);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const bf16x8 w = *reinterpret_cast<const bf16x8*>(p.value_w + ((ks * 2 + nt) * 64 + lane) * 16);
                va[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xb, va[nt], 0, 0, 0);
            }
        }
        char* out = reinterpret_cast<char*>(p.v_act) + (size_t)(pos >> 4) * (VALUE_K / 32) * 1024 + (pos & 15) * 64 + px * 1024;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            *reinterpret_cast<uint2*>(out + (nt * 16 + q * 4) * 2) =
                make_uint2(relu_bf16x2(pack_bf16x2(va[nt][0], va[nt][1])), relu_bf16x2(pack_bf16x2(va[nt][2], va[nt][3])));
    }
    // ---- policy conv: ReLU -> image B -> p_act (k = px * C + ch, tiled for the head GEMM)
    conv3x3(2 * p.blocks, regA);
    store_relu(regB);
    __syncthreads();
    {
        const int per_px = C >> 3;   // 16-byte granules per pixel
        char* out = reinterpret_cast<char*>(p.p_act) + (size_t)(pos >> 4) * (2 * C) * 1024 + (pos & 15) * 64;   // K / 32 = 2 C
        for (int g = tid; g < 64 * per_px; g += 256) {
            const int gp = g / per_px, cg = g - gp * per_px;
            const int k = gp * C + cg * 8;
            const uint4 v = *reinterpret_cast<const uint4*>(regB + gp * P + cg * 16);
            *reinterpret_cast<uint4*>(out + (size_t)(k >> 5) * 1024 + (k & 31) * 2) = v;
        }
    }
}

void launch_tower_any(const TowerParams& tp, int B, hipStream_t s) {
    switch (tp.ch / 32) {   // az_net_create admits exactly these widths
        case 1: hipLaunchKernelGGL(k_tower_any<2>, dim3(B), dim3(256), 0, s, tp); break;
        case 2: hipLaunchKernelGGL(k_tower_any<4>, dim3(B), dim3(256), 0, s, tp); break;
        case 3: hipLaunchKernelGGL(k_tower_any<6>, dim3(B), dim3(256), 0, s, tp); break;
        case 4: hipLaunchKernelGGL(k_tower_any<8>, dim3(B), dim3(256), 0, s, tp); break;
        case 5: hipLaunchKernelGGL(k_tower_any<10>, dim3(B), dim3(256), 0, s, tp); break;
        case 6: hipLaunchKernelGGL(k_tower_any<12>, dim3(B), dim3(256), 0, s, tp); break;
        case 7: hipLaunchKernelGGL(k_tower_any<14>, dim3(B), dim3(256), 0, s, tp); break;
        default: hipLaunchKernelGGL(k_tower_any<16>, dim3(B), dim3(256), 0, s, tp); break;
    }
}

}  // namespace az
