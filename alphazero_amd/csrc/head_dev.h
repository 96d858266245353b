// Final stage of the network heads, shared by k_head_finish (net.hip) and the fused
// per-simulation tree kernel (selfplay.hip): split-K partial sums + bias -> policy
// softmax; value FC2 + softmax -> P(win) - P(loss)  (src/mcts.py:252-256).
// Floating-point contraction is pinned off inside so that both translation units
// produce bit-identical probabilities whatever their command-line flags are.
#pragma once
#include "az_common.h"

namespace az {

constexpr int HEAD_KCHUNK = 1024;
constexpr int POLICY_CHUNKS = 64 * 128 / HEAD_KCHUNK;  // 8
constexpr int VALUE_CHUNKS = 64 * 32 / HEAD_KCHUNK;    // 2

struct FinishParams {
    const float* part_p; const float* part_v;
    const float* bp;   // [192]
    const float* bv1;  // [128]
    const float* wv2;  // [2][128]
    const float* bv2;  // [2]
    float* probs; float* value; float* policy_logits; float* wl_logits;
    int B;                   // rows the partial sums are strided by
    const int32_t* n_rows;   // rows in use when the tower assigned them on the device (else NULL: B)
};

__device__ __forceinline__ float head_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float head_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// The global-memory operands of one position's head finish, held in registers so that a
// caller can issue the loads long before it needs the result.
struct HeadOperands {
    float bp[3], part_p[3][POLICY_CHUNKS];
    float bv1[2], part_v[2][VALUE_CHUNKS], wv2[2][2];
    float bv2[2];
};

__device__ __forceinline__ HeadOperands head_load(const FinishParams& p, int pos, int lane) {
    HeadOperands h;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int n = lane + 64 * i;
        h.bp[i] = p.bp[n];
#pragma unroll
        for (int c = 0; c < POLICY_CHUNKS; ++c) h.part_p[i][c] = p.part_p[((size_t)c * p.B + pos) * NUM_ACTIONS + n];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = lane + 64 * i;
        h.bv1[i] = p.bv1[n];
#pragma unroll
        for (int c = 0; c < VALUE_CHUNKS; ++c) h.part_v[i][c] = p.part_v[((size_t)c * p.B + pos) * 128 + n];
        h.wv2[i][0] = p.wv2[n];
        h.wv2[i][1] = p.wv2[128 + n];
    }
    h.bv2[0] = p.bv2[0];
    h.bv2[1] = p.bv2[1];
    return h;
}

// One wave per position.  probs_out[i] = probability of action lane + 64*i; returns
// the value (identical in every lane).
__device__ __forceinline__ float head_compute(const FinishParams& p, const HeadOperands& h, int pos, int lane,
                                              float (&probs_out)[3]) {
#pragma clang fp contract(off)
    float lg[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int n = lane + 64 * i;
        float s = h.bp[i];
#pragma unroll
        for (int c = 0; c < POLICY_CHUNKS; ++c) s += h.part_p[i][c];
        lg[i] = s;
        if (p.policy_logits) p.policy_logits[(size_t)pos * NUM_ACTIONS + n] = s;
    }
    const float mx = head_wave_max(fmaxf(lg[0], fmaxf(lg[1], lg[2])));
    float e[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) e[i] = __expf(lg[i] - mx);
    const float inv = 1.0f / head_wave_sum(e[0] + e[1] + e[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) probs_out[i] = e[i] * inv;
    // value head: hidden = relu(fc1), wl = fc2(hidden)
    float w0 = 0.f, w1 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float s = h.bv1[i];
#pragma unroll
        for (int c = 0; c < VALUE_CHUNKS; ++c) s += h.part_v[i][c];
        s = fmaxf(s, 0.f);
        w0 += s * h.wv2[i][0];
        w1 += s * h.wv2[i][1];
    }
    w0 = head_wave_sum(w0) + h.bv2[0];
    w1 = head_wave_sum(w1) + h.bv2[1];
    if (lane == 0 && p.wl_logits) { p.wl_logits[2 * pos] = w0; p.wl_logits[2 * pos + 1] = w1; }
    const float m = fmaxf(w0, w1);
    const float e0 = __expf(w0 - m), e1 = __expf(w1 - m);
    const float is = 1.0f / (e0 + e1);
    return e0 * is - e1 * is;   // P(win) - P(loss), src/mcts.py:256
}

__device__ inline float dev_head_finish(const FinishParams& p, int pos, int lane, float (&probs_out)[3]) {
    const HeadOperands h = head_load(p, pos, lane);
    return head_compute(p, h, pos, lane, probs_out);
}

}  // namespace az
