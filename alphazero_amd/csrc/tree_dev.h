// Device-side building blocks of the tree search, shared by the step kernels
// (tree.hip) and the fused per-simulation kernel (selfplay.hip).
//
// Every translation unit that includes this must be compiled with
// -ffp-contract=off: q + u and the prior arithmetic are separately rounded
// float32 operations in the reference (src/mcts.py:299-305, 264-279).
#pragma once
#include "az_engine.h"

namespace az {

// planes [3][8][8] float32 of a canonical position: 48 lanes x 16 B
__device__ inline void write_planes(float* dst, uint64_t mine, uint64_t theirs, int lane) {
    if (lane < 48) {
        const int p = lane >> 4;
        const int s0 = (4 * lane) & 63;
        const uint64_t src = (p == 0) ? mine : ((p == 1) ? theirs : ~0ull);
        float4 v;
        v.x = (float)((src >> (s0 + 0)) & 1ull);
        v.y = (float)((src >> (s0 + 1)) & 1ull);
        v.z = (float)((src >> (s0 + 2)) & 1ull);
        v.w = (float)((src >> (s0 + 3)) & 1ull);
        reinterpret_cast<float4*>(dst)[lane] = v;
    }
}

// game g's live arena (two arenas per game; re-rooting copies the kept subtree
// into the other one)
__device__ inline size_t pool_base(const EngineView& E, int g) {
    return ((size_t)g * 2 + E.arena[g]) * E.cap;
}

__device__ inline double wave_max_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(v, o);
        v = (other > v) ? other : v;
    }
    return v;
}

__device__ inline uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ inline double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------
// One PUCT descent of game g by one wave: src/mcts.py:153-171 with _select_child (:281-312)
// Per-game values the descent starts from; loading them early (before other work of
// the same kernel) takes their latency off the critical path.
struct SelectStart {
    uint32_t root;
    int rmode, turn, winner;
    uint64_t white, black;
};

__device__ inline SelectStart load_select_start(const EngineView& E, int g) {
    SelectStart s;
    s.root = E.root[g]; s.rmode = E.root_mode[g]; s.turn = E.turn[g]; s.winner = E.winner[g];
    s.white = E.white[g]; s.black = E.black[g];
    return s;
}

// Statistics of the children of one node as the descent reads them, one child per lane.
struct ChildRegs {
    uint32_t cN, cinfo, cfc;
    float cW, cP;
    double p64;      // noisy root prior (root_p64), only read for a root in float64 mode
};

// What a caller may load ahead of dev_select: the root's own fields, its children, and the
// per-game counters the descent updates.  The caller patches these registers with the
// tree updates it made after loading them.
struct RootPrefetch {
    uint32_t inf, fc, pn;
    ChildRegs c;
    unsigned long long stat_depth, stat_terminal;
};

__device__ inline ChildRegs load_children(const EngineView& E, size_t pool, int g, uint32_t fc, int k, int lane,
                                          bool want_p64) {
    ChildRegs c{0, 0, 0, 0.f, 0.f, 0.0};
    if (lane < k) {
        c.cN = E.N[pool + fc + lane];
        c.cW = E.W[pool + fc + lane];
        c.cinfo = E.IN[pool + fc + lane];
        c.cfc = E.FC[pool + fc + lane];
        c.cP = E.P[pool + fc + lane];
        if (want_p64) c.p64 = E.root_p64[(size_t)g * 48 + lane];
    }
    return c;
}

// The caller has checked E.active[g].  Each level costs ONE round of dependent loads:
// a node's own (info, first child, visits) arrive with its parent's child statistics.
// With `pre` the root level costs none.
__device__ inline void dev_select(const EngineView& E, int g, int lane, const SelectStart& st,
                                  float* planes, uint8_t* need_eval, const RootPrefetch* pre = nullptr) {
    const size_t pool = pool_base(E, g);
    const uint32_t r = st.root;
    const int rmode = st.rmode;
    int t = st.turn;
    int win = st.winner;
    uint64_t mine = (t > 0) ? st.white : bitrev64(st.black);
    uint64_t theirs = (t > 0) ? st.black : bitrev64(st.white);
    uint32_t node = r;
    uint32_t* path = E.path + (size_t)g * MAX_DEPTH;
    int depth = 1;
    if (lane == 0) path[0] = node;
    const float cp32 = (float)E.c_puct;
    uint32_t inf, fc, pn;
    if (pre) { inf = pre->inf; fc = pre->fc; pn = pre->pn; }
    else { inf = E.IN[pool + node]; fc = E.FC[pool + node]; pn = E.N[pool + node]; }
    while (true) {
        const int k = (inf >> 8) & 0xFF;
        if (k == 0) break;
        if (depth >= MAX_DEPTH) { if (lane == 0) atomicOr(E.err, ERR_DEPTH); break; }
        const double sqrt_parent = sqrt((double)(pn > 1 ? pn : 1));
        const bool root64 = (node == r) && rmode;
        const bool f64mode = root64 || ((inf >> 16) & FLAG_UNIFORM64);
        double score = -INFINITY;
        const ChildRegs c = (pre && depth == 1) ? pre->c : load_children(E, pool, g, fc, k, lane, root64);
        const uint32_t cinfo = c.cinfo, cfc = c.cfc, cN = c.cN;
        if (lane < k) {
            const float cW = c.cW;
            if (!f64mode) {
                const float cP = c.cP;
                const float q = (cN == 0) ? 0.0f : -(cW / (float)cN);
                float u = cp32 * cP;
                u = u * (float)sqrt_parent;
                u = u / (float)(1u + cN);
                const float s = q + u;
                score = (double)s;
            } else {
                const double p64 = root64 ? c.p64 : 1.0 / (double)k;
                const double q = (cN == 0) ? 0.0 : (double)(-(cW / (float)cN));
                double u = E.c_puct * p64;
                u = u * sqrt_parent;
                u = u / (double)(1u + cN);
                score = q + u;
            }
        }
        // argmax with ties to the lowest child slot (strict '>' in insertion order)
        const double best = wave_max_f64(score);
        const unsigned long long tie = __ballot(score == best);
        const int bl = (tie == 0) ? 0 : (__ffsll((long long)tie) - 1);
        const int a = __shfl((int)(cinfo & 0xFF), bl);
        inf = __shfl(cinfo, bl); pn = __shfl(cN, bl);
        const uint32_t next_fc = __shfl(cfc, bl);
        // decode_action + step in canonical coordinates (src/game.py:257-278, 156-187)
        const int sq = a / 3, d = a - 3 * sq;
        const int to = sq + ((d == 0) ? 8 : ((d == 1) ? 7 : 9));
        const uint64_t fb = 1ull << sq, tb = 1ull << (to & 63);
        mine = (mine & ~fb) | tb;
        theirs &= ~tb;
        if (to >= 56) win = t;
        if (win == 0 && theirs == 0) win = t;
        const uint64_t nm = bitrev64(theirs), nt = bitrev64(mine);
        mine = nm; theirs = nt; t = -t;
        node = fc + bl;
        fc = next_fc;
        if (lane == 0) path[depth] = node;
        ++depth;
    }
    const Legal3 l = legal_sources(mine, theirs);
    const bool terminal = (win != 0) || ((l.fwd | l.dl | l.dr) == 0);
    if (lane == 0) {
        E.leaf_node[g] = node; E.leaf_mine[g] = mine; E.leaf_theirs[g] = theirs;
        E.leaf_turn[g] = (int8_t)t; E.path_len[g] = (uint16_t)depth;
        E.leaf_need[g] = terminal ? 0 : 1;
        // src/mcts.py:328-335: result seen by the side to move at the leaf
        E.leaf_tv[g] = (win != 0) ? ((win == t) ? 1.0f : -1.0f) : -1.0f;
        if (need_eval) need_eval[g] = terminal ? 0 : 1;
        const unsigned long long sd = pre ? pre->stat_depth : E.stat_depth[g];
        E.stat_depth[g] = sd + (unsigned long long)(depth - 1);
        if (terminal) E.stat_terminal[g] = (pre ? pre->stat_terminal : E.stat_terminal[g]) + 1ull;
    }
    if (!terminal && planes) write_planes(planes + (size_t)g * 192, mine, theirs, lane);
}

// ---------------------------------------------------------------------------
// _expand_node (src/mcts.py:260-279).  pr: the 192 softmaxed policy values of this
// leaf (global or LDS); s_p / s_a: 48-entry LDS scratch.  One wave, all lanes.
// The leaf a simulation stopped at, and what its expansion wrote into it.
struct LeafState {
    uint32_t node;
    uint64_t mine, theirs;
    int turn;
    uint32_t used;       // E.used[g] before the expansion
    uint32_t info;       // E.IN[leaf] before the expansion
};
struct Expansion {
    int k;               // children created (0: none)
    uint32_t first;      // pool index of the first one
    uint32_t info;       // the leaf's new E.IN word
};

__device__ inline LeafState load_leaf(const EngineView& E, int g, size_t pool) {
    LeafState s;
    s.node = E.leaf_node[g]; s.mine = E.leaf_mine[g]; s.theirs = E.leaf_theirs[g];
    s.turn = E.leaf_turn[g]; s.used = E.used[g];
    s.info = E.IN[pool + s.node];
    return s;
}

__device__ inline Expansion dev_expand(const EngineView& E, int g, int lane, size_t pool, const LeafState& lf,
                                       const float* pr, float* s_p, uint8_t* s_a) {
    const uint64_t mine = lf.mine, theirs = lf.theirs;
    const int t = lf.turn;
    const uint32_t leaf = lf.node;
    Expansion ex{0, 0, lf.info};
    const Legal3 l = legal_sources(mine, theirs);
    const int k = __popcll(l.fwd) + __popcll(l.dl) + __popcll(l.dr);
    // insertion order (src/game.py:117-118 scans the REAL board row-major):
    // White = ascending canonical squares, Black = descending
    const int sq = lane;
    const uint64_t before = (t > 0) ? ((1ull << sq) - 1ull)
                                    : ((sq == 63) ? 0ull : (~0ull << (sq + 1)));
    int rank = __popcll(l.fwd & before) + __popcll(l.dl & before) + __popcll(l.dr & before);
    if ((l.fwd >> sq) & 1ull) { s_a[rank] = (uint8_t)(sq * 3 + 0); s_p[rank] = pr[sq * 3 + 0]; ++rank; }
    if ((l.dl >> sq) & 1ull)  { s_a[rank] = (uint8_t)(sq * 3 + 1); s_p[rank] = pr[sq * 3 + 1]; ++rank; }
    if ((l.dr >> sq) & 1ull)  { s_a[rank] = (uint8_t)(sq * 3 + 2); s_p[rank] = pr[sq * 3 + 2]; ++rank; }
    __syncthreads();
    const float p_r = (lane < k) ? s_p[lane] : 0.f;
    const uint32_t a_r = (lane < k) ? s_a[lane] : 0;
    float prior_sum = 0.0f;                       // sequential float32 sum, insertion order
    for (int i = 0; i < k; ++i) prior_sum = prior_sum + __shfl(p_r, i);
    if (k > 0) {
        const uint32_t base = lf.used;
        if ((uint64_t)base + (uint64_t)k > (uint64_t)E.cap) {
            if (lane == 0) atomicOr(E.err, ERR_POOL);
        } else {
            if (lane < k) {
                const size_t c = pool + base + lane;
                E.N[c] = 0; E.W[c] = 0.f; E.FC[c] = 0; E.IN[c] = a_r;
                E.P[c] = (prior_sum > 0) ? (p_r / prior_sum) : (float)(1.0 / (double)k);
            }
            const uint32_t flags = (prior_sum > 0) ? 0u : FLAG_UNIFORM64;
            ex.k = k; ex.first = base;
            ex.info = (lf.info & 0xFF) | ((uint32_t)k << 8) | (flags << 16);
            if (lane == 0) {
                E.used[g] = base + k;
                E.FC[pool + leaf] = base;
                E.IN[pool + leaf] = ex.info;
            }
        }
    }
    return ex;
}

__device__ inline void dev_expand(const EngineView& E, int g, int lane, const float* pr, float* s_p,
                                  uint8_t* s_a) {
    const size_t pool = pool_base(E, g);
    const LeafState lf = load_leaf(E, g, pool);
    (void)dev_expand(E, g, lane, pool, lf, pr, s_p, s_a);
}

// _backpropagate (src/mcts.py:337-342): v is seen by the side to move at the leaf
__device__ inline void dev_backup(const EngineView& E, int g, int lane, float v) {
    const size_t pool = pool_base(E, g);
    const int len = E.path_len[g];
    const uint32_t* path = E.path + (size_t)g * MAX_DEPTH;
    for (int i = lane; i < len; i += 64) {
        const size_t n = pool + path[i];
        const float vv = ((len - 1 - i) & 1) ? -v : v;   // value = -value per level
        E.N[n] = E.N[n] + 1;
        E.W[n] = E.W[n] + vv;
    }
    if (lane == 0) E.stat_descents[g] += 1ull;
}

}  // namespace az
