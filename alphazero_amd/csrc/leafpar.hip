// Leaf-parallel search of a few games: the latency mode of MCTS.search (what the reference's
// arena and web front end call for one game at a time, src/arena.py:248-265, src/web.py:447-470).
//
// The reference evaluates ONE leaf per game and simulation (src/mcts.py:153-187), so a single
// game's search is a chain of network calls on one position each.  Here K descents of a game are
// made before the network runs, the K leaves are evaluated in one batch (K compute units instead
// of one), then expanded and backed up in descent order.  While a descent's evaluation is
// pending, every node on its path carries one "in flight" count (bits 24-31 of the node's info
// word): selection reads a child's statistics as N + in_flight visits and W + in_flight value,
// i.e. as if each pending descent had ended in a loss for the player choosing that child
// (virtual loss), which spreads the K descents over the tree.  Backup adds the real value to W
// and 1 to N and takes the in-flight count away again, so N and W always hold exact sums.
// Two descents that end at the same leaf share its evaluation (the second is backed up with
// the first one's value).
//
// With K = 1 there is no virtual loss and this would be the reference's order; the caller uses
// the exact kernels (az_engine_search) for that and this file only for K > 1, where results
// differ from the reference's by construction.
//
// Compiled with -ffp-contract=off like the other tree code (q + u stays two rounded operations).
#include <type_traits>

#include "az_engine.h"
#include "head_dev.h"
#include "net_internal.h"
#include "tree_dev.h"

namespace az {

constexpr uint32_t VL_ONE = 1u << 24, VL_MASK = 0xFF000000u;
constexpr int LP_MAX_K = 32;          // descents per game and batch
constexpr int LP_MAX_ROWS = 256;      // network rows per batch (games x K)

// one wave per game: K descents, one after the other
__global__ __launch_bounds__(64) void k_lp_select(EngineView E, LeafParView L, int kmax, int K) {
    __shared__ uint32_t s_leaf[LP_MAX_K];
    const int lane = threadIdx.x;
    const int g = L.slots[blockIdx.x];
    const size_t pool = pool_base(E, g);
    const SelectStart st = load_select_start(E, g);
    const float cp32 = (float)E.c_puct;
    if (st.root == NODE_NONE) {   // nothing to descend from: no row of this game holds a descent
        if (lane == 0) atomicOr(E.err, ERR_NOROOT);
        K = 0;
    }
    for (int j = 0; j < K; ++j) {
        const int row = blockIdx.x * kmax + j;
        uint32_t* path = L.path + (size_t)row * MAX_DEPTH;
        int t = st.turn, win = st.winner;
        uint64_t mine = (t > 0) ? st.white : bitrev64(st.black);
        uint64_t theirs = (t > 0) ? st.black : bitrev64(st.white);
        uint32_t node = st.root;
        int depth = 1;
        uint32_t inf = E.IN[pool + node], fc = E.FC[pool + node];
        uint32_t pn = E.N[pool + node] + (inf >> 24);
        if (lane == 0) { path[0] = node; E.IN[pool + node] = inf + VL_ONE; }
        while (true) {
            const int k = (inf >> 8) & 0xFF;
            if (k == 0) break;
            if (depth >= MAX_DEPTH) { if (lane == 0) atomicOr(E.err, ERR_DEPTH); break; }
            const double sqrt_parent = sqrt((double)(pn > 1 ? pn : 1));
            const bool root64 = (node == st.root) && st.rmode;
            const bool f64mode = root64 || ((inf >> 16) & FLAG_UNIFORM64);
            double score = -INFINITY;
            const ChildRegs c = load_children(E, pool, g, fc, k, lane, root64);
            const uint32_t vl = c.cinfo >> 24;
            const uint32_t n_eff = c.cN + vl;
            if (lane < k) {
                const float w_eff = c.cW + (float)vl;
                if (!f64mode) {
                    const float q = (n_eff == 0) ? 0.0f : -(w_eff / (float)n_eff);
                    float u = cp32 * c.cP;
                    u = u * (float)sqrt_parent;
                    u = u / (float)(1u + n_eff);
                    const float s = q + u;
                    score = (double)s;
                } else {
                    const double p64 = root64 ? c.p64 : 1.0 / (double)k;
                    const double q = (n_eff == 0) ? 0.0 : (double)(-(w_eff / (float)n_eff));
                    double u = E.c_puct * p64;
                    u = u * sqrt_parent;
                    u = u / (double)(1u + n_eff);
                    score = q + u;
                }
            }
            const double best = wave_max_f64(score);
            const unsigned long long tie = __ballot(score == best);
            const int bl = (tie == 0) ? 0 : (__ffsll((long long)tie) - 1);
            if (lane == bl) E.IN[pool + fc + bl] = c.cinfo + VL_ONE;   // this descent is in flight through the child
            const int a = __shfl((int)(c.cinfo & 0xFF), bl);
            inf = __shfl(c.cinfo, bl);
            pn = __shfl(n_eff, bl);
            const uint32_t next_fc = __shfl(c.cfc, bl);
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
        int same = -1;
        for (int i = 0; i < j; ++i)
            if (s_leaf[i] == node && same < 0) same = i;
        if (lane == 0) {
            s_leaf[j] = node;
            L.node[row] = node; L.mine[row] = mine; L.theirs[row] = theirs; L.turn[row] = (int8_t)t;
            L.len[row] = (uint16_t)depth;
            L.tv[row] = (win != 0) ? ((win == t) ? 1.0f : -1.0f) : -1.0f;
            L.kind[row] = (same >= 0) ? 2 : (terminal ? 1 : 0);
            L.first[row] = (uint8_t)(same >= 0 ? same : j);
            E.stat_depth[g] += (unsigned long long)(depth - 1);
            if (terminal) E.stat_terminal[g] += 1ull;
        }
        __syncthreads();   // the in-flight counts and s_leaf are visible to the next descent
    }
    // rows of this game that hold no descent of this batch: any valid position for the network
    for (int j = K + lane; j < kmax; j += 64) {
        const int row = blockIdx.x * kmax + j;
        L.mine[row] = 0xFFFFull; L.theirs[row] = 0xFFFF000000000000ull; L.kind[row] = 3;
    }
}

// one wave per game: evaluate, expand and back up the K leaves in descent order
__global__ __launch_bounds__(64) void k_lp_finish(EngineView E, LeafParView L, FinishParams F, int kmax, int K) {
    __shared__ float s_probs[192];
    __shared__ float s_p[MAX_CHILDREN];
    __shared__ uint8_t s_a[MAX_CHILDREN];
    __shared__ float s_val[LP_MAX_K];
    const int lane = threadIdx.x;
    const int g = L.slots[blockIdx.x];
    const size_t pool = pool_base(E, g);
    for (int j = 0; j < K; ++j) {
        const int row = blockIdx.x * kmax + j;
        const int kind = L.kind[row];
        if (kind == 3) continue;   // the game had no root (reported by k_lp_select)
        float v;
        if (kind == 0) {
            float pr[3];
            v = dev_head_finish(F, row, lane, pr);
            s_probs[lane] = pr[0]; s_probs[lane + 64] = pr[1]; s_probs[lane + 128] = pr[2];
            __syncthreads();
            LeafState lf;
            lf.node = L.node[row]; lf.mine = L.mine[row]; lf.theirs = L.theirs[row]; lf.turn = L.turn[row];
            lf.used = E.used[g]; lf.info = E.IN[pool + lf.node];
            const Expansion ex = dev_expand(E, g, lane, pool, lf, s_probs, s_p, s_a);
            // dev_expand rebuilt the leaf's info word without the in-flight count
            if (lane == 0 && ex.k > 0) E.IN[pool + lf.node] = ex.info | (lf.info & VL_MASK);
        } else if (kind == 1) {
            v = L.tv[row];
        } else {
            v = s_val[L.first[row]];
        }
        if (lane == 0) s_val[j] = v;
        __syncthreads();   // the expansion is in memory before the in-flight counts are read back
        const int len = L.len[row];
        const uint32_t* path = L.path + (size_t)row * MAX_DEPTH;
        for (int i = lane; i < len; i += 64) {
            const size_t n = pool + path[i];
            const float vv = ((len - 1 - i) & 1) ? -v : v;
            E.N[n] = E.N[n] + 1;
            E.W[n] = E.W[n] + vv;
            E.IN[n] = E.IN[n] - VL_ONE;
        }
        if (lane == 0) E.stat_descents[g] += 1ull;
        __syncthreads();
    }
}

}  // namespace az

using namespace az;

// one set of buffers per engine, made on first use and freed by az_engine_destroy with the
// engine's other device memory
static int lp_buffers(az_engine* e, LeafParView* out) {
    if (e->lp_ready) { *out = e->lp; return AZ_OK; }
    LeafParView v{};
    auto alloc = [&](auto*& ptr, size_t count) -> int {
        void* p = nullptr;
        AZ_HIP(hipMalloc(&p, count * sizeof(*ptr)));
        e->allocs.push_back(p);
        AZ_HIP(hipMemset(p, 0, count * sizeof(*ptr)));
        ptr = (std::remove_reference_t<decltype(ptr)>)p;
        return AZ_OK;
    };
    int rc = AZ_OK;
    if (rc == AZ_OK) rc = alloc(v.mine, LP_MAX_ROWS);
    if (rc == AZ_OK) rc = alloc(v.theirs, LP_MAX_ROWS);
    if (rc == AZ_OK) rc = alloc(v.node, LP_MAX_ROWS);
    if (rc == AZ_OK) rc = alloc(v.turn, LP_MAX_ROWS);
    if (rc == AZ_OK) rc = alloc(v.len, LP_MAX_ROWS);
    if (rc == AZ_OK) rc = alloc(v.kind, LP_MAX_ROWS);
    if (rc == AZ_OK) rc = alloc(v.first, LP_MAX_ROWS);
    if (rc == AZ_OK) rc = alloc(v.tv, LP_MAX_ROWS);
    if (rc == AZ_OK) rc = alloc(v.path, (size_t)LP_MAX_ROWS * MAX_DEPTH);
    if (rc == AZ_OK) rc = alloc(v.slots, LP_MAX_ROWS);
    if (rc != AZ_OK) return rc;   // what was allocated stays in e->allocs and goes with the engine
    e->lp = v;
    e->lp_ready = true;
    *out = v;
    return AZ_OK;
}

extern "C" int az_engine_search_leaf_parallel(az_engine* e, int32_t n, const int32_t* slots_host, int32_t num_simulations,
                                              int32_t leaf_batch, void* stream) {
    AZ_CHECK_ARG(e && slots_host, "az_engine_search_leaf_parallel: null argument");
    AZ_CHECK_ARG(leaf_batch >= 1 && leaf_batch <= LP_MAX_K, "az_engine_search_leaf_parallel: leaf_batch %d is not in 1..%d",
                 leaf_batch, LP_MAX_K);
    AZ_CHECK_ARG(n >= 1 && n * leaf_batch <= LP_MAX_ROWS, "az_engine_search_leaf_parallel: %d games x %d leaves exceed %d rows",
                 n, leaf_batch, LP_MAX_ROWS);
    AZ_CHECK_ARG(num_simulations >= 0, "az_engine_search_leaf_parallel: negative simulation count");
    if (!e->net) AZ_FAIL(AZ_ERR_STATE, "no network attached (az_engine_attach_net)");
    AZ_CHECK_ARG(az_net_max_batch(e->net) >= n * leaf_batch, "az_engine_search_leaf_parallel: the network takes %d rows, %d needed",
                 az_net_max_batch(e->net), n * leaf_batch);
    for (int i = 0; i < n; ++i) {
        AZ_CHECK_ARG(slots_host[i] >= 0 && slots_host[i] < e->v.B, "az_engine_search_leaf_parallel: game %d out of range", slots_host[i]);
        for (int j = 0; j < i; ++j)
            AZ_CHECK_ARG(slots_host[i] != slots_host[j], "az_engine_search_leaf_parallel: game %d appears twice", slots_host[i]);
    }
    hipStream_t s = (hipStream_t)stream;
    LeafParView L;
    int rc = lp_buffers(e, &L);
    if (rc != AZ_OK) return rc;
    AZ_HIP(hipMemcpyAsync(L.slots, slots_host, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    AZ_HIP(hipStreamSynchronize(s));   // slots_host is free again
    const int rows = n * leaf_batch;
    const FinishParams fin = az::net_finish_params(e->net, rows, nullptr);
    for (int done = 0; done < num_simulations; done += leaf_batch) {
        const int K = num_simulations - done < leaf_batch ? num_simulations - done : leaf_batch;
        hipLaunchKernelGGL(k_lp_select, dim3(n), dim3(64), 0, s, e->v, L, leaf_batch, K);
        if ((rc = az::net_forward_partial(e->net, rows, L.mine, L.theirs, nullptr, s))) return rc;
        hipLaunchKernelGGL(k_lp_finish, dim3(n), dim3(64), 0, s, e->v, L, fin, leaf_batch, K);
    }
    AZ_HIP(hipGetLastError());
    return az::check_engine_error(e, s);
}
