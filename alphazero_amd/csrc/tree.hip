// PUCT tree search over an SoA node pool in HBM, one wavefront per game.
//
// Compile with -ffp-contract=off: the reference evaluates q + u as two
// separately rounded float32 operations (src/mcts.py:299-305) and visit counts
// are only bit-exact if this code does the same.
//
// Node pool layout (per engine, all arrays [B * cap], game g owns the slice
// [g*cap, (g+1)*cap)); a node is also the edge from its parent:
//   N  u32   visit_count            (src/mcts.py:27)
//   W  f32   value_sum              (float32 in the reference under numpy >= 2)
//   P  f32   prior                  (src/mcts.py:274-279)
//   FC u32   index of first child; children are contiguous, in insertion order
//   IN u32   action | n_child << 8 | flags << 16
// Each game owns two arenas of `cap` nodes.  Nodes are bump-allocated in the
// live arena; re-rooting (tree reuse) copies the kept subtree breadth-first into
// the other arena and flips, so memory is bounded by the live tree.  Noisy root priors (float64 in the reference after
// src/mcts.py:326) live in a per-game side array of 48 doubles.
#include <vector>

#include "tree_dev.h"

namespace az {

// ---------------------------------------------------------------------------
// src/mcts.py:117-133
__global__ __launch_bounds__(64) void k_roots_prepare(EngineView E, float* planes,
                                                      uint8_t* need_eval) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (!E.active[g]) {
        if (lane == 0) { E.leaf_need[g] = 0; if (need_eval) need_eval[g] = 0; }
        return;
    }
    const size_t pool = pool_base(E, g);
    uint32_t r = E.root[g];
    const bool fresh = (r == NODE_NONE);
    if (fresh) {                        // Node(0): fresh tree, allocator reset
        r = 0;
        if (lane == 0) {
            E.root[g] = 0; E.used[g] = 1; E.root_mode[g] = 0;
            E.N[pool] = 0; E.W[pool] = 0.f; E.P[pool] = 0.f; E.FC[pool] = 0; E.IN[pool] = 0;
        }
    }
    const uint32_t k = fresh ? 0 : ((E.IN[pool + r] >> 8) & 0xFF);
    const int need = (k == 0);
    if (need) {
        const uint64_t w = E.white[g], b = E.black[g];
        const int t = E.turn[g];
        const uint64_t mine = (t > 0) ? w : bitrev64(b);
        const uint64_t theirs = (t > 0) ? b : bitrev64(w);
        if (lane == 0) {
            E.leaf_node[g] = r; E.leaf_mine[g] = mine; E.leaf_theirs[g] = theirs;
            E.leaf_turn[g] = (int8_t)t; E.path_len[g] = 1; E.path[(size_t)g * MAX_DEPTH] = r;
        }
        if (planes) write_planes(planes + (size_t)g * 192, mine, theirs, lane);
    }
    if (lane == 0) {
        E.leaf_need[g] = (uint8_t)need; if (need_eval) need_eval[g] = (uint8_t)need;
        if (need) E.stat_rootexp[g] += 1ull;
    }
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_select(EngineView E, float* planes, uint8_t* need_eval) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (!E.active[g]) {
        if (lane == 0) { E.leaf_need[g] = 0; if (need_eval) need_eval[g] = 0; }
        return;
    }
    dev_select(E, g, lane, load_select_start(E, g), planes, need_eval);
}

__global__ __launch_bounds__(64) void k_expand_backup(EngineView E, const float* probs,
                                                      const float* values, int do_backup) {
    __shared__ float s_p[MAX_CHILDREN];
    __shared__ uint8_t s_a[MAX_CHILDREN];
    const int g = blockIdx.x, lane = threadIdx.x;
    if (!E.active[g]) return;
    const int need = E.need_cur[g];
    const int row = (E.row_of && need) ? E.row_of[g] : g;
    if (need) dev_expand(E, g, lane, probs + (size_t)row * 192, s_p, s_a);
    if (do_backup) dev_backup(E, g, lane, need ? values[row] : E.leaf_tv[g]);
    if (lane == 0) E.need_cur[g] = 0;
}

// ---------------------------------------------------------------------------
// _add_dirichlet_noise (src/mcts.py:314-326)
__global__ __launch_bounds__(64) void k_add_noise(EngineView E, const double* noise,
                                                  uint64_t call_id) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (!E.active[g]) return;
    const uint32_t r = E.root[g];
    if (r == NODE_NONE) return;
    const size_t pool = pool_base(E, g);
    const uint32_t inf = E.IN[pool + r];
    const int k = (inf >> 8) & 0xFF;
    if (k == 0) return;
    const uint32_t fc = E.FC[pool + r];
    double nz;
    if (noise) {
        nz = (lane < k) ? noise[(size_t)g * 48 + lane] : 0.0;
    } else {
        Rng rng(E.seed, (uint64_t)g, call_id, (uint64_t)lane);
        const double gm = (lane < k) ? rng.gamma(E.alpha) : 0.0;
        const double tot = wave_sum_f64(gm);
        nz = (tot > 0) ? gm / tot : 1.0 / (double)k;
    }
    if (lane < k) {
        const double add = E.eps * nz;
        double p;
        if (E.root_mode[g]) {
            p = (1.0 - E.eps) * E.root_p64[(size_t)g * 48 + lane] + add;
        } else if ((inf >> 16) & FLAG_UNIFORM64) {
            p = (1.0 - E.eps) * (1.0 / (double)k) + add;
        } else {
            const float keep = (float)(1.0 - E.eps) * E.P[pool + fc + lane];  // float32 product
            p = (double)keep + add;
        }
        E.root_p64[(size_t)g * 48 + lane] = p;
    }
    __syncthreads();
    if (lane == 0) E.root_mode[g] = 1;
}

// ---------------------------------------------------------------------------
// get_action_probs (src/mcts.py:191-225).  Temperatures 0 and 1 (the trainer's schedule,
// src/parallel_trainer.py:164) are bit-exact: an argmax, and count / total with both exactly
// representable.  Any other temperature follows the same formula, count ** (1 / T) / sum, in
// float32 with the device's powf and a fixed summation order; numpy's powf and pairwise
// sum differ from it in the last bits, so that case is equal to the reference within
// rounding (relative 1e-5), not bit for bit.
__global__ __launch_bounds__(64) void k_action_probs(EngineView E, const float* temps,
                                                     float* probs_out) {
    __shared__ float row[192];
    const int g = blockIdx.x, lane = threadIdx.x;
    if (!E.active[g]) return;
    const size_t pool = pool_base(E, g);
    const uint32_t r = E.root[g];
    int k = 0; uint32_t fc = 0;
    if (r != NODE_NONE) { k = (E.IN[pool + r] >> 8) & 0xFF; fc = E.FC[pool + r]; }
    const float temp = temps ? temps[g] : ((E.ply[g] >= E.temp_thr) ? 0.0f : 1.0f);
    const bool t0 = (temp == 0.0f);
    for (int i = lane; i < 192; i += 64) row[i] = 0.f;
    __syncthreads();
    uint32_t cnt = 0; int a = 0;
    if (lane < k) { cnt = E.N[pool + fc + lane]; a = E.IN[pool + fc + lane] & 0xFF; }
    const uint32_t total = wave_sum_u32(cnt);
    if (t0) {
        // np.argmax over the 192-vector: highest count, lowest action index
        unsigned long long key = (lane < k) ? (((unsigned long long)cnt << 8) | (unsigned)(255 - a)) : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(key, o);
            key = other > key ? other : key;
        }
        const int best = ((key >> 8) == 0) ? 0 : (255 - (int)(key & 0xFF));
        if (lane == 0) row[best] = 1.0f;
    } else if (temp == 1.0f) {
        if (lane < k) row[a] = (total > 0) ? ((float)cnt / (float)total) : (float)(1.0 / (double)k);
    } else {
        const float powered = (lane < k) ? powf((float)cnt, 1.0f / temp) : 0.0f;
        float sum = powered;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane < k) row[a] = (sum > 0) ? (powered / sum) : (float)(1.0 / (double)k);
    }
    __syncthreads();
    for (int i = lane; i < 192; i += 64) probs_out[(size_t)g * 192 + i] = row[i];
}

// ---------------------------------------------------------------------------
// Re-root game g at node `src` of its live arena: breadth-first copy of that
// subtree into the other arena (children stay contiguous and in insertion
// order, N/W/P bit patterns unchanged), then flip arenas.  One wave.
__device__ void reroot_subtree(EngineView E, int g, uint32_t src, int lane) {
    const size_t from = pool_base(E, g);
    const size_t to = ((size_t)g * 2 + (E.arena[g] ^ 1)) * E.cap;
    if (lane == 0) {
        E.N[to] = E.N[from + src]; E.W[to] = E.W[from + src]; E.P[to] = E.P[from + src];
        E.FC[to] = E.FC[from + src]; E.IN[to] = E.IN[from + src];
    }
    __syncthreads();
    uint32_t count = 1, head = 0;
    while (head < count) {
        const uint32_t nb = min(64u, count - head);
        uint32_t k = 0, old_fc = 0;
        if ((uint32_t)lane < nb) {
            k = (E.IN[to + head + lane] >> 8) & 0xFF;
            old_fc = E.FC[to + head + lane];
        }
        uint32_t incl = k;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const uint32_t total = __shfl(incl, 63);
        const uint32_t off = count + incl - k;
        if ((uint32_t)lane < nb && k > 0) E.FC[to + head + lane] = off;
        for (uint32_t j = 0; j < nb; ++j) {
            const uint32_t kj = __shfl(k, (int)j), oj = __shfl(old_fc, (int)j), dj = __shfl(off, (int)j);
            if ((uint32_t)lane < kj) {
                const size_t a = from + oj + lane, d = to + dj + lane;
                E.N[d] = E.N[a]; E.W[d] = E.W[a]; E.P[d] = E.P[a]; E.FC[d] = E.FC[a]; E.IN[d] = E.IN[a];
            }
        }
        __syncthreads();  // the copied nodes are read back as the next frontier
        count += total;
        head += nb;
    }
    if (lane == 0) { E.arena[g] ^= 1; E.root[g] = 0; E.used[g] = count; }
}

// game.step + tree reuse (src/parallel_trainer.py:190-198, src/game.py:156-187)
__global__ __launch_bounds__(64) void k_advance(EngineView E, const int16_t* actions) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int a = actions[g];
    if (a < 0) return;
    const size_t pool = pool_base(E, g);
    const uint32_t r = E.root[g];
    uint32_t newroot = NODE_NONE;
    if (r != NODE_NONE) {
        const int k = (E.IN[pool + r] >> 8) & 0xFF;
        const uint32_t fc = E.FC[pool + r];
        const bool hit = (lane < k) && ((int)(E.IN[pool + fc + lane] & 0xFF) == a);
        const unsigned long long m = __ballot(hit);
        if (m) newroot = fc + (__ffsll((long long)m) - 1);
    }
    if (lane == 0) {
        E.root_mode[g] = 0;
        if (newroot == NODE_NONE) { E.root[g] = NODE_NONE; E.used[g] = 0; }
        uint64_t w = E.white[g], b = E.black[g];
        const int t = E.turn[g];
        int win = E.winner[g];
        const int sq = a / 3, d = a - 3 * sq;
        const int to = sq + ((d == 0) ? 8 : ((d == 1) ? 7 : 9));
        const int rf = (t > 0) ? sq : 63 - sq, rt = (t > 0) ? to : 63 - to;
        const uint64_t fb = 1ull << (rf & 63), tb = 1ull << (rt & 63);
        if (t > 0) { w = (w & ~fb) | tb; b &= ~tb; if ((rt >> 3) == 7) win = 1; if (win == 0 && b == 0) win = 1; }
        else       { b = (b & ~fb) | tb; w &= ~tb; if ((rt >> 3) == 0) win = -1; if (win == 0 && w == 0) win = -1; }
        E.white[g] = w; E.black[g] = b; E.turn[g] = (int8_t)(-t); E.winner[g] = (int8_t)win;
        E.ply[g] = E.ply[g] + 1;
    }
    if (newroot != NODE_NONE) reroot_subtree(E, g, newroot, lane);
}

__global__ void k_set_positions(EngineView E, int n, const int32_t* slots, const uint64_t* w,
                                const uint64_t* b, const int8_t* t, const int8_t* win, int keep_tree) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = slots[i];
    E.white[g] = w[i]; E.black[g] = b[i]; E.turn[g] = t[i]; E.winner[g] = win[i];
    if (!keep_tree) {
        E.root[g] = NODE_NONE; E.used[g] = 0; E.root_mode[g] = 0; E.ply[g] = 0; E.leaf_need[g] = 0;
    }
}

__global__ void k_root_counts(EngineView E, int32_t* out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.B) return;
    const uint32_t r = E.root[g];
    out[g] = (r == NODE_NONE) ? -1 : (int32_t)((E.IN[pool_base(E, g) + r] >> 8) & 0xFF);
}

__global__ __launch_bounds__(64) void k_set_root(EngineView E, int g, int64_t node) {
    const int lane = threadIdx.x;
    if (lane == 0) E.root_mode[g] = 0;
    if (node < 0) { if (lane == 0) { E.root[g] = NODE_NONE; E.used[g] = 0; } return; }
    reroot_subtree(E, g, (uint32_t)node, lane);
}

// az_engine_import_tree: the arrays are in the live arena already; publish the tree
__global__ void k_import_tree(EngineView E, int g, uint32_t n_nodes, int root64) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        E.root[g] = 0; E.used[g] = n_nodes; E.root_mode[g] = (uint8_t)(root64 ? 1 : 0); E.leaf_need[g] = 0;
    }
}
__global__ void k_read_arena(EngineView E, int g, uint8_t* out) { *out = E.arena[g]; }

// batched form: block i re-roots game slots[i] at nodes[i] (negative: drop the tree)
__global__ __launch_bounds__(64) void k_set_roots(EngineView E, int n, const int32_t* slots, const int64_t* nodes) {
    const int lane = threadIdx.x;
    const int g = slots[blockIdx.x];
    const int64_t node = nodes[blockIdx.x];
    if (g < 0 || g >= E.B || node >= (int64_t)E.used[g]) { if (lane == 0) atomicOr(E.err, ERR_NODE); return; }
    if (lane == 0) E.root_mode[g] = 0;
    if (node < 0) { if (lane == 0) { E.root[g] = NODE_NONE; E.used[g] = 0; } return; }
    reroot_subtree(E, g, (uint32_t)node, lane);
}

__device__ void node_query(const EngineView& E, int g, int64_t node_in, NodeQuery* out, int lane) {
    const size_t pool = pool_base(E, g);
    const uint32_t r = E.root[g];
    const uint32_t node = (node_in < 0) ? r : (uint32_t)node_in;
    if (node == NODE_NONE) { if (lane == 0) { out->k = -1; out->node_index = -1; } return; }
    const uint32_t inf = E.IN[pool + node];
    const int k = (inf >> 8) & 0xFF;
    const uint32_t fc = E.FC[pool + node];
    const bool root64 = (node == r) && E.root_mode[g];
    const bool uni = (inf >> 16) & FLAG_UNIFORM64;
    if (lane == 0) {
        out->k = k; out->node_visits = E.N[pool + node]; out->node_value_sum = E.W[pool + node];
        out->node_index = node;
    }
    if (lane < k) {
        out->actions[lane] = (uint8_t)(E.IN[pool + fc + lane] & 0xFF);
        out->visits[lane] = E.N[pool + fc + lane];
        out->value_sum[lane] = E.W[pool + fc + lane];
        out->prior64[lane] = root64 ? E.root_p64[(size_t)g * 48 + lane]
                                    : (uni ? 1.0 / (double)k : (double)E.P[pool + fc + lane]);
        out->prior_is64[lane] = (root64 || uni) ? 1 : 0;
        out->child_index[lane] = (int64_t)(fc + lane);
    }
}

__global__ __launch_bounds__(64) void k_node_children(EngineView E, int g, int64_t node_in,
                                                      NodeQuery* out) {
    node_query(E, g, node_in, out, threadIdx.x);
}

// block i reports the root of game slots[i]
__global__ __launch_bounds__(64) void k_roots_children(EngineView E, const int32_t* slots, NodeQuery* out) {
    node_query(E, slots[blockIdx.x], -1, out + blockIdx.x, threadIdx.x);
}

}  // namespace az

using namespace az;

// ---------------------------------------------------------------------------
static int engine_alloc(az_engine* e) {
    EngineView& V = e->v;
    const size_t B = (size_t)V.B, cap = (size_t)V.cap;
#define AZ_ALLOC(ptr, count)                                                          \
    do {                                                                              \
        void* _p = nullptr;                                                           \
        AZ_HIP(hipMalloc(&_p, (count) * sizeof(*(ptr))));                             \
        e->allocs.push_back(_p);                                                      \
        AZ_HIP(hipMemset(_p, 0, (count) * sizeof(*(ptr))));                           \
        (ptr) = (decltype(ptr))_p;                                                    \
    } while (0)
    AZ_ALLOC(V.white, B); AZ_ALLOC(V.black, B); AZ_ALLOC(V.turn, B); AZ_ALLOC(V.winner, B);
    AZ_ALLOC(V.active, B); AZ_ALLOC(V.ply, B); AZ_ALLOC(V.root, B); AZ_ALLOC(V.used, B);
    AZ_ALLOC(V.root_mode, B); AZ_ALLOC(V.root_p64, B * 48);
    AZ_ALLOC(V.leaf_node, B); AZ_ALLOC(V.leaf_mine, B); AZ_ALLOC(V.leaf_theirs, B);
    AZ_ALLOC(V.leaf_turn, B); AZ_ALLOC(V.leaf_tv, B);
    // two flag buffers, each readable as zeros up to the next multiple of 16 games
    uint8_t *flags0 = nullptr, *flags1 = nullptr;
    AZ_ALLOC(flags0, (B + 31) & ~(size_t)15); AZ_ALLOC(flags1, (B + 31) & ~(size_t)15);
    e->need_buf[0] = flags0; e->need_buf[1] = flags1;
    V.leaf_need = V.need_cur = flags0;
    AZ_ALLOC(V.path_len, B); AZ_ALLOC(V.path, B * MAX_DEPTH);
    AZ_ALLOC(V.arena, B);
#define AZ_ALLOC_RAW(ptr, count)                                                      \
    do {                                                                              \
        void* _p = nullptr;                                                           \
        AZ_HIP(hipMalloc(&_p, (count) * sizeof(*(ptr))));                             \
        (ptr) = (decltype(ptr))_p;                                                    \
        e->allocs.push_back(_p);                                                      \
    } while (0)
    // node pool: two arenas per game; nodes are initialised when allocated
    AZ_ALLOC_RAW(V.N, 2 * B * cap); AZ_ALLOC_RAW(V.W, 2 * B * cap); AZ_ALLOC_RAW(V.P, 2 * B * cap);
    AZ_ALLOC_RAW(V.FC, 2 * B * cap); AZ_ALLOC_RAW(V.IN, 2 * B * cap);
#undef AZ_ALLOC_RAW
    AZ_ALLOC(V.err, 1); AZ_ALLOC(V.stat_depth, B); AZ_ALLOC(V.stat_descents, B);
    AZ_ALLOC(V.stat_terminal, B); AZ_ALLOC(V.stat_rootexp, B);
    AZ_ALLOC(e->stage_slots, B); AZ_ALLOC(e->stage_w, B); AZ_ALLOC(e->stage_b, B);
    AZ_ALLOC(e->stage_t, B); AZ_ALLOC(e->stage_win, B); AZ_ALLOC(e->stage_act, B);
    AZ_ALLOC(e->stage_temps, B); AZ_ALLOC(e->query_dev, 1);
#undef AZ_ALLOC
    AZ_HIP(hipHostMalloc((void**)&e->query_host, sizeof(NodeQuery), hipHostMallocDefault));
    AZ_HIP(hipMemset(V.active, 1, B));
    AZ_HIP(hipMemset(V.root, 0xFF, B * sizeof(uint32_t)));
    return AZ_OK;
}

extern "C" int az_engine_create(az_engine** out, const az_engine_config* cfg) {
    AZ_CHECK_ARG(out && cfg, "az_engine_create: null argument");
    AZ_CHECK_ARG(cfg->n_games > 0 && cfg->n_games <= (1 << 20), "az_engine_create: n_games = %d out of range",
                 cfg->n_games);
    if (cfg->device >= 0) AZ_HIP(hipSetDevice(cfg->device));
    int device = 0;
    AZ_HIP(hipGetDevice(&device));
    int64_t cap = cfg->pool_nodes_per_game;
    if (cap <= 0) cap = 1 << 18;
    AZ_CHECK_ARG(cap >= 64 && cap < (1ll << 31), "az_engine_create: pool_nodes_per_game out of range");
    az_engine* e = new az_engine();
    e->device = device;
    EngineView& V = e->v;
    V.B = cfg->n_games;
    V.c_puct = cfg->c_puct;
    V.alpha = cfg->dirichlet_alpha;
    V.eps = cfg->dirichlet_epsilon;
    V.temp_thr = cfg->temperature_threshold;
    V.seed = cfg->seed;
    V.cap = (uint32_t)cap;
    const int rc = engine_alloc(e);
    if (rc != AZ_OK) { az_engine_destroy(e); return rc; }
    *out = e;
    return AZ_OK;
}

extern "C" int az_engine_destroy(az_engine* e) {
    if (!e) return AZ_OK;
    for (void* p : e->allocs) (void)hipFree(p);
    if (e->query_host) (void)hipHostFree(e->query_host);
    if (e->query_batch_host) (void)hipHostFree(e->query_batch_host);
    delete e;
    return AZ_OK;
}

// Reads the device error word (synchronises).  A condition is reported once: the word is
// cleared behind the read, so one tree that outgrew its pool, or one bad index, does not
// fail every later call on an engine whose other games are fine.  What the failed step
// left behind is consistent: a leaf that could not be expanded stays a leaf, a game whose
// samples did not fit the queue is dropped whole.
int az::check_engine_error(az_engine* e, hipStream_t s) {
    int32_t flag = 0;
    AZ_HIP(hipMemcpyAsync(&flag, e->v.err, sizeof flag, hipMemcpyDeviceToHost, s));
    AZ_HIP(hipStreamSynchronize(s));
    if (flag == 0) return AZ_OK;
    AZ_HIP(hipMemsetAsync(e->v.err, 0, sizeof(int32_t), s));
    AZ_HIP(hipStreamSynchronize(s));
    if (flag & ERR_POOL) AZ_FAIL(AZ_ERR_CAPACITY, "node pool exhausted (pool_nodes_per_game = %u)", e->v.cap);
    if (flag & ERR_DEPTH) AZ_FAIL(AZ_ERR_CAPACITY, "selection path deeper than %d", MAX_DEPTH);
    if (flag & ERR_NODE) AZ_FAIL(AZ_ERR_ARG, "az_engine_set_roots: a game slot or node index was out of range");
    if (flag & ERR_NOROOT)
        AZ_FAIL(AZ_ERR_STATE, "az_engine_search_leaf_parallel: a game has no root (az_engine_search with 0 simulations makes one)");
    if (flag & ERR_HISTORY)
        AZ_FAIL(AZ_ERR_CAPACITY, "sample queue full or a game longer than %d plies: samples were dropped "
                                 "(drain the queue before it holds %u samples)", MAX_PLIES, e->sp.q_cap);
    AZ_FAIL(AZ_ERR_STATE, "unknown device error flag %d", flag);
}

static int check_err(az_engine* e, hipStream_t s) { return az::check_engine_error(e, s); }

static int set_positions_impl(az_engine* e, int32_t n, const int32_t* slots_host,
                              const uint64_t* white_host, const uint64_t* black_host,
                              const int8_t* turn_host, const int8_t* winner_host, void* stream,
                              int keep_tree);

extern "C" int az_engine_set_positions(az_engine* e, int32_t n, const int32_t* slots_host,
                                       const uint64_t* white_host, const uint64_t* black_host,
                                       const int8_t* turn_host, const int8_t* winner_host,
                                       void* stream) {
    return set_positions_impl(e, n, slots_host, white_host, black_host, turn_host, winner_host, stream, 0);
}

extern "C" int az_engine_update_positions(az_engine* e, int32_t n, const int32_t* slots_host,
                                          const uint64_t* white_host, const uint64_t* black_host,
                                          const int8_t* turn_host, const int8_t* winner_host,
                                          void* stream) {
    return set_positions_impl(e, n, slots_host, white_host, black_host, turn_host, winner_host, stream, 1);
}

extern "C" int az_engine_root_counts(az_engine* e, int32_t* counts_host, void* stream) {
    AZ_CHECK_ARG(e && counts_host, "az_engine_root_counts: null argument");
    hipStream_t s = (hipStream_t)stream;
    int32_t* tmp = reinterpret_cast<int32_t*>(e->stage_slots);
    hipLaunchKernelGGL(k_root_counts, dim3((e->v.B + 255) / 256), dim3(256), 0, s, e->v, tmp);
    AZ_HIP(hipGetLastError());
    AZ_HIP(hipMemcpyAsync(counts_host, tmp, e->v.B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    AZ_HIP(hipStreamSynchronize(s));
    return AZ_OK;
}

static int set_positions_impl(az_engine* e, int32_t n, const int32_t* slots_host,
                              const uint64_t* white_host, const uint64_t* black_host,
                              const int8_t* turn_host, const int8_t* winner_host, void* stream,
                              int keep_tree) {
    AZ_CHECK_ARG(e && n >= 0 && n <= e->v.B, "az_engine_set_positions: bad n");
    if (n == 0) return AZ_OK;
    AZ_CHECK_ARG(slots_host && white_host && black_host && turn_host, "az_engine_set_positions: null pointer");
    for (int i = 0; i < n; ++i)
        AZ_CHECK_ARG(slots_host[i] >= 0 && slots_host[i] < e->v.B, "az_engine_set_positions: slot %d out of range",
                     slots_host[i]);
    hipStream_t s = (hipStream_t)stream;
    std::vector<int8_t> zero;
    if (!winner_host) { zero.assign(n, 0); winner_host = zero.data(); }
    AZ_HIP(hipMemcpyAsync(e->stage_slots, slots_host, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    AZ_HIP(hipMemcpyAsync(e->stage_w, white_host, n * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    AZ_HIP(hipMemcpyAsync(e->stage_b, black_host, n * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    AZ_HIP(hipMemcpyAsync(e->stage_t, turn_host, n, hipMemcpyHostToDevice, s));
    AZ_HIP(hipMemcpyAsync(e->stage_win, winner_host, n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_set_positions, dim3((n + 255) / 256), dim3(256), 0, s, e->v, n, e->stage_slots,
                       e->stage_w, e->stage_b, e->stage_t, e->stage_win, keep_tree);
    AZ_HIP(hipGetLastError());
    AZ_HIP(hipStreamSynchronize(s));  // the staging buffers and host arrays are free again
    return AZ_OK;
}

extern "C" int az_engine_get_positions(az_engine* e, uint64_t* white_host, uint64_t* black_host,
                                       int8_t* turn_host, int8_t* winner_host, void* stream) {
    AZ_CHECK_ARG(e, "az_engine_get_positions: null engine");
    hipStream_t s = (hipStream_t)stream;
    const size_t B = e->v.B;
    if (white_host) AZ_HIP(hipMemcpyAsync(white_host, e->v.white, B * 8, hipMemcpyDeviceToHost, s));
    if (black_host) AZ_HIP(hipMemcpyAsync(black_host, e->v.black, B * 8, hipMemcpyDeviceToHost, s));
    if (turn_host) AZ_HIP(hipMemcpyAsync(turn_host, e->v.turn, B, hipMemcpyDeviceToHost, s));
    if (winner_host) AZ_HIP(hipMemcpyAsync(winner_host, e->v.winner, B, hipMemcpyDeviceToHost, s));
    AZ_HIP(hipStreamSynchronize(s));
    return AZ_OK;
}

extern "C" int az_engine_set_active(az_engine* e, const uint8_t* active_host, void* stream) {
    AZ_CHECK_ARG(e && active_host, "az_engine_set_active: null argument");
    hipStream_t s = (hipStream_t)stream;
    AZ_HIP(hipMemcpyAsync(e->v.active, active_host, e->v.B, hipMemcpyHostToDevice, s));
    AZ_HIP(hipStreamSynchronize(s));
    return AZ_OK;
}

extern "C" int az_engine_roots_prepare(az_engine* e, float* planes_dev, uint8_t* need_eval_dev,
                                       void* stream) {
    AZ_CHECK_ARG(e, "az_engine_roots_prepare: null engine");
    AZ_CHECK_ARG(((uintptr_t)planes_dev & 15) == 0, "az_engine_roots_prepare: planes must be 16-byte aligned");
    hipLaunchKernelGGL(k_roots_prepare, dim3(e->v.B), dim3(64), 0, (hipStream_t)stream, e->v, planes_dev,
                       need_eval_dev);
    AZ_HIP(hipGetLastError());
    e->phase = PHASE_ROOTS;
    return AZ_OK;
}

extern "C" int az_engine_roots_expand(az_engine* e, const float* probs_dev, void* stream) {
    AZ_CHECK_ARG(e && probs_dev, "az_engine_roots_expand: null argument");
    if (e->phase != PHASE_ROOTS) AZ_FAIL(AZ_ERR_STATE, "az_engine_roots_expand without az_engine_roots_prepare");
    hipLaunchKernelGGL(k_expand_backup, dim3(e->v.B), dim3(64), 0, (hipStream_t)stream, e->v, probs_dev,
                       (const float*)nullptr, 0);
    AZ_HIP(hipGetLastError());
    e->phase = PHASE_IDLE;
    return AZ_OK;
}

extern "C" int az_engine_add_noise(az_engine* e, const double* noise_dev, void* stream) {
    AZ_CHECK_ARG(e, "az_engine_add_noise: null engine");
    hipLaunchKernelGGL(k_add_noise, dim3(e->v.B), dim3(64), 0, (hipStream_t)stream, e->v, noise_dev,
                       ++e->noise_calls);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

extern "C" int az_engine_select(az_engine* e, float* planes_dev, uint8_t* need_eval_dev, void* stream) {
    AZ_CHECK_ARG(e, "az_engine_select: null engine");
    AZ_CHECK_ARG(((uintptr_t)planes_dev & 15) == 0, "az_engine_select: planes must be 16-byte aligned");
    hipLaunchKernelGGL(k_select, dim3(e->v.B), dim3(64), 0, (hipStream_t)stream, e->v, planes_dev,
                       need_eval_dev);
    AZ_HIP(hipGetLastError());
    e->phase = PHASE_SELECTED;
    return AZ_OK;
}

extern "C" int az_engine_expand_backup(az_engine* e, const float* probs_dev, const float* values_dev,
                                       void* stream) {
    AZ_CHECK_ARG(e && probs_dev && values_dev, "az_engine_expand_backup: null argument");
    if (e->phase != PHASE_SELECTED) AZ_FAIL(AZ_ERR_STATE, "az_engine_expand_backup without az_engine_select");
    hipLaunchKernelGGL(k_expand_backup, dim3(e->v.B), dim3(64), 0, (hipStream_t)stream, e->v, probs_dev,
                       values_dev, 1);
    AZ_HIP(hipGetLastError());
    e->phase = PHASE_IDLE;
    return AZ_OK;
}

extern "C" int az_engine_node_children(az_engine* e, int32_t g, int64_t node, int32_t* k_host,
                                       uint8_t* actions_host, uint32_t* visits_host,
                                       float* value_sum_host, double* prior64_host,
                                       uint8_t* prior_is64_host, int64_t* child_index_host,
                                       uint32_t* node_visits_host, float* node_value_sum_host,
                                       int64_t* node_index_host, void* stream) {
    AZ_CHECK_ARG(e && g >= 0 && g < e->v.B, "az_engine_node_children: game %d out of range", g);
    AZ_CHECK_ARG(node < (int64_t)e->v.cap, "az_engine_node_children: node out of range");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_node_children, dim3(1), dim3(64), 0, s, e->v, g, node, e->query_dev);
    AZ_HIP(hipGetLastError());
    AZ_HIP(hipMemcpyAsync(e->query_host, e->query_dev, sizeof(NodeQuery), hipMemcpyDeviceToHost, s));
    const int rc = check_err(e, s);  // synchronises
    if (rc != AZ_OK) return rc;
    const NodeQuery& q = *e->query_host;
    const int k = q.k < 0 ? 0 : q.k;
    if (k_host) *k_host = q.k;
    if (node_visits_host) *node_visits_host = q.node_visits;
    if (node_value_sum_host) *node_value_sum_host = q.node_value_sum;
    if (node_index_host) *node_index_host = q.node_index;
    for (int i = 0; i < k; ++i) {
        if (actions_host) actions_host[i] = q.actions[i];
        if (visits_host) visits_host[i] = q.visits[i];
        if (value_sum_host) value_sum_host[i] = q.value_sum[i];
        if (prior64_host) prior64_host[i] = q.prior64[i];
        if (prior_is64_host) prior_is64_host[i] = q.prior_is64[i];
        if (child_index_host) child_index_host[i] = q.child_index[i];
    }
    return AZ_OK;
}

extern "C" int az_engine_roots_children(az_engine* e, int32_t n, const int32_t* slots_host, int32_t* k_host,
                                        uint8_t* actions_host, uint32_t* visits_host, float* value_sum_host,
                                        double* prior64_host, uint8_t* prior_is64_host,
                                        int64_t* child_index_host, uint32_t* node_visits_host,
                                        float* node_value_sum_host, int64_t* node_index_host, void* stream) {
    AZ_CHECK_ARG(e && n >= 0 && n <= e->v.B, "az_engine_roots_children: n = %d out of range", n);
    if (n == 0) return AZ_OK;
    AZ_CHECK_ARG(slots_host, "az_engine_roots_children: null slots");
    for (int i = 0; i < n; ++i)
        AZ_CHECK_ARG(slots_host[i] >= 0 && slots_host[i] < e->v.B, "az_engine_roots_children: game %d out of range", slots_host[i]);
    hipStream_t s = (hipStream_t)stream;
    if (!e->query_batch_dev) {   // B records, allocated on first use
        if (!e->query_batch_host)
            AZ_HIP(hipHostMalloc((void**)&e->query_batch_host, (size_t)e->v.B * sizeof(NodeQuery), hipHostMallocDefault));
        void* p = nullptr;
        AZ_HIP(hipMalloc(&p, (size_t)e->v.B * sizeof(NodeQuery)));
        e->allocs.push_back(p);
        e->query_batch_dev = (NodeQuery*)p;   // set last: both buffers exist whenever this one does
    }
    AZ_HIP(hipMemcpyAsync(e->stage_slots, slots_host, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_roots_children, dim3(n), dim3(64), 0, s, e->v, e->stage_slots, e->query_batch_dev);
    AZ_HIP(hipGetLastError());
    AZ_HIP(hipMemcpyAsync(e->query_batch_host, e->query_batch_dev, (size_t)n * sizeof(NodeQuery), hipMemcpyDeviceToHost, s));
    const int rc = check_err(e, s);  // synchronises
    if (rc != AZ_OK) return rc;
    for (int i = 0; i < n; ++i) {
        const NodeQuery& q = e->query_batch_host[i];
        const int k = q.k < 0 ? 0 : q.k;
        if (k_host) k_host[i] = q.k;
        if (node_visits_host) node_visits_host[i] = q.k < 0 ? 0 : q.node_visits;
        if (node_value_sum_host) node_value_sum_host[i] = q.k < 0 ? 0.f : q.node_value_sum;
        if (node_index_host) node_index_host[i] = q.node_index;
        const size_t o = (size_t)i * MAX_CHILDREN;
        for (int j = 0; j < k; ++j) {
            if (actions_host) actions_host[o + j] = q.actions[j];
            if (visits_host) visits_host[o + j] = q.visits[j];
            if (value_sum_host) value_sum_host[o + j] = q.value_sum[j];
            if (prior64_host) prior64_host[o + j] = q.prior64[j];
            if (prior_is64_host) prior_is64_host[o + j] = q.prior_is64[j];
            if (child_index_host) child_index_host[o + j] = q.child_index[j];
        }
    }
    return AZ_OK;
}

extern "C" int az_engine_action_probs(az_engine* e, const float* temps_host, float* probs_dev,
                                      void* stream) {
    AZ_CHECK_ARG(e && probs_dev, "az_engine_action_probs: null argument");
    hipStream_t s = (hipStream_t)stream;
    const float* temps_dev = nullptr;
    if (temps_host) {
        for (int i = 0; i < e->v.B; ++i)
            AZ_CHECK_ARG(temps_host[i] >= 0.0f, "az_engine_action_probs: negative temperature %g", temps_host[i]);
        AZ_HIP(hipMemcpyAsync(e->stage_temps, temps_host, e->v.B * sizeof(float), hipMemcpyHostToDevice, s));
        temps_dev = e->stage_temps;
    }
    hipLaunchKernelGGL(k_action_probs, dim3(e->v.B), dim3(64), 0, s, e->v, temps_dev, probs_dev);
    AZ_HIP(hipGetLastError());
    if (temps_host) AZ_HIP(hipStreamSynchronize(s));
    return AZ_OK;
}

extern "C" int az_engine_advance(az_engine* e, const int16_t* actions_host, void* stream) {
    AZ_CHECK_ARG(e && actions_host, "az_engine_advance: null argument");
    for (int i = 0; i < e->v.B; ++i)
        AZ_CHECK_ARG(actions_host[i] < 192, "az_engine_advance: action %d out of range", actions_host[i]);
    hipStream_t s = (hipStream_t)stream;
    AZ_HIP(hipMemcpyAsync(e->stage_act, actions_host, e->v.B * sizeof(int16_t), hipMemcpyHostToDevice, s));
    const int rc = az::advance_dev(e, e->stage_act, s);
    if (rc != AZ_OK) return rc;
    AZ_HIP(hipStreamSynchronize(s));
    return AZ_OK;
}

int az::advance_dev(az_engine* e, const int16_t* actions_dev, hipStream_t s) {
    hipLaunchKernelGGL(k_advance, dim3(e->v.B), dim3(64), 0, s, e->v, actions_dev);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

extern "C" int az_engine_set_root(az_engine* e, int32_t g, int64_t node, void* stream) {
    AZ_CHECK_ARG(e && g >= 0 && g < e->v.B, "az_engine_set_root: game %d out of range", g);
    AZ_CHECK_ARG(node < (int64_t)e->v.cap, "az_engine_set_root: node out of range");
    hipLaunchKernelGGL(k_set_root, dim3(1), dim3(64), 0, (hipStream_t)stream, e->v, g, node);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

extern "C" int az_engine_set_roots(az_engine* e, int32_t n, const int32_t* slots_host, const int64_t* nodes_host,
                                   void* stream) {
    AZ_CHECK_ARG(e && n >= 0 && n <= e->v.B, "az_engine_set_roots: n = %d out of range", n);
    if (n == 0) return AZ_OK;
    AZ_CHECK_ARG(slots_host && nodes_host, "az_engine_set_roots: null array");
    for (int i = 0; i < n; ++i) {
        AZ_CHECK_ARG(slots_host[i] >= 0 && slots_host[i] < e->v.B, "az_engine_set_roots: game %d out of range", slots_host[i]);
        AZ_CHECK_ARG(nodes_host[i] < (int64_t)e->v.cap, "az_engine_set_roots: node out of range");
    }
    hipStream_t s = (hipStream_t)stream;
    int64_t* nodes_dev = reinterpret_cast<int64_t*>(e->stage_w);
    AZ_HIP(hipMemcpyAsync(e->stage_slots, slots_host, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    AZ_HIP(hipMemcpyAsync(nodes_dev, nodes_host, n * sizeof(int64_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_set_roots, dim3(n), dim3(64), 0, s, e->v, n, e->stage_slots, nodes_dev);
    AZ_HIP(hipGetLastError());
    AZ_HIP(hipStreamSynchronize(s));  // the staging buffers and host arrays are free again
    return AZ_OK;
}

extern "C" int az_engine_import_tree(az_engine* e, int32_t g, int32_t n_nodes, const uint32_t* visits_host,
                                     const float* value_sum_host, const float* prior_host,
                                     const uint32_t* first_child_host, const uint8_t* n_children_host,
                                     const uint8_t* action_host, const uint8_t* uniform64_host,
                                     const double* root_prior64_host, void* stream) {
    AZ_CHECK_ARG(e, "az_engine_import_tree: null engine");
    AZ_CHECK_ARG(g >= 0 && g < e->v.B, "az_engine_import_tree: game %d out of range", g);
    AZ_CHECK_ARG(n_nodes >= 1 && (uint32_t)n_nodes <= e->v.cap, "az_engine_import_tree: %d nodes, the pool holds %u per game",
                 n_nodes, e->v.cap);
    AZ_CHECK_ARG(visits_host && value_sum_host && prior_host && first_child_host && n_children_host && action_host,
                 "az_engine_import_tree: null array");
    // every node's children must be one block behind the node itself, inside the tree; every
    // block disjoint from the others (breadth-first order gives that: blocks in ascending order)
    std::vector<uint32_t> info(n_nodes);
    uint32_t next_block = 1;
    for (int i = 0; i < n_nodes; ++i) {
        const uint32_t k = n_children_host[i], fc = first_child_host[i];
        AZ_CHECK_ARG(k <= (uint32_t)MAX_CHILDREN, "az_engine_import_tree: node %d has %u children (at most %d)", i, k, MAX_CHILDREN);
        AZ_CHECK_ARG(action_host[i] < NUM_ACTIONS || i == 0, "az_engine_import_tree: node %d has action %u", i, action_host[i]);
        if (k) {
            AZ_CHECK_ARG(fc == next_block && (uint64_t)fc + k <= (uint64_t)n_nodes,
                         "az_engine_import_tree: the children of node %d are not the next block of the breadth-first order", i);
            next_block = fc + k;
        }
        const uint32_t flags = (uniform64_host && uniform64_host[i] && k) ? FLAG_UNIFORM64 : 0u;
        info[i] = (uint32_t)(action_host[i] % NUM_ACTIONS) | (k << 8) | (flags << 16);
    }
    AZ_CHECK_ARG(next_block == (uint32_t)n_nodes, "az_engine_import_tree: %d nodes given, %u reachable", n_nodes, next_block);
    hipStream_t s = (hipStream_t)stream;
    uint8_t* arena_dev = reinterpret_cast<uint8_t*>(e->stage_win);
    uint8_t arena = 0;
    hipLaunchKernelGGL(k_read_arena, dim3(1), dim3(1), 0, s, e->v, g, arena_dev);
    AZ_HIP(hipGetLastError());
    AZ_HIP(hipMemcpyAsync(&arena, arena_dev, 1, hipMemcpyDeviceToHost, s));
    AZ_HIP(hipStreamSynchronize(s));
    const size_t base = ((size_t)g * 2 + (arena & 1)) * e->v.cap;
    const size_t n = (size_t)n_nodes;
    AZ_HIP(hipMemcpyAsync(e->v.N + base, visits_host, n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    AZ_HIP(hipMemcpyAsync(e->v.W + base, value_sum_host, n * sizeof(float), hipMemcpyHostToDevice, s));
    AZ_HIP(hipMemcpyAsync(e->v.P + base, prior_host, n * sizeof(float), hipMemcpyHostToDevice, s));
    AZ_HIP(hipMemcpyAsync(e->v.FC + base, first_child_host, n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    AZ_HIP(hipMemcpyAsync(e->v.IN + base, info.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (root_prior64_host)
        AZ_HIP(hipMemcpyAsync(e->v.root_p64 + (size_t)g * MAX_CHILDREN, root_prior64_host,
                              (size_t)n_children_host[0] * sizeof(double), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_import_tree, dim3(1), dim3(64), 0, s, e->v, g, (uint32_t)n_nodes, root_prior64_host ? 1 : 0);
    AZ_HIP(hipGetLastError());
    AZ_HIP(hipStreamSynchronize(s));  // the host arrays are free again
    return AZ_OK;
}

extern "C" int az_engine_pool_usage(az_engine* e, uint32_t* used_host, void* stream) {
    AZ_CHECK_ARG(e, "az_engine_pool_usage: null engine");
    hipStream_t s = (hipStream_t)stream;
    if (used_host) AZ_HIP(hipMemcpyAsync(used_host, e->v.used, e->v.B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    return check_err(e, s);
}

extern "C" int az_engine_stats(az_engine* e, uint64_t* depth_sum_host, uint64_t* descents_host,
                               void* stream) {
    AZ_CHECK_ARG(e, "az_engine_stats: null engine");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long c[4];
    const int rc = az::engine_counters(e, c, s);
    if (rc != AZ_OK) return rc;
    if (depth_sum_host) *depth_sum_host = c[0];
    if (descents_host) *descents_host = c[1];
    return AZ_OK;
}

int az::engine_counters(az_engine* e, unsigned long long out[4], hipStream_t s) {
    const size_t B = e->v.B;
    std::vector<unsigned long long> host(4 * B);
    const unsigned long long* src[4] = {e->v.stat_depth, e->v.stat_descents, e->v.stat_terminal, e->v.stat_rootexp};
    for (int i = 0; i < 4; ++i)
        AZ_HIP(hipMemcpyAsync(host.data() + i * B, src[i], B * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    AZ_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < 4; ++i) {
        unsigned long long t = 0;
        for (size_t g = 0; g < B; ++g) t += host[i * B + g];
        out[i] = t;
    }
    return AZ_OK;
}
