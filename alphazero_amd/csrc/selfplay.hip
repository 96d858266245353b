// Fused self-play loop: search with the attached network, then the per-move
// bookkeeping of ParallelTrainer.execute_parallel_episodes
// (src/parallel_trainer.py:158-217) entirely on the device.
//
// Compile with -ffp-contract=off (shares the exact float32 rules of tree.hip).
#include "net_internal.h"
#include "tree_dev.h"

extern "C" int az_net_forward(struct az_net* n, int32_t B, const uint64_t* mine_dev,
                              const uint64_t* theirs_dev, float* probs_dev, float* value_dev,
                              float* policy_logits_dev, float* wl_logits_dev, void* stream);
extern "C" int az_net_max_batch(struct az_net* n);

namespace az {

// Record (state, pi, turn) before moving (src/parallel_trainer.py:185-186) and
// choose the move: np.random.choice(192, p=pi) (:189) drawn on the device, or an
// injected action.
__global__ __launch_bounds__(64) void k_record_and_choose(EngineView E, SelfplayView S,
                                                          const int16_t* injected, uint64_t call_id) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (!E.active[g]) { if (lane == 0) S.chosen[g] = -1; return; }
    const int ply = E.ply[g];
    const float* pi = S.pi + (size_t)g * 192;
    const float p0 = pi[3 * lane], p1 = pi[3 * lane + 1], p2 = pi[3 * lane + 2];
    if (ply < MAX_PLIES) {
        const size_t hrow = (size_t)g * MAX_PLIES + ply;
        float* hp = S.hist_pi + hrow * 192;
        hp[lane] = pi[lane]; hp[64 + lane] = pi[64 + lane]; hp[128 + lane] = pi[128 + lane];
        if (lane == 0) {
            const uint64_t w = E.white[g], b = E.black[g];
            const int t = E.turn[g];
            S.hist_mine[hrow] = (t > 0) ? w : bitrev64(b);
            S.hist_theirs[hrow] = (t > 0) ? b : bitrev64(w);
            S.hist_turn[hrow] = (int8_t)t;
        }
    } else if (lane == 0) {
        atomicOr(E.err, ERR_HISTORY);
    }
    int action;
    if (injected) {
        action = injected[g];
    } else {
        // inverse-CDF draw in float64 over the 192 probabilities (numpy's choice
        // normalises the cumulative sum by its last entry, then searches it)
        const double d0 = (double)p0, d1 = (double)p1, d2 = (double)p2;
        double incl = d0 + d1 + d2;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const double total = __shfl(incl, 63);
        const double excl = incl - (d0 + d1 + d2);
        Rng rng(E.seed ^ 0xA5A5A5A5ull, (uint64_t)g, call_id, 0x77ull);
        const double u = rng.uniform() * total;
        // first index whose cumulative sum exceeds u, among entries with p > 0
        int local = -1;
        if (d0 > 0 && excl + d0 > u) local = 0;
        else if (d1 > 0 && excl + d0 + d1 > u) local = 1;
        else if (d2 > 0 && incl > u) local = 2;
        const unsigned long long hit = __ballot(local >= 0);
        int src;
        if (hit) src = __ffsll((long long)hit) - 1;
        else {  // u landed on the upper edge: take the last entry with mass
            const unsigned long long any = __ballot(d0 + d1 + d2 > 0);
            src = any ? 63 - __clzll((long long)any) : 0;
            local = (d2 > 0) ? 2 : ((d1 > 0) ? 1 : 0);
        }
        action = __shfl(3 * lane + (local < 0 ? 0 : local), src);
    }
    if (lane == 0) S.chosen[g] = (int16_t)action;
}

// After the move: which games have finished, and where their samples go in the queue.
// The reference walks its active list in game order and appends a finished game's samples
// as it meets it (src/parallel_trainer.py:176-217), so within a ply samples are queued in
// ascending game index: queue space is handed out by an exclusive scan over the games,
// never by the order in which workgroups happen to arrive.  One workgroup of 1024 lanes,
// 1024 games per pass.  A game whose samples do not fit the queue is marked as such
// (ERR_HISTORY) and the queue length is left at the first game that did not fit; space is
// never reserved and given back.
__global__ __launch_bounds__(1024) void k_finish_scan(EngineView E, SelfplayView S) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry, s_limit;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_carry = *S.q_count; s_limit = 0xFFFFFFFFu; }
    __syncthreads();
    for (int g0 = 0; g0 < E.B; g0 += 1024) {
        const int g = g0 + tid;
        uint32_t n = 0;
        int fin = FIN_NONE;
        if (g < E.B && E.active[g] && S.chosen[g] >= 0) {
            const uint64_t w = E.white[g], b = E.black[g];
            const int t = E.turn[g], win = E.winner[g];
            bool terminal = (win != 0);
            int white_won = (win > 0);
            if (!terminal) {
                const uint64_t mine = (t > 0) ? w : bitrev64(b), theirs = (t > 0) ? b : bitrev64(w);
                const Legal3 l = legal_sources(mine, theirs);
                if ((l.fwd | l.dl | l.dr) == 0) { terminal = true; white_won = (t < 0); }  // side to move loses
            }
            if (terminal) { fin = white_won ? FIN_WHITE : FIN_BLACK; n = min((uint32_t)E.ply[g], (uint32_t)MAX_PLIES); }
        }
        uint32_t incl = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const uint32_t x = s_wave[i]; total += x; if (i < wave) before += x; }
        const uint32_t base = s_carry + before + incl - n;
        if (g < E.B) {
            const bool fits = (uint64_t)base + n <= (uint64_t)S.q_cap;
            if (fin != FIN_NONE && !fits) { fin |= FIN_DROPPED; atomicMin(&s_limit, base); }
            S.fin_flag[g] = (uint8_t)fin;
            S.fin_base[g] = base;
        }
        __syncthreads();
        if (tid == 0) s_carry += total;
        __syncthreads();
    }
    if (tid == 0) {
        if (s_limit != 0xFFFFFFFFu) { atomicOr(E.err, ERR_HISTORY); *S.q_count = s_limit; }
        else *S.q_count = s_carry;
    }
}

// Finished games emit their samples (src/parallel_trainer.py:201-213) at the queue position
// k_finish_scan gave them, and either restart from the initial position or leave the
// active set (:216-217).
__global__ __launch_bounds__(64) void k_finish_games(EngineView E, SelfplayView S, int restart) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (!E.active[g] || S.chosen[g] < 0) return;
    if (lane == 0) atomicAdd(&S.stat[0], 1ull);
    const int fin = S.fin_flag[g];
    if (fin == FIN_NONE) return;
    if (!(fin & FIN_DROPPED)) {
        const int n = min((int)E.ply[g], MAX_PLIES);
        const uint32_t base = S.fin_base[g];
        const bool white_won = (fin & FIN_WHITE) != 0;
        const float wv = white_won ? 1.f : 0.f, lv = white_won ? 0.f : 1.f;
        for (int i = 0; i < n; ++i) {
            const size_t hrow = (size_t)g * MAX_PLIES + i, q = (size_t)base + i;
            const float* hp = S.hist_pi + hrow * 192;
            float* qp = S.q_pi + q * 192;
            qp[lane] = hp[lane]; qp[64 + lane] = hp[64 + lane]; qp[128 + lane] = hp[128 + lane];
            if (lane == 0) {
                S.q_mine[q] = S.hist_mine[hrow]; S.q_theirs[q] = S.hist_theirs[hrow];
                const bool white_moved = S.hist_turn[hrow] > 0;
                S.q_wl[2 * q] = white_moved ? wv : lv;
                S.q_wl[2 * q + 1] = white_moved ? lv : wv;
            }
        }
    }
    if (lane == 0) {
        atomicAdd(&S.stat[1], 1ull);
        if (restart) {
            E.white[g] = 0x000000000000FFFFull; E.black[g] = 0xFFFF000000000000ull;
            E.turn[g] = 1; E.winner[g] = 0; E.ply[g] = 0;
            E.root[g] = NODE_NONE; E.used[g] = 0; E.root_mode[g] = 0;
        } else {
            E.active[g] = 0;
        }
    }
}

// Unpack queued samples into the reference's float layout (states [n][3][8][8],
// policies [n][192], wls [n][2]; src/parallel_trainer.py:380-382).
__global__ void k_unpack_samples(SelfplayView S, uint32_t n, float* states, float* policies, float* wls) {
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    if (q >= n) return;
    const uint64_t mine = S.q_mine[q], theirs = S.q_theirs[q];
    if (lane < 48) {
        const int pl = lane >> 4, s0 = (4 * lane) & 63;
        const uint64_t src = (pl == 0) ? mine : ((pl == 1) ? theirs : ~0ull);
        float4 v;
        v.x = (float)((src >> (s0 + 0)) & 1ull); v.y = (float)((src >> (s0 + 1)) & 1ull);
        v.z = (float)((src >> (s0 + 2)) & 1ull); v.w = (float)((src >> (s0 + 3)) & 1ull);
        reinterpret_cast<float4*>(states + (size_t)q * 192)[lane] = v;
    }
    for (int i = lane; i < 192; i += 64) policies[(size_t)q * 192 + i] = S.q_pi[(size_t)q * 192 + i];
    if (lane < 2) wls[2 * (size_t)q + lane] = S.q_wl[2 * (size_t)q + lane];
}

// Wire format of one sample for the exchange between GPUs: 792 bytes =
// mine u64 | theirs u64 | wl f32[2] | pi f32[192]  (half the 1544 bytes of the float planes).
__global__ __launch_bounds__(64) void k_pack_samples(SelfplayView S, uint32_t n, uint8_t* packed) {
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    if (q >= n) return;
    uint8_t* row = packed + (size_t)q * AZ_PACKED_SAMPLE_BYTES;
    if (lane == 0) {
        reinterpret_cast<uint64_t*>(row)[0] = S.q_mine[q];
        reinterpret_cast<uint64_t*>(row)[1] = S.q_theirs[q];
    }
    float* f = reinterpret_cast<float*>(row + 16);
    if (lane < 2) f[lane] = S.q_wl[2 * (size_t)q + lane];
    for (int i = lane; i < 192; i += 64) f[2 + i] = S.q_pi[(size_t)q * 192 + i];
}

__global__ __launch_bounds__(64) void k_unpack_packed(const uint8_t* packed, uint32_t n, float* states, float* policies,
                                                      float* wls) {
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    if (q >= n) return;
    const uint8_t* row = packed + (size_t)q * AZ_PACKED_SAMPLE_BYTES;
    const uint64_t mine = reinterpret_cast<const uint64_t*>(row)[0], theirs = reinterpret_cast<const uint64_t*>(row)[1];
    write_planes(states + (size_t)q * 192, mine, theirs, lane);
    const float* f = reinterpret_cast<const float*>(row + 16);
    if (lane < 2) wls[2 * (size_t)q + lane] = f[lane];
    for (int i = lane; i < 192; i += 64) policies[(size_t)q * 192 + i] = f[2 + i];
}

// One simulation's tree work for one game in ONE launch: finish the network heads for
// the leaf (softmax / value), expand it, back the value up, then run the next
// simulation's PUCT descent (src/mcts.py:153-187).
//
// The kernel is a chain of dependent memory round trips, so every load whose address is
// known is issued up front, in three stages: (1) the game's scalars, the recorded path and
// the head partial sums; (2) the root's fields and the visit/value words of the path nodes;
// (3) the root's child statistics.  Expansion and backup then only store, and the next
// descent starts from registers patched with those stores; its first memory round trip
// is the second tree level.
__global__ __launch_bounds__(64) void k_tree_step(EngineView E, FinishParams F, int do_select) {
    __shared__ float s_probs[192];
    __shared__ float s_p[MAX_CHILDREN];
    __shared__ uint8_t s_a[MAX_CHILDREN];
    const int g = blockIdx.x, lane = threadIdx.x;
    // ---- load stage 1
    const int active = E.active[g];
    const int need = E.need_cur[g];
    float v = E.leaf_tv[g];
    const SelectStart st = load_select_start(E, g);
    const size_t pool = pool_base(E, g);
    const int len = E.path_len[g];
    const uint32_t* path = E.path + (size_t)g * MAX_DEPTH;
    const uint32_t my_path = path[lane];           // MAX_DEPTH >= 64: always in bounds
    unsigned long long stat_descents = 0;
    RootPrefetch pre;
    pre.stat_depth = E.stat_depth[g]; pre.stat_terminal = E.stat_terminal[g];
    if (lane == 0) stat_descents = E.stat_descents[g];
    if (!active) return;
    LeafState lf;
    lf.node = E.leaf_node[g]; lf.mine = E.leaf_mine[g]; lf.theirs = E.leaf_theirs[g];
    lf.turn = E.leaf_turn[g]; lf.used = E.used[g];
    HeadOperands ho;
    const int row = (E.row_of && need) ? E.row_of[g] : g;   // written by this simulation's network launch
    if (need) ho = head_load(F, row, lane);
    // ---- load stage 2
    const bool on_path = lane < len;
    const size_t my_node = pool + (on_path ? my_path : st.root);
    uint32_t bN = E.N[my_node];
    float bW = E.W[my_node];
    lf.info = E.IN[pool + lf.node];
    pre.inf = E.IN[pool + st.root]; pre.fc = E.FC[pool + st.root];
    // ---- load stage 3
    const int rk = (pre.inf >> 8) & 0xFF;
    if (do_select) pre.c = load_children(E, pool, g, pre.fc, rk, lane, st.rmode != 0);

    Expansion ex{0, 0, lf.info};
    if (need) {
        float pr[3];
        v = head_compute(F, ho, row, lane, pr);
        s_probs[lane] = pr[0]; s_probs[lane + 64] = pr[1]; s_probs[lane + 128] = pr[2];
        __syncthreads();
        ex = dev_expand(E, g, lane, pool, lf, s_probs, s_p, s_a);
    }
    // _backpropagate (src/mcts.py:337-342) on the preloaded words; deeper levels (rare) as usual
    if (on_path) {
        const float vv = ((len - 1 - lane) & 1) ? -v : v;
        bN = bN + 1; bW = bW + vv;
        E.N[my_node] = bN; E.W[my_node] = bW;
    }
    for (int i = lane + 64; i < len; i += 64) {
        const size_t n = pool + path[i];
        const float vv = ((len - 1 - i) & 1) ? -v : v;
        E.N[n] = E.N[n] + 1;
        E.W[n] = E.W[n] + vv;
    }
    if (lane == 0) { E.stat_descents[g] = stat_descents + 1ull; E.need_cur[g] = 0; }
    if (!do_select) return;
    __syncthreads();  // this wave's tree updates are visible to its own descent
    if (len == 1) {
        // the leaf was the root itself: its children did not exist when load stage 3 was issued
        pre.inf = ex.info; pre.fc = ex.first; pre.pn = __shfl(bN, 0);
        const int k = (pre.inf >> 8) & 0xFF;
        pre.c = load_children(E, pool, g, pre.fc, k, lane, st.rmode != 0);
    } else {
        // patch the prefetched registers with what this launch stored: the root's visit, the
        // visit and value of the root child on the path, and that child's expansion
        pre.pn = __shfl(bN, 0);
        const uint32_t child = __shfl(my_path, 1);
        const uint32_t cN1 = __shfl(bN, 1);
        const float cW1 = __shfl(bW, 1);
        if (pre.fc + (uint32_t)lane == child) {
            pre.c.cN = cN1; pre.c.cW = cW1;
            if (len == 2 && ex.k > 0) { pre.c.cinfo = ex.info; pre.c.cfc = ex.first; }
        }
    }
    dev_select(E, g, lane, st, nullptr, nullptr, &pre);
}

}  // namespace az

using namespace az;

static int selfplay_alloc(az_engine* e) {
    if (e->sp_ready) return AZ_OK;
    SelfplayView& S = e->sp;
    const size_t B = e->v.B;
    S.q_cap = (uint32_t)(B * 160 < (1u << 16) ? (1u << 16) : B * 160);
#define AZ_ALLOC(ptr, count)                                       \
    do {                                                           \
        void* _p = nullptr;                                        \
        AZ_HIP(hipMalloc(&_p, (count) * sizeof(*(ptr))));          \
        e->allocs.push_back(_p);                                   \
        AZ_HIP(hipMemset(_p, 0, (count) * sizeof(*(ptr))));        \
        (ptr) = (decltype(ptr))_p;                                 \
    } while (0)
    AZ_ALLOC(S.pi, B * 192); AZ_ALLOC(S.chosen, B);
    AZ_ALLOC(S.hist_mine, B * MAX_PLIES); AZ_ALLOC(S.hist_theirs, B * MAX_PLIES);
    AZ_ALLOC(S.hist_turn, B * MAX_PLIES); AZ_ALLOC(S.hist_pi, B * MAX_PLIES * 192);
    AZ_ALLOC(S.q_mine, (size_t)S.q_cap); AZ_ALLOC(S.q_theirs, (size_t)S.q_cap);
    AZ_ALLOC(S.q_wl, (size_t)S.q_cap * 2); AZ_ALLOC(S.q_pi, (size_t)S.q_cap * 192);
    AZ_ALLOC(S.q_count, 1); AZ_ALLOC(S.stat, 4);
    AZ_ALLOC(S.fin_base, B); AZ_ALLOC(S.fin_flag, B);
    AZ_ALLOC(e->nn_probs, B * 192); AZ_ALLOC(e->nn_values, B);
    AZ_ALLOC(e->row_of_dev, B); AZ_ALLOC(e->n_rows_dev, 1); AZ_ALLOC(e->rows_total_dev, 1);
#undef AZ_ALLOC
    e->sp_ready = true;
    return AZ_OK;
}

extern "C" int az_engine_attach_net(az_engine* e, az_net* n) {
    AZ_CHECK_ARG(e && n, "az_engine_attach_net: null argument");
    AZ_CHECK_ARG(az_net_max_batch(n) >= e->v.B, "az_engine_attach_net: network max_batch %d < n_games %d",
                 az_net_max_batch(n), e->v.B);
    // the fused search hands the engine's device arrays to the network's kernels
    AZ_CHECK_ARG(az::net_device(n) == e->device, "az_engine_attach_net: the network lives on device %d, the engine on device %d",
                 az::net_device(n), e->device);
    const int rc = selfplay_alloc(e);
    if (rc == AZ_OK) e->net = n;  // a network counts as attached only with the self-play buffers in place
    return rc;
}

static int check_sticky(az_engine* e, hipStream_t s) { return az::check_engine_error(e, s); }

// MCTS.search_batch (src/mcts.py:94-189) with the attached network.  Per simulation
// three launches: tower, head GEMMs, k_tree_step.
// The network evaluates only the leaves that need it: the tower assigns its rows from the
// games' flags on the device (RowAssign), so neither inactive games nor terminal leaves
// (src/mcts.py:183-187: backed up without an evaluation) occupy rows.  The host only
// sizes the launches, by the number of active games.
static int count_active(az_engine* e, hipStream_t s, int* rows_out) {
    const int B = e->v.B;
    std::vector<uint8_t> active(B);
    AZ_HIP(hipMemcpyAsync(active.data(), e->v.active, B, hipMemcpyDeviceToHost, s));
    AZ_HIP(hipStreamSynchronize(s));
    int M = 0;
    for (int g = 0; g < B; ++g) M += active[g] ? 1 : 0;
    *rows_out = M;
    return AZ_OK;
}

static int search_async(az_engine* e, int num_simulations, int add_noise, hipStream_t s) {
    if (!e->net) AZ_FAIL(AZ_ERR_STATE, "no network attached (az_engine_attach_net)");
    const int B = e->v.B;
    int rc, M = 0;
    if ((rc = count_active(e, s, &M))) return rc;
    if (M == 0) return AZ_OK;                       // no active game: nothing to search
    EngineView& V = e->v;
    // every kernel launched below takes V by value; the step API's view (one flag buffer, rows
    // are games) is put back before returning
    struct Restore {
        az_engine* e;
        ~Restore() { e->v.row_of = nullptr; e->v.leaf_need = e->v.need_cur = e->need_buf[0]; }
    } restore{e};
    V.row_of = e->row_of_dev;
    RowAssign rows{nullptr, B, e->row_of_dev, e->n_rows_dev, e->rows_total_dev};
    // roots: flags in buffer 1, which is all zero again once the roots are expanded
    V.leaf_need = V.need_cur = e->need_buf[1];
    rows.need = e->need_buf[1];
    if ((rc = az_engine_roots_prepare(e, nullptr, nullptr, s))) return rc;
    if ((rc = az::net_forward_full(e->net, M, V.leaf_mine, V.leaf_theirs, &rows, e->nn_probs, e->nn_values,
                                   nullptr, nullptr, s))) return rc;
    if ((rc = az_engine_roots_expand(e, e->nn_probs, s))) return rc;
    if (add_noise && (rc = az_engine_add_noise(e, nullptr, s))) return rc;
    // simulation i reads its flags from buffer i & 1 and selection writes the next ones into the
    // other buffer: the network launch of simulation i + 1 reads ALL games' flags while
    // k_tree_step blocks of simulation i may still be running ahead of one another
    V.leaf_need = V.need_cur = e->need_buf[0];
    if (num_simulations > 0 && (rc = az_engine_select(e, nullptr, nullptr, s))) return rc;
    const FinishParams fin = az::net_finish_params(e->net, M, e->n_rows_dev);
    for (int i = 0; i < num_simulations; ++i) {
        rows.need = e->need_buf[i & 1];
        if ((rc = az::net_forward_partial(e->net, M, V.leaf_mine, V.leaf_theirs, &rows, s))) return rc;
        V.need_cur = e->need_buf[i & 1];
        V.leaf_need = e->need_buf[(i + 1) & 1];
        hipLaunchKernelGGL(k_tree_step, dim3(B), dim3(64), 0, s, V, fin, i + 1 < num_simulations ? 1 : 0);
    }
    AZ_HIP(hipGetLastError());
    e->phase = PHASE_IDLE;
    return AZ_OK;
}

extern "C" int az_engine_search(az_engine* e, int32_t num_simulations, int32_t add_noise, void* stream) {
    AZ_CHECK_ARG(e && num_simulations >= 0, "az_engine_search: bad argument");
    const int rc = search_async(e, num_simulations, add_noise, (hipStream_t)stream);
    if (rc != AZ_OK) return rc;
    return check_sticky(e, (hipStream_t)stream);
}

extern "C" int az_engine_play_move(az_engine* e, int32_t num_simulations, int32_t add_noise,
                                   const int16_t* actions_host, int32_t restart_finished, void* stream) {
    AZ_CHECK_ARG(e && num_simulations >= 0, "az_engine_play_move: bad argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = search_async(e, num_simulations, add_noise, s);
    if (rc != AZ_OK) return rc;
    if ((rc = az_engine_action_probs(e, nullptr, e->sp.pi, s))) return rc;
    const int16_t* inj = nullptr;
    if (actions_host) {
        AZ_HIP(hipMemcpyAsync(e->stage_act, actions_host, e->v.B * sizeof(int16_t), hipMemcpyHostToDevice, s));
        inj = e->stage_act;
    }
    hipLaunchKernelGGL(k_record_and_choose, dim3(e->v.B), dim3(64), 0, s, e->v, e->sp, inj, ++e->move_calls);
    if ((rc = az::advance_dev(e, e->sp.chosen, s))) return rc;
    hipLaunchKernelGGL(k_finish_scan, dim3(1), dim3(1024), 0, s, e->v, e->sp);
    hipLaunchKernelGGL(k_finish_games, dim3(e->v.B), dim3(64), 0, s, e->v, e->sp, restart_finished);
    AZ_HIP(hipGetLastError());
    return check_sticky(e, s);
}

extern "C" int az_engine_chosen_actions(az_engine* e, int16_t* actions_host, void* stream) {
    AZ_CHECK_ARG(e && actions_host && e->sp_ready, "az_engine_chosen_actions: bad argument");
    hipStream_t s = (hipStream_t)stream;
    AZ_HIP(hipMemcpyAsync(actions_host, e->sp.chosen, e->v.B * sizeof(int16_t), hipMemcpyDeviceToHost, s));
    AZ_HIP(hipStreamSynchronize(s));
    return AZ_OK;
}

extern "C" int az_engine_drain_samples(az_engine* e, int64_t max_samples, float* states_dev,
                                       float* policies_dev, float* wls_dev, int64_t* n_out, void* stream) {
    AZ_CHECK_ARG(e && n_out && e->sp_ready, "az_engine_drain_samples: bad argument");
    hipStream_t s = (hipStream_t)stream;
    uint32_t n = 0;
    AZ_HIP(hipMemcpyAsync(&n, e->sp.q_count, sizeof n, hipMemcpyDeviceToHost, s));
    AZ_HIP(hipStreamSynchronize(s));
    *n_out = n;
    if (n == 0 || !states_dev) return AZ_OK;  // states_dev == NULL: just report the count
    AZ_CHECK_ARG(policies_dev && wls_dev, "az_engine_drain_samples: null output");
    AZ_CHECK_ARG((int64_t)n <= max_samples, "az_engine_drain_samples: %u samples queued, room for %lld", n,
                 (long long)max_samples);
    AZ_CHECK_ARG(((uintptr_t)states_dev & 15) == 0, "az_engine_drain_samples: states must be 16-byte aligned");
    hipLaunchKernelGGL(k_unpack_samples, dim3(n), dim3(64), 0, s, e->sp, n, states_dev, policies_dev, wls_dev);
    AZ_HIP(hipGetLastError());
    AZ_HIP(hipMemsetAsync(e->sp.q_count, 0, sizeof(uint32_t), s));
    AZ_HIP(hipStreamSynchronize(s));
    return AZ_OK;
}

extern "C" int az_engine_drain_packed(az_engine* e, int64_t max_samples, void* packed_dev, int64_t* n_out, void* stream) {
    AZ_CHECK_ARG(e && n_out && e->sp_ready, "az_engine_drain_packed: bad argument");
    hipStream_t s = (hipStream_t)stream;
    uint32_t n = 0;
    AZ_HIP(hipMemcpyAsync(&n, e->sp.q_count, sizeof n, hipMemcpyDeviceToHost, s));
    AZ_HIP(hipStreamSynchronize(s));
    *n_out = n;
    if (n == 0 || !packed_dev) return AZ_OK;
    AZ_CHECK_ARG((int64_t)n <= max_samples, "az_engine_drain_packed: %u samples queued, room for %lld", n,
                 (long long)max_samples);
    AZ_CHECK_ARG(((uintptr_t)packed_dev & 7) == 0, "az_engine_drain_packed: the buffer must be 8-byte aligned");
    hipLaunchKernelGGL(k_pack_samples, dim3(n), dim3(64), 0, s, e->sp, n, (uint8_t*)packed_dev);
    AZ_HIP(hipGetLastError());
    AZ_HIP(hipMemsetAsync(e->sp.q_count, 0, sizeof(uint32_t), s));
    AZ_HIP(hipStreamSynchronize(s));
    return AZ_OK;
}

extern "C" int az_unpack_samples(int64_t n, const void* packed_dev, float* states_dev, float* policies_dev,
                                 float* wls_dev, void* stream) {
    AZ_CHECK_ARG(n >= 0 && n < (1ll << 31), "az_unpack_samples: n = %lld out of range", (long long)n);
    if (n == 0) return AZ_OK;
    AZ_CHECK_ARG(packed_dev && states_dev && policies_dev && wls_dev, "az_unpack_samples: null pointer");
    AZ_CHECK_ARG(((uintptr_t)packed_dev & 7) == 0 && ((uintptr_t)states_dev & 15) == 0,
                 "az_unpack_samples: packed rows must be 8-byte aligned, states 16-byte aligned");
    hipLaunchKernelGGL(k_unpack_packed, dim3((uint32_t)n), dim3(64), 0, (hipStream_t)stream, (const uint8_t*)packed_dev,
                       (uint32_t)n, states_dev, policies_dev, wls_dev);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

extern "C" int az_engine_selfplay_stats(az_engine* e, az_selfplay_stats* out, void* stream) {
    AZ_CHECK_ARG(e && out && e->sp_ready, "az_engine_selfplay_stats: bad argument");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long st[4] = {0, 0, 0, 0}, sims = 0;
    uint32_t q = 0;
    AZ_HIP(hipMemcpyAsync(st, e->sp.stat, sizeof st, hipMemcpyDeviceToHost, s));
    AZ_HIP(hipMemcpyAsync(&q, e->sp.q_count, sizeof q, hipMemcpyDeviceToHost, s));
    AZ_HIP(hipStreamSynchronize(s));
    unsigned long long c[4];
    const int rc = az::engine_counters(e, c, s);
    if (rc != AZ_OK) return rc;
    sims = c[1];
    const unsigned long long term = c[2], rootexp = c[3];
    // every descent that did not end in a terminal leaf was evaluated, plus the root expansions
    out->simulations = sims; out->nn_evals = sims - term + rootexp; out->moves = st[0]; out->games_finished = st[1];
    out->samples_pending = q;
    unsigned long long rows = 0;
    AZ_HIP(hipMemcpyAsync(&rows, e->rows_total_dev, sizeof rows, hipMemcpyDeviceToHost, s));
    AZ_HIP(hipStreamSynchronize(s));
    out->network_rows = rows;
    return AZ_OK;
}
