// Engine state shared by the tree kernels (tree.hip) and the fused self-play
// loop (selfplay.hip).
#pragma once
#include <vector>

#include "az_common.h"

namespace az {

constexpr uint32_t NODE_NONE = 0xFFFFFFFFu;
constexpr int MAX_DEPTH = 256;          // longest selection path (a game lasts < 200 plies)
constexpr uint32_t FLAG_UNIFORM64 = 1;  // children hold the float64 prior 1/k (src/mcts.py:277-278)
constexpr int ERR_POOL = 1, ERR_DEPTH = 2, ERR_HISTORY = 4, ERR_NODE = 8, ERR_NOROOT = 16;

// Device pointers + scalars, passed to kernels by value.
struct EngineView {
    int32_t B;
    uint32_t cap;
    double c_puct, alpha, eps;
    int32_t temp_thr;
    uint64_t seed;
    // per game: position on the real board, search bookkeeping
    uint64_t *white, *black;
    int8_t *turn, *winner;
    uint8_t* active;
    uint16_t* ply;
    uint32_t *root, *used;
    uint8_t* arena;       // which of the game's two node arenas is live
    uint8_t* root_mode;   // 1: root children priors are the doubles in root_p64
    double* root_p64;     // [B][48]
    // leaf of the current simulation
    uint32_t* leaf_node;
    uint64_t *leaf_mine, *leaf_theirs;
    int8_t* leaf_turn;
    uint8_t* leaf_need;   // [B] leaf must be evaluated; the buffer selection WRITES (and the step kernels read)
    uint8_t* need_cur;    // [B] the same flags of the simulation being backed up.  The step API uses one
                          // buffer for both; the fused search alternates between two, because the network
                          // launch of simulation i+1 reads every game's new flag while it assigns its rows
    float* leaf_tv;
    uint16_t* path_len;
    uint32_t* path;       // [B][MAX_DEPTH]
    // node pool, SoA
    uint32_t* N;
    float* W;
    float* P;
    uint32_t* FC;
    uint32_t* IN;
    int32_t* err;
    // per-game counters [B] (summed on the host when asked for: one shared counter would
    // serialise 1024 atomics per simulation at the L2, ~12 us)
    unsigned long long *stat_depth, *stat_descents;
    unsigned long long *stat_terminal, *stat_rootexp;  // terminal leaves backed up, roots expanded
    // Row of game g in the network's outputs (probabilities, values, head partial sums), or NULL
    // when rows are games.  The fused search evaluates only the leaves that need it (active
    // games whose leaf is not terminal), packed densely in game order by the network launch
    // itself, so that a shrinking set of games runs smaller, faster network launches and a
    // terminal leaf costs no network row.
    const int32_t* row_of;
};

struct NodeQuery {
    int32_t k;
    uint32_t node_visits;
    float node_value_sum;
    int64_t node_index;
    uint8_t actions[MAX_CHILDREN];
    uint8_t prior_is64[MAX_CHILDREN];
    uint32_t visits[MAX_CHILDREN];
    float value_sum[MAX_CHILDREN];
    double prior64[MAX_CHILDREN];
    int64_t child_index[MAX_CHILDREN];
};

#ifdef __HIPCC__
// counter-based generator for the device-drawn Dirichlet noise and move sampling
__device__ inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t key, ctr;
    __device__ Rng(uint64_t seed, uint64_t a, uint64_t b, uint64_t c)
        : key(mix64(seed ^ mix64(a * 0x9E3779B97F4A7C15ull + b) ^ (c << 32))), ctr(0) {}
    __device__ uint64_t next() { return mix64(key + 0x9E3779B97F4A7C15ull * (++ctr)); }
    __device__ double uniform() { return ((double)(next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
    __device__ double normal() {
        const double u1 = uniform(), u2 = uniform();
        return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
    // Marsaglia-Tsang, boosted for shape < 1
    __device__ double gamma(double a) {
        const double boost = (a < 1.0) ? pow(uniform(), 1.0 / a) : 1.0;
        const double aa = (a < 1.0) ? a + 1.0 : a;
        const double d = aa - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
        for (int it = 0; it < 64; ++it) {
            const double x = normal();
            double v = 1.0 + c * x;
            if (v <= 0) continue;
            v = v * v * v;
            const double u = uniform();
            if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return d * v * boost;
        }
        return d * boost;
    }
};

#endif

enum Phase { PHASE_IDLE = 0, PHASE_ROOTS = 1, PHASE_SELECTED = 2 };

// what k_finish_scan found for a game after a move
constexpr int FIN_NONE = 0, FIN_BLACK = 1, FIN_WHITE = 2, FIN_DROPPED = 4;  // DROPPED: queue full

constexpr int MAX_PLIES = 256;  // per-game history depth (a game lasts < 200 plies)

// Self-play bookkeeping on the device (selfplay.hip): per-game move history and
// the queue of finished training samples (src/parallel_trainer.py:185-213).
struct SelfplayView {
    float* pi;              // [B][192] visit-count policy of the current move
    int16_t* chosen;        // [B] action played this move
    uint64_t *hist_mine, *hist_theirs;  // [B][MAX_PLIES] canonical bitboards
    int8_t* hist_turn;      // [B][MAX_PLIES]
    float* hist_pi;         // [B][MAX_PLIES][192]
    // finished samples, packed
    uint64_t *q_mine, *q_theirs;  // [Q]
    float* q_wl;            // [Q][2]
    float* q_pi;            // [Q][192]
    uint32_t* q_count;      // samples in the queue
    uint32_t* fin_base;     // [B] queue position of a game that finished with this move
    uint8_t* fin_flag;      // [B] FIN_* of this move
    uint32_t q_cap;
    unsigned long long* stat;  // [0] moves, [1] games finished
};

// Buffers of the leaf-parallel latency search (leafpar.hip), one row per pending descent.
struct LeafParView {
    uint64_t *mine, *theirs;   // [rows] canonical leaf positions = the network's input
    uint32_t* node;            // [rows]
    int8_t* turn;              // [rows]
    uint16_t* len;             // [rows] path length
    uint8_t* kind;             // [rows] 0: evaluate, 1: terminal, 2: same leaf as row first[row], 3: no descent
    uint8_t* first;            // [rows]
    float* tv;                 // [rows] terminal value
    uint32_t* path;            // [rows][MAX_DEPTH]
    int32_t* slots;            // [games]
};

}  // namespace az

struct az_engine {
    az::EngineView v{};
    int device = 0;
    int phase = az::PHASE_IDLE;
    uint64_t noise_calls = 0;
    std::vector<void*> allocs;
    // staging for host <-> device hand-off of small per-game arrays
    int32_t* stage_slots = nullptr;
    uint64_t *stage_w = nullptr, *stage_b = nullptr;
    int8_t *stage_t = nullptr, *stage_win = nullptr;
    int16_t* stage_act = nullptr;
    float* stage_temps = nullptr;
    az::NodeQuery* query_dev = nullptr;
    az::NodeQuery* query_host = nullptr;
    az::NodeQuery* query_batch_dev = nullptr;    // [B], for az_engine_roots_children
    az::NodeQuery* query_batch_host = nullptr;
    // fused self-play (selfplay.hip)
    struct az_net* net = nullptr;
    az::SelfplayView sp{};
    bool sp_ready = false;
    float* nn_probs = nullptr;   // [B][192]
    float* nn_values = nullptr;  // [B]
    uint8_t* need_buf[2] = {nullptr, nullptr};  // [B rounded up to 16] each; v.leaf_need / v.need_cur point into them
    int32_t* row_of_dev = nullptr;   // [B] row of each evaluated game in the packed network batch
    int32_t* n_rows_dev = nullptr;   // rows of the last network launch
    unsigned long long* rows_total_dev = nullptr;  // rows of all network launches of the fused search
    uint64_t move_calls = 0;
    // leaf-parallel latency search (leafpar.hip): allocated on first use, freed with the engine
    az::LeafParView lp{};
    bool lp_ready = false;
};

namespace az {
// tree.hip: report and clear the device error word; synchronises
int check_engine_error(az_engine* e, hipStream_t s);
// tree.hip: sums of the per-game counters {depth, descents, terminal, rootexp}; synchronises
int engine_counters(az_engine* e, unsigned long long out[4], hipStream_t s);
// tree.hip: play the actions held in a device array (same as az_engine_advance)
int advance_dev(az_engine* e, const int16_t* actions_dev, hipStream_t s);
}  // namespace az
