// Engine state shared by the tree kernels (tree.hip) and the fused self-play
// loop (selfplay.hip).
#pragma once
#include <vector>

#include "az_common.h"

namespace az {

constexpr uint32_t NODE_NONE = 0xFFFFFFFFu;
constexpr int MAX_DEPTH = 256;          // longest selection path (a game lasts < 200 plies)
constexpr uint32_t FLAG_UNIFORM64 = 1;  // children hold the float64 prior 1/k (src/mcts.py:277-278)
constexpr int ERR_POOL = 1, ERR_DEPTH = 2, ERR_HISTORY = 4;

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
    uint8_t* root_mode;   // 1: root children priors are the doubles in root_p64
    double* root_p64;     // [B][48]
    // leaf of the current simulation
    uint32_t* leaf_node;
    uint64_t *leaf_mine, *leaf_theirs;
    int8_t* leaf_turn;
    uint8_t* leaf_need;
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
    unsigned long long *stat_depth, *stat_descents;
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

enum Phase { PHASE_IDLE = 0, PHASE_ROOTS = 1, PHASE_SELECTED = 2 };

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
};
