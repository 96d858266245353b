/*
 * az_oracle.c — CPU restatement of the reference self-play hot path
 * (game rules, batched PUCT MCTS).  TEST INFRASTRUCTURE ONLY: see az_oracle.h.
 *
 * The game half works on an int8 8x8 board exactly like the reference does
 * (not on bitboards), so that it is an independent check of the GPU's
 * bitboard formulation.  The search half keeps float32 / float64 types exactly
 * where numpy >= 2 promotion rules put them in the reference.
 *
 * Build with -ffp-contract=off and without -ffast-math (oracle/Makefile).
 * Parity status: pinned against the imported reference (tests/golden).
 */
#include "az_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define WHITE 1
#define BLACK (-1)
#define EMPTY 0

/* game.py:41-45 */
static const int DIRS[3][2] = {{1, 0}, {1, -1}, {1, 1}};

typedef struct {
    int8_t board[8][8];
    int turn;
    int winner; /* 0 = None */
} game_t;

static void game_from_pos(game_t* g, const azo_pos* p) {
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
            uint64_t bit = 1ull << (r * 8 + c);
            g->board[r][c] = (p->white & bit) ? WHITE : ((p->black & bit) ? BLACK : EMPTY);
        }
    g->turn = p->turn;
    g->winner = p->winner;
}

static void game_to_pos(const game_t* g, azo_pos* p) {
    p->white = p->black = 0;
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
            uint64_t bit = 1ull << (r * 8 + c);
            if (g->board[r][c] == WHITE) p->white |= bit;
            else if (g->board[r][c] == BLACK) p->black |= bit;
        }
    p->turn = g->turn;
    p->winner = g->winner;
}

/* game.py:90-99 */
static void game_init(game_t* g) {
    memset(g->board, 0, sizeof g->board);
    for (int c = 0; c < 8; ++c) {
        g->board[0][c] = g->board[1][c] = WHITE;
        g->board[6][c] = g->board[7][c] = BLACK;
    }
    g->turn = WHITE;
    g->winner = 0;
}

/* game.py:109-150 — row-major scan of the real board, directions 0,1,2. */
static int game_legal_moves(const game_t* g, int mv[48][4]) {
    int n = 0;
    for (int row = 0; row < 8; ++row)
        for (int col = 0; col < 8; ++col) {
            if (g->board[row][col] != g->turn) continue;
            for (int d = 0; d < 3; ++d) {
                int dr = DIRS[d][0], dc = DIRS[d][1];
                int tr, tc;
                if (g->turn == BLACK) { tr = row - dr; tc = col - dc; }
                else                  { tr = row + dr; tc = col + dc; }
                if (tr < 0 || tr >= 8 || tc < 0 || tc >= 8) continue;
                int target = g->board[tr][tc];
                if (d == 0) { if (target != EMPTY) continue; }
                else        { if (target == g->turn) continue; }
                mv[n][0] = row; mv[n][1] = col; mv[n][2] = tr; mv[n][3] = tc;
                ++n;
            }
        }
    return n;
}

/* game.py:156-187 */
static void game_step(game_t* g, const int mv[4]) {
    g->board[mv[2]][mv[3]] = g->board[mv[0]][mv[1]];
    g->board[mv[0]][mv[1]] = EMPTY;
    int to_row = mv[2];
    if (g->turn == WHITE && to_row == 7) g->winner = WHITE;
    else if (g->turn == BLACK && to_row == 0) g->winner = BLACK;
    if (g->winner == 0) {
        int opp = -g->turn, any = 0;
        for (int r = 0; r < 8 && !any; ++r)
            for (int c = 0; c < 8; ++c)
                if (g->board[r][c] == opp) { any = 1; break; }
        if (!any) g->winner = g->turn;
    }
    g->turn = -g->turn;
}

/* game.py:189-194 */
static int game_is_terminal(const game_t* g) {
    if (g->winner != 0) return 1;
    int mv[48][4];
    return game_legal_moves(g, mv) == 0;
}

/* game.py:196-215 */
static void game_result(const game_t* g, float* w, float* l) {
    if (!game_is_terminal(g)) { *w = 0.f; *l = 0.f; return; }
    if (g->winner == WHITE) { *w = 1.f; *l = 0.f; }
    else if (g->winner == BLACK) { *w = 0.f; *l = 1.f; }
    else if (g->turn == WHITE) { *w = 0.f; *l = 1.f; }
    else { *w = 1.f; *l = 0.f; }
}

/* game.py:225-255 */
static int game_encode_action(const game_t* g, const int mv[4]) {
    int fr = mv[0], fc = mv[1], tr = mv[2], tc = mv[3];
    int dc;
    if (g->turn == WHITE) {
        dc = tc - fc;
    } else {
        fr = 7 - fr; fc = 7 - fc;
        int tcf = 7 - tc;
        (void)tr;
        dc = tcf - fc;
    }
    int direction = (dc == 0) ? 0 : ((dc == -1) ? 1 : 2);
    return (fr * 8 + fc) * 3 + direction; /* game.py:48-51 */
}

/* game.py:257-278 */
static void game_decode_action(const game_t* g, int action, int mv[4]) {
    int square = action / 3, direction = action % 3;
    int fr = square / 8, fc = square % 8;
    int tr = fr + DIRS[direction][0], tc = fc + DIRS[direction][1];
    if (g->turn == BLACK) { fr = 7 - fr; fc = 7 - fc; tr = 7 - tr; tc = 7 - tc; }
    mv[0] = fr; mv[1] = fc; mv[2] = tr; mv[3] = tc;
}

/* game.py:280-307 */
static void game_encode_state(const game_t* g, float planes[192]) {
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
            int v = (g->turn == WHITE) ? g->board[r][c] : g->board[7 - r][7 - c];
            int mine = (g->turn == WHITE) ? WHITE : BLACK;
            planes[0 * 64 + r * 8 + c] = (v == mine) ? 1.f : 0.f;
            planes[1 * 64 + r * 8 + c] = (v == -mine) ? 1.f : 0.f;
            planes[2 * 64 + r * 8 + c] = 1.f;
        }
}

static int game_ordered_actions(const game_t* g, uint8_t actions[48]) {
    int mv[48][4];
    int n = game_legal_moves(g, mv);
    for (int i = 0; i < n; ++i) actions[i] = (uint8_t)game_encode_action(g, mv[i]);
    return n;
}

/* ------------------------------------------------------------------ */
/* public game API                                                     */
/* ------------------------------------------------------------------ */

void azo_start_position(azo_pos* p) {
    game_t g; game_init(&g); game_to_pos(&g, p);
}

int azo_legal_actions(const azo_pos* p, uint8_t actions[48]) {
    game_t g; game_from_pos(&g, p);
    return game_ordered_actions(&g, actions);
}

void azo_step_action(azo_pos* p, int action) {
    game_t g; game_from_pos(&g, p);
    int mv[4]; game_decode_action(&g, action, mv);
    game_step(&g, mv);
    game_to_pos(&g, p);
}

int azo_is_terminal(const azo_pos* p) {
    game_t g; game_from_pos(&g, p);
    return game_is_terminal(&g);
}

void azo_get_result(const azo_pos* p, float* w, float* l) {
    game_t g; game_from_pos(&g, p);
    game_result(&g, w, l);
}

void azo_encode_state(const azo_pos* p, float planes[192]) {
    game_t g; game_from_pos(&g, p);
    game_encode_state(&g, planes);
}

void azo_legal_mask(const azo_pos* p, uint8_t mask[192]) {
    uint8_t a[48];
    memset(mask, 0, 192);
    int n = azo_legal_actions(p, a);
    for (int i = 0; i < n; ++i) mask[a[i]] = 1;
}

int azo_status(const azo_pos* p) {
    if (p->winner != 0) return p->winner;
    uint8_t a[48];
    return azo_legal_actions(p, a) == 0 ? 2 : 0;
}

void azo_movegen_batch(int64_t n, const uint64_t* white, const uint64_t* black,
                       const int8_t* turn, uint64_t* legal3, uint8_t* mask192,
                       float* planes, int8_t* status) {
    for (int64_t i = 0; i < n; ++i) {
        azo_pos p = {white[i], black[i], turn[i], 0};
        /* the batch kernel derives "winner by board" itself: a side standing on
         * its goal row, or the opponent of an extinct side (game.py:175-187). */
        uint8_t a[48];
        int k = azo_legal_actions(&p, a);
        if (legal3) {
            uint64_t l0 = 0, l1 = 0, l2 = 0;
            for (int j = 0; j < k; ++j) {
                int sq = a[j] / 3, d = a[j] % 3;
                if (d == 0) l0 |= 1ull << sq; else if (d == 1) l1 |= 1ull << sq; else l2 |= 1ull << sq;
            }
            legal3[3 * i + 0] = l0; legal3[3 * i + 1] = l1; legal3[3 * i + 2] = l2;
        }
        if (mask192) {
            memset(mask192 + 192 * i, 0, 192);
            for (int j = 0; j < k; ++j) mask192[192 * i + a[j]] = 1;
        }
        if (planes) azo_encode_state(&p, planes + 192 * i);
        if (status) {
            int s = 0;
            if (white[i] & 0xFF00000000000000ull) s = 1;
            else if (black[i] & 0xFFull) s = -1;
            else if (black[i] == 0) s = 1;
            else if (white[i] == 0) s = -1;
            else if (k == 0) s = 2;
            status[i] = (int8_t)s;
        }
    }
}

static uint64_t splitmix64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void azo_random_positions(int64_t n, uint64_t seed, uint64_t* white,
                          uint64_t* black, int8_t* turn, int8_t* winner) {
    uint64_t s = seed;
    game_t g; game_init(&g);
    for (int64_t i = 0; i < n; ++i) {
        azo_pos p; game_to_pos(&g, &p);
        white[i] = p.white; black[i] = p.black;
        turn[i] = (int8_t)p.turn;
        if (winner) winner[i] = (int8_t)p.winner;
        int mv[48][4];
        int k = game_legal_moves(&g, mv);
        if (g.winner != 0 || k == 0) { game_init(&g); continue; }
        game_step(&g, mv[splitmix64(&s) % (uint64_t)k]);
    }
}

/* ------------------------------------------------------------------ */
/* MCTS                                                                */
/* ------------------------------------------------------------------ */

/* mcts.py:21-40.  visit_count int; value_sum float32 (np.float32 after the
 * first network value is added, integer-valued Python float before that —
 * identical in float32); prior np.float32 from _expand_node, or float64 after
 * _add_dirichlet_noise / the uniform fallback. */
typedef struct {
    uint32_t N;
    float    W;
    float    P32;
    double   P64;
    int32_t  first_child; /* -1: not expanded */
    uint8_t  n_child;
    uint8_t  action;
    uint8_t  p_is64;
} node_t;

typedef struct {
    node_t* nodes;
    int32_t n, cap;
    int32_t root; /* -1 = None */
    game_t  game;
    game_t  scratch;
    int32_t path[256];
    int32_t path_len;
    int     leaf_need_eval;
    int     root_need_eval;
} tree_t;

struct azo_mcts {
    int      n_games;
    double   c_puct;
    tree_t*  trees;
    uint8_t* active;
    int64_t  depth_sum, descents;
};

static int32_t tree_alloc(tree_t* t, int count) {
    if (t->n + count > t->cap) {
        int32_t ncap = t->cap ? t->cap * 2 : 1024;
        while (ncap < t->n + count) ncap *= 2;
        t->nodes = (node_t*)realloc(t->nodes, (size_t)ncap * sizeof(node_t));
        t->cap = ncap;
    }
    int32_t at = t->n;
    t->n += count;
    return at;
}

static void node_init(node_t* nd, int action) {
    nd->N = 0; nd->W = 0.f; nd->P32 = 0.f; nd->P64 = 0.0;
    nd->first_child = -1; nd->n_child = 0; nd->action = (uint8_t)action; nd->p_is64 = 0;
}

azo_mcts* azo_mcts_create(int n_games, double c_puct) {
    azo_mcts* m = (azo_mcts*)calloc(1, sizeof *m);
    m->n_games = n_games;
    m->c_puct = c_puct;
    m->trees = (tree_t*)calloc((size_t)n_games, sizeof(tree_t));
    m->active = (uint8_t*)malloc((size_t)n_games);
    memset(m->active, 1, (size_t)n_games);
    for (int g = 0; g < n_games; ++g) { m->trees[g].root = -1; game_init(&m->trees[g].game); }
    return m;
}

void azo_mcts_destroy(azo_mcts* m) {
    if (!m) return;
    for (int g = 0; g < m->n_games; ++g) free(m->trees[g].nodes);
    free(m->trees); free(m->active); free(m);
}

void azo_mcts_set_game(azo_mcts* m, int g, const azo_pos* p) {
    tree_t* t = &m->trees[g];
    game_from_pos(&t->game, p);
    t->root = -1; t->n = 0;
}

void azo_mcts_get_game(const azo_mcts* m, int g, azo_pos* p) {
    game_to_pos(&m->trees[g].game, p);
}

void azo_mcts_set_active(azo_mcts* m, const uint8_t* active) {
    memcpy(m->active, active, (size_t)m->n_games);
}

/* mcts.py:260-279 */
static void expand_node(tree_t* t, int32_t ni, const game_t* g, const float* probs) {
    uint8_t actions[48];
    int k = game_ordered_actions(g, actions);
    float prior_sum = 0.0f; /* 0.0 + np.float32 -> np.float32, sequential */
    for (int i = 0; i < k; ++i) prior_sum = prior_sum + probs[actions[i]];
    if (k == 0) return;
    int32_t base = tree_alloc(t, k);
    node_t* nd = &t->nodes[ni];
    nd->first_child = base;
    nd->n_child = (uint8_t)k;
    for (int i = 0; i < k; ++i) {
        node_t* c = &t->nodes[base + i];
        node_init(c, actions[i]);
        if (prior_sum > 0) {
            c->P32 = probs[actions[i]] / prior_sum;
            c->P64 = (double)c->P32;
        } else {
            c->p_is64 = 1;                 /* 1.0 / len(legal_actions): Python float */
            c->P64 = 1.0 / (double)k;
            c->P32 = (float)c->P64;
        }
    }
}

/* mcts.py:281-312 */
static int32_t select_child(const azo_mcts* m, const tree_t* t, int32_t ni) {
    const node_t* nd = &t->nodes[ni];
    uint32_t pn = nd->N > 1 ? nd->N : 1;
    double sqrt_parent = sqrt((double)pn);
    double best_score = -INFINITY;
    int32_t best = -1;
    for (int i = 0; i < nd->n_child; ++i) {
        const node_t* c = &t->nodes[nd->first_child + i];
        double score;
        if (!c->p_is64) {
            float q = (c->N == 0) ? 0.0f : -(c->W / (float)c->N);
            float u = (float)m->c_puct * c->P32;
            u = u * (float)sqrt_parent;
            u = u / (float)(1u + c->N);
            float s = q + u;
            score = (double)s;
        } else {
            double q = (c->N == 0) ? 0.0 : (double)(-(c->W / (float)c->N));
            double u = m->c_puct * c->P64;
            u = u * sqrt_parent;
            u = u / (double)(1u + c->N);
            score = q + u;
        }
        if (score > best_score) { best_score = score; best = nd->first_child + i; }
    }
    return best;
}

/* mcts.py:337-342 */
static void backpropagate(tree_t* t, float value) {
    for (int i = t->path_len - 1; i >= 0; --i) {
        node_t* nd = &t->nodes[t->path[i]];
        nd->N += 1;
        nd->W = nd->W + value;
        value = -value;
    }
}

int azo_mcts_roots_prepare(azo_mcts* m, float* planes, uint8_t* need_eval) {
    int cnt = 0;
    for (int g = 0; g < m->n_games; ++g) {
        tree_t* t = &m->trees[g];
        need_eval[g] = 0; t->root_need_eval = 0;
        if (!m->active[g]) continue;
        if (t->root < 0) {
            t->n = 0;
            t->root = tree_alloc(t, 1);
            node_init(&t->nodes[t->root], 0);
        }
        if (t->nodes[t->root].n_child == 0) {
            need_eval[g] = 1; t->root_need_eval = 1;
            game_encode_state(&t->game, planes + 192 * (size_t)g);
            ++cnt;
        }
    }
    return cnt;
}

void azo_mcts_roots_expand(azo_mcts* m, const float* probs) {
    for (int g = 0; g < m->n_games; ++g) {
        tree_t* t = &m->trees[g];
        if (!m->active[g] || !t->root_need_eval) continue;
        expand_node(t, t->root, &t->game, probs + 192 * (size_t)g);
        t->root_need_eval = 0;
    }
}

/* mcts.py:314-326 */
void azo_mcts_add_noise(azo_mcts* m, const double* noise, double epsilon) {
    for (int g = 0; g < m->n_games; ++g) {
        tree_t* t = &m->trees[g];
        if (!m->active[g] || t->root < 0) continue;
        node_t* r = &t->nodes[t->root];
        for (int i = 0; i < r->n_child; ++i) {
            node_t* c = &t->nodes[r->first_child + i];
            double add = epsilon * noise[48 * (size_t)g + i];
            if (!c->p_is64) {
                float keep = (float)(1.0 - epsilon) * c->P32; /* float32 product */
                c->P64 = (double)keep + add;
            } else {
                c->P64 = (1.0 - epsilon) * c->P64 + add;
            }
            c->p_is64 = 1;
            c->P32 = (float)c->P64;
        }
    }
}

int azo_mcts_select(azo_mcts* m, float* planes, uint8_t* need_eval) {
    int cnt = 0;
    for (int g = 0; g < m->n_games; ++g) {
        tree_t* t = &m->trees[g];
        need_eval[g] = 0; t->leaf_need_eval = 0;
        if (!m->active[g]) continue;
        int32_t ni = t->root;
        t->scratch = t->game;
        t->path_len = 0;
        t->path[t->path_len++] = ni;
        while (t->nodes[ni].n_child > 0) {
            ni = select_child(m, t, ni);
            int mv[4];
            game_decode_action(&t->scratch, t->nodes[ni].action, mv);
            game_step(&t->scratch, mv);
            t->path[t->path_len++] = ni;
        }
        m->depth_sum += t->path_len - 1; m->descents += 1;
        if (!game_is_terminal(&t->scratch)) {
            t->leaf_need_eval = 1; need_eval[g] = 1;
            game_encode_state(&t->scratch, planes + 192 * (size_t)g);
            ++cnt;
        }
    }
    return cnt;
}

void azo_mcts_expand_backup(azo_mcts* m, const float* probs, const float* values) {
    for (int g = 0; g < m->n_games; ++g) {
        tree_t* t = &m->trees[g];
        if (!m->active[g]) continue;
        if (t->leaf_need_eval) {
            expand_node(t, t->path[t->path_len - 1], &t->scratch, probs + 192 * (size_t)g);
            backpropagate(t, values[g]);
        } else {
            /* mcts.py:328-335 */
            float w, l; game_result(&t->scratch, &w, &l);
            float white_value = w - l;
            float v = (t->scratch.turn == WHITE) ? white_value : -white_value;
            backpropagate(t, v);
        }
        t->leaf_need_eval = 0;
    }
}

int azo_mcts_root_children(const azo_mcts* m, int g, uint8_t* actions,
                           uint32_t* N, float* W, double* p64, uint8_t* p_is64,
                           uint32_t* root_N, float* root_W) {
    const tree_t* t = &m->trees[g];
    if (t->root < 0) return -1;
    const node_t* r = &t->nodes[t->root];
    if (root_N) *root_N = r->N;
    if (root_W) *root_W = r->W;
    for (int i = 0; i < r->n_child; ++i) {
        const node_t* c = &t->nodes[r->first_child + i];
        if (actions) actions[i] = c->action;
        if (N) N[i] = c->N;
        if (W) W[i] = c->W;
        if (p64) p64[i] = c->p_is64 ? c->P64 : (double)c->P32;
        if (p_is64) p_is64[i] = c->p_is64;
    }
    return r->n_child;
}

/* mcts.py:191-225 for temperature 0 and 1 */
void azo_mcts_action_probs(const azo_mcts* m, int g, int temperature_is_zero,
                           float probs[192]) {
    const tree_t* t = &m->trees[g];
    memset(probs, 0, 192 * sizeof(float));
    if (t->root < 0) return;
    const node_t* r = &t->nodes[t->root];
    for (int i = 0; i < r->n_child; ++i) {
        const node_t* c = &t->nodes[r->first_child + i];
        probs[c->action] = (float)c->N;
    }
    if (temperature_is_zero) {
        int best = 0;
        for (int a = 1; a < 192; ++a) if (probs[a] > probs[best]) best = a;
        memset(probs, 0, 192 * sizeof(float));
        probs[best] = 1.0f;
    } else {
        /* counts are integers < 2^24: any summation order is exact in float32 */
        float total = 0.f;
        for (int a = 0; a < 192; ++a) total += probs[a];
        if (total > 0) {
            for (int a = 0; a < 192; ++a) probs[a] = probs[a] / total;
        } else if (r->n_child) {
            float u = (float)(1.0 / (double)r->n_child);
            for (int i = 0; i < r->n_child; ++i) probs[t->nodes[r->first_child + i].action] = u;
        }
    }
}

/* copy the subtree under `src` into a fresh array, children kept contiguous
 * and in insertion order */
static void reroot(tree_t* t, int32_t src) {
    int32_t cap = 1024;
    node_t* out = (node_t*)malloc((size_t)cap * sizeof(node_t));
    int32_t n = 1, head = 0;
    out[0] = t->nodes[src];
    while (head < n) {
        node_t* nd = &out[head++];
        if (nd->n_child == 0) { nd->first_child = -1; continue; }
        int32_t old = nd->first_child, k = nd->n_child;
        if (n + k > cap) {
            while (n + k > cap) cap *= 2;
            out = (node_t*)realloc(out, (size_t)cap * sizeof(node_t));
            nd = &out[head - 1];
        }
        nd->first_child = n;
        memcpy(out + n, t->nodes + old, (size_t)k * sizeof(node_t));
        n += k;
    }
    free(t->nodes);
    t->nodes = out; t->n = n; t->cap = cap; t->root = 0;
}

/* parallel_trainer.py:190-198 */
void azo_mcts_advance(azo_mcts* m, int g, int action) {
    tree_t* t = &m->trees[g];
    int mv[4];
    game_decode_action(&t->game, action, mv);
    game_step(&t->game, mv);
    if (t->root < 0) return;
    const node_t* r = &t->nodes[t->root];
    int32_t found = -1;
    for (int i = 0; i < r->n_child; ++i)
        if (t->nodes[r->first_child + i].action == action) { found = r->first_child + i; break; }
    if (found < 0) { t->root = -1; t->n = 0; return; }
    reroot(t, found);
}

double azo_mcts_mean_depth(const azo_mcts* m) {
    return m->descents ? (double)m->depth_sum / (double)m->descents : 0.0;
}

int64_t azo_mcts_total_nodes(const azo_mcts* m) {
    int64_t s = 0;
    for (int g = 0; g < m->n_games; ++g) s += m->trees[g].n;
    return s;
}
