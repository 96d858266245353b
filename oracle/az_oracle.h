/*
 * az_oracle.h — CPU restatement of the reference self-play hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may include, link
 * or load this.  It is used by tests/, by __graft_entry__.smoke() and by the
 * cpu_baseline leg of bench.py as the checker / reported CPU baseline.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against
 * the imported Python reference (tests/golden/make_golden.py generated the
 * fixtures under tests/golden/ by running the reference itself; see
 * tests/test_oracle_golden.py).
 *
 * Citations are file:line under the reference checkout (src/...).
 */
#ifndef AZ_ORACLE_H
#define AZ_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A position: bit (row*8+col) of white/black (tests/test_compatibility.py:8-17),
 * turn +1 = WHITE, -1 = BLACK, winner 0 = None / +1 / -1 (game.py:75-88). */
typedef struct {
    uint64_t white;
    uint64_t black;
    int32_t  turn;
    int32_t  winner;
} azo_pos;

void azo_start_position(azo_pos* p);
/* game.py:109-150 then encode_action (225-255) per move: ordered action list. */
int  azo_legal_actions(const azo_pos* p, uint8_t actions[48]);
/* game.py:257-278 + 156-187: decode the action for the side to move and apply it. */
void azo_step_action(azo_pos* p, int action);
int  azo_is_terminal(const azo_pos* p);                     /* game.py:189-194 */
void azo_get_result(const azo_pos* p, float* w, float* l);  /* game.py:196-215 */
void azo_encode_state(const azo_pos* p, float planes[192]); /* game.py:280-307 */
void azo_legal_mask(const azo_pos* p, uint8_t mask[192]);   /* game.py:309-315 */
/* status byte as the GPU move-gen kernel reports it: 0 live, 1/-1(=255) winner
 * recorded on the position, 2 no legal moves for the side to move. */
int  azo_status(const azo_pos* p);

/* Batched forms used by the movegen parity tests and the CPU baseline. */
void azo_movegen_batch(int64_t n, const uint64_t* white, const uint64_t* black,
                       const int8_t* turn, uint64_t* legal3, uint8_t* mask192,
                       float* planes, int8_t* status);
/* Seeded uniform-random legal self-play from the start position; records every
 * position (terminal ones included) until n are filled. */
void azo_random_positions(int64_t n, uint64_t seed, uint64_t* white,
                          uint64_t* black, int8_t* turn, int8_t* winner);

/* ---- batched PUCT MCTS (mcts.py:94-189), driven step by step ---- */
typedef struct azo_mcts azo_mcts;

azo_mcts* azo_mcts_create(int n_games, double c_puct);
void      azo_mcts_destroy(azo_mcts* m);
/* Set game g's position and drop its root (root = None). */
void azo_mcts_set_game(azo_mcts* m, int g, const azo_pos* p);
void azo_mcts_get_game(const azo_mcts* m, int g, azo_pos* p);
void azo_mcts_set_active(azo_mcts* m, const uint8_t* active);
/* mcts.py:117-133: create missing roots; report which roots need expansion and
 * their encoded states.  planes [n_games][192], need_eval [n_games]. */
int  azo_mcts_roots_prepare(azo_mcts* m, float* planes, uint8_t* need_eval);
/* mcts.py:135-138: expand those roots from probs [n_games][192]; value unused. */
void azo_mcts_roots_expand(azo_mcts* m, const float* probs);
/* mcts.py:141-144,314-326: noise [n_games][48] doubles, one Dirichlet draw per
 * expanded root, in child insertion order. */
void azo_mcts_add_noise(azo_mcts* m, const double* noise, double epsilon);
/* mcts.py:153-171: one descent per active game. */
int  azo_mcts_select(azo_mcts* m, float* planes, uint8_t* need_eval);
/* mcts.py:174-187: expand + back up evaluated leaves, back up terminal ones. */
void azo_mcts_expand_backup(azo_mcts* m, const float* probs, const float* values);
/* Root statistics, children in insertion order.  p64 receives the prior as a
 * double; p_is64 says whether the reference holds it as float64 (noisy root or
 * uniform fallback) or float32. Returns child count, or -1 for no root. */
int  azo_mcts_root_children(const azo_mcts* m, int g, uint8_t* actions,
                            uint32_t* N, float* W, double* p64, uint8_t* p_is64,
                            uint32_t* root_N, float* root_W);
/* mcts.py:191-225 with temperature 0 or 1. */
void azo_mcts_action_probs(const azo_mcts* m, int g, int temperature_is_zero,
                           float probs[192]);
/* parallel_trainer.py:190-198: step the game and re-root at the chosen child. */
void azo_mcts_advance(azo_mcts* m, int g, int action);
/* Mean selection path length (nodes below the root) over all descents so far. */
double azo_mcts_mean_depth(const azo_mcts* m);
int64_t azo_mcts_total_nodes(const azo_mcts* m);

#ifdef __cplusplus
}
#endif
#endif
