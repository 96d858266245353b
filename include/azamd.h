/*
 * azamd.h — C ABI of libazamd.so, the gfx950 (MI355X) self-play library.
 *
 * The reference (pierre-roth/AlphaZero) is pure Python and has no FFI of its
 * own; each entry point below names the reference function(s) it replaces as
 * file:line under the reference checkout.  All pointers named *_dev are device
 * (HBM) pointers owned by the caller; *_host are host pointers.  `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).  Every function
 * returns 0 on success or a negative AZ_ERR_* code; az_last_error() then
 * returns a thread-local message.  Nothing throws across this boundary.
 *
 * Board convention: bit (row*8+col) of a 64-bit word; row 0 is White's home
 * row; turn +1 = WHITE, -1 = BLACK (src/game.py:28-31,
 * tests/test_compatibility.py:8-17).  "Canonical" means seen by the side to
 * move: for Black the board is rotated by 180 degrees (src/game.py:299-302),
 * which on bitboards is a 64-bit bit reversal with the colours swapped.
 */
#ifndef AZAMD_H
#define AZAMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_OK 0
#define AZ_ERR_ARG (-1)      /* bad argument (null pointer, size out of range) */
#define AZ_ERR_HIP (-2)      /* a HIP runtime call failed */
#define AZ_ERR_STATE (-3)    /* call sequence violated */
#define AZ_ERR_CAPACITY (-4) /* node pool / history exhausted */

const char* az_last_error(void);
int az_version(void);

/* ------------------------------------------------------------------------
 * Stateless batch move generation / terminal detection / encoding.
 * Replaces, per position: BreakthroughGame.get_legal_moves (src/game.py:109-150),
 * get_legal_action_mask (:309-315), get_encoded_state (:280-307) and the
 * board-derivable part of _check_winner / is_terminal (:175-194).
 *
 *   white_dev, black_dev [n] u64, turn_dev [n] i8 (+1 / -1)        inputs
 *   legal3_dev  [n][3] u64  canonical source squares that may move forward /
 *                           diagonal-left / diagonal-right (action = sq*3+dir)
 *   mask192_dev [n][192] u8 the reference's boolean mask, one byte per action
 *   planes_dev  [n][3][8][8] f32 the reference's encoded state
 *   status_dev  [n] i8  0 live; +1 / -1 a side has won on the board (a piece
 *                       on its goal row, or the other side extinct); 2 the
 *                       side to move has no legal move
 * Any output pointer may be NULL.  n may be 0.  mask192_dev and planes_dev must
 * be 16-byte aligned, legal3_dev 8-byte aligned (AZ_ERR_ARG otherwise).
 */
int az_movegen_encode(int64_t n, const uint64_t* white_dev, const uint64_t* black_dev,
                      const int8_t* turn_dev, uint64_t* legal3_dev, uint8_t* mask192_dev,
                      float* planes_dev, int8_t* status_dev, void* stream);

/* ------------------------------------------------------------------------
 * SE-ResNet inference (bf16 MFMA).  Replaces AlphaZeroNet.forward in eval mode
 * (src/model.py:144-180) together with the softmax / value recipe of
 * MCTS._batch_evaluate (src/mcts.py:248-256).
 */
typedef struct az_net az_net;

typedef struct {
    int32_t num_blocks;   /* src/config.py:26 */
    int32_t num_filters;  /* src/config.py:27, src/model.py:110: a multiple of 32 from 32 to 256.  128 with se_ratio 8
                             (the reference's configuration) runs the tuned tower kernels, every other
                             combination one any-width kernel (k_tower_any) with the same arithmetic */
    int32_t se_ratio;     /* src/config.py:28, src/model.py:33: hidden units = num_filters / se_ratio (floor), >= 1 */
    int32_t max_batch;
    int32_t device;       /* HIP device ordinal, -1 = current */
} az_net_config;

int az_net_create(az_net** out, const az_net_config* cfg);
int az_net_destroy(az_net* n);
int az_net_max_batch(az_net* n);
/* state_host: every floating-point tensor of AlphaZeroNet.state_dict()
 * (src/model.py:123-142), flattened and concatenated in state_dict order
 * (num_batches_tracked entries skipped), float32, `count` floats in total.
 * BatchNorm is folded (eval mode) and the weights are re-laid-out for the MFMA
 * kernels here; call again after every training step that changes the weights.
 * Synchronises the device first (forwards still running on any stream keep the old
 * weights) and returns with the new ones in place. */
int az_net_load_state(az_net* n, const float* state_host, int64_t count);
/* mine_dev / theirs_dev [B] u64 canonical bitboards (planes 0 and 1 of
 * src/game.py:280-307; plane 2 is implicit).  Outputs: probs_dev [B][192]
 * softmaxed policy, value_dev [B] = P(win) - P(loss); the raw logits
 * (policy_logits_dev [B][192], wl_logits_dev [B][2]) are optional (NULL). */
int az_net_forward(az_net* n, int32_t B, const uint64_t* mine_dev, const uint64_t* theirs_dev,
                   float* probs_dev, float* value_dev, float* policy_logits_dev,
                   float* wl_logits_dev, void* stream);
/* az_net_forward without any allocation.  No counterpart in the reference: it is for
 * callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops,
 * harnesses that forbid it).  The head activations and split-K partial sums of the call,
 * which az_net_forward keeps in buffers az_net_create allocated, live here in a static
 * device array inside the library (6.75 MiB on every GPU the library is used on, held
 * until the process ends), so B is limited to AZ_NET_NOALLOC_MAX_BATCH whatever max_batch
 * the network has; a larger B fails with AZ_ERR_ARG, and so does a network of more than 128
 * filters (its head activations do not fit the array).  Inputs, outputs and results are
 * those of az_net_forward, bit for bit.  Every call on a GPU shares that GPU's array:
 * calls that overlap on one GPU (two streams, two threads, two networks) race.  The
 * network must have been created and loaded beforehand (that is where the weights are
 * allocated and the array's device address is looked up). */
#define AZ_NET_NOALLOC_MAX_BATCH 256
int az_net_forward_noalloc(az_net* n, int32_t B, const uint64_t* mine_dev, const uint64_t* theirs_dev,
                           float* probs_dev, float* value_dev, float* policy_logits_dev,
                           float* wl_logits_dev, void* stream);
/* Same as az_net_forward, from the reference's float planes [B][3][8][8]; fails with AZ_ERR_ARG
 * if the input is not an encoded state (0/1 planes, third plane all ones).
 * Synchronises the stream. */
int az_net_forward_planes(az_net* n, int32_t B, const float* planes_dev, float* probs_dev,
                          float* value_dev, float* policy_logits_dev, float* wl_logits_dev,
                          void* stream);

/* Measurement aid (no reference counterpart): with timing enabled, every 8th
 * tower-kernel launch is bracketed by HIP events on the stream it runs on (up
 * to 512 samples); az_net_tower_time waits for them, returns their mean
 * duration in milliseconds and the sample count, and clears the pool. */
int az_net_set_timing(az_net* n, int32_t enable);
/* Form of the tower kernel: 0 (default) chooses by batch size - one position per
 * workgroup up to 256 positions, two up to 512, one position per wave above; 1, 2, 4
 * force one of these (4: on v_mfma_f32_32x32x16_bf16; 16: on v_mfma_f32_16x16x32_bf16, the one
 * position per wave default), 8 the eight-wave channel-split form.  Every form computes the
 * same network (src/model.py:144-180); they differ in how a batch spreads over the chip.
 * The forms are variants of the 128-filter kernel: a network of another width takes 0 only. */
int az_net_set_tower_form(az_net* n, int32_t form);
int az_net_tower_time(az_net* n, double* mean_ms, int64_t* samples);
/* Measurement aid (no reference counterpart): with AZ_TOWER_STAMPS=1 in the environment at
 * az_net_create, the one-position-per-wave tower records per workgroup ten 64-bit counters
 * (cycles in conv loops / conv epilogues / block epilogues, 100 MHz ticks, four parts of the
 * block epilogue, first and last tick); this copies workgroups x 10 of them to the host. */
int az_net_debug_stamps(az_net* n, unsigned long long* out_host, int32_t workgroups);

/* ------------------------------------------------------------------------
 * Search engine: B independent games, one tree each, nodes in an SoA pool in
 * HBM, one wavefront per game.  Replaces MCTS.search_batch and its helpers
 * (src/mcts.py:94-189, 260-342) plus the per-move bookkeeping of
 * ParallelTrainer.execute_parallel_episodes (src/parallel_trainer.py:176-198).
 */
typedef struct az_engine az_engine;

typedef struct {
    int32_t n_games;              /* B */
    double  c_puct;               /* src/config.py:35 (a Python float) */
    double  dirichlet_alpha;      /* src/config.py:40 */
    double  dirichlet_epsilon;    /* src/config.py:41 */
    int32_t temperature_threshold;/* src/config.py:42 */
    int64_t pool_nodes_per_game;  /* 0 = default */
    uint64_t seed;                /* device RNG (noise, move sampling) */
    int32_t device;               /* HIP device ordinal, -1 = current */
} az_engine_config;

int az_engine_create(az_engine** out, const az_engine_config* cfg);
int az_engine_destroy(az_engine* e);

/* Set game g's position and drop its tree (root = None).  Host arrays of
 * length n; slots_host[i] is the game index.  winner_host may be NULL (= 0). */
int az_engine_set_positions(az_engine* e, int32_t n, const int32_t* slots_host,
                            const uint64_t* white_host, const uint64_t* black_host,
                            const int8_t* turn_host, const int8_t* winner_host, void* stream);
/* Same, but keep each game's tree: for callers that step the game themselves
 * and pass a child of the old root back as the new root. */
int az_engine_update_positions(az_engine* e, int32_t n, const int32_t* slots_host,
                               const uint64_t* white_host, const uint64_t* black_host,
                               const int8_t* turn_host, const int8_t* winner_host, void* stream);
int az_engine_get_positions(az_engine* e, uint64_t* white_host, uint64_t* black_host,
                            int8_t* turn_host, int8_t* winner_host, void* stream);
/* Games with active == 0 are skipped by every step below (the shrinking
 * active list of src/parallel_trainer.py:158-217). */
int az_engine_set_active(az_engine* e, const uint8_t* active_host, void* stream);

/* One search_batch call, step by step (so that any evaluator can sit between
 * the steps).  planes_dev [B][3][8][8] f32, need_eval_dev [B] u8,
 * probs_dev [B][192] f32 (softmaxed policy), values_dev [B] f32.
 *   roots_prepare  src/mcts.py:117-133   create missing roots, emit those to expand
 *   roots_expand   src/mcts.py:135-138   expand them; the value is not backed up
 *   add_noise      src/mcts.py:141-144, 314-326; noise_dev [B][48] f64 Dirichlet
 *                  draws in child insertion order, or NULL to draw on the device
 *   select         src/mcts.py:153-171   one PUCT descent per game
 *   expand_backup  src/mcts.py:174-187   expand + back up evaluated leaves,
 *                                        back up terminal leaves
 */
int az_engine_roots_prepare(az_engine* e, float* planes_dev, uint8_t* need_eval_dev, void* stream);
int az_engine_roots_expand(az_engine* e, const float* probs_dev, void* stream);
int az_engine_add_noise(az_engine* e, const double* noise_dev, void* stream);
int az_engine_select(az_engine* e, float* planes_dev, uint8_t* need_eval_dev, void* stream);
int az_engine_expand_backup(az_engine* e, const float* probs_dev, const float* values_dev,
                            void* stream);

/* Children of a node in insertion order (Node.children, src/mcts.py:21-40).
 * node = -1 means game g's root.  Host outputs, each with room for 48 entries;
 * prior64 holds the prior widened to double, prior_is64 says whether the
 * reference would hold it as float64.  child_index receives pool indices that
 * can be passed back as `node`.  Synchronises the stream. */
int az_engine_node_children(az_engine* e, int32_t g, int64_t node, int32_t* k_host,
                            uint8_t* actions_host, uint32_t* visits_host, float* value_sum_host,
                            double* prior64_host, uint8_t* prior_is64_host,
                            int64_t* child_index_host, uint32_t* node_visits_host,
                            float* node_value_sum_host, int64_t* node_index_host, void* stream);

/* The same for the roots of n games in one launch and one copy (what a caller
 * of search_batch reads next from every returned root, src/parallel_trainer.py:182-196).
 * Per-child outputs are [n][48], per-node outputs [n]; k_host[i] = -1 for a game
 * without a root.  Synchronises the stream. */
int az_engine_roots_children(az_engine* e, int32_t n, const int32_t* slots_host, int32_t* k_host,
                             uint8_t* actions_host, uint32_t* visits_host, float* value_sum_host,
                             double* prior64_host, uint8_t* prior_is64_host,
                             int64_t* child_index_host, uint32_t* node_visits_host,
                             float* node_value_sum_host, int64_t* node_index_host, void* stream);

/* Number of children of every game's root (-1: no root), host array [B]. */
int az_engine_root_counts(az_engine* e, int32_t* counts_host, void* stream);

/* MCTS.get_action_probs (src/mcts.py:191-225) for every active game, with the
 * temperature schedule of src/parallel_trainer.py:164 when temps_host is NULL
 * (1.0 while the game's ply count < temperature_threshold, else 0), or the
 * given per-game temperatures (>= 0).  Temperatures 0 and 1 reproduce the reference bit for
 * bit; any other one evaluates count ** (1 / T) / sum in float32 on the device and equals
 * numpy's result within rounding (relative 1e-5).  probs_dev [B][192] f32. */
int az_engine_action_probs(az_engine* e, const float* temps_host, float* probs_dev, void* stream);

/* src/parallel_trainer.py:190-198: play action actions_host[g] (< 0: leave the
 * game alone) and re-root the tree at that child (None if absent). */
int az_engine_advance(az_engine* e, const int16_t* actions_host, void* stream);
/* Re-root game g at an arbitrary pool node without touching the position
 * (callers that pass Node handles back as roots). node = -1 drops the tree. */
int az_engine_set_root(az_engine* e, int32_t g, int64_t node, void* stream);
/* The same for n games in one launch: game slots_host[i] is re-rooted at
 * nodes_host[i] (what search_batch does with a list of reused roots,
 * src/mcts.py:117-125).  A slot may appear once.  A node index that is not in
 * the game's tree is reported by the next call that checks the engine's error flag. */
int az_engine_set_roots(az_engine* e, int32_t n, const int32_t* slots_host, const int64_t* nodes_host, void* stream);

/* A tree the caller built on the host becomes game g's tree (search_batch accepts any Node as a
 * root, src/mcts.py:117-125; Node, src/mcts.py:21-40).  Nodes in breadth-first order, node 0 the
 * root, the children of a node one contiguous block in insertion order: first_child_host[i],
 * n_children_host[i] (0: a leaf).  visits / value_sum / prior / action as the reference's
 * visit_count, value_sum (held as float32, what a search leaves there), prior (float32) and the
 * key in the parent's children (ignored for the root).  uniform64_host[i] != 0 (array may be
 * NULL): node i's children hold the float64 prior 1/k of src/mcts.py:277-278.
 * root_prior64_host (NULL, or one double per child of the root): the root's children hold
 * float64 priors, as after _add_dirichlet_noise (src/mcts.py:314-326).  The game's position is
 * whatever az_engine_set_positions gave it.  Synchronises the stream. */
int az_engine_import_tree(az_engine* e, int32_t g, int32_t n_nodes, const uint32_t* visits_host,
                          const float* value_sum_host, const float* prior_host,
                          const uint32_t* first_child_host, const uint8_t* n_children_host,
                          const uint8_t* action_host, const uint8_t* uniform64_host,
                          const double* root_prior64_host, void* stream);

/* Diagnostics: nodes in use per game.  Like every call that synchronises, it reports a
 * pending device-side error (node pool exhausted, path too deep, bad node index, sample
 * queue full).  A condition is reported once and then cleared: the engine stays usable,
 * the game that hit the limit keeps a consistent but incomplete tree. */
int az_engine_pool_usage(az_engine* e, uint32_t* used_host, void* stream);
/* Sum of selection path lengths (nodes below the root) and number of descents
 * since creation: mean depth = depth_sum / descents. */
int az_engine_stats(az_engine* e, uint64_t* depth_sum_host, uint64_t* descents_host, void* stream);

/* ------------------------------------------------------------------------
 * Fused self-play: the whole of ParallelTrainer.execute_parallel_episodes
 * (src/parallel_trainer.py:137-223) on the device, evaluating leaves with an
 * attached az_net (whose max_batch must be >= n_games).
 */
typedef struct {
    uint64_t simulations;     /* descents backed up since creation */
    uint64_t nn_evals;        /* leaves + roots that needed an evaluation, whichever API ran the search
                                 (az_engine_search / az_engine_play_move, or the step calls
                                 az_engine_roots_prepare ... az_engine_expand_backup) */
    uint64_t moves;           /* plies played */
    uint64_t games_finished;
    uint64_t samples_pending; /* finished-game samples waiting in the queue */
    uint64_t network_rows;    /* positions the network launches of the fused search (az_engine_search,
                                 az_engine_play_move) actually computed; terminal leaves and idle games
                                 take no row, so it equals nn_evals as long as every search of this
                                 engine ran through those two calls (evaluations the step calls asked the
                                 caller for count in nn_evals only) */
} az_selfplay_stats;

int az_engine_attach_net(az_engine* e, az_net* n);
/* One MCTS.search_batch call (src/mcts.py:94-189): expand missing roots,
 * optional device-drawn Dirichlet noise, num_simulations simulations. */
int az_engine_search(az_engine* e, int32_t num_simulations, int32_t add_noise, void* stream);
/* Latency mode of MCTS.search (one game at a time, src/arena.py:248-265, src/web.py:447-470):
 * num_simulations simulations of the n games slots_host[0..n), leaf_batch descents of a game per
 * network call instead of the reference's one (src/mcts.py:153-187).  Descents whose evaluation
 * is pending count as one visit with a lost outcome on every node of their path (virtual loss)
 * until they are backed up; two descents of a batch that reach the same leaf share its
 * evaluation.  Visit counts and value sums stay exact sums, but the order of the descents is
 * not the reference's: results differ from az_engine_search for leaf_batch > 1 (use that call
 * for the reference's results).  Roots must exist (az_engine_search with 0 simulations expands
 * them); a game without one makes no descent and the call fails with AZ_ERR_STATE.
 * n * leaf_batch <= 256 rows, leaf_batch <= 32; the attached network must take that
 * many rows.  No counterpart in the reference. */
int az_engine_search_leaf_parallel(az_engine* e, int32_t n, const int32_t* slots_host, int32_t num_simulations,
                                   int32_t leaf_batch, void* stream);
/* One iteration of the self-play loop (src/parallel_trainer.py:158-217) for
 * every active game: search, visit-count policy with the temperature schedule,
 * record (state, pi, turn), choose the move (actions_host[g] if given, else a
 * device draw from pi), step + re-root, and for finished games emit their
 * samples into the queue and restart them (restart_finished != 0) or drop them
 * from the active set.  Samples are queued in the reference's order: move by move,
 * within a move by ascending game index, within a game by ply
 * (src/parallel_trainer.py:176-217); a game that does not fit the queue is dropped
 * whole and reported as AZ_ERR_CAPACITY. */
int az_engine_play_move(az_engine* e, int32_t num_simulations, int32_t add_noise,
                        const int16_t* actions_host, int32_t restart_finished, void* stream);
int az_engine_chosen_actions(az_engine* e, int16_t* actions_host, void* stream);
/* Move the queued samples into the caller's device buffers in the layout of
 * src/parallel_trainer.py:380-382 (states [n][3][8][8], policies [n][192],
 * wls [n][2], float32) and empty the queue; *n_out = samples written.  With
 * states_dev == NULL only the count is reported. */
int az_engine_drain_samples(az_engine* e, int64_t max_samples, float* states_dev,
                            float* policies_dev, float* wls_dev, int64_t* n_out, void* stream);
/* The same queue in the wire format of the exchange between GPUs (no counterpart in the
 * reference, which is one process: its src/parallel_trainer.py:201-213 appends to a Python
 * list): rows of AZ_PACKED_SAMPLE_BYTES = canonical bitboards mine u64 | theirs u64 |
 * wl f32[2] | pi f32[192], about half the size of the float planes.  packed_dev has room
 * for max_samples rows and is 8-byte aligned; with packed_dev == NULL only the count is
 * reported.  az_unpack_samples expands n such rows (from any rank) into the layout of
 * az_engine_drain_samples; it needs no engine. */
#define AZ_PACKED_SAMPLE_BYTES 792
int az_engine_drain_packed(az_engine* e, int64_t max_samples, void* packed_dev, int64_t* n_out, void* stream);
int az_unpack_samples(int64_t n, const void* packed_dev, float* states_dev, float* policies_dev,
                      float* wls_dev, void* stream);
int az_engine_selfplay_stats(az_engine* e, az_selfplay_stats* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
