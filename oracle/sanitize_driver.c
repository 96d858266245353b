/*
 * sanitize_driver.c — runs the CPU oracle (az_oracle.c) through every entry point under
 * AddressSanitizer and UndefinedBehaviorSanitizer.  TEST INFRASTRUCTURE ONLY, like the oracle
 * itself: the oracle is what the GPU results are checked against, so an out-of-bounds read or a
 * signed overflow inside it would be a wrong answer that nothing else checks.
 *
 *   make -C oracle sanitize && oracle/_san/sanitize_driver
 *
 * Scenario: 60,000 positions of seeded random play through the batched move generator and the
 * per-position functions (both must agree); then 6 games of self-play to the end, 48
 * simulations per move, with a synthetic evaluator (probabilities from a hash of the position:
 * peaked, flat, and all-zero rows that force the uniform-prior fallback), Dirichlet-like noise on
 * every root, tree reuse across plies, one game re-started in the middle.  Exit status 0 and the
 * line "sanitize_driver ok" mean that neither sanitizer reported anything and the self-checks hold.
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "az_oracle.h"

static uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* a deterministic "network": row kind by hash (0: all zero, 1: flat, other: peaked) */
static void evaluate(const float* planes, float* probs, float* value) {
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 128; ++i) h = mix(h ^ (uint64_t)(planes[i] != 0.0f) ^ ((uint64_t)i << 8));
    const int kind = (int)(h % 7);
    float sum = 0.f;
    for (int a = 0; a < 192; ++a) {
        const uint64_t r = mix(h + (uint64_t)a * 0x9E3779B97F4A7C15ull);
        float p = (kind == 0) ? 0.f : ((kind == 1) ? 1.f : (float)((r >> 40) & 0xFFFF) * (float)((r >> 20) & 7));
        probs[a] = p;
        sum += p;
    }
    if (sum > 0.f)
        for (int a = 0; a < 192; ++a) probs[a] /= sum;
    *value = (float)((double)(mix(h ^ 77) >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { fprintf(stderr, "sanitize_driver: check failed at line %d: %s\n", __LINE__, #cond); exit(2); } \
    } while (0)

static void movegen_part(void) {
    const int64_t n = 60000;
    uint64_t* w = malloc(n * 8); uint64_t* b = malloc(n * 8);
    int8_t* t = malloc(n); int8_t* win = malloc(n);
    uint64_t* legal3 = malloc(n * 24); uint8_t* mask = malloc(n * 192);
    float* planes = malloc(n * 192 * sizeof(float)); int8_t* status = malloc(n);
    CHECK(w && b && t && win && legal3 && mask && planes && status);
    azo_random_positions(n, 12345, w, b, t, win);
    azo_movegen_batch(n, w, b, t, legal3, mask, planes, status);
    azo_movegen_batch(n, w, b, t, legal3, NULL, NULL, status);   /* optional outputs absent */
    for (int64_t i = 0; i < n; i += 7) {
        azo_pos p = {w[i], b[i], t[i], win[i]};
        uint8_t m1[192], acts[48];
        float pl[192];
        azo_legal_mask(&p, m1);
        azo_encode_state(&p, pl);
        CHECK(memcmp(m1, mask + i * 192, 192) == 0);
        CHECK(memcmp(pl, planes + i * 192, sizeof pl) == 0);
        const int k = azo_legal_actions(&p, acts);
        int set = 0;
        for (int a = 0; a < 192; ++a) set += m1[a];
        CHECK(k == set && k <= 48);
        for (int j = 0; j < k; ++j) CHECK(m1[acts[j]] == 1);
        float rw, rl;
        azo_get_result(&p, &rw, &rl);
        (void)azo_is_terminal(&p);
        (void)azo_status(&p);
    }
    free(w); free(b); free(t); free(win); free(legal3); free(mask); free(planes); free(status);
}

static void search_part(void) {
    enum { G = 6, SIMS = 48 };
    azo_mcts* m = azo_mcts_create(G, 1.5);
    CHECK(m);
    azo_pos start;
    azo_start_position(&start);
    float* planes = malloc(G * 192 * sizeof(float));
    float* probs = malloc(G * 192 * sizeof(float));
    float values[G];
    uint8_t need[G], active[G];
    double* noise = malloc(G * 48 * sizeof(double));
    CHECK(planes && probs && noise);
    for (int g = 0; g < G; ++g) { azo_mcts_set_game(m, g, &start); active[g] = 1; }
    azo_mcts_set_active(m, active);
    int live = G, plies = 0, restarted = 0;
    uint64_t rng = 99;
    while (live > 0 && plies < 400) {
        memset(probs, 0, G * 192 * sizeof(float));
        azo_mcts_roots_prepare(m, planes, need);
        for (int g = 0; g < G; ++g)
            if (need[g]) evaluate(planes + g * 192, probs + g * 192, &values[g]);
        azo_mcts_roots_expand(m, probs);
        for (int g = 0; g < G; ++g) {         /* any positive vector that sums to one will do here */
            double s = 0;
            for (int j = 0; j < 48; ++j) { rng = mix(rng + j); noise[g * 48 + j] = (double)(rng >> 11) + 1.0; s += noise[g * 48 + j]; }
            for (int j = 0; j < 48; ++j) noise[g * 48 + j] /= s;
        }
        azo_mcts_add_noise(m, noise, 0.25);
        for (int s = 0; s < SIMS; ++s) {
            azo_mcts_select(m, planes, need);
            for (int g = 0; g < G; ++g)
                if (need[g]) evaluate(planes + g * 192, probs + g * 192, &values[g]);
            azo_mcts_expand_backup(m, probs, values);
        }
        for (int g = 0; g < G; ++g) {
            if (!active[g]) continue;
            uint8_t acts[48], is64[48];
            uint32_t N[48], rootN;
            float W[48], rootW, pi[192];
            double p64[48];
            const int k = azo_mcts_root_children(m, g, acts, N, W, p64, is64, &rootN, &rootW);
            CHECK(k > 0 && k <= 48);
            uint32_t tot = 0;
            for (int j = 0; j < k; ++j) tot += N[j];
            CHECK(tot >= SIMS && rootN >= SIMS);
            azo_mcts_action_probs(m, g, plies >= 16, pi);
            float s = 0;
            for (int a = 0; a < 192; ++a) s += pi[a];
            CHECK(s > 0.999f && s < 1.001f);
            rng = mix(rng + g);
            int pick = (int)(rng % (uint64_t)k);           /* also unvisited children: the tree is then dropped */
            azo_mcts_advance(m, g, acts[pick]);
            azo_pos p;
            azo_mcts_get_game(m, g, &p);
            if (azo_is_terminal(&p)) {
                if (!restarted && g == 2) { azo_mcts_set_game(m, g, &start); restarted = 1; }
                else { active[g] = 0; --live; }
            }
        }
        azo_mcts_set_active(m, active);
        ++plies;
    }
    CHECK(live == 0);
    CHECK(azo_mcts_mean_depth(m) >= 0.0 && azo_mcts_total_nodes(m) > 0);
    printf("search part: %d plies, mean depth %.2f, %" PRId64 " nodes\n", plies, azo_mcts_mean_depth(m), azo_mcts_total_nodes(m));
    azo_mcts_destroy(m);
    free(planes); free(probs); free(noise);
}

int main(void) {
    movegen_part();
    search_part();
    printf("sanitize_driver ok\n");
    return 0;
}
