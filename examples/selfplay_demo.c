/* Plain-C use of the drop-in boundary (include/dbaz.h, libdbaz_hip.so): plays N complete 3x3
 * Dots & Boxes games with the formula evaluator (no network weights needed) and prints the
 * dataset rows' summary -- the call sequence a non-Python host of the reference's self-play
 * path (self_play.py:291-306) would use.
 *
 *   gcc -std=c11 -Iinclude examples/selfplay_demo.c -o selfplay_demo \
 *       -Ldotsboxesaz_amd -ldbaz_hip -Wl,-rpath,$PWD/dotsboxesaz_amd
 *   ./selfplay_demo 64
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dbaz.h"

#define CHECK(e, call)                                                            \
    do {                                                                          \
        int _rc = (call);                                                         \
        if (_rc != DBAZ_OK) {                                                     \
            fprintf(stderr, "%s failed (%d): %s\n", #call, _rc, dbaz_last_error(e)); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

int main(int argc, char **argv)
{
    const int n_games = argc > 1 ? atoi(argv[1]) : 64;
    const int rows = 3, cols = 3, H = rows + 1, W = cols + 1, A = 2 * H * W, F = 3 * H * W;
    dbaz_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.rows = rows; cfg.cols = cols; cfg.n_slots = 32; cfg.mcts_num_read = 50;
    cfg.cpuct = 1.25; cfg.cpuct_base = 19652; cfg.noise_alpha = 0.8; cfg.noise_coeff = 0.25;
    cfg.reuse_tree = 1; cfg.n_temp = 2; cfg.temp_idx[0] = 0; cfg.temp_val[0] = 1.0; cfg.temp_idx[1] = 6; cfg.temp_val[1] = 0.02;
    cfg.evaluator = DBAZ_EVAL_FORMULA_HASH; cfg.seed = 7;
    dbaz_engine *e = NULL;
    if (dbaz_create(&cfg, &e) != DBAZ_OK) {
        fprintf(stderr, "dbaz_create: %s\n", dbaz_last_error(NULL));
        return 2;
    }
    CHECK(e, dbaz_selfplay_start(e, n_games, 0));
    CHECK(e, dbaz_run(e, 0));
    dbaz_counters c;
    CHECK(e, dbaz_get_counters(e, &c));
    int32_t n = 0;
    CHECK(e, dbaz_fetch_samples(e, 0, &n, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL));
    int32_t *game = malloc(sizeof(int32_t) * n), *visits = malloc(sizeof(int32_t) * (size_t)n * A);
    int32_t *tsize = malloc(sizeof(int32_t) * n), *tcount = malloc(sizeof(int32_t) * n);
    int16_t *midx = malloc(sizeof(int16_t) * n), *move = malloc(sizeof(int16_t) * n), *x = malloc(sizeof(int16_t) * (size_t)n * F);
    int16_t *maxd = malloc(sizeof(int16_t) * n), *played = malloc(sizeof(int16_t) * n);
    int8_t *player = malloc(n), *z = malloc(n);
    double *pi = malloc(sizeof(double) * (size_t)n * A);
    float *q = malloc(sizeof(float) * n);
    int32_t got = 0;
    CHECK(e, dbaz_fetch_samples(e, n, &got, game, midx, move, player, x, visits, pi, z, maxd, tsize, tcount, q, played));
    long wins = 0, losses = 0, draws = 0;
    double pisum = 0;
    for (int i = 0; i < got; i++) {
        if (midx[i] == 0) { if (z[i] > 0) wins++; else if (z[i] < 0) losses++; else draws++; }
        for (int a = 0; a < A; a++) pisum += pi[(size_t)i * A + a];
    }
    printf("games %lld rows %d expansions %lld first-player wins/losses/draws %ld/%ld/%ld mean(sum pi) %.6f\n",
           (long long)c.games_finished, got, (long long)c.expansions, wins, losses, draws, got ? pisum / got : 0.0);
    dbaz_destroy(e);
    return (c.games_finished == n_games && got > 0) ? 0 : 3;
}
