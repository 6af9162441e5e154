/*
 * dbaz.h -- C ABI of the MI355X-native self-play rollout engine (libdbaz_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of damlobster/DotsBoxesAZ:
 * rules + sequential PUCT search + batched policy/value evaluation + the
 * self-play driver.  The reference has no FFI layer of its own (it is pure
 * Python); each entry point below names the reference interface it replaces
 * (file:line relative to the reference repo).  The ctypes binding a maintainer
 * would add on the reference side is shown in INTEGRATION.md and shipped in
 * dotsboxesaz_amd/_lib.py.
 *
 * Conventions
 *   - every function returns 0 on success, a DBAZ_E* code otherwise;
 *     dbaz_last_error() returns the message (maps to ValueError for
 *     DBAZ_EILLEGAL, RuntimeError otherwise);
 *   - all pointers are HOST pointers unless the name ends in _dev; the caller
 *     owns every buffer; no torch types, no C++ types;
 *   - a handle is single-threaded; one handle per GPU / rank; no process-global
 *     state (board size is per handle, unlike BoxesState.init_static_fields; the
 *     message of a call that has no handle -- a failed create -- is kept per thread);
 *   - there is NO CPU fallback: without a HIP device dbaz_create fails.
 *
 * Board geometry: rows x cols boxes, H = rows+1, W = cols+1, A = 2*H*W action
 * slots, action = p*H*W + l*W + c (dots_boxes_game.py:62).  A <= 256.
 */
#ifndef DBAZ_H
#define DBAZ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DBAZ_OK 0
#define DBAZ_EINVAL 1   /* bad argument */
#define DBAZ_EILLEGAL 2 /* illegal move (reference: ValueError, dots_boxes_game.py:63-65) */
#define DBAZ_EDEVICE 3  /* HIP runtime error / no device */
#define DBAZ_EPOOL 4    /* per-game node pool exhausted: raise nodes_per_slot */
#define DBAZ_ESTATE 5   /* call sequence error */

#define DBAZ_MAX_A 256
#define DBAZ_ABI_VERSION 3 /* dbaz_version(): bumped whenever dbaz_config / dbaz_counters change layout or meaning */

/* dbaz_config.debug_flags */
#define DBAZ_DBG_EARLY_JOIN 1u  /* join the driver pass in front of the network launch (round 2's first order) */
#define DBAZ_DBG_NO_FALLBACK 2u /* skip the exact-f32 safety-net launch of nn_precision = 1 (timing runs only) */
#define DBAZ_DBG_LAZY_GC 4u     /* node collector recycles dropped nodes only when fewer than 64 indices are available (A/B) */
#define DBAZ_RESULT_NONE 2

/* evaluator kinds (what plays the role of async_nn, mcts.py:187) */
#define DBAZ_EVAL_FORMULA_HASH 0    /* integer-hash priors/value (bit-exact test evaluator) */
#define DBAZ_EVAL_FORMULA_UNIFORM 1 /* uniform priors, v = 0 (tree + rules only) */
#define DBAZ_EVAL_RESNET 2          /* ResNetZero, HIP MFMA kernels (nn.py:108-122) */
#define DBAZ_EVAL_SIMPLENN 3        /* SimpleNN, 3x3 boards (dots_boxes_nn.py:61-98) */
#define DBAZ_EVAL_EXTERNAL 4        /* caller supplies (p, v) per leaf: dbaz_select / dbaz_expand_backup */

typedef struct dbaz_engine dbaz_engine;

/* configuration.py:82-100 values + sizing */
typedef struct {
    int32_t rows, cols;
    int32_t n_slots;        /* concurrent games resident on the GPU */
    int32_t nodes_per_slot; /* node pool per game; 0 = 10*(mcts_num_read+2), up to 4x that while all pools fit 96 GiB */
    int32_t mcts_num_read;  /* self_play.mcts.mcts_num_read */
    double cpuct, cpuct_base; /* self_play.mcts.mcts_cpuct */
    double noise_alpha, noise_coeff; /* self_play.noise */
    int32_t reuse_tree;     /* self_play.reuse_mcts_tree */
    int32_t n_temp;         /* self_play.mcts.temperature {move_idx: T} */
    int32_t temp_idx[8];
    double temp_val[8];
    int32_t evaluator;      /* DBAZ_EVAL_* */
    int32_t device;         /* HIP device ordinal */
    uint64_t seed;          /* Philox key for move sampling / Dirichlet noise */
    int32_t max_out_rows;   /* capacity of the finished-sample buffer; 0 = max(4096, 2*n_slots*(E+1)) */
    int32_t nn_precision;   /* 0 = exact f32 MFMA; 1 = f16x3 split MFMA (f32-grade; an evaluation whose activations leave
                             * f16's range is redone in exact f32 on the device, counters.f32_fallback_evals).  (A library built
                             * with -DDBAZ_DEBUG also accepts the A/B tilings 2, 3, 4 of tools/ab_tilings.sh.) */
    int32_t match_play;     /* two-model match play (self_play.compute_elo, :309-344): the evaluator of a move's
                               search is model (root.to_play XOR game_idx&1) */
    int32_t evaluator2;     /* DBAZ_EVAL_* of model 1 (match play) */
    int32_t transposition_cache; /* 0 = on for network evaluators (default), 1 = off, 2 = on for network AND formula
                             * evaluators (parity tests of the hit path).  The reference caches (p, v) by position hash
                             * (utils/proxies.py:35-43); results are identical either way */
    int32_t max_pending_evals; /* UCT_search's max_pending_evals (mcts.py:183) = self_play.mcts.max_async_searches
                             * (self_play.py:27-30): simulations of ONE tree in flight.  0 / 1 = the sequential search (all
                             * parity paths); K > 1 = waves of up to K selections with virtual loss, one batched evaluation per
                             * wave (dbaz_search / dbaz_search_timed, and the self-play driver when selfplay_pending != 0) */
    int32_t selfplay_pending;  /* self-play driver (dbaz_run / dbaz_step): 0 = one simulation per game and step (default: the
                             * batch comes from the concurrent games); 1 = every search of a game runs in waves of
                             * max_pending_evals simulations with the reference's bookkeeping (visits added at backup,
                             * mcts.py:105-132), as self_play.py:27-30 does with max_async_searches */
    int32_t eval_round;     /* "full rounds only" (DESIGN 4): 0 = the network's own round size (workgroups per launch round x
                             * samples per workgroup), -1 = off (every leaf is evaluated in the step that selected it),
                             * r > 0 = rounds of r leaves (tests: small runs through the same path) */
    int32_t eval_defer_max; /* with eval_round > 0: the largest left-over that is put off to the next step (0 = r - 1) */
    uint32_t debug_flags;   /* measurement aids, 0 in production: DBAZ_DBG_* */
    int32_t reserved[3];
} dbaz_config;

typedef struct {
    int64_t steps;          /* simulation steps launched */
    int64_t expansions;     /* completed _search calls (node expansions) */
    int64_t nn_evals;       /* leaves sent to the evaluator */
    int64_t terminal_leaves;
    int64_t sum_path;       /* sum over expansions of the search-path length (nodes) */
    int64_t games_finished;
    int64_t rows_ready;     /* sample rows waiting in the output buffer */
    int64_t moves_played;
    int64_t pool_high_water;/* max nodes in use in any slot */
    int32_t active_slots;   /* slots still playing */
    int32_t error_slots;    /* slots stopped by an error (pool exhausted) */
    int32_t blocked_slots;  /* finished games waiting for room in the output buffer: fetch samples */
    int32_t f32_fallback_evals; /* nn_precision = 1, ResNetZero: leaves whose f16x3 evaluation left f16's range and were
                                 * re-evaluated by the exact-f32 tower in the same step (normally 0) */
    /* HIP-event timing of the last timed region (dbaz_timing_begin/_end) */
    double ms_total, ms_tree, ms_nn;
    int64_t nn_launches;    /* conv-tower launches inside the timed region */
    double ms_nn_tower;     /* summed duration of the dominant conv kernel */
    int64_t cache_hits;     /* leaves whose (p, v) came from an already evaluated twin position of the same tree */
    int64_t pool_resets;    /* self-play driver: moves that started from a fresh root because the subtree kept by tree reuse would
                             * not have left mcts_num_read + 2 nodes of the slot's pool free (the reference's trees are unbounded;
                             * 0 unless a network concentrates its visits for many plies in a row: raise nodes_per_slot) */
} dbaz_counters;

const char *dbaz_last_error(const dbaz_engine *e); /* e may be NULL: error of the last dbaz_create */
int dbaz_version(void);            /* DBAZ_ABI_VERSION the library was built with */
const char *dbaz_build_info(void); /* "src=<sha256[:16] of csrc/ + include/dbaz.h> nn=<the same of the network kernels>[ debug]": ties profiles/ counter files to a build */
int dbaz_nodes_per_slot(const dbaz_engine *e); /* the node pool size in effect (dbaz_config.nodes_per_slot = 0: the default rule) */

int dbaz_create(const dbaz_config *cfg, dbaz_engine **out);
void dbaz_destroy(dbaz_engine *e);
int dbaz_sync(dbaz_engine *e);

/* ---- G1-G6: batched rules on SoA states (dots_boxes_game.py:30-109) --------------
 * state i = { edges[4*i..4*i+3] : bitmask of played edges (= hash[0], :106-109),
 *             b2c2[2*i..2*i+1]  : 2*boxes_to_close (integers; :39,:86),
 *             to_play[i], just_played[i] (-1 = None) } */
int dbaz_rules_init(dbaz_engine *e, int32_t n, uint64_t *edges, int16_t *b2c2, int8_t *to_play,
                    int8_t *just_played);                                   /* __init__ :30-39 */
int dbaz_rules_valid_moves(dbaz_engine *e, int32_t n, const uint64_t *edges,
                           uint8_t *valid /*[n*A]*/);                       /* get_valid_moves :44-49 */
int dbaz_rules_play(dbaz_engine *e, int32_t n, uint64_t *edges, int16_t *b2c2, int8_t *to_play,
                    int8_t *just_played, const int32_t *moves,
                    int8_t *n_closed /*[n]; -1 = illegal, state untouched*/,
                    int8_t *closed_lc /*[n*4] (l,c) pairs, -1 padded; may be NULL*/); /* play_ :61-89 */
int dbaz_rules_result(dbaz_engine *e, int32_t n, const int16_t *b2c2, const int8_t *to_play,
                      int8_t *result /* 1,0,-1 or DBAZ_RESULT_NONE */);    /* get_result :51-59 */
int dbaz_rules_features(dbaz_engine *e, int32_t n, const uint64_t *edges, const int16_t *b2c2,
                        const int8_t *to_play, int16_t *x /*[n*3*H*W]*/);  /* get_features :96-100 */

/* ---- N1-N3: policy/value network (nn.py:108-129,155-160; dots_boxes_nn.py:61-105) ----
 * Weights arrive as state_dict entries under the reference's key names
 * ("resnet.resblocks.3.conv1.weight", ...), any order; dbaz_nn_commit folds the
 * eval-mode BatchNorms and uploads. */
/* kind DBAZ_EVAL_RESNET: channels <= 128 (zero-padded to 16/32/64/128), blocks, head_channels, value_fc as in
 * configuration.py:134-156.  kind DBAZ_EVAL_SIMPLENN: the other arguments are ignored; 3x3 boards only. */
int dbaz_nn_configure(dbaz_engine *e, int32_t kind /*DBAZ_EVAL_RESNET|SIMPLENN*/, int32_t channels,
                      int32_t blocks, int32_t head_channels, int32_t value_fc);
/* match play: subsequent dbaz_nn_configure/_set_tensor/_commit/_predict address model 0 or 1 */
int dbaz_nn_select_model(dbaz_engine *e, int32_t model);
int dbaz_nn_set_tensor(dbaz_engine *e, const char *key, const float *data, int64_t numel);
int dbaz_nn_commit(dbaz_engine *e);
/* NeuralNetWrapper.predict_sync: X float32 [n,3,H,W] -> softmax p [n,A], tanh v [n] */
int dbaz_nn_predict(dbaz_engine *e, int32_t n, const float *X, float *p, float *v);

/* ---- M1-M9: batched sequential search, one tree per slot (mcts.py) ----------------- */
/* per-call arguments of UCT_search: cpuct=(c, base), dirichlet=(alpha, coeff) (mcts.py:183,205,211) */
int dbaz_set_search_params(dbaz_engine *e, double cpuct, double cpuct_base, double noise_alpha, double noise_coeff);
/* create_root_uct_node for every slot: slot i starts from the position reached by
 * moves[offsets[i]..offsets[i+1]) (NULL, NULL = empty boards).  mcts.py:156-160 */
int dbaz_set_positions(dbaz_engine *e, const int16_t *moves, const int32_t *offsets);
/* UCT_search on every slot with max_pending_evals=1 semantics (mcts.py:183-244).
 *   num_reads[n_slots]  NULL = the driver rule min(4*n_valid!, mcts_num_read) (self_play.py:64-65)
 *   noise[n_slots*A]    Dirichlet sample per slot over ALL A slots (used if noise_alpha>0);
 *                       NULL = drawn on the device (Philox)
 * Blocks until every slot finished its reads.  Not for DBAZ_EVAL_EXTERNAL. */
int dbaz_search(dbaz_engine *e, const int32_t *num_reads, const double *noise);
/* the same with UCT_search's wall-clock cut-off (mcts.py:201-203,232-233; players.py:55-69 calls
 * UCT_search(node, int(1e12), ..., time_limit)): reads that have not started when time_limit_s (<= 0: 120 s) has
 * elapsed are dropped; the root expansion is never cut; the clock is read between waves */
int dbaz_search_timed(dbaz_engine *e, const int32_t *num_reads, const double *noise, double time_limit_s);
/* per-call max_pending_evals of a handle created with max_pending_evals = K > 1: 1 <= k <= K simulations in flight
 * (k = 1 through the same kernels reproduces the sequential search bit for bit).
 * virtual_visits = 0: the reference's bookkeeping -- a simulation's visits are added at backup, only `total_value -=
 * VIRTUAL_LOSS` marks its path while it is pending (mcts.py:105-132); with an evaluator that suspends once per call
 * the reference's searches run in exactly these waves, and the results are equal bit for bit (tests/golden/pending.npz).
 * virtual_visits = 1 (default): the visit is counted on every edge of the path at selection time as well, so the
 * simulations of a wave spread over the tree (the reference's in-flight leaves are mostly duplicates). */
int dbaz_set_pending(dbaz_engine *e, int32_t k, int32_t virtual_visits);
/* external-evaluator form of the same call: begin, then loop
 *   dbaz_select -> (evaluate leaves on the host) -> dbaz_expand_backup  until *n_active == 0 */
int dbaz_search_begin(dbaz_engine *e, const int32_t *num_reads, const double *noise);
int dbaz_select(dbaz_engine *e, int32_t *n_active,
                int16_t *leaf_x /*[n_slots*3HW] get_features of each leaf*/,
                uint8_t *need_eval /*[n_slots] 1 = non-terminal leaf of an active slot*/);
int dbaz_expand_backup(dbaz_engine *e, const float *p /*[n_slots*A]*/, const float *v /*[n_slots]*/);
/* root arrays after a search (what UCT_search returns / tests inspect); any pointer may be NULL */
int dbaz_get_roots(dbaz_engine *e, double *priors /*[n*A]*/, float *total_value /*[n*A]*/,
                   int32_t *visits /*[n*A]*/, int32_t *changed /*[n*A]*/,
                   int32_t *stats /*[n*3]: max_deepness, tree_size, terminal_count*/,
                   float *q_value /*[n]*/, float *root_tv /*[n]*/, int32_t *root_nv /*[n]*/);
/* root game states (edges/b2c2/to_play/just_played/result/expanded) */
int dbaz_get_root_states(dbaz_engine *e, uint64_t *edges, int16_t *b2c2, int8_t *to_play,
                         int8_t *just_played, int8_t *result, int8_t *expanded);
/* init_mcts_tree on every slot (mcts.py:163-180); moves[i] < 0 leaves slot i alone */
int dbaz_advance(dbaz_engine *e, const int32_t *moves, int32_t reuse_tree);

/* ---- D1-D3: self-play driver (self_play.py:19-156) ---------------------------------- */
/* Start playing games first_game_idx .. first_game_idx+n_games-1; every slot takes the next
 * unplayed index when its game ends (generate_games' chunking, self_play.py:291-306). */
int dbaz_selfplay_start(dbaz_engine *e, int64_t n_games, int64_t first_game_idx);
/* teacher forcing for parity tests: moves / Dirichlet vectors for game `game_idx`
 * (noise may be NULL).  Must be called before dbaz_selfplay_start. */
int dbaz_selfplay_script(dbaz_engine *e, int64_t game_idx, const int16_t *moves, int32_t n_moves,
                         const double *noise /*[n_moves*A]*/);
/* synthetic mid-game population for benchmarking: slot i is advanced by plies[i]
 * uniformly random legal moves before its first search. */
int dbaz_selfplay_fastforward(dbaz_engine *e, const int32_t *plies);
/* benchmark population, second knob: the FIRST search of slot i is cut to min(rule, first_reads[i]) reads
 * (0 = the rule), so that the slots' move boundaries -- re-rooting, tree reuse, game turnover -- are spread over a
 * whole search instead of falling into the same steps.  Like dbaz_selfplay_fastforward it applies to the next
 * dbaz_selfplay_start only and is never used on the parity paths. */
int dbaz_selfplay_stagger(dbaz_engine *e, const int32_t *first_reads);
/* benchmark population, third knob: the first plies[i] plies of slot i's FIRST game are searched with `reads` simulations
 * per move instead of mcts_num_read -- real search, network, temperature schedule and tree reuse, only cheaper -- so that
 * the slots reach mid-game positions of the kind search-based play produces (uniformly random plies, the fastforward knob,
 * give positions with more terminal leaves and transpositions than games do).  Applies to the next dbaz_selfplay_start. */
int dbaz_selfplay_quickplay(dbaz_engine *e, const int32_t *plies, int32_t reads);
int dbaz_step(dbaz_engine *e, int32_t k);             /* k simulation steps, asynchronous */
/* until all games are finished, max_steps (if > 0) are done, or every remaining slot is
 * blocked on a full output buffer (counters.blocked_slots == active_slots: fetch and call again) */
int dbaz_run(dbaz_engine *e, int64_t max_steps);
int dbaz_get_counters(dbaz_engine *e, dbaz_counters *out);
int dbaz_timing_begin(dbaz_engine *e);
int dbaz_timing_end(dbaz_engine *e);
/* SelfPlay.get_datasets rows of finished games (self_play.py:95-156), sorted by
 * (game_idx, move_idx).  Drains the output buffer. */
int dbaz_fetch_samples(dbaz_engine *e, int32_t max_rows, int32_t *n_rows,
                       int32_t *game_idx, int16_t *move_idx, int16_t *move, int8_t *player,
                       int16_t *x /*[rows*3HW]*/, int32_t *visits /*[rows*A]*/, double *pi /*[rows*A]*/,
                       int8_t *z, int16_t *max_deepness, int32_t *tree_size, int32_t *terminal_count,
                       float *q_value, int16_t *played);
/* device-resident replay rows for the RCCL all-gather (multi-GPU iteration end):
 * fixed-stride packed rows, see DESIGN.md "replay row".  Returns a DEVICE pointer. */
int dbaz_replay_rows_dev(dbaz_engine *e, void **rows_dev, int32_t *n_rows, int32_t *row_bytes);
/* empties the finished-row buffer without a host copy (the caller took the rows on the device) */
int dbaz_replay_rows_clear(dbaz_engine *e);

/* ---- training data path: replay rows in HBM -> dataset -> batches in HBM --------
 * Replaces utils.HDFStoreDataset's array build (utils/utils.py:61-80), torch's DataLoader gather
 * and SymmetriesGenerator (dots_boxes/dots_boxes_nn.py:11-58) of the reference's train loop
 * (nn.py:186-216).  Rows are the packed replay rows of dbaz_replay_rows_dev (this engine's, or
 * the RCCL all-gathered ones of all ranks); they never leave HBM. */
#define DBAZ_MAX_DATASETS 4
/* a handle keeps up to DBAZ_MAX_DATASETS datasets resident (train + validation of the reference's
 * loop); the dataset calls below address the selected one (default 0) */
int dbaz_dataset_select(dbaz_engine *e, int32_t which);
int dbaz_dataset_begin(dbaz_engine *e);
/* append rows sel[0..n_sel) of the packed device rows (sel = HOST int32 indices in the order
 * the reference's DataFrame would have: where-clause, training flag, df.sample; NULL = all) */
int dbaz_dataset_add_rows(dbaz_engine *e, const void *rows_dev, int64_t n_rows, int32_t row_bytes,
                          const int32_t *sel, int64_t n_sel);
/* order: dataset order as a HOST permutation of the staged rows (NULL = staging order), for
 * selections that interleave several row buffers.  pos_average != 0: merge rows with identical
 * features (groupby(x).mean(): Kahan float64 means of pi and z in dataset order; groups in
 * ascending lexicographic order of the feature columns) */
int dbaz_dataset_finish(dbaz_engine *e, int32_t pos_average, const int32_t *order, int64_t *n_out);
/* host copies of the dataset arrays (features int16 [n,3HW], policy float32 [n,A], value [n]) */
int dbaz_dataset_fetch(dbaz_engine *e, int16_t *x, float *pi, float *z);
/* one batch: rows idx[0..n) (HOST indices) under symmetry sym (0..7 = SymmetriesGenerator.IDXS)
 * into caller-owned DEVICE buffers boards float32 [n,3,H,W], pi [n,A], z [n,1]; complete on return */
int dbaz_dataset_batch(dbaz_engine *e, const int32_t *idx, int32_t n, int32_t sym, float *boards_dev,
                       float *pi_dev, float *z_dev);
/* the same QUEUED on the caller's own stream (a hipStream_t; e.g. torch's current stream) and returning at once: the indices are
 * validated on the host, nothing is synchronised.  Use this form when the output buffers come from a stream-ordered caching
 * allocator: dbaz_dataset_batch writes them from the handle's stream, i.e. possibly while work the caller has queued on the
 * buffers' previous owner is still pending. */
int dbaz_dataset_batch_on(dbaz_engine *e, const int32_t *idx, int32_t n, int32_t sym, float *boards_dev, float *pi_dev,
                          float *z_dev, void *stream);
/* SymmetriesGenerator.forward on DEVICE tensors boards [n,3,H,W] / policies [n,A] (either may be
 * NULL); out-of-place */
int dbaz_symmetry_apply(dbaz_engine *e, int32_t sym, const float *boards_in_dev, const float *pol_in_dev,
                        int64_t n, float *boards_out_dev, float *pol_out_dev);
/* host only (no handle, no GPU): src[a'] with out[a'] = in[src[a']] over the two edge planes */
int dbaz_symmetry_table(int32_t rows, int32_t cols, int32_t sym, int32_t *lut_out);

/* ---- training-mode tower: forward and backward of ResNetZero's residual blocks under model.train(True) ----
 * Replaces, inside NeuralNetWrapper.train's step (nn.py:203-221: p, v = self.model(boards); loss.backward()), the
 * ResBlock stack of ResNet.forward (nn.py:23-28,48-57) with BatchNorm2d in training mode (batch statistics, running-stat
 * update, gradient through the statistics) -- 98 % of the step's FLOPs.  bn_input, conv0, the heads, AlphaZeroLoss and
 * torch.optim.SGD stay with the caller.  All tensor arguments are DEVICE pointers; calls are asynchronous on `stream`
 * (a hipStream_t, NULL = the default stream).  One handle holds the activations of ONE forward pass for its backward. */
typedef struct dbaz_trainer dbaz_trainer;
const char *dbaz_trainer_last_error(const dbaz_trainer *t); /* t == NULL: why dbaz_trainer_create failed */
int dbaz_trainer_create(int32_t rows, int32_t cols, int32_t channels /* 64 */, int32_t blocks, int32_t max_batch,
                        int32_t device, dbaz_trainer **out);
void dbaz_trainer_destroy(dbaz_trainer *t);
/* x / out: float32 [n][64][H][W] (torch NCHW).  conv_w[l] [64][64][3][3], conv_b[l], bn_w[l], bn_b[l], run_mean[l],
 * run_var[l] [64]: HOST arrays of DEVICE pointers, layer l = 2*block + (0: conv1/bn1, 1: conv2/bn2); run_mean / run_var
 * are updated in place (momentum 0.1, unbiased variance), NULL skips that. */
int dbaz_trainer_forward(dbaz_trainer *t, int32_t n, const float *x, const float *const *conv_w, const float *const *conv_b,
                         const float *const *bn_w, const float *const *bn_b, float *const *run_mean, float *const *run_var,
                         float *out, void *stream);
/* backward of the forward pass the handle holds: grad_out / grad_x [n][64][H][W]; parameter gradients are WRITTEN to
 * g_conv_w[l], g_conv_b[l], g_bn_w[l], g_bn_b[l] (shapes of the parameters); bn_w as in the forward call */
int dbaz_trainer_backward(dbaz_trainer *t, const float *grad_out, const float *const *bn_w, float *grad_x,
                          float *const *g_conv_w, float *const *g_conv_b, float *const *g_bn_w, float *const *g_bn_b,
                          void *stream);

/* ---- the whole network of the optimizer step: `p, v = self.model(boards)` and `loss.backward()` of
 * NeuralNetWrapper.train (nn.py:203-221) for a ResNetZero under model.train(True) -- bn_input, conv0 + bn0 (nn.py:19-21,48-50), the
 * residual tower, both heads (nn.py:74-105: conv 1x1 + bn + ReLU + fc (+ ReLU + fc1) -> log_softmax / tanh) -- on one
 * dbaz_trainer handle.  A dbaz_net_tensors holds DEVICE pointers to the parameters (or, in the backward call, to where their
 * gradients are WRITTEN), shapes as torch's; blk_* are HOST arrays [2*blocks] as in dbaz_trainer_forward.  Head channels: 16. */
typedef struct dbaz_net_tensors {
    float *bn_input_w, *bn_input_b;                                    /* [3] */
    float *conv0_w, *conv0_b;                                          /* [64][3][3][3], [64] */
    float *bn0_w, *bn0_b;                                              /* [64] */
    float *const *blk_conv_w, *const *blk_conv_b, *const *blk_bn_w, *const *blk_bn_b;
    float *ph_conv_w, *ph_conv_b, *ph_bn_w, *ph_bn_b;                  /* [16][64][1][1], [16], [16], [16] */
    float *ph_fc_w, *ph_fc_b;                                          /* [A][16*H*W], [A] */
    float *vh_conv_w, *vh_conv_b, *vh_bn_w, *vh_bn_b;
    float *vh_fc0_w, *vh_fc0_b;                                        /* [value_fc][16*H*W], [value_fc] */
    float *vh_fc1_w, *vh_fc1_b;                                        /* [1][value_fc], [1] */
} dbaz_net_tensors;
typedef struct dbaz_net_running { /* BatchNorm running statistics, updated in place (any pointer may be NULL) */
    float *bn_input_mean, *bn_input_var, *bn0_mean, *bn0_var;
    float *const *blk_mean, *const *blk_var;                           /* HOST arrays [2*blocks] or NULL */
    float *ph_mean, *ph_var, *vh_mean, *vh_var;
} dbaz_net_running;
/* x float32 [n][3][H][W] -> logp [n][n_actions] (log_softmax), v [n] (tanh).  `running` may be NULL. */
int dbaz_trainer_net_forward(dbaz_trainer *t, int32_t n, const float *x, const dbaz_net_tensors *params, const dbaz_net_running *running,
                             int32_t head_channels, int32_t n_actions, int32_t value_fc, float *logp, float *v, void *stream);
/* backward of the net_forward pass the handle holds; x and params as in that call; d_logp [n][n_actions], d_v [n] */
int dbaz_trainer_net_backward(dbaz_trainer *t, const float *x, const float *d_logp, const float *d_v, const dbaz_net_tensors *params,
                              const dbaz_net_tensors *grads, void *stream);

/* ---- training-mode BatchNorm2d (+ ReLU) on NCHW tensors of any channel count: bn_input, bn0 and the heads' bn0
 * of ResNetZero under model.train(True) (nn.py:19-21,81-83,98-100,114; torch.nn.BatchNorm2d: batch statistics, biased variance,
 * running-stat update with momentum, gradient through the statistics).  Stateless: all pointers DEVICE memory, `workspace` of
 * dbaz_bn2d_workspace_bytes(channels) bytes owned by the caller (the backward call may reuse the forward call's), asynchronous
 * on `stream`; errors: dbaz_trainer_last_error(NULL). */
int64_t dbaz_bn2d_workspace_bytes(int32_t channels);
int dbaz_bn2d_forward(const float *x /*[n][C][H*W]*/, int32_t n, int32_t channels, int32_t hw, const float *gamma, const float *beta,
                      float *run_mean, float *run_var, float eps, float momentum, int32_t relu, float *out, float *save_mean /*[C]*/,
                      float *save_invstd /*[C]*/, void *workspace, void *stream);
int dbaz_bn2d_backward(const float *dout, const float *out, const float *x, int32_t n, int32_t channels, int32_t hw, const float *gamma,
                       const float *save_mean, const float *save_invstd, int32_t relu, float *dx, float *dgamma, float *dbeta,
                       void *workspace, void *stream);

/* ---- AlphaZeroLoss and the SGD update of the optimizer step (nn.py:131-138,179,203-221) --------------------
 * Stateless, all pointers DEVICE memory, asynchronous on `stream`; errors: dbaz_trainer_last_error(NULL).
 * dbaz_az_loss: loss_v = mean((z - v)^2), loss_pi = -mean_n(sum_a pi * logp) (nn.py:133-135); loss3 = {loss_v + loss_pi,
 * loss_pi, loss_v}; d_logp [n][A] / d_v [n] (either may be NULL) receive grad_scale * d(loss_v + loss_pi)/d(.).
 * workspace: dbaz_az_loss_workspace_bytes() bytes. */
int64_t dbaz_az_loss_workspace_bytes(void);
int dbaz_az_loss(const float *logp /*[n][A]*/, const float *v /*[n]*/, const float *pi /*[n][A]*/, const float *z /*[n]*/, int32_t n,
                 int32_t n_actions, float grad_scale, float *loss3, float *d_logp, float *d_v, void *workspace, void *stream);
/* torch.optim.SGD.step (dampening 0, no nesterov) for ALL parameter tensors at once: d = g + weight_decay * p;
 * buf = momentum * buf + d; p -= lr * buf (momentum = 0: p -= lr * d, bufs may be NULL).  params / grads / bufs: HOST arrays of
 * n_tensors DEVICE pointers (float32, contiguous), numels their element counts; the pointers reach the device as kernel arguments
 * (no host -> device copy on the stream).  table_dev: n_tensors * 32 bytes of device scratch; chunk c (2048 elements) belongs to
 * tensor chunk_tensor_dev[c] at chunk index chunk_off_dev[c] within it (device arrays the caller builds once per parameter set).
 * A zero-initialised buf equals torch's first step. */
int dbaz_sgd_step(int32_t n_tensors, const void *const *params, const void *const *grads, const void *const *bufs, const int64_t *numels,
                  void *table_dev, const int32_t *chunk_tensor_dev, const int32_t *chunk_off_dev, int32_t n_chunks, float lr,
                  float momentum, float weight_decay, void *stream);

#ifdef __cplusplus
}
#endif
#endif
