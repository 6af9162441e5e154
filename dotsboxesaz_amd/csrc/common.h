// common.h -- shared definitions of the HIP engine (host + device).
// Layouts are documented in DESIGN.md ("Data layout in HBM").
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dbaz.h"

#define WAVE 64

// ---- board geometry, passed by value to kernels -------------------------------------
struct Geo {
    int rows, cols, H, W, HW, A, AS /* A rounded up to 4 */, B, E /* real edges */;
    int node_dw;       // dwords per tree node: META_DW + 4*AS
    int cap;           // nodes per slot
    int dmax;          // max search-path length (E + 2)
    uint64_t sentinel[4]; // board[1,H-1,:] and board[0,:,W-1] (dots_boxes_game.py:34-35)
    uint64_t amask[4];    // bits < A
};

// ---- tree node (one contiguous block of node_dw dwords) -----------------------------
//  dw 0..7   edges[4] u64   played-edge bitmask of the node's state
//  dw 8      parent (i32, -1 for the root)
//  dw 9      move (i16) | to_play (u8) << 16 | just_played+1 (u8) << 24
//  dw 10     b2c2[0] (i16) | b2c2[1] (i16) << 16
//  dw 11     flags (u8: 1 expanded, 2 terminal) | result+1 (u8) << 8 | deepness (u16) << 16
//  dw 12     network value v of the node's position (written at expansion; a transposition twin reads it)
//  dw 13..15 reserved
//  then rows P[AS] f32, W[AS] f32, NS[AS] (visits | same_player << 30), C[AS] i32 child index
#define META_DW 16
#define NF_EXPANDED 1u
#define NF_TERMINAL 2u
#define NF_INFLIGHT 4u   // K > 1 search: selected as a leaf in the current wave, evaluation pending
#define NS_MASK 0x3FFFFFFFu
#define NS_SAME 0x40000000u

// ---- per-slot search state ----------------------------------------------------------
enum Phase : int32_t {
    PH_IDLE = 0,        // nothing to do (manual mode between calls / finished)
    PH_EXPAND_ROOT = 1, // root unexpanded: one uncounted _search (mcts.py:207-208)
    PH_SIMS = 2,        // sims_left > 0
    PH_READY = 3,       // reads done, waiting for the move (k_advance)
    PH_EMIT = 4,        // game finished, rows waiting for space in the output buffer
    PH_ERROR = 5,
};

struct Slot {
    int32_t phase;
    int32_t ready_at;        // engine step at which k_expand_backup set PH_READY (same 8-byte word as phase, written together)
    int32_t n_nodes;
    int32_t sims_left;
    int32_t root_N;          // TreeRoot.child_number_visits[move]
    float root_W;            // TreeRoot.child_total_value[move]
    int32_t root_prepped;    // root priors already renormalised/mixed by a UCT_search call
    int32_t root_prior_f64;  // dtype of root child_priors in the reference (0 = float32 values)
    int32_t deepness_correction, max_deepness, terminal_count, tree_size;
    int32_t leaf, path_len, leaf_terminal, leaf_result, leaf_to_play;
    int32_t move_idx;        // ply within the current game
    int32_t n_rows;          // sample rows recorded for the current game
    int64_t game_idx;        // -1: no game
    double temperature;
    int32_t error;
    int32_t pool_high;
    int64_t n_search, sum_path, n_eval, n_term;
    int32_t ff_plies;        // pending fast-forward plies (benchmark population)
    int32_t model;           // match play: which of the two evaluators serves this slot's current search
    int32_t leaf_hit;        // transposition cache: expanded node of this tree with the leaf's position, or -1
    int32_t tt_epoch;        // 1..255, bumped at every re-root / new game (stale table entries are preferred victims)
    int64_t n_hit;
    int32_t sel_step;        // engine step of the k_select that left the current leaf (k_expand_backup takes only those)
    int32_t ff_reads;        // benchmark population: read budget of the slot's FIRST search (0 = the driver rule)
    // node pool bookkeeping (re-rooting moves nothing: the dropped part of a tree is collected incrementally)
    int32_t root;            // node index of the current root
    int32_t n_free;          // entries on the free stack (recycled node indices)
    int32_t pend_head, pend_tail; // ring of dropped nodes whose children have not been enumerated yet
    int32_t quick_until;     // benchmark population: plies of the slot's FIRST game that are searched with SearchCfg.quick_reads
    int32_t wave_sims;       // K > 1 search: simulations selected in the current wave, waiting for expand/backup
    int32_t first_wave;      // K > 1 search: the first wave of a UCT_search call is min(K, A) wide (mcts.py:228-229)
    int32_t pool_resets;     // self-play driver: moves whose kept subtree would not have left room for the next search (fresh root instead)
    int32_t eval_pos;        // position of this step's leaf in the evaluation list (k_select)
    int32_t pending;         // 1 = the leaf was selected in an earlier step and its evaluation put off (SearchCfg.eval_round)
};

// device-side reduction of the Slot array (dbaz_get_counters / dbaz_run poll this instead of copying every Slot)
struct SlotSummary {
    unsigned long long n_search, n_eval, n_hit, n_term, sum_path, n_reset;
    int32_t active, error, blocked, pool_high;
    int32_t first_error_slot, first_error_code; // lowest slot index in PH_ERROR (or 0x7fffffff) and its code
};

// one in-flight simulation of a tree searched with several pending evaluations (k_select_multi / k_expand_backup_multi)
struct SimRec {
    int32_t leaf, path_len, terminal, result, to_play;
    int32_t dup;   // its leaf was already selected by an earlier simulation of the same wave (one evaluation serves both)
};

struct PathEnt {
    int32_t node;
    int16_t move_in; // move from path[d-1] to this node
    int16_t to_play;
};

// per-row metadata kept while a game is in flight / in the output buffer
struct RowMeta {
    int32_t game_idx;
    int16_t move_idx, move, played, max_deepness;
    int32_t tree_size, terminal_count;
    float q_value;
    int8_t player, z;
    int16_t pad;
};

// ---- search parameters ---------------------------------------------------------------
struct SearchCfg {
    double cpuct, cpuct_base, alpha, coeff;
    int mcts_num_read, reuse_tree, n_temp;
    int temp_idx[8];
    double temp_val[8];
    int evaluator;
    int match_play;          // two-model match play (self_play.compute_elo): evaluator by root to_play
    int evaluator2;          // evaluator of model 1
    uint64_t seed;
    int table_n;             // entries in pbc/sqrt tables
    int step;                // engine step counter of this launch
    int driver_concurrent;   // this k_select runs next to the driver pass of the same step (self-play stepping)
    int quick_reads;         // read budget of the quick plies (dbaz_selfplay_quickplay)
    int pending;             // K > 1 search: width of a wave (<= TreeBufs.kmax), dbaz_set_pending
    int virtual_visits;      // K > 1 search: 1 = a simulation's visit is counted on its path at selection, 0 = at backup (the reference)
    // full rounds only (self-play stepping with a network evaluator): the evaluation list is cut back to a multiple of eval_round
    // leaves whenever at most eval_defer_max would be left over; the slots behind the cut keep their leaf and ask again next step
    int eval_round, eval_defer_max; // 0: every leaf is evaluated in the step that selected it
    int gc_lazy;             // node collector: > 0 = recycle dropped nodes only while fewer than this many indices are available
                             // (dropped nodes then live on as transposition twins until their memory is needed); 0 = two per simulation
};

// device buffer bundle handed to the tree kernels
struct TreeBufs {
    uint32_t *nodes;   // [n_slots][cap][node_dw]
    Slot *slots;       // [n_slots]
    PathEnt *path;     // [n_slots][dmax]
    double *root_prior;// [n_slots][AS]
    double *noise_in;  // [n_slots][AS] externally supplied Dirichlet vectors
    int32_t *noise_valid; // [n_slots] 1 = noise_in holds a vector for the next prep
    float *feat;       // [n_slots][3*HW] leaf features (float32 planes, predict_sync layout)
    float *evalP;      // [n_slots][AS]
    float *evalV;      // [n_slots]
    int32_t *eval_list;// [n_slots] compacted slots needing an NN evaluation (model 0)
    int32_t *eval_list2;// [n_slots] same for model 1 (match play)
    int32_t *n_eval;   // [4] list lengths of the two models; [2] leaves model 0's network took this step (k_head_fc, full rounds only)
    int32_t *pend;     // [n_slots][cap] ring: dropped nodes waiting for the collector (their child rows are still needed)
    int32_t *freel;    // [n_slots][cap] stack: node indices ready for reuse
    // search with K > 1 pending evaluations per tree (buffers exist when dbaz_config.max_pending_evals > 1)
    int32_t kmax;         // simulations in flight per tree
    SimRec *simrec;       // [n_slots][kmax]
    PathEnt *path_m;      // [n_slots][kmax][dmax]
    float *feat_m;        // [n_slots * kmax][3*HW]
    float *evalP_m;       // [n_slots * kmax][AS]
    float *evalV_m;       // [n_slots * kmax]
    int32_t *list_m;      // [n_slots * kmax] entries slot * kmax + k needing an evaluation
    int32_t *drv_list;  // [n_slots] slots that need the driver this step (k_driver_scan)
    int32_t *drv_count; // [1]
    unsigned long long *tt; // [n_slots][tt_mask+1] transposition table (tag24 | epoch8 | node index), nullptr = off
    int32_t tt_mask;
    const double *pbc_table;  // log((N+base+1)/base)+cpuct for N < table_n (host libm)
    const double *sqrt_table; // sqrt(N)
    // self-play rows of the game in flight
    int16_t *row_x;    // [n_slots][E+1][3HW]
    int32_t *row_vis;  // [n_slots][E+1][A]
    RowMeta *row_meta; // [n_slots][E+1]
    // finished rows
    int16_t *out_x;    // [max_out][3HW]
    int32_t *out_vis;  // [max_out][A]
    RowMeta *out_meta; // [max_out]
    int32_t *out_count;// [1]
    int32_t max_out;
    // game dispenser
    long long *next_game; // [1]
    long long last_game;  // exclusive
    long long first_game;
    long long *games_finished; // [1]
    long long *moves_played;   // [1]
    // teacher-forcing scripts (tests)
    const int16_t *script_moves; // [n_script][E+1], -1 = sample
    const double *script_noise;  // [n_script][E+1][A] or null
    const uint8_t *script_has_noise; // [n_script]
    int n_script;
};

#define HIP_CHECK_RET(e, call)                                                                 \
    do {                                                                                        \
        hipError_t _err = (call);                                                               \
        if (_err != hipSuccess) {                                                               \
            set_error(e, DBAZ_EDEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_err),\
                      __FILE__, __LINE__);                                                      \
            return DBAZ_EDEVICE;                                                                \
        }                                                                                       \
    } while (0)
