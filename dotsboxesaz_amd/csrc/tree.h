// tree.h -- host launchers of the tree / rules kernels (tree.hip)
#pragma once
#include "common.h"

void tree_launch_search_begin(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots,
                              const int32_t *num_reads_dev);
void tree_launch_select(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots);
void tree_launch_select_multi(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots);
void tree_launch_expand_backup_multi(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots);
void tree_launch_expand_backup(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots);
void tree_launch_set_positions(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots,
                               const int16_t *moves_dev, const int32_t *offsets_dev);
void tree_launch_advance_manual(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots,
                                const int32_t *moves_dev, int reuse);
void tree_launch_selfplay_start(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots);
void tree_launch_advance_auto(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots);
void tree_launch_get_roots(hipStream_t s, const Geo &g, const TreeBufs &B, int n_slots, double *priors, float *tv,
                           int32_t *nv, int32_t *changed, int32_t *stats, float *q, float *root_tv, int32_t *root_nv,
                           uint64_t *edges, int16_t *b2c2, int8_t *to_play, int8_t *just_played, int8_t *result,
                           int8_t *expanded);
void tree_launch_get_leaves(hipStream_t s, const Geo &g, const TreeBufs &B, int n_slots, int16_t *leaf_x,
                            uint8_t *need_eval, int32_t *n_active);
void tree_launch_slot_summary(hipStream_t s, const TreeBufs &B, int n_slots, SlotSummary *out_dev);
void tree_launch_stop_search(hipStream_t s, const TreeBufs &B, int n_slots);
void tree_launch_count_active(hipStream_t s, const TreeBufs &B, int n_slots, int32_t *out3);
void tree_launch_rules(hipStream_t s, const Geo &g, int op, int n, uint64_t *edges, int16_t *b2c2, int8_t *to_play,
                       int8_t *just_played, const int32_t *moves, int8_t *n_closed, int8_t *closed_lc, uint8_t *valid,
                       int8_t *result, int16_t *x);
