// replay.h -- training DATA path on HIP (replay.hip): dataset build from packed replay rows that
// already sit in HBM (HDFStoreDataset's array build, utils/utils.py:61-80, incl. pos_average) and
// batch assembly with the SymmetriesGenerator transforms (dots_boxes/dots_boxes_nn.py:11-58).
#pragma once
#include <string>

#include "common.h"

struct ReplayDS;

ReplayDS *rds_create(const Geo &g);
void rds_destroy(ReplayDS *d);
int rds_begin(ReplayDS *d, std::string &err);
// appends rows sel[0..n_sel) (host indices into the packed rows; nullptr = all rows in order)
int rds_add_rows(ReplayDS *d, hipStream_t s, const void *rows_dev, int64_t n_rows, int row_bytes, const int32_t *sel_host,
                 int64_t n_sel, std::string &err);
// order_host: dataset order as a permutation of the staged rows (nullptr = staging order)
int rds_finish(ReplayDS *d, hipStream_t s, int pos_average, const int32_t *order_host, int64_t *n_out, std::string &err);
int64_t rds_size(const ReplayDS *d);
int rds_fetch(ReplayDS *d, hipStream_t s, int16_t *x, float *pi, float *z, std::string &err);
// on_caller_stream: queued on `s` = the caller's own stream, nothing synchronised (indices validated on the host)
int rds_batch(ReplayDS *d, hipStream_t s, const int32_t *idx_host, int n, int sym, float *boards_dev, float *pi_dev,
              float *z_dev, std::string &err, bool on_caller_stream = false);
int rds_symmetry_apply(ReplayDS *d, hipStream_t s, int sym, const float *boards_in, const float *pol_in, int64_t n,
                       float *boards_out, float *pol_out, std::string &err);
// src[a'] with out[a'] = in[src[a']] over the two edge planes; 0 on success
int rds_symmetry_lut(const Geo &g, int sym, int32_t *lut, std::string &err);
