// nn.h -- policy/value network on HIP (nn.hip): ResNetZero (nn.py:108-122) and
// SimpleNN (dots_boxes_nn.py:61-98) forward passes, eval-mode BN folded on the host.
#pragma once
#include <string>

#include "common.h"

struct NNState;

NNState *nn_create(const Geo &g, int max_batch, int precision, bool no_fallback = false);
void nn_destroy(NNState *nn);
int nn_configure(NNState *nn, int kind, int channels, int blocks, int head_channels, int value_fc, std::string &err);
int nn_set_tensor(NNState *nn, const char *key, const float *data, int64_t numel, std::string &err);
int nn_commit(NNState *nn, hipStream_t s, std::string &err);
bool nn_ready(const NNState *nn);
// Evaluate samples feat[list[j]] (float32 planes [3][H][W], j < *n_dev <= max_n) and scatter
// softmax policy to P[list[j]*AS + a] and tanh value to V[list[j]].  list == nullptr: identity.
// ev_begin/ev_end (optional) are recorded around the conv tower on `s`.
// cut_round / cut_defer (self-play stepping; 0 = off): only cut_n(*n_dev) leaves of the list are evaluated -- full rounds of
// workgroups -- and that count is left in *n_used for k_expand_backup ("full rounds only")
void nn_forward(NNState *nn, hipStream_t s, const float *feat, const int32_t *list_dev, const int32_t *n_dev,
                int max_n, float *P, float *V, int AS, hipEvent_t ev_begin, hipEvent_t ev_end, int cut_round = 0, int cut_defer = 0,
                int32_t *n_used = nullptr);
double nn_flops_per_sample(const NNState *nn);
// samples of one full round of the main tower launch (CUs x samples per workgroup), and the largest left-over that nn_forward
// would hand to a remainder launch (0: no remainder launches for this geometry)
void nn_round_info(const NNState *nn, int *round, int *rem_max);
const char *nn_tower_kernel_name(const NNState *nn);
// f16x3 mode: non-zero once an activation exceeded f16's range (results invalid: use precision 0)
int nn_overflowed(NNState *nn);
// f16x3 mode of ResNetZero: samples whose f16x3 evaluation left f16's range and were redone in exact f32
long long nn_fallback_evals(NNState *nn);
