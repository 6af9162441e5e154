// train.h -- what csrc/train.hip (the residual tower, BatchNorm2d, loss, SGD) and csrc/train_net.hip (stem, heads, the whole-network
// entry points) share: the trainer handle, constants, small device helpers and the host functions one unit calls in the other.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/dbaz.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
union u128h { f32x4 f; f16x8 h; };

#define TT 512          // threads per workgroup of the conv / wgrad kernels: 8 waves, two per SIMD
#define TC 64           // channels (the two-cout-tile MFMA tiling is written for 64)
#define TL_MAX 64       // conv layers of a tower (2 * blocks)
#define RED_BLOCKS 256  // workgroups of the column-sum kernels

struct dbaz_net_buffers;
void net_free(struct dbaz_trainer *t); // train_net.hip

struct dbaz_trainer {
    int dev = 0, H = 0, W = 0, HW = 0, L = 0, maxN = 0, n = 0;
    int S = 1, Sw = 1, cus = 256;
#ifdef DBAZ_STAMP
    unsigned long long *stamps = nullptr, *stamps_wg = nullptr; // diagnostic build only (k_conv_t, k_wgrad_h3)
#endif
    float eps = 1e-5f, momentum = 0.1f;
    size_t conv_lds = 0, wgrad_lds = 0;
    bool have_fwd = false;
    bool net_fwd = false;  // the held forward pass is a dbaz_trainer_net_forward (whole network)
    int wgrad_h3 = 1, Swh = 1; // k_wgrad_h3 (f16x3) and its samples per chunk; 0: the exact-f32 k_wgrad
    size_t wgrad_h3_lds = 0;
    std::string err;
    float *A = nullptr, *Y = nullptr, *G = nullptr, *dA[2] = {nullptr, nullptr}, *dY = nullptr;
    _Float16 *wpk = nullptr;     // [2][L][C*C*9*2] halves: forward and transposed (dgrad) fragments
    float *wsc = nullptr;        // [2][L] 2^-sw of the packed weights
    unsigned *amax = nullptr;    // [L+1] bits of max|A[l]|, [L+1] = max|dY| of the layer in flight
    float *mean = nullptr, *invstd = nullptr; // [L][C]
    double *part = nullptr;      // partial column sums: [RED_BLOCKS][<= 4][C] rows, or one [2][C] row per workgroup of a conv launch
    double *part_bs = nullptr;   // [conv workgroups or RED_BLOCKS][2][C]: BatchNorm-backward partial rows (see tower_backward_rows)
    double *sums = nullptr;      // [4][C]
    float *wg_part = nullptr;    // [cus][9][C][C]
    unsigned long long *relu_mask = nullptr; // [L][maxN*HW]: sign bits of A[l+1] (64 channels per row)
    dbaz_net_buffers *net = nullptr; // stem and heads (dbaz_trainer_net_forward), allocated on first use
};

int terr(dbaz_trainer *t, int code, const char *fmt, ...); // train.hip: records the message, returns code

#define HIPCHK(t, call)                                                                                       \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) return terr(t, DBAZ_EDEVICE, "%s: %s", #call, hipGetErrorString(e_));          \
    } while (0)

// power of two that brings a tensor whose largest magnitude has the float bits `bits` into [2^13, 2^14)
__device__ __forceinline__ float scale_from_max(unsigned bits)
{
    if (bits == 0u) return 1.0f;
    int k = 13 - ((int)((bits >> 23) & 0xffu) - 127);
    k = max(-100, min(100, k));
    return __uint_as_float((unsigned)(k + 127) << 23);
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}


#define FIN_BLOCKS TC   // workgroups of the *_fin kernels: one per channel
#define BN_NB 64        // k_bn2d_*: sample slices (blockIdx.y) per channel

static inline size_t act_elems(const dbaz_trainer *t) { return (size_t)t->maxN * t->HW * TC; }
static inline int red_blocks(long long M) { return (int)std::max(1LL, std::min((long long)RED_BLOCKS, (M + 31) / 32)); }

// ---- train.hip, called by train_net.hip
// The tower on rows: A[0] (t->A, NHWC rows, max|A[0]| in t->amax[0]) -> A[L].  t->amax[1..] must be zero.
void tower_forward_rows(dbaz_trainer *t, int n, const float *const *conv_w, const float *const *conv_b, const float *const *bn_w,
                        const float *const *bn_b, float *const *run_mean, float *const *run_var, hipStream_t s);
// what the bottom layer's input-gradient conv needs to leave the BatchNorm-backward sums of the layer BELOW the tower (the stem's
// bn0 in dbaz_trainer_net_backward): that layer's ReLU mask, conv output and batch statistics
struct BelowTower {
    const unsigned long long *mask = nullptr;
    const float *y = nullptr, *mean = nullptr, *invstd = nullptr;
    float *g_w = nullptr, *g_b = nullptr; // where that layer's dgamma / dbeta go
};
// Backward of the tower on rows: dA[L] in t->dA[0] -> dA[0] in t->dA[returned index]; parameter gradients written.
int tower_backward_rows(dbaz_trainer *t, const float *const *bn_w, float *const *g_conv_w, float *const *g_conv_b, float *const *g_bn_w,
                        float *const *g_bn_b, const BelowTower &below, hipStream_t s);
// launches of train.hip's kernels on row tensors [M][64]:
//   batch statistics from `nparts` partial rows [2][64] (k_bn_stats_fin), then out = relu(bn(y) (+ res)), max|out|, ReLU mask (k_bn_apply)
void train_bn_forward_rows(dbaz_trainer *t, hipStream_t s, const double *part, int nparts, long long M, float *mean, float *invstd, float *run_mean,
                           float *run_var, const float *y, const float *res, float *out, const float *gamma, const float *beta, unsigned *amax,
                           unsigned long long *mask);
//   dY = BatchNorm backward of (dA, mask, y) with the totals in `sums`; max|dY|; partial rows of sum(dY) into `part` (k_bn_bwd_apply)
void train_bn_backward_apply_rows(dbaz_trainer *t, hipStream_t s, const float *dA, const unsigned long long *mask, const float *y, long long M,
                                  const float *mean, const float *invstd, const float *gamma, const double *sums, float *dY, unsigned *dymax,
                                  double *part);
//   batch statistics of an NCHW tensor [n][C][HW] (k_bn2d_stats + _fin); ws: C * BN_NB * 2 doubles
void train_bn2d_statistics(dbaz_trainer *t, hipStream_t s, const float *x, int n, int C, int HW, double *ws, float *mean, float *invstd,
                           float *run_mean, float *run_var);
