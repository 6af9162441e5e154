// train_net.hip -- the rest of ResNetZero's training-mode pass around the residual tower of train.hip, and the two entry points
// that run the WHOLE network of the optimizer step: dbaz_trainer_net_forward / dbaz_trainer_net_backward.
#include "train.h"

// ====================================================================================
// The whole network of the training step on this library (nn.py:108-122 under model.train(True)):
//   x [n][3][H][W] -> bn_input -> conv0 3x3 (3 -> 64) -> bn0 -> ReLU -> the residual tower (train.hip) -> per head conv 1x1
//   (64 -> 16) -> bn -> ReLU -> flatten -> policy: fc -> log_softmax; value: fc0 -> ReLU -> fc1 -> tanh
// and its backward.  Everything stays in the tower's row layout ([n*HW][channels]): no NCHW <-> rows transposes, the two
// heads share one 32-channel row (policy channels 0..15, value 16..31) and one fully connected GEMM over the combined row
// (weights of the other head's channels are zero), exact f32 throughout (v_mfma_f32_16x16x4_f32 or FMA) -- 2 % of the step's
// FLOPs.  bn_input is folded into the stem conv; its gradient and conv0's weight gradient come from ONE set of correlations
// of dY0 with the normalized input (see k_stem_val).
// ====================================================================================
#define HC 16          // channels per head (configuration.py: inner_channels 16)
#define HC2 (2 * HC)   // the two heads side by side in one row
#define NET_WG 256

struct dbaz_net_buffers {
    int hc = 0, A = 0, VF = 0, NO = 0, NOp = 0, KF = 0, maxN = 0;
    float *Y0 = nullptr;                 // [maxN*HW][64] conv0 output
    unsigned long long *mask0 = nullptr; // ReLU mask of the stem
    float *st = nullptr;                 // small floats (offsets ST_*): in_mean[4] in_invstd[4] mean0[64] invstd0[64] mean_h[32] invstd_h[32] bh[32] Wh[64][32]
    double *ws = nullptr;                // bn2d workspace of bn_input + partial rows of the head kernels
    float *Yh = nullptr, *Hh = nullptr;  // [maxN*HW][32] head conv output, after bn + ReLU
    float *dHh = nullptr;                // [maxN*HW][32] gradient of Hh, then of Yh (in place)
    float *Wc = nullptr, *bc = nullptr;  // [NO][KF] combined FC weights in row order, [NOp] biases
    float *logits = nullptr, *dlg = nullptr; // [maxN][NOp]
    float *gemm_part = nullptr;          // split-K partials
    float *logp = nullptr, *v = nullptr; // copies of the outputs for the backward pass
    double *sums_h = nullptr;            // [2][32]
    float *stem_part = nullptr;          // [STEM_SPLITS][36][64]
    float *val = nullptr;                // [maxN*HW][36]: see k_stem_val
    float *Gsum = nullptr;               // [36][64]
    size_t gemm_part_elems = 0;
};

// ---- generic small GEMM on v_mfma_f32_16x16x4_f32: C[m][n] (+)= sum_k A(m,k) B(k,n), A(m,k) = A[m*sam + k*sak], B(k,n) =
// B[k*sbk + n*sbn]; 64 x 64 tile per workgroup (4 waves, 32 x 32 each), K in steps of 32 through LDS; blockIdx.z = split of K
// (partial results at C + z * c_split, summed by the caller's fin kernel).  The staging picks the thread order along the
// operand's unit stride.
struct GemmArgs {
    const float *A; long long sam, sak;
    const float *B; long long sbk, sbn;
    float *C; long long ldc, c_split;
    const float *bias; // [N] or nullptr
    int M, N, K, kchunk;
};

__global__ void __launch_bounds__(NET_WG) k_gemm_f32(GemmArgs g)
{
    constexpr int KT = 32;
    __shared__ float As[64][KT + 1];
    __shared__ float Bs[KT][64 + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, gq = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int k_begin = blockIdx.z * g.kchunk, k_end = min(g.K, k_begin + g.kchunk);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool a_k_unit = g.sak == 1, b_n_unit = g.sbn == 1;
    // operands travel global -> registers TWO K steps ahead of their MFMAs (fetched inside the step their latency was exposed
    // once per step -- 55 us per launch on the heads' GEMMs; one step ahead still left half of it)
    float av[2][8], bv[2][8];
    // per element: offset at k = k_begin (or -1: outside M / N) and its k index within the step; a step then costs one add per
    // element (two 64-bit multiplies per element and step were as long as the step's MFMAs)
    long long aoff[8], boff[8];
    int ak[8], bk[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int i = tid + e * NET_WG;
        int m, k;
        if (a_k_unit) { m = i >> 5; k = i & 31; } else { k = i >> 6; m = i & 63; }
        ak[e] = k;
        aoff[e] = m0 + m < g.M ? (long long)(m0 + m) * g.sam + (long long)(k_begin + k) * g.sak : -1;
        int kb, nb;
        if (b_n_unit) { kb = i >> 6; nb = i & 63; } else { nb = i >> 5; kb = i & 31; }
        bk[e] = kb;
        boff[e] = n0 + nb < g.N ? (long long)(k_begin + kb) * g.sbk + (long long)(n0 + nb) * g.sbn : -1;
    }
    const long long a_step = (long long)KT * g.sak, b_step = (long long)KT * g.sbk;
    auto fetch = [&](float (&a)[8], float (&b)[8], int k0) {
        const long long sa = (long long)((k0 - k_begin) / KT) * a_step, sb = (long long)((k0 - k_begin) / KT) * b_step;
        // (unconditional loads from a clamped offset, zeroed afterwards: behind a branch the compiler waits for ALL outstanding
        // loads at the join and the two-step prefetch is lost)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const bool va = aoff[e] >= 0 && k0 + ak[e] < k_end, vb = boff[e] >= 0 && k0 + bk[e] < k_end;
            const float x = g.A[va ? aoff[e] + sa : 0], y = g.B[vb ? boff[e] + sb : 0];
            a[e] = va ? x : 0.0f;
            b[e] = vb ? y : 0.0f;
        }
    };
    auto step = [&](float (&a_)[8], float (&b_)[8], int knext) {
        __syncthreads(); // the previous step's fragment reads are done
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int i = tid + e * NET_WG;
            if (a_k_unit) As[i >> 5][i & 31] = a_[e]; else As[i & 63][i >> 6] = a_[e];
            if (b_n_unit) Bs[i >> 6][i & 63] = b_[e]; else Bs[i & 31][i >> 5] = b_[e];
        }
        __syncthreads();
        if (knext < k_end) fetch(a_, b_, knext);
#pragma unroll
        for (int kk = 0; kk < KT / 4; kk++) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; i++) a[i] = As[wm + i * 16 + m16][kk * 4 + gq];
#pragma unroll
            for (int j = 0; j < 2; j++) b[j] = Bs[kk * 4 + gq][wn + j * 16 + m16];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    if (k_begin < k_end) fetch(av[0], bv[0], k_begin);
    if (k_begin + KT < k_end) fetch(av[1], bv[1], k_begin + KT);
    for (int k0 = k_begin; k0 < k_end; k0 += 2 * KT) {
        step(av[0], bv[0], k0 + 2 * KT);
        if (k0 + KT < k_end) step(av[1], bv[1], k0 + 3 * KT);
    }
    float *C = g.C + (long long)blockIdx.z * g.c_split;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int n = n0 + wn + j * 16 + m16;
            const float bz = (g.bias && n < g.N && blockIdx.z == 0) ? g.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + wm + i * 16 + 4 * gq + r;
                if (m < g.M && n < g.N) C[(long long)m * g.ldc + n] = acc[i][j][r] + bz;
            }
        }
}

static void launch_gemm(hipStream_t s, const float *A, long long sam, long long sak, const float *B, long long sbk, long long sbn, float *C,
                        long long ldc, int M, int N, int K, const float *bias, int splits, long long c_split)
{
    GemmArgs g;
    g.A = A; g.sam = sam; g.sak = sak; g.B = B; g.sbk = sbk; g.sbn = sbn; g.C = C; g.ldc = ldc; g.c_split = c_split; g.bias = bias;
    g.M = M; g.N = N; g.K = K;
    g.kchunk = ((K + splits - 1) / splits + 31) / 32 * 32;
    const int z = (K + g.kchunk - 1) / g.kchunk;
    hipLaunchKernelGGL(k_gemm_f32, dim3((N + 63) / 64, (M + 63) / 64, z), dim3(NET_WG), 0, s, g);
}
static int gemm_splits(int K, int splits) // the z extent launch_gemm uses
{
    const int kchunk = ((K + splits - 1) / splits + 31) / 32 * 32;
    return (K + kchunk - 1) / kchunk;
}

// column sums of a NET_WG-thread workgroup whose thread (rl, cq) = (tid / CQ, tid % CQ) holds K x 4 doubles of channel quad cq:
// one partial row [K][4 CQ] per workgroup
template <int K, int CQ>
__device__ __forceinline__ void net_colsum_store(double (&s)[K][4], double *part)
{
    constexpr int RL = NET_WG / CQ;
    __shared__ double red[RL][K][4 * CQ + 1];
    const int tid = threadIdx.x, cq = tid % CQ, rl = tid / CQ;
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
        for (int e = 0; e < 4; e++) red[rl][k][cq * 4 + e] = s[k][e];
    __syncthreads();
    if (tid < K * 4 * CQ) {
        const int k = tid / (4 * CQ), c = tid - k * 4 * CQ;
        double v = 0.0;
        for (int r = 0; r < RL; r++) v += red[r][k][c];
        part[((size_t)blockIdx.x * K + k) * 4 * CQ + c] = v;
    }
}
// total of column c of the partial rows part[nparts][ncols] (all threads of a NET_WG workgroup call; result in every thread)
__device__ __forceinline__ double net_col_total(const double *part, int nparts, int ncols, int c)
{
    __shared__ double red[NET_WG / 64];
    __syncthreads();
    double v = 0.0;
    for (int b = threadIdx.x; b < nparts; b += NET_WG) v += part[(size_t)b * ncols + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < NET_WG / 64; w++) t += red[w];
    return t;
}

// total of column c of part[nparts][ncols] by one wave (every lane gets it)
__device__ __forceinline__ double wave_col_total(const double *part, int nparts, int ncols, int c)
{
    double v = 0.0;
    for (int b = threadIdx.x & 63; b < nparts; b += 64) v += part[(size_t)b * ncols + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- stem forward: Y0[row][o] = b0[o] + sum_{ci,tap} W0[o][ci][tap] * xhat[n][ci][pos + tap], xhat = bn_input(x) inside the
// board and 0 outside (the conv pads the NORMALIZED input); partial sums of Y0, Y0^2 for bn0.  Thread = (row lane, channel quad)
// with its quad's 27 x 4 weights in registers; a workgroup normalizes STEM_S samples at a time into LDS (read from global by
// each of a row's 16 threads, the 27 inputs made the kernel load-issue-bound: 109 us).
#define STEM_S 4
__global__ void __launch_bounds__(NET_WG) k_stem_conv(const float *__restrict__ x, const float *in_mean, const float *in_invstd,
                                                      const float *in_w, const float *in_b, const float *__restrict__ w0,
                                                      const float *__restrict__ b0, f32x4 *__restrict__ y4, int n, int H, int W, double *part)
{
    extern __shared__ float sxs[]; // xh[STEM_S][3][HW] | nb[HW][9]
    const int HW = H * W;
    float *xh = sxs;
    int *nb = reinterpret_cast<int *>(sxs + STEM_S * 3 * HW);
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    for (int i = tid; i < HW * 9; i += NET_WG) {
        const int pos = i / 9, t = i - pos * 9, y0 = pos / W, x0 = pos - y0 * W;
        const int yy = y0 + t / 3 - 1, xx = x0 + t % 3 - 1;
        nb[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? yy * W + xx : -1;
    }
    f32x4 w[27];
#pragma unroll
    for (int k = 0; k < 27; k++)
#pragma unroll
        for (int e = 0; e < 4; e++) w[k][e] = w0[(cq * 4 + e) * 27 + k]; // w0[o][ci][tap], k = ci*9 + tap
    float sc[3], sh[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { sc[c] = in_invstd[c] * in_w[c]; sh[c] = in_b[c] - in_mean[c] * sc[c]; }
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(b0 + cq * 4);
    double s[2][4] = {};
    for (int s0 = blockIdx.x * STEM_S; s0 < n; s0 += gridDim.x * STEM_S) {
        const int ns = min(STEM_S, n - s0);
        __syncthreads();
        for (int i = tid; i < ns * 3 * HW; i += NET_WG) {
            const int c = (i / HW) % 3;
            xh[i] = x[(size_t)s0 * 3 * HW + i] * sc[c] + sh[c];
        }
        __syncthreads();
        for (int lr = rl; lr < ns * HW; lr += 16) {
            const int sidx = lr / HW, pos = lr - sidx * HW;
            const float *xs = xh + sidx * 3 * HW;
            f32x4 acc = bias;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int q = nb[pos * 9 + t];
                const int qq = max(q, 0);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float xv = q < 0 ? 0.0f : xs[c * HW + qq];
                    acc += xv * w[c * 9 + t];
                }
            }
            y4[((size_t)s0 * HW + lr) * 16 + cq] = acc;
#pragma unroll
            for (int e = 0; e < 4; e++) { const double d = acc[e]; s[0][e] += d; s[1][e] += d * d; }
        }
    }
    net_colsum_store<2, 16>(s, part);
}

// ---- heads: packed parameters.  Wh[c][o] = conv weight of output o (policy 0..15, value 16..31) and input c; bh[o];
// Wc[oo][pos*32 + c] = fc weight of output oo (policy actions, then the value head's fc0 units) for row element (pos, c) -- torch
// flattens NCHW, [c*HW + pos] -- and 0 for the other head's channels; bc[oo].
__global__ void __launch_bounds__(NET_WG) k_head_pack(const float *ph_cw, const float *ph_cb, const float *vh_cw, const float *vh_cb,
                                                      const float *ph_fw, const float *ph_fb, const float *vh_fw, const float *vh_fb,
                                                      float *Wh, float *bh, float *Wc, float *bc, int HW, int A, int VF, int NOp)
{
    const int KF = HW * HC2, NO = A + VF;
    const long long total = (long long)NO * KF;
    for (long long i = (long long)blockIdx.x * NET_WG + threadIdx.x; i < total; i += (long long)gridDim.x * NET_WG) {
        const int oo = (int)(i / KF), k = (int)(i - (long long)oo * KF), pos = k / HC2, c = k - pos * HC2;
        float v = 0.0f;
        if (oo < A) { if (c < HC) v = ph_fw[(size_t)oo * HC * HW + c * HW + pos]; }
        else if (c >= HC) v = vh_fw[(size_t)(oo - A) * HC * HW + (c - HC) * HW + pos];
        Wc[i] = v;
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < TC * HC2; i += NET_WG) {
            const int c = i / HC2, o = i - c * HC2;
            Wh[i] = o < HC ? ph_cw[o * TC + c] : vh_cw[(o - HC) * TC + c];
        }
        for (int i = threadIdx.x; i < HC2; i += NET_WG) bh[i] = i < HC ? ph_cb[i] : vh_cb[i - HC];
        for (int i = threadIdx.x; i < NOp; i += NET_WG) bc[i] = i < A ? ph_fb[i] : (i < NO ? vh_fb[i - A] : 0.0f);
    }
}

// ---- head conv 1x1 (both heads): Yh[row][o] = bh[o] + sum_c A[row][c] Wh[c][o]; partial sums of Yh, Yh^2.  Thread = (row lane,
// output quad), four rows per pass.
__global__ void __launch_bounds__(NET_WG) k_head_conv(const f32x4 *__restrict__ a4, const float *__restrict__ Wh, const float *__restrict__ bh,
                                                      f32x4 *__restrict__ yh4, long long M, double *part)
{
    __shared__ __attribute__((aligned(16))) float wt[TC][HC2];
    const int tid = threadIdx.x, oq = tid & 7, rl = tid >> 3;
    for (int i = tid; i < TC * HC2; i += NET_WG) wt[i / HC2][i % HC2] = Wh[i];
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(bh + oq * 4);
    __syncthreads();
    double s[2][4] = {};
    constexpr int RP = 4; // rows per thread and pass: each weight quad read from LDS serves all of them
    for (long long r0 = (long long)blockIdx.x * (32 * RP) + rl; r0 < M; r0 += (long long)gridDim.x * (32 * RP)) {
        f32x4 acc[RP];
#pragma unroll
        for (int u = 0; u < RP; u++) acc[u] = bias;
#pragma unroll 2
        for (int c4 = 0; c4 < TC / 4; c4++) {
            f32x4 x[RP];
#pragma unroll
            for (int u = 0; u < RP; u++) x[u] = r0 + 32 * u < M ? a4[(r0 + 32 * u) * 16 + c4] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(&wt[c4 * 4 + e][oq * 4]);
#pragma unroll
                for (int u = 0; u < RP; u++) acc[u] += x[u][e] * w;
            }
        }
#pragma unroll
        for (int u = 0; u < RP; u++)
            if (r0 + 32 * u < M) {
                yh4[(r0 + 32 * u) * 8 + oq] = acc[u];
#pragma unroll
                for (int e = 0; e < 4; e++) { const double d = acc[u][e]; s[0][e] += d; s[1][e] += d * d; }
            }
    }
    net_colsum_store<2, 8>(s, part);
}

// batch statistics of the 32 head channels (grid: 32 workgroups); running statistics of the policy head's bn (channels 0..15) and
// the value head's (16..31)
__global__ void __launch_bounds__(NET_WG) k_head_stats_fin(const double *part, int nparts, long long M, float eps, float momentum, float *mean,
                                                           float *invstd, float *ph_rm, float *ph_rv, float *vh_rm, float *vh_rv)
{
    const int c = blockIdx.x;
    const double t0 = net_col_total(part, nparts, 2 * HC2, c), t1 = net_col_total(part, nparts, 2 * HC2, HC2 + c);
    if (threadIdx.x == 0) {
        const double m = t0 / (double)M;
        double var = t1 / (double)M - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        float *rm = c < HC ? ph_rm : vh_rm, *rv = c < HC ? ph_rv : vh_rv;
        const int cc = c < HC ? c : c - HC;
        if (rm) rm[cc] = (float)((1.0 - momentum) * (double)rm[cc] + (double)momentum * m);
        if (rv) {
            const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
            rv[cc] = (float)((1.0 - momentum) * (double)rv[cc] + (double)momentum * unb);
        }
    }
}

// Hh = relu(bn(Yh)) on [M][32] rows
__global__ void __launch_bounds__(NET_WG) k_head_bn_apply(const f32x4 *__restrict__ yh4, f32x4 *__restrict__ hh4, long long n4, const float *mean,
                                                          const float *invstd, const float *ph_w, const float *ph_b, const float *vh_w,
                                                          const float *vh_b)
{
    const int oq = threadIdx.x & 7;
    f32x4 mu, sc, be;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int c = oq * 4 + e;
        mu[e] = mean[c];
        sc[e] = invstd[c] * (c < HC ? ph_w[c] : vh_w[c - HC]);
        be[e] = c < HC ? ph_b[c] : vh_b[c - HC];
    }
    for (long long i = (long long)blockIdx.x * NET_WG + threadIdx.x; i < n4; i += (long long)gridDim.x * NET_WG) {
        const f32x4 y = yh4[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = fmaxf((y[e] - mu[e]) * sc[e] + be[e], 0.0f);
        hh4[i] = o;
    }
}

// per sample (one wave): logits = sum of the fc GEMM's split-K partials (bias in the first); logp = log_softmax(logits[0..A));
// h = relu(logits[A..A+VF)); v = tanh(b1 + sum_j W1[j] h[j])
__global__ void __launch_bounds__(NET_WG) k_head_out(const float *__restrict__ lpart, int splits, long long split_stride, float *__restrict__ logits,
                                                     int n, int A, int VF, int NOp, const float *__restrict__ w1, const float *__restrict__ b1,
                                                     float *__restrict__ logp, float *__restrict__ v, float *__restrict__ logp_keep,
                                                     float *__restrict__ v_keep)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NO = A + VF;
    for (int s = blockIdx.x * 4 + wave; s < n; s += gridDim.x * 4) {
        float *lg = logits + (size_t)s * NOp;
        // (every pass re-adds the partials instead of reading back what another lane has just stored)
        auto val = [&](int a) {
            float t = lpart[(size_t)s * NOp + a];
            for (int z = 1; z < splits; z++) t += lpart[(long long)z * split_stride + (size_t)s * NOp + a];
            return t;
        };
        float mx = -INFINITY;
        for (int a = lane; a < NO; a += 64) {
            const float t = val(a);
            lg[a] = t;
            if (a < A) mx = fmaxf(mx, t);
        }
        mx = wave_max(mx);
        float sum = 0.0f;
        for (int a = lane; a < A; a += 64) sum += expf(val(a) - mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float lse = mx + logf(sum);
        for (int a = lane; a < A; a += 64) {
            const float lp = val(a) - lse;
            logp[(size_t)s * A + a] = lp;
            logp_keep[(size_t)s * A + a] = lp;
        }
        float hv = 0.0f;
        for (int j = lane; j < VF; j += 64) hv += w1[j] * fmaxf(val(A + j), 0.0f);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hv += __shfl_xor(hv, o);
        if (lane == 0) {
            const float t = tanhf(hv + b1[0]);
            v[s] = t;
            v_keep[s] = t;
        }
    }
}

// backward of k_head_out: d_logits[a] = d_logp[a] - exp(logp[a]) sum_a d_logp; dz = d_v (1 - v^2); d_logits[A+j] = dz W1[j] (h_j > 0);
// partial rows [NOp | VF | 1] of the column sums of d_logits (the fc biases' gradients), of dz h_j (fc1's weight) and of dz (its bias)
__global__ void __launch_bounds__(NET_WG) k_head_out_bwd(const float *__restrict__ d_logp, const float *__restrict__ d_v,
                                                         const float *__restrict__ logp, const float *__restrict__ v,
                                                         const float *__restrict__ logits, int n, int A, int VF, int NOp,
                                                         const float *__restrict__ w1, float *__restrict__ dlg, double *part)
{
    extern __shared__ double sred[]; // [4 waves][NOp + VF + 1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncol = NOp + VF + 1;
    for (int i = threadIdx.x; i < 4 * ncol; i += NET_WG) sred[i] = 0.0;
    __syncthreads();
    double *my = sred + wave * ncol; // (each column of a wave's row is owned by one lane: no conflicts)
    for (int s = blockIdx.x * 4 + wave; s < n; s += gridDim.x * 4) {
        const float *dp = d_logp + (size_t)s * A, *lp = logp + (size_t)s * A, *lg = logits + (size_t)s * NOp;
        float sd = 0.0f;
        for (int a = lane; a < A; a += 64) sd += dp[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sd += __shfl_xor(sd, o);
        for (int a = lane; a < A; a += 64) {
            const float d = dp[a] - expf(lp[a]) * sd;
            dlg[(size_t)s * NOp + a] = d;
            my[a] += (double)d;
        }
        const float vv = v[s];
        const float dz = d_v[s] * (1.0f - vv * vv);
        for (int j = lane; j < NOp - A; j += 64) {
            float d = 0.0f;
            if (j < VF) {
                const float h = fmaxf(lg[A + j], 0.0f);
                d = h > 0.0f ? dz * w1[j] : 0.0f;
                my[NOp + j] += (double)dz * (double)h;
            }
            dlg[(size_t)s * NOp + A + j] = d;
            my[A + j] += (double)d;
        }
        if (lane == 0) my[NOp + VF] += (double)dz;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ncol; c += NET_WG)
        part[(size_t)blockIdx.x * ncol + c] = (sred[c] + sred[ncol + c]) + (sred[2 * ncol + c] + sred[3 * ncol + c]);
}

// totals of k_head_out_bwd's partial rows (grid: one 64-thread workgroup per column) -> gradients of the policy fc bias [A], the
// value fc0 bias [VF], fc1's weight [VF] and bias [1]
__global__ void __launch_bounds__(64) k_head_small_fin(const double *part, int nparts, int A, int VF, int NOp, float *g_ph_fb, float *g_vh_f0b,
                                                       float *g_vh_f1w, float *g_vh_f1b)
{
    const int ncol = NOp + VF + 1, c = blockIdx.x;
    const double t = wave_col_total(part, nparts, ncol, c);
    if (threadIdx.x == 0) {
        if (c < A) g_ph_fb[c] = (float)t;
        else if (c < A + VF) g_vh_f0b[c - A] = (float)t;
        else if (c >= NOp && c < NOp + VF) g_vh_f1w[c - NOp] = (float)t;
        else if (c == NOp + VF) g_vh_f1b[0] = (float)t;
    }
}

// sums the split-K partials of dWc[NO][KF] and scatters them into torch's layouts: policy fc [A][hc*HW], value fc0 [VF][hc*HW]
__global__ void __launch_bounds__(NET_WG) k_fc_wgrad_fin(const float *__restrict__ part, int splits, long long split_stride, int HW, int A, int VF,
                                                         float *__restrict__ g_ph_fw, float *__restrict__ g_vh_fw)
{
    const int KF = HW * HC2;
    const long long total = (long long)(A + VF) * KF;
    for (long long i = (long long)blockIdx.x * NET_WG + threadIdx.x; i < total; i += (long long)gridDim.x * NET_WG) {
        const int oo = (int)(i / KF), k = (int)(i - (long long)oo * KF), pos = k / HC2, c = k - pos * HC2;
        const bool pol = oo < A;
        if (pol != (c < HC)) continue; // the other head's channels: structural zeros of Wc
        double t = 0.0;
        for (int z = 0; z < splits; z++) t += (double)part[(long long)z * split_stride + i];
        if (pol) g_ph_fw[(size_t)oo * HC * HW + c * HW + pos] = (float)t;
        else g_vh_fw[(size_t)(oo - A) * HC * HW + (c - HC) * HW + pos] = (float)t;
    }
}

// heads' BatchNorm backward, pass 1: g = dHh (Hh > 0); partial sums of g and g * yhat over the rows
__global__ void __launch_bounds__(NET_WG) k_head_bn_bwd_sums(const f32x4 *__restrict__ dh4, const f32x4 *__restrict__ hh4,
                                                             const f32x4 *__restrict__ yh4, long long M, const float *mean, const float *invstd,
                                                             double *part)
{
    const int oq = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + oq * 4), is = *reinterpret_cast<const f32x4 *>(invstd + oq * 4);
    double s[2][4] = {};
    for (long long r = (long long)blockIdx.x * 32 + rl; r < M; r += (long long)gridDim.x * 32) {
        const f32x4 d = dh4[r * 8 + oq], h = hh4[r * 8 + oq], y = yh4[r * 8 + oq];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float g = h[e] > 0.0f ? d[e] : 0.0f;
            const float yh = (y[e] - mu[e]) * is[e];
            s[0][e] += (double)g;
            s[1][e] += (double)g * (double)yh;
        }
    }
    net_colsum_store<2, 8>(s, part);
}
__global__ void __launch_bounds__(NET_WG) k_head_bn_bwd_fin(const double *part, int nparts, double *sums, float *g_ph_w, float *g_ph_b, float *g_vh_w,
                                                            float *g_vh_b)
{
    const int c = blockIdx.x;
    const double t0 = net_col_total(part, nparts, 2 * HC2, c), t1 = net_col_total(part, nparts, 2 * HC2, HC2 + c);
    if (threadIdx.x == 0) {
        sums[c] = t0;
        sums[HC2 + c] = t1;
        if (c < HC) { g_ph_b[c] = (float)t0; g_ph_w[c] = (float)t1; }
        else { g_vh_b[c - HC] = (float)t0; g_vh_w[c - HC] = (float)t1; }
    }
}
// pass 2 (in place): dYh = gamma invstd (g - sum(g)/M - yhat sum(g yhat)/M); partial sums of dYh (the conv biases' gradient)
__global__ void __launch_bounds__(NET_WG) k_head_bn_bwd_apply(f32x4 *__restrict__ dh4, const f32x4 *__restrict__ hh4, const f32x4 *__restrict__ yh4,
                                                              long long M, const float *mean, const float *invstd, const float *ph_w,
                                                              const float *vh_w, const double *sums, double *part)
{
    const int oq = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + oq * 4), is = *reinterpret_cast<const f32x4 *>(invstd + oq * 4);
    f32x4 ga, mg, mgy;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int c = oq * 4 + e;
        ga[e] = c < HC ? ph_w[c] : vh_w[c - HC];
        mg[e] = (float)(sums[c] / (double)M);
        mgy[e] = (float)(sums[HC2 + c] / (double)M);
    }
    double s[1][4] = {};
    for (long long r = (long long)blockIdx.x * 32 + rl; r < M; r += (long long)gridDim.x * 32) {
        const f32x4 d = dh4[r * 8 + oq], h = hh4[r * 8 + oq], y = yh4[r * 8 + oq];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float g = h[e] > 0.0f ? d[e] : 0.0f;
            const float yh = (y[e] - mu[e]) * is[e];
            o[e] = ga[e] * is[e] * (g - mg[e] - yh * mgy[e]);
            s[0][e] += (double)o[e];
        }
        dh4[r * 8 + oq] = o;
    }
    net_colsum_store<1, 8>(s, part);
}

// dA[row][c] = sum_o dYh[row][o] Wh[c][o] (the gradient entering the tower, in rows); two rows per pass
__global__ void __launch_bounds__(NET_WG) k_head_conv_bwd_data(const f32x4 *__restrict__ dyh4, const float *__restrict__ Wh, f32x4 *__restrict__ da4,
                                                               long long M)
{
    __shared__ __attribute__((aligned(16))) float wo[HC2][TC]; // wo[o][c]
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    for (int i = tid; i < TC * HC2; i += NET_WG) wo[i % HC2][i / HC2] = Wh[i];
    __syncthreads();
    for (long long r0 = (long long)blockIdx.x * 32 + rl; r0 < M; r0 += (long long)gridDim.x * 32) {
        const long long r1 = r0 + 16;
        const bool two = r1 < M;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int o4 = 0; o4 < HC2 / 4; o4++) {
            const f32x4 d0 = dyh4[r0 * 8 + o4];
            const f32x4 d1 = two ? dyh4[r1 * 8 + o4] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(&wo[o4 * 4 + e][cq * 4]);
                acc0 += d0[e] * w;
                acc1 += d1[e] * w;
            }
        }
        da4[r0 * 16 + cq] = acc0;
        if (two) da4[r1 * 16 + cq] = acc1;
    }
}

// sums the split-K partials of dWh[o][c] -> the two heads' conv weight gradients [hc][64] (workgroup = 16 outputs x 16 partial
// lanes); the last 8 workgroups total k_head_bn_bwd_apply's partial rows -> the conv biases' gradients (a wave per channel)
__global__ void __launch_bounds__(NET_WG) k_head_wgrad_fin(const float *__restrict__ part, int splits, const double *bias_part, int bias_nparts,
                                                           float *g_ph_cw, float *g_vh_cw, float *g_ph_cb, float *g_vh_cb)
{
    __shared__ double red[16][17];
    const int nb_w = HC2 * TC / 16;
    if ((int)blockIdx.x >= nb_w) {
        const int c = ((int)blockIdx.x - nb_w) * 4 + (threadIdx.x >> 6);
        const double t = wave_col_total(bias_part, bias_nparts, HC2, c);
        if ((threadIdx.x & 63) == 0) { if (c < HC) g_ph_cb[c] = (float)t; else g_vh_cb[c - HC] = (float)t; }
        return;
    }
    const int oi = threadIdx.x & 15, lanep = threadIdx.x >> 4, i = blockIdx.x * 16 + oi; // i = o * 64 + c
    double t = 0.0;
    for (int z = lanep; z < splits; z += 16) t += (double)part[(size_t)z * HC2 * TC + i];
    red[lanep][oi] = t;
    __syncthreads();
    if (lanep == 0) {
        double v = 0.0;
        for (int r = 0; r < 16; r++) v += red[r][oi];
        const int o = i / TC, c = i - o * TC;
        if (o < HC) g_ph_cw[o * TC + c] = (float)v; else g_vh_cw[(o - HC) * TC + c] = (float)v;
    }
}

// ---- stem backward.  With xn = (x - mean) invstd (0 outside the board) and I = 1 inside / 0 outside:
//   Gx[ci][tap][o] = sum_rows dY0[row][o] xn[ci](row + tap),   Cn[tap][o] = sum_rows dY0[row][o] I(row + tap)
// give conv0's weight gradient  dW0[o][ci][tap] = gamma_ci Gx + beta_ci Cn  (the conv saw gamma xn + beta) AND bn_input's:
// d gamma_ci = sum_{o,tap} W0[o][ci][tap] Gx[ci][tap][o], d beta_ci = sum_{o,tap} W0[o][ci][tap] Cn[tap][o] (the input-gradient
// conv followed by the sums over positions, with the order of the two sums exchanged) -- no input-gradient conv is run.
// The correlations are one GEMM [36 x rows] x [rows x 64] (k_gemm_f32, split over the rows) on the matrix val[row][k] below.
// val[row][ci*9 + tap] = xn[ci](row + tap), val[row][27 + tap] = I(row + tap)
__global__ void __launch_bounds__(NET_WG) k_stem_val(const float *__restrict__ x, const float *in_mean, const float *in_invstd, long long M, int H,
                                                     int W, float *__restrict__ val)
{
    const int HW = H * W;
    const long long total = M * 36;
    for (long long i = (long long)blockIdx.x * NET_WG + threadIdx.x; i < total; i += (long long)gridDim.x * NET_WG) {
        const long long row = i / 36;
        const int k = (int)(i - row * 36), g = k / 9, t = k - g * 9;
        const int n = (int)(row / HW), pos = (int)(row - (long long)n * HW), y0 = pos / W, x0 = pos - y0 * W;
        const int yy = y0 + t / 3 - 1, xx = x0 + t % 3 - 1;
        float v = 0.0f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            v = g < 3 ? (x[((size_t)n * 3 + g) * HW + yy * W + xx] - in_mean[g]) * in_invstd[g] : 1.0f;
        val[i] = v;
    }
}
// Gsum[k][o] = sum over the workgroups' partials (grid: 36 x 4 workgroups, thread = (partial lane of 16, o of 16))
__global__ void __launch_bounds__(NET_WG) k_stem_fin1(const float *__restrict__ part, int nparts, float *__restrict__ Gsum)
{
    __shared__ double red[16][17];
    const int k = blockIdx.x >> 2, o = (blockIdx.x & 3) * 16 + (threadIdx.x & 15), j = threadIdx.x >> 4;
    double t = 0.0;
    for (int b = j; b < nparts; b += 16) t += (double)part[((size_t)b * 36 + k) * TC + o];
    red[j][threadIdx.x & 15] = t;
    __syncthreads();
    if (j == 0) {
        double v = 0.0;
        for (int r = 0; r < 16; r++) v += red[r][threadIdx.x & 15];
        Gsum[k * TC + o] = (float)v;
    }
}
// conv0's weight and bias gradients, bn_input's weight and bias gradients
__global__ void __launch_bounds__(NET_WG) k_stem_fin2(const float *__restrict__ Gsum, const float *__restrict__ w0, const float *in_w, const float *in_b,
                                                      const double *bias_part, int bias_nparts, float *g_w0, float *g_b0, float *g_in_w,
                                                      float *g_in_b)
{
    // grid of 5: workgroups 0..2 = bn_input's gradient for input channel ci, 3 = conv0's weight gradient, 4 = its bias gradient
    // (one workgroup doing the five in turn was five global round trips behind each other: 23 us)
    __shared__ double rg[NET_WG], rb[NET_WG / 64];
    const int tid = threadIdx.x;
    if (blockIdx.x < 3) {
        const int ci = blockIdx.x;
        double a = 0.0, b = 0.0;
        for (int i = tid; i < TC * 9; i += NET_WG) {
            const int o = i / 9, t = i - o * 9;
            const double w = (double)w0[o * 27 + ci * 9 + t];
            a += w * (double)Gsum[(ci * 9 + t) * TC + o];
            b += w * (double)Gsum[(27 + t) * TC + o];
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) { a += __shfl_xor(a, sh); b += __shfl_xor(b, sh); }
        if ((tid & 63) == 0) { rg[tid >> 6] = a; rb[tid >> 6] = b; }
        __syncthreads();
        if (tid == 0) {
            g_in_w[ci] = (float)((rg[0] + rg[1]) + (rg[2] + rg[3]));
            g_in_b[ci] = (float)((rb[0] + rb[1]) + (rb[2] + rb[3]));
        }
    } else if (blockIdx.x == 3) {
        for (int i = tid; i < TC * 27; i += NET_WG) {
            const int o = i / 27, k = i - o * 27, ci = k / 9, t = k - ci * 9;
            g_w0[i] = in_w[ci] * Gsum[(ci * 9 + t) * TC + o] + in_b[ci] * Gsum[(27 + t) * TC + o];
        }
    } else {
        const int c = tid & 63, jj = tid >> 6;
        double t = 0.0;
#pragma unroll 8
        for (int b = jj; b < bias_nparts; b += NET_WG / 64) t += bias_part[(size_t)b * TC + c];
        rg[tid] = t;
        __syncthreads();
        if (jj == 0) g_b0[c] = (float)((rg[c] + rg[64 + c]) + (rg[128 + c] + rg[192 + c]));
    }
}

void net_free(dbaz_trainer *t)
{
    dbaz_net_buffers *b = t->net;
    if (!b) return;
    void *ptrs[] = {b->Y0, b->mask0, b->st, b->ws, b->Yh, b->Hh, b->dHh, b->Wc, b->bc, b->logits, b->dlg, b->gemm_part, b->logp, b->v,
                    b->sums_h, b->stem_part, b->Gsum, b->val};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete b;
    t->net = nullptr;
}

// offsets (floats) into dbaz_net_buffers::st and (doubles) into ::ws
enum { ST_IN_MEAN = 0, ST_IN_INVSTD = 4, ST_MEAN0 = 8, ST_INVSTD0 = 72, ST_MEAN_H = 136, ST_INVSTD_H = 168, ST_BH = 200, ST_WH = 256,
       ST_FLOATS = 256 + TC * HC2 };
#define NET_HB 1024  // workgroups of the head row kernels
#define NET_OB 256   // workgroups of k_head_out_bwd
#define NET_SB 2048  // workgroups of k_stem_conv
#define FC_SPLITS 16   // fc weight gradient: K = batch
#define FCF_SPLITS 4   // fc forward: K = 32 HW
#define HW_SPLITS 392
#define STEM_SPLITS 392
enum { WS_IN = 0, WS_HP1 = 3 * BN_NB * 2 + 16, WS_HP2 = WS_HP1 + NET_SB * 2 * TC }; // (HP1 also holds k_stem_conv's NET_SB rows of [2][64])

static int net_alloc(dbaz_trainer *t, int A, int VF)
{
    if (t->net && t->net->A == A && t->net->VF == VF) return DBAZ_OK;
    net_free(t);
    dbaz_net_buffers *b = new dbaz_net_buffers();
    t->net = b;
    b->hc = HC; b->A = A; b->VF = VF; b->NO = A + VF; b->NOp = (b->NO + 3) & ~3; b->KF = t->HW * HC2; b->maxN = t->maxN;
    const size_t rows = (size_t)t->maxN * t->HW;
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc((void **)&b->Y0, rows * TC * 4);
    alloc((void **)&b->mask0, rows * 8);
    alloc((void **)&b->st, (size_t)ST_FLOATS * 4);
    const size_t ws_doubles = (size_t)WS_HP2 + (size_t)NET_HB * HC2 + (size_t)NET_OB * (b->NOp + VF + 1);
    alloc((void **)&b->ws, ws_doubles * 8);
    alloc((void **)&b->Yh, rows * HC2 * 4);
    alloc((void **)&b->Hh, rows * HC2 * 4);
    alloc((void **)&b->dHh, rows * HC2 * 4);
    alloc((void **)&b->Wc, (size_t)b->NO * b->KF * 4);
    alloc((void **)&b->bc, (size_t)b->NOp * 4);
    alloc((void **)&b->logits, (size_t)t->maxN * b->NOp * 4);
    alloc((void **)&b->dlg, (size_t)t->maxN * b->NOp * 4);
    b->gemm_part_elems = std::max(std::max((size_t)FC_SPLITS * b->NO * b->KF, (size_t)HW_SPLITS * HC2 * TC), (size_t)FCF_SPLITS * t->maxN * b->NOp);
    alloc((void **)&b->gemm_part, b->gemm_part_elems * 4);
    alloc((void **)&b->logp, (size_t)t->maxN * A * 4);
    alloc((void **)&b->v, (size_t)t->maxN * 4);
    alloc((void **)&b->sums_h, (size_t)2 * HC2 * 8);
    alloc((void **)&b->stem_part, (size_t)STEM_SPLITS * 36 * TC * 4);
    alloc((void **)&b->val, rows * 36 * 4);
    alloc((void **)&b->Gsum, (size_t)36 * TC * 4);
    if (e != hipSuccess) {
        const std::string msg = hipGetErrorString(e);
        net_free(t);
        return terr(t, DBAZ_EDEVICE, "allocation of the stem / head buffers failed: %s", msg.c_str());
    }
    return DBAZ_OK;
}

static bool net_tensors_ok(const dbaz_net_tensors *p)
{
    return p && p->bn_input_w && p->bn_input_b && p->conv0_w && p->conv0_b && p->bn0_w && p->bn0_b && p->blk_conv_w && p->blk_conv_b &&
           p->blk_bn_w && p->blk_bn_b && p->ph_conv_w && p->ph_conv_b && p->ph_bn_w && p->ph_bn_b && p->ph_fc_w && p->ph_fc_b &&
           p->vh_conv_w && p->vh_conv_b && p->vh_bn_w && p->vh_bn_b && p->vh_fc0_w && p->vh_fc0_b && p->vh_fc1_w && p->vh_fc1_b;
}

// p, v = model(x) of a ResNetZero in training mode (nn.py:108-122): see the section comment.  x [n][3][H][W]; logp [n][A]
// (log_softmax of the policy head), v [n] (tanh of the value head).  The handle keeps what the backward pass needs.
extern "C" int dbaz_trainer_net_forward(dbaz_trainer *t, int32_t n, const float *x, const dbaz_net_tensors *P, const dbaz_net_running *R,
                                        int32_t head_channels, int32_t n_actions, int32_t value_fc, float *logp, float *v, void *stream)
{
    if (!t) return DBAZ_EINVAL;
    if (n < 1 || n > t->maxN) return terr(t, DBAZ_EINVAL, "batch %d outside 1..%d", n, t->maxN);
    if (!x || !logp || !v || !net_tensors_ok(P)) return terr(t, DBAZ_EINVAL, "null argument");
    if (head_channels != HC) return terr(t, DBAZ_EINVAL, "the heads are built for %d channels (got %d)", HC, head_channels);
    if (n_actions < 1 || n_actions > 1024 || value_fc < 1 || value_fc > 256) // (k_head_out_bwd's column sums: 4 (A + 2 vf + 1) doubles of LDS)
        return terr(t, DBAZ_EINVAL, "n_actions %d / value_fc %d unsupported", n_actions, value_fc);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(t, hipSetDevice(t->dev));
    const int rc = net_alloc(t, n_actions, value_fc);
    if (rc != DBAZ_OK) return rc;
    dbaz_net_buffers *b = t->net;
    const int L = t->L, HW = t->HW, A = b->A, VF = b->VF, NOp = b->NOp, KF = b->KF, NO = b->NO;
    const long long M = (long long)n * HW;
    const size_t ae = act_elems(t);
    t->have_fwd = false;
    HIPCHK(t, hipMemsetAsync(t->amax, 0, (size_t)(L + 2) * 4, s));
    float *st = b->st;
    // stem: bn_input's batch statistics, conv0 on the normalized input (+ bn0's statistics), bn0 + ReLU -> A[0]
    train_bn2d_statistics(t, s, x, n, 3, HW, b->ws + WS_IN, st + ST_IN_MEAN, st + ST_IN_INVSTD, R ? R->bn_input_mean : nullptr,
                          R ? R->bn_input_var : nullptr);
    const int sb = std::min(NET_SB, (n + STEM_S - 1) / STEM_S);
    hipLaunchKernelGGL(k_stem_conv, dim3(sb), dim3(NET_WG), (size_t)(STEM_S * 3 + 9) * HW * 4, s, x, st + ST_IN_MEAN, st + ST_IN_INVSTD, P->bn_input_w,
                       P->bn_input_b, P->conv0_w, P->conv0_b, reinterpret_cast<f32x4 *>(b->Y0), n, t->H, t->W, b->ws + WS_HP1);
    train_bn_forward_rows(t, s, b->ws + WS_HP1, sb, M, st + ST_MEAN0, st + ST_INVSTD0, R ? R->bn0_mean : nullptr, R ? R->bn0_var : nullptr, b->Y0,
                          nullptr, t->A, P->bn0_w, P->bn0_b, t->amax, b->mask0);
    tower_forward_rows(t, n, P->blk_conv_w, P->blk_conv_b, P->blk_bn_w, P->blk_bn_b, R ? R->blk_mean : nullptr, R ? R->blk_var : nullptr, s);
    // heads
    hipLaunchKernelGGL(k_head_pack, dim3(256), dim3(NET_WG), 0, s, P->ph_conv_w, P->ph_conv_b, P->vh_conv_w, P->vh_conv_b, P->ph_fc_w, P->ph_fc_b,
                       P->vh_fc0_w, P->vh_fc0_b, st + ST_WH, st + ST_BH, b->Wc, b->bc, HW, A, VF, NOp);
    const int hb = (int)std::min<long long>(NET_HB, (M + 127) / 128);
    hipLaunchKernelGGL(k_head_conv, dim3(hb), dim3(NET_WG), 0, s, reinterpret_cast<const f32x4 *>(t->A + ae * L), st + ST_WH, st + ST_BH,
                       reinterpret_cast<f32x4 *>(b->Yh), M, b->ws + WS_HP1);
    hipLaunchKernelGGL(k_head_stats_fin, dim3(HC2), dim3(NET_WG), 0, s, b->ws + WS_HP1, hb, M, t->eps, t->momentum, st + ST_MEAN_H, st + ST_INVSTD_H,
                       R ? R->ph_mean : nullptr, R ? R->ph_var : nullptr, R ? R->vh_mean : nullptr, R ? R->vh_var : nullptr);
    hipLaunchKernelGGL(k_head_bn_apply, dim3((int)std::min<long long>(1024, (M * 8 + NET_WG - 1) / NET_WG)), dim3(NET_WG), 0, s,
                       reinterpret_cast<const f32x4 *>(b->Yh), reinterpret_cast<f32x4 *>(b->Hh), M * 8, st + ST_MEAN_H, st + ST_INVSTD_H, P->ph_bn_w,
                       P->ph_bn_b, P->vh_bn_w, P->vh_bn_b);
    launch_gemm(s, b->Hh, KF, 1, b->Wc, 1, KF, b->gemm_part, NOp, n, NO, KF, b->bc, FCF_SPLITS, (long long)t->maxN * NOp);
    hipLaunchKernelGGL(k_head_out, dim3(std::min(1024, (n + 3) / 4)), dim3(NET_WG), 0, s, b->gemm_part, gemm_splits(KF, FCF_SPLITS),
                       (long long)t->maxN * NOp, b->logits, n, A, VF, NOp, P->vh_fc1_w, P->vh_fc1_b, logp, v, b->logp, b->v);
    HIPCHK(t, hipGetLastError());
    t->n = n;
    t->have_fwd = true;
    t->net_fwd = true;
    return DBAZ_OK;
}

// Backward of the dbaz_trainer_net_forward pass the handle holds: d_logp [n][A], d_v [n] -> every parameter gradient of `grads`
// WRITTEN (same layout as the parameters).  x and params: as in the forward call.
extern "C" int dbaz_trainer_net_backward(dbaz_trainer *t, const float *x, const float *d_logp, const float *d_v, const dbaz_net_tensors *P,
                                         const dbaz_net_tensors *G, void *stream)
{
    if (!t) return DBAZ_EINVAL;
    if (!t->have_fwd || !t->net_fwd || !t->net) return terr(t, DBAZ_ESTATE, "dbaz_trainer_net_backward without a dbaz_trainer_net_forward pass");
    if (!x || !d_logp || !d_v || !net_tensors_ok(P) || !net_tensors_ok(G)) return terr(t, DBAZ_EINVAL, "null argument");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(t, hipSetDevice(t->dev));
    dbaz_net_buffers *b = t->net;
    const int L = t->L, HW = t->HW, n = t->n, A = b->A, VF = b->VF, NOp = b->NOp, KF = b->KF, NO = b->NO;
    const long long M = (long long)n * HW;
    const size_t ae = act_elems(t);
    float *st = b->st;
    double *hp1 = b->ws + WS_HP1, *hp2 = b->ws + WS_HP2, *hp3 = hp2 + (size_t)NET_HB * HC2;
    // heads
    hipLaunchKernelGGL(k_head_out_bwd, dim3(NET_OB), dim3(NET_WG), (size_t)4 * (NOp + VF + 1) * 8, s, d_logp, d_v, b->logp, b->v, b->logits, n, A, VF, NOp,
                       P->vh_fc1_w, b->dlg, hp3);
    hipLaunchKernelGGL(k_head_small_fin, dim3(NOp + VF + 1), dim3(64), 0, s, hp3, NET_OB, A, VF, NOp, G->ph_fc_b, G->vh_fc0_b, G->vh_fc1_w, G->vh_fc1_b);
    launch_gemm(s, b->dlg, 1, NOp, b->Hh, KF, 1, b->gemm_part, KF, NO, KF, n, nullptr, FC_SPLITS, (long long)NO * KF);
    hipLaunchKernelGGL(k_fc_wgrad_fin, dim3(512), dim3(NET_WG), 0, s, b->gemm_part, gemm_splits(n, FC_SPLITS), (long long)NO * KF, HW, A, VF, G->ph_fc_w,
                       G->vh_fc0_w);
    launch_gemm(s, b->dlg, NOp, 1, b->Wc, KF, 1, b->dHh, KF, n, KF, NO, nullptr, 1, 0);
    const int hb2 = (int)std::min<long long>(NET_HB, (M + 31) / 32);
    hipLaunchKernelGGL(k_head_bn_bwd_sums, dim3(hb2), dim3(NET_WG), 0, s, reinterpret_cast<const f32x4 *>(b->dHh), reinterpret_cast<const f32x4 *>(b->Hh),
                       reinterpret_cast<const f32x4 *>(b->Yh), M, st + ST_MEAN_H, st + ST_INVSTD_H, hp1);
    hipLaunchKernelGGL(k_head_bn_bwd_fin, dim3(HC2), dim3(NET_WG), 0, s, hp1, hb2, b->sums_h, G->ph_bn_w, G->ph_bn_b, G->vh_bn_w, G->vh_bn_b);
    hipLaunchKernelGGL(k_head_bn_bwd_apply, dim3(hb2), dim3(NET_WG), 0, s, reinterpret_cast<f32x4 *>(b->dHh), reinterpret_cast<const f32x4 *>(b->Hh),
                       reinterpret_cast<const f32x4 *>(b->Yh), M, st + ST_MEAN_H, st + ST_INVSTD_H, P->ph_bn_w, P->vh_bn_w, b->sums_h, hp2);
    const int hsplits = (int)std::min<long long>(HW_SPLITS, (M + 31) / 32);
    launch_gemm(s, b->dHh, 1, HC2, t->A + ae * L, TC, 1, b->gemm_part, TC, HC2, TC, (int)M, nullptr, hsplits, (long long)HC2 * TC);
    hipLaunchKernelGGL(k_head_wgrad_fin, dim3(HC2 * TC / 16 + HC2 / 4), dim3(NET_WG), 0, s, b->gemm_part, gemm_splits((int)M, hsplits), hp2, hb2, G->ph_conv_w,
                       G->vh_conv_w, G->ph_conv_b, G->vh_conv_b);
    hipLaunchKernelGGL(k_head_conv_bwd_data, dim3((int)std::min<long long>(1024, (M + 31) / 32)), dim3(NET_WG), 0, s,
                       reinterpret_cast<const f32x4 *>(b->dHh), st + ST_WH, reinterpret_cast<f32x4 *>(t->dA[0]), M);
    // tower; its bottom conv leaves bn0's backward sums
    BelowTower below;
    below.mask = b->mask0; below.y = b->Y0; below.mean = st + ST_MEAN0; below.invstd = st + ST_INVSTD0;
    below.g_w = G->bn0_w; below.g_b = G->bn0_b;
    const int cur = tower_backward_rows(t, P->blk_bn_w, G->blk_conv_w, G->blk_conv_b, G->blk_bn_w, G->blk_bn_b, below, s);
    // stem (bn0's backward sums, dgamma and dbeta were finished by the tower's last k_wgrad_reduce launch)
    const int rb = red_blocks(M);
    unsigned *dymax = t->amax + L + 1;
    train_bn_backward_apply_rows(t, s, t->dA[cur], b->mask0, b->Y0, M, st + ST_MEAN0, st + ST_INVSTD0, P->bn0_w, t->sums, t->dY, dymax, t->part);
    hipLaunchKernelGGL(k_stem_val, dim3(2048), dim3(NET_WG), 0, s, x, st + ST_IN_MEAN, st + ST_IN_INVSTD, M, t->H, t->W, b->val);
    const int ssplits = (int)std::min<long long>(STEM_SPLITS, (M + 31) / 32);
    launch_gemm(s, b->val, 1, 36, t->dY, TC, 1, b->stem_part, TC, 36, TC, (int)M, nullptr, ssplits, (long long)36 * TC);
    hipLaunchKernelGGL(k_stem_fin1, dim3(36 * 4), dim3(NET_WG), 0, s, b->stem_part, gemm_splits((int)M, ssplits), b->Gsum);
    hipLaunchKernelGGL(k_stem_fin2, dim3(5), dim3(NET_WG), 0, s, b->Gsum, P->conv0_w, P->bn_input_w, P->bn_input_b, t->part, rb, G->conv0_w, G->conv0_b,
                       G->bn_input_w, G->bn_input_b);
    HIPCHK(t, hipGetLastError());
    t->have_fwd = false;
    return DBAZ_OK;
}
