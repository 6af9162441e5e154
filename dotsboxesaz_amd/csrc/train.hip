// train.hip -- TRAINING-mode forward and backward of ResNetZero's residual tower on gfx950.
//
// Reference: ResBlock.forward (nn.py:48-57: relu(bn1(conv1(x))) -> bn2(conv2(.)) -> += x -> relu) under
// model.train(True) inside NeuralNetWrapper.train (nn.py:203-221): BatchNorm2d normalises with the statistics of the
// BATCH (biased variance), updates running_mean / running_var (momentum 0.1, unbiased variance), and
// loss.backward() differentiates through those statistics.  The tower is 2*blocks conv3x3 (64 -> 64) layers, 98 % of the
// step's FLOPs; the stem, the heads and the whole-network entry points live in train_net.hip; BatchNorm2d on NCHW tensors, the
// loss and the SGD update are at the end of this file.
//
// Data layout: activations and gradients live in HBM as f32 NHWC -- row = sample * HW + position, 64 channels = 256 B per
// row -- which is the row layout of the conv kernels' LDS images.  Per layer l (input A[l], l = 0 .. L-1):
//   forward    Y[l]   = conv3x3(A[l]) + bias                         k_conv_t   (f16x3 MFMA, as the self-play tower)
//              mean, invstd over the n*HW rows of Y[l]                k_conv_t's epilogue (f64 sums per workgroup) + k_bn_stats_fin
//              A[l+1] = relu(bn(Y[l]) (+ A[l-1] for the second conv of a block))   k_bn_apply
//   backward   g = dA[l+1] * (A[l+1] > 0); sums of g and g*yhat      epilogue of the k_conv_t that wrote dA[l+1] (top layer: k_bn_bwd_sums)
//                                                                    + k_bn_bwd_sums_fin (-> dgamma, dbeta)
//              dY[l] = gamma*invstd*(g - mean(g) - yhat*mean(g*yhat)) k_bn_bwd_apply (also keeps g for the skip path)
//              dW[l] = sum_rows A[l](row + tap) x dY[l](row)          k_wgrad_h3 (f16x3 MFMA; k_wgrad: exact f32) + k_wgrad_reduce
//              dA[l] = conv3x3^T(dY[l]) (+ g of the block's end)      k_conv_t with flipped / transposed fragments
// Every f32 operand of k_conv_t is an error-compensated (hi, lo) pair of halves scaled by a power of two taken from the
// tensor's own maximum (tracked by the kernel that produced it), so gradients of any magnitude keep f32-grade products.
#include "train.h"

static thread_local std::string g_train_error; // message of a failed dbaz_trainer_create / dbaz_bn2d_* call; per thread

int terr(dbaz_trainer *t, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    (t ? t->err : g_train_error) = buf;
    return code;
}

// ------------------------------------------------------------------------------------
// NCHW (torch) <-> NHWC rows; one workgroup per sample
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_nchw_to_rows(const float *__restrict__ in, float *__restrict__ out, int HW, unsigned *amax)
{
    extern __shared__ float sm[]; // [C][HW + 1]
    const int s = blockIdx.x, tid = threadIdx.x;
    const float *src = in + (size_t)s * TC * HW;
    float mx = 0.0f;
    for (int i = tid; i < TC * HW; i += 256) {
        const int c = i / HW, p = i - c * HW;
        const float v = src[i];
        mx = fmaxf(mx, fabsf(v));
        sm[c * (HW + 1) + p] = v;
    }
    __syncthreads();
    float *dst = out + (size_t)s * HW * TC;
    for (int i = tid; i < TC * HW; i += 256) {
        const int p = i >> 6, c = i & 63;
        dst[i] = sm[c * (HW + 1) + p];
    }
    if (amax) { // one atomic per workgroup (one per wave of 4 096 workgroups serialises for 160 us)
        __shared__ float s_mx[4];
        mx = wave_max(mx);
        if ((tid & 63) == 0) s_mx[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) {
            const float m = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
            if (m > 0.0f) atomicMax(amax, __float_as_uint(m));
        }
    }
}

__global__ void __launch_bounds__(256) k_rows_to_nchw(const float *__restrict__ in, float *__restrict__ out, int HW)
{
    extern __shared__ float sm[]; // [HW][C + 1]
    const int s = blockIdx.x, tid = threadIdx.x;
    const float *src = in + (size_t)s * HW * TC;
    for (int i = tid; i < TC * HW; i += 256) sm[(i >> 6) * (TC + 1) + (i & 63)] = src[i];
    __syncthreads();
    float *dst = out + (size_t)s * TC * HW;
    for (int i = tid; i < TC * HW; i += 256) {
        const int c = i / HW, p = i - c * HW;
        dst[i] = sm[p * (TC + 1) + c];
    }
}

// ------------------------------------------------------------------------------------
// weights [cout][cin][3][3] f32 (torch) -> (hi, lo) MFMA fragments [ct][tap][ks][hi|lo][lane][8 halves], scaled by 2^sw
// with max|w| * 2^sw in [2^13, 2^14) (the layout nn.hip's pack_conv builds on the host).  blockIdx.y = 1 packs the
// transposed convolution of the backward pass: out channel = cin, in channel = cout, tap flipped.
// ------------------------------------------------------------------------------------
struct PackArgs { const float *w[TL_MAX]; };

__global__ void __launch_bounds__(TT) k_pack_w(PackArgs pa, _Float16 *__restrict__ wpk, float *__restrict__ wsc, int L)
{
    __shared__ float red[TT / 64];
    __shared__ float s_scale;
    const int l = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const float *w = pa.w[l];
    constexpr int NW = TC * TC * 9;
    float mx = 0.0f;
    for (int i = tid; i < NW; i += TT) mx = fmaxf(mx, fabsf(w[i]));
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        float m = 0.0f;
        for (int i = 0; i < TT / 64; i++) m = fmaxf(m, red[i]);
        const float sc = scale_from_max(__float_as_uint(m));
        s_scale = sc;
        if (blockIdx.z == 0) wsc[dir * L + l] = 1.0f / sc;
    }
    __syncthreads();
    const float sc = s_scale;
    _Float16 *o = wpk + ((size_t)dir * L + l) * NW * 2;
    constexpr int KS = TC / 32;
    for (int i = tid + blockIdx.z * TT; i < NW; i += TT * gridDim.z) { // (every z slice finds the layer's maximum itself)
        // i = (((ct * 9 + tap) * KS + ks) * 64 + lane) * 8 + e
        const int e = i & 7, lane = (i >> 3) & 63, r = i >> 9, ks = r % KS, r2 = r / KS, tap = r2 % 9, ct = r2 / 9;
        const int oc = ct * 16 + (lane & 15), ic = ks * 32 + 8 * (lane >> 4) + e;
        const float v = (dir == 0 ? w[((size_t)oc * TC + ic) * 9 + tap] : w[((size_t)ic * TC + oc) * 9 + (8 - tap)]) * sc;
        const _Float16 h = (_Float16)v;
        const size_t base = ((size_t)r * 2) * 64 * 8 + (size_t)lane * 8 + e;
        o[base] = h;
        o[base + 64 * 8] = (_Float16)(v - (float)h);
    }
}

// ------------------------------------------------------------------------------------
// conv3x3 64 -> 64 over NHWC rows, f16x3 on v_mfma_f32_16x16x32_f16 with two cout tiles per wave (the tiling of
// nn.hip's conv_lds_h3_c2): a workgroup stages the S samples' rows as (hi, lo) halves in an LDS image of (C+8)-dword rows,
// wave w owns couts [32 (w & 1), +32) and position tiles [4 (w >> 1), +4); out-of-image taps read a zero region at the
// lane's own bank slot.  out = acc * 2^-(sx+sw) (+ bias) (+ add).  128 registers: two workgroups per CU.
// ------------------------------------------------------------------------------------
// ReLU mask, one bit per element: (row, channel) is set where the layer's output is > 0 -- what the backward pass needs of A --
// at 8 bytes per row instead of 256; bit 16 e + cq of mask[row] belongs to element e of channel quad cq (four 16-lane ballots)
__device__ __forceinline__ unsigned quad_mask(unsigned long long m, int cq)
{
    const unsigned lo = (unsigned)(m >> cq), hi = (unsigned)(m >> (32 + cq));
    return (lo & 1u) | ((lo >> 15) & 2u) | ((hi & 1u) << 2) | ((hi >> 13) & 8u);
}

struct ConvArgs {
    const float *in;          // [n*HW][C]
    const unsigned *in_max;   // bits of max|in|
    const _Float16 *wpk;      // this layer's fragments
    const float *wsc;         // this layer's 2^-sw
    const float *bias;        // [C] or nullptr
    const float *add;         // [n*HW][C] added to the result, or nullptr
    float *out;               // [n*HW][C]
    double *stat_part;        // [grid][2][C]: the workgroup's sums of out and out^2 over its rows (the forward convs: BatchNorm's
                              // batch statistics without another pass over the tensor), or nullptr
    // input-gradient convs: out (+ add) is dA of the layer below; its BatchNorm backward needs sum(g), sum(g * yhat) over the rows
    double *bs_part;                      // [grid][2][C], or nullptr
    const unsigned long long *bs_mask;    // that layer's ReLU mask, one word per row
    const float *bs_y, *bs_mean, *bs_invstd; // its conv output Y and batch statistics
    int n, S, H, W;
#ifdef DBAZ_STAMP
    unsigned long long *stamp_out; // diagnostic build only: [grid][8 waves][8]
#endif
};

#ifdef DBAZ_STAMP
#define TSTAMP(var)                                                                     \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
#define TSTAMP_RT(var)                                                                  \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
#else
#define TSTAMP(var) do { } while (0)
#define TSTAMP_RT(var) do { } while (0)
#endif

__global__ void __launch_bounds__(TT, 4) k_conv_t(ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int C = TC, S4 = (C + 8) / 4, KS = C / 32, LO = C / 8, N = 9 * KS, NTT = 4;
    const int HW = a.H * a.W, W = a.W, H = a.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s0 = blockIdx.x * a.S;
    const int ns = min(a.S, a.n - s0);
    if (ns <= 0) return;
#ifdef DBAZ_STAMP
    unsigned long long ts0, ts1, ts2, ts3, ts4, ts5, tr0, tr1;
    TSTAMP_RT(tr0);
    TSTAMP(ts0);
#endif
    const int R = ns * HW;
    const int zu = (a.S * HW * S4 + 15) & ~15;
    f32x4 *X4 = reinterpret_cast<f32x4 *>(lds);
    const f32x4 *wpk = reinterpret_cast<const f32x4 *>(a.wpk);
    const int ct0 = (wave & 1) * 2;
    // weight fragments (L2 -> registers) run RD - 1 steps ahead in an RD-deep ring; the first RD - 1 steps are fetched before
    // the staging so that their latency hides behind it
    constexpr int RD = 2;
    const f32x4 *wb0 = wpk + (size_t)ct0 * N * 2 * 64 + lane;
    const f32x4 *wb1 = wb0 + (size_t)N * 2 * 64;
    u128h a_h[2][RD], a_l[2][RD];
#pragma unroll
    for (int j = 0; j < RD - 1; j++) {
        a_h[0][j].f = wb0[(size_t)j * 128]; a_l[0][j].f = wb0[(size_t)j * 128 + 64];
        a_h[1][j].f = wb1[(size_t)j * 128]; a_l[1][j].f = wb1[(size_t)j * 128 + 64];
    }
    const float sx = scale_from_max(*a.in_max);
    {
        // all of this thread's rows are requested before the first one is converted (one HBM latency per workgroup, not eight)
        const f32x4 *in4 = reinterpret_cast<const f32x4 *>(a.in) + (size_t)s0 * HW * (C / 4);
        _Float16 *img = reinterpret_cast<_Float16 *>(lds);
        constexpr int NLD = 256 * (C / 4) / TT; // <= 256 rows per workgroup
        f32x4 pf[NLD];
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int i = tid + j * TT;
            pf[j] = i < R * (C / 4) ? in4[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int i = tid + j * TT;
            const int row = i >> 4, c4 = i & 15;
            const f32x4 v = pf[j] * sx;
            union { h2v h[2]; u32x2 u; } oh, ol;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const f2v x = {v[2 * q], v[2 * q + 1]};
                const h2v h = __builtin_convertvector(x, h2v);
                oh.h[q] = h;
                ol.h[q] = __builtin_convertvector(x - __builtin_convertvector(h, f2v), h2v);
            }
            if (i < R * (C / 4)) {
                _Float16 *ph = img + (size_t)row * (S4 * 8) + c4 * 4;
                *reinterpret_cast<u32x2 *>(ph) = oh.u;
                *reinterpret_cast<u32x2 *>(ph + C) = ol.u;
            }
        }
        if (tid < 3 * S4) X4[zu + tid] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (tid == 0) *reinterpret_cast<unsigned *>(X4 + zu + 3 * S4 + (TT / 64) * 2 * C / 2) = 0u; // arrival counter of the epilogue's column sums
    }
    TSTAMP(ts1);
    __syncthreads();
    TSTAMP(ts2);
    const int jrow = lane & 15, gq = lane >> 4;
    const int tbase = (wave >> 1) * NTT;
    int vm[NTT];
#pragma unroll
    for (int t = 0; t < NTT; t++) {
        const int row = (tbase + t) * 16 + jrow;
        const int pos = row % HW, y = pos / W, x = pos - y * W;
        int m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            m |= ((yy >= 0) && (yy < H) && (xx >= 0) && (xx < W)) ? (1 << tap) : 0;
        }
        vm[t] = row < R ? m : 0;
    }
    const int rowbase = (tbase * 16 + jrow) * S4 + gq;
    const int zbase = zu;
    f32x4 acc[2][NTT];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int t = 0; t < NTT; t++) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u128h bh[NTT], bl[NTT];
    const char *sb = reinterpret_cast<const char *>(lds);
    int ab[NTT];
#pragma unroll
    for (int t = 0; t < NTT; t++)
        ab[t] = ((vm[t] & 1) ? rowbase + (-W - 1) * S4 : zbase + ((rowbase + (-W - 1) * S4) & 15) - t * 16 * S4) * 16;
#pragma unroll
    for (int t = 0; t < NTT; t++) bh[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4);
#pragma unroll
    for (int t = 0; t < NTT; t++) bl[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4 + LO * 16);
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int cur = i % RD, nxt = (i + RD - 1) % RD;
        const int ni = i + 1, ntap = ni / KS, nks = ni % KS;
        if (i + RD - 1 < N) {
            a_h[0][nxt].f = wb0[(size_t)(i + RD - 1) * 128];
            a_l[0][nxt].f = wb0[(size_t)(i + RD - 1) * 128 + 64];
            a_h[1][nxt].f = wb1[(size_t)(i + RD - 1) * 128];
            a_l[1][nxt].f = wb1[(size_t)(i + RD - 1) * 128 + 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NTT; t++) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[0][cur].h, bh[t].h, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[1][cur].h, bh[t].h, acc[1][t], 0, 0, 0);
        }
        if (ni < N && nks == 0) {
            const int off = ((ntap / 3 - 1) * W + (ntap % 3 - 1)) * S4;
            const int zt = zbase + ((rowbase + off) & 15);
#pragma unroll
            for (int t = 0; t < NTT; t++) ab[t] = (((vm[t] >> ntap) & 1) ? rowbase + off : zt - t * 16 * S4) * 16;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NTT; t++) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l[0][cur].h, bh[t].h, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l[1][cur].h, bh[t].h, acc[1][t], 0, 0, 0);
            if (ni < N) bh[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4 + nks * 64);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NTT; t++) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[0][cur].h, bl[t].h, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[1][cur].h, bl[t].h, acc[1][t], 0, 0, 0);
            if (ni < N) bl[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4 + nks * 64 + LO * 16);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- epilogue: lane holds couts (ct0 + c) * 16 + 4 gq .. +3 of position row (tbase + t) * 16 + jrow.  The results go
    // through the (now idle) LDS image as f32 rows of C + 4 dwords and leave as whole 256-byte rows: written straight from the
    // accumulator layout they were 64-byte pieces of 16 different rows per instruction (13 us of a 55 us launch)
    const float osc = (1.0f / sx) * *a.wsc;
    constexpr int OS = C + 4; // dwords per staged output row: the 16 rows of a tile fall on disjoint banks
    // what the store pass adds to / reads beside the result is requested HERE, before the barriers: thread tid stores quads
    // tid + j TT (channel quad tid & 15 of rows (tid >> 4) + 32 j), so its operands are known now and their HBM latency hides
    // behind the wait for the slowest wave and the staging (fetched inside the store loop they were one latency per pass)
    constexpr int NST = 256 * (C / 4) / TT;
    int te = tid;
    asm volatile("" : "+v"(te)); // the epilogue's addresses are formed HERE: hoisted above the MFMA loop they cost 29 spilled registers
    const size_t g0 = (size_t)s0 * HW * (C / 4);
    const int nq = R * (C / 4);
    f32x4 pa[NST], py[NST];
    unsigned m4 = 0;
    if (a.bs_part) {
        unsigned long long mw[NST];
#pragma unroll
        for (int j = 0; j < NST; j++) {
            const int i = te + j * TT;
            mw[j] = i < nq ? a.bs_mask[(size_t)s0 * HW + (i >> 4)] : 0ull;
        }
        const f32x4 *y4 = reinterpret_cast<const f32x4 *>(a.bs_y) + g0;
#pragma unroll
        for (int j = 0; j < NST; j++) {
            const int i = te + j * TT;
            py[j] = i < nq ? y4[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NST; j++) m4 |= quad_mask(mw[j], te & 15) << (4 * j);
    }
    if (a.add) {
        const f32x4 *add4 = reinterpret_cast<const f32x4 *>(a.add) + g0;
#pragma unroll
        for (int j = 0; j < NST; j++) {
            const int i = te + j * TT;
            pa[j] = i < nq ? add4[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    TSTAMP(ts3);
    __syncthreads();          // every wave has left the MFMA loop
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int cq = (ct0 + c) * 4 + gq;
        const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + cq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NTT; t++) {
            const int row = (tbase + t) * 16 + jrow;
            if (row < R) *reinterpret_cast<f32x4 *>(lds + (size_t)row * OS + cq * 4) = acc[c][t] * osc + bv;
        }
    }
    __syncthreads();
    TSTAMP(ts4);
    {
        f32x4 *out4 = reinterpret_cast<f32x4 *>(a.out) + g0;
        // column sums in f64 beside the store: (out, out^2) for the forward convs -- BatchNorm's batch statistics -- or, for the
        // input-gradient convs, (g, g * yhat) of the NEXT layer down (g = out where that layer's ReLU let the value through):
        // what k_bn_stats / k_bn_bwd_sums read the tensor again for (9.8 and 20.3 us per layer)
        double st[2][4] = {};
        f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu;
        if (a.bs_part) {
            mu = *reinterpret_cast<const f32x4 *>(a.bs_mean + (te & 15) * 4);
            is = *reinterpret_cast<const f32x4 *>(a.bs_invstd + (te & 15) * 4);
        }
#pragma unroll
        for (int j = 0; j < NST; j++) {
            const int i = te + j * TT;
            if (i < nq) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(lds + (size_t)(i >> 4) * OS + (i & 15) * 4);
                if (a.add) v += pa[j];
                out4[i] = v;
                if (a.stat_part) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { const double d = v[e]; st[0][e] += d; st[1][e] += d * d; }
                }
                if (a.bs_part) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float g = ((m4 >> (4 * j + e)) & 1u) ? v[e] : 0.0f;
                        const float yh = (py[j][e] - mu[e]) * is[e];
                        st[0][e] += (double)g;
                        st[1][e] += (double)g * (double)yh;
                    }
                }
            }
        }
        double *sp = a.stat_part ? a.stat_part : a.bs_part;
        if (sp) {
            // the thread's channel quad is tid & 15 on every pass (TT is a multiple of 16): the wave's four row lanes of a quad
            // are added up by two exchanges, lanes 0..15 leave the WAVE's sums in its own LDS slot, and the last wave to arrive
            // (an LDS counter, no workgroup barrier: one through LDS behind two barriers cost 6.8 us per launch -- a
            // workgroup's tail is the launch's critical path twice over) adds the 8 slots in wave order and writes the
            // workgroup's row
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    st[k][e] += __shfl_xor(st[k][e], 16);
                    st[k][e] += __shfl_xor(st[k][e], 32);
                }
            double *slots = reinterpret_cast<double *>(X4 + zu + 3 * S4);       // [8 waves][2][C]
            unsigned *arrived = reinterpret_cast<unsigned *>(slots + (TT / 64) * 2 * C);
            if (lane < 16) {
                double *o = slots + wave * 2 * C + lane * 4;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    *reinterpret_cast<double2 *>(o + k * C) = make_double2(st[k][0], st[k][1]);
                    *reinterpret_cast<double2 *>(o + k * C + 2) = make_double2(st[k][2], st[k][3]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            unsigned old = 0;
            if (lane == 0) old = atomicAdd(arrived, 1u);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == TT / 64 - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                double2 v = {0.0, 0.0};
#pragma unroll
                for (int w = 0; w < TT / 64; w++) {
                    const double2 x = *reinterpret_cast<const double2 *>(slots + w * 2 * C + lane * 2);
                    v.x += x.x; v.y += x.y;
                }
                *reinterpret_cast<double2 *>(sp + (size_t)blockIdx.x * 2 * C + lane * 2) = v;
            }
        }
    }
#ifdef DBAZ_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TSTAMP(ts5);
    TSTAMP_RT(tr1);
    if (a.stamp_out && lane == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *o = a.stamp_out + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = ts1 - ts0; o[1] = ts2 - ts1; o[2] = ts3 - ts2; o[3] = ts4 - ts3; o[4] = ts5 - ts4; o[5] = tr0; o[6] = tr1;
        o[7] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
}

// ------------------------------------------------------------------------------------
// column sums over the rows of [M][C] tensors, f64: the workgroup's 512 threads = 32 row lanes x 16 channel quads write one
// partial row per workgroup; the *_fin kernels (colsum_total) add the rows up.
// ------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void block_colsum_store(double (&s)[K][4], double *part /*[blocks][K][C]*/)
{
    __shared__ double red[TT / 16][K][4 * 16 + 1];
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
        for (int e = 0; e < 4; e++) red[rl][k][cq * 4 + e] = s[k][e];
    __syncthreads();
    if (tid < K * TC) {
        const int k = tid / TC, c = tid - k * TC;
        double v = 0.0;
        for (int r = 0; r < TT / 16; r++) v += red[r][k][c];
        part[((size_t)blockIdx.x * K + k) * TC + c] = v;
    }
}

// the partial rows' totals of channel blockIdx.x (grid: C workgroups of 512 threads, one row lane each): tot[k].  One level, no
// last-arriver: a __threadfence costs ~20 us here too (it writes back what the previous kernel left dirty in the XCD's L2).
template <int K>
__device__ __forceinline__ void colsum_total(const double *part, int nparts, double *tot /* LDS [K] */)
{
    __shared__ double red[TT / 64][K];
    const int tid = threadIdx.x, c = blockIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double v = 0.0;
        for (int bb = tid; bb < nparts; bb += TT) v += part[((size_t)bb * K + k) * TC + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((tid & 63) == 0) red[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < K) {
        double v = 0.0;
        for (int w = 0; w < TT / 64; w++) v += red[w][tid];
        tot[tid] = v;
    }
    __syncthreads();
}

__device__ __forceinline__ void block_atomic_max(float mx, unsigned *amax)
{
    __shared__ float s_mx[TT / 64];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.0f;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) m = fmaxf(m, s_mx[i]);
        if (m > 0.0f) atomicMax(amax, __float_as_uint(m));
    }
}

// -> batch mean / invstd and the running statistics (BatchNorm2d training mode: momentum 0.1, unbiased running variance)
__global__ void __launch_bounds__(TT) k_bn_stats_fin(const double *part, int nparts, long long M, float eps, float momentum, float *mean,
                                                     float *invstd, float *run_mean, float *run_var)
{
    __shared__ double tot[2];
    colsum_total<2>(part, nparts, tot);
    if (threadIdx.x == 0) {
        const int c = blockIdx.x;
        const double m = tot[0] / (double)M;
        double var = tot[1] / (double)M - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean) run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + (double)momentum * m);
        if (run_var) {
            const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
            run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + (double)momentum * unb);
        }
    }
}

// A_out = relu(gamma * (y - mean) * invstd + beta (+ res)); tracks max(A_out); writes the ReLU mask (quad_mask)
__global__ void __launch_bounds__(256) k_bn_apply(const f32x4 *__restrict__ y4, const f32x4 *__restrict__ res4, f32x4 *__restrict__ out4,
                                                  long long n4, const float *mean, const float *invstd, const float *gamma,
                                                  const float *beta, unsigned *amax, unsigned long long *__restrict__ mask)
{
    const int cq = threadIdx.x & 15; // (blockDim and the grid stride are multiples of 16)
    const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + cq * 4);
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(invstd + cq * 4) * *reinterpret_cast<const f32x4 *>(gamma + cq * 4);
    const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + cq * 4);
    float mx = 0.0f;
    // four quads per thread and pass, all loads issued before the first use (1-2 loads in flight per thread left the kernel at
    // 4.4 TB/s)
    const long long stride = (long long)gridDim.x * 256;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride) {
        f32x4 y[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long i = i0 + u * stride;
            y[u] = i < n4 ? y4[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            r[u] = (res4 && i < n4) ? res4[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long i = i0 + u * stride;
            f32x4 v = (y[u] - mu) * sc + be + r[u];
            unsigned long long bits = 0;
            const int sh = threadIdx.x & 48; // first lane of this row's 16 lanes within the wave
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v[e] = fmaxf(v[e], 0.0f);
                mx = fmaxf(mx, v[e]);
                bits |= ((__ballot(v[e] > 0.0f && i < n4) >> sh) & 0xffffull) << (16 * e);
            }
            if (i < n4) out4[i] = v;
            if (cq == 0 && i < n4) mask[i >> 4] = bits;
        }
    }
    block_atomic_max(mx, amax); // one atomic per workgroup (one per wave of a 4 096-workgroup grid serialised for 160 us)
}

// backward, pass 1: g = dA * (A_out > 0); partials of sum(g) and sum(g * yhat); the _fin kernel turns them into dbeta, dgamma
// and `sums` for pass 2 and clears the max|dY| word pass 2 accumulates into
__global__ void __launch_bounds__(TT) k_bn_bwd_sums(const f32x4 *__restrict__ dA4, const unsigned long long *__restrict__ mask, const f32x4 *__restrict__ y4,
                                                    long long M, const float *mean, const float *invstd, double *part)
{
    double s[2][4] = {};
    const int cq = threadIdx.x & 15;
    const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + cq * 4);
    const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + cq * 4);
    for (long long r = (long long)blockIdx.x * 32 + (threadIdx.x >> 4); r < M; r += (long long)gridDim.x * 32) {
        const f32x4 d = dA4[r * 16 + cq], y = y4[r * 16 + cq];
        const unsigned m = quad_mask(mask[r], cq);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float g = ((m >> e) & 1u) ? d[e] : 0.0f;
            const float yh = (y[e] - mu[e]) * is[e];
            s[0][e] += (double)g;
            s[1][e] += (double)g * (double)yh;
        }
    }
    block_colsum_store<2>(s, part);
}

__global__ void __launch_bounds__(TT) k_bn_bwd_sums_fin(const double *part, int nparts, double *sums, float *dbeta, float *dgamma, unsigned *zero)
{
    __shared__ double tot[2];
    colsum_total<2>(part, nparts, tot);
    if (threadIdx.x == 0) {
        const int c = blockIdx.x;
        sums[c] = tot[0];
        sums[TC + c] = tot[1];
        dbeta[c] = (float)tot[0];
        dgamma[c] = (float)tot[1];
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *zero = 0u;
}

// backward, pass 2: dY = gamma * invstd * (g - sum(g)/M - yhat * sum(g*yhat)/M); keeps g (skip path of a block's end);
// tracks max|dY|; partials of sum(dY) (the conv bias gradient: totalled by k_wgrad_reduce's last workgroups)
__global__ void __launch_bounds__(TT) k_bn_bwd_apply(const f32x4 *__restrict__ dA4, const unsigned long long *__restrict__ mask, const f32x4 *__restrict__ y4,
                                                     long long M, const float *mean, const float *invstd, const float *gamma,
                                                     const double *sums, f32x4 *__restrict__ dY4, f32x4 *__restrict__ g4,
                                                     unsigned *amax, double *part)
{
    double s[1][4] = {};
    const int cq = threadIdx.x & 15;
    const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + cq * 4);
    const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + cq * 4);
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + cq * 4);
    f32x4 mg, mgy;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        mg[e] = (float)(sums[cq * 4 + e] / (double)M);
        mgy[e] = (float)(sums[TC + cq * 4 + e] / (double)M);
    }
    float mx = 0.0f;
    for (long long r = (long long)blockIdx.x * 32 + (threadIdx.x >> 4); r < M; r += (long long)gridDim.x * 32) {
        const f32x4 d = dA4[r * 16 + cq], y = y4[r * 16 + cq];
        const unsigned m = quad_mask(mask[r], cq);
        f32x4 g, o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            g[e] = ((m >> e) & 1u) ? d[e] : 0.0f;
            const float yh = (y[e] - mu[e]) * is[e];
            o[e] = ga[e] * is[e] * (g[e] - mg[e] - yh * mgy[e]);
            mx = fmaxf(mx, fabsf(o[e]));
            s[0][e] += (double)o[e];
        }
        dY4[r * 16 + cq] = o;
        if (g4) g4[r * 16 + cq] = g;
    }
    block_atomic_max(mx, amax);
    block_colsum_store<1>(s, part);
}

// ------------------------------------------------------------------------------------
// weight gradient dW[tap][cin][cout] = sum over rows of A(row + tap offset)[cin] * dY(row)[cout], exact f32 on
// v_mfma_f32_16x16x4_f32 (m = cin, n = cout, k = 4 consecutive rows).  A workgroup stages chunks of Sw samples: dY as plain
// rows, A into a ZERO-PADDED image (one pad column per line, one pad line per sample, guard rows in front), so that a tap is a
// constant row offset and needs no border mask; tab[row] holds the byte offset of a dY row's window in that image.  Rows are
// C+16 dwords (the two rows of a 32-lane read group fall on disjoint banks).  The whole 9 x 64 x 64 gradient stays in
// registers: wave w owns cin tile w & 3 and cout tiles 2 (w >> 2), +1 for all 9 taps (18 accumulator tiles); the operands
// of K-step k+1 are read from LDS while the 18 MFMAs of step k issue, and the next chunk's rows travel HBM -> registers
// under the whole loop.  The workgroups' partial gradients are summed by k_wgrad_reduce.
// ------------------------------------------------------------------------------------
#define WG_STRIDE (TC + 16)
#define WG_MAXLD 13 // float4 per thread and chunk: 2 images x <= 208 rows x 16 quads / 512 threads

struct WgGeo { int RW, RA, PW, G; };
__host__ __device__ inline WgGeo wg_geo(int Sw, int H, int W)
{
    WgGeo g;
    g.PW = W + 1;
    g.G = g.PW + 1;
    g.RW = (Sw * H * W + 3) & ~3;                 // dY rows per chunk, padded to the K step
    g.RA = 2 * g.G + Sw * (H + 1) * g.PW;          // rows of the padded A image
    return g;
}
static size_t wg_lds_bytes(int Sw, int H, int W)
{
    const WgGeo g = wg_geo(Sw, H, W);
    return (size_t)(g.RA + g.RW) * WG_STRIDE * 4 + (size_t)g.RW * 4;
}

__global__ void __launch_bounds__(TT, 2) k_wgrad(const float *__restrict__ act, const float *__restrict__ dy, int n, int Sw, int H, int W,
                                                 float *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int HW = H * W;
    const WgGeo geo = wg_geo(Sw, H, W);
    const int RW = geo.RW, PW = geo.PW;
    float *Ai = lds;
    float *Di = lds + (size_t)geo.RA * WG_STRIDE;
    int *tab = reinterpret_cast<int *>(Di + (size_t)RW * WG_STRIDE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cit = wave & 3, ch = wave >> 2;
    const int m16 = lane & 15, gq = lane >> 4;
    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; t++) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int i = tid; i < geo.RA * (WG_STRIDE / 4); i += TT) reinterpret_cast<f32x4 *>(Ai)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = tid; r < RW; r += TT) {
        int v = 0;
        if (r < Sw * HW) {
            const int sidx = r / HW, pos = r - sidx * HW, y = pos / W, x = pos - y * W;
            v = (geo.G + sidx * (H + 1) * PW + y * PW + x - PW - 1) * WG_STRIDE * 4; // window start: tap (0,0) = row - PW - 1
        }
        tab[r] = v;
    }
    const int nchunks = (n + Sw - 1) / Sw;
    f32x4 pf[WG_MAXLD];
    const int img4 = RW * 16; // float4 per image
    auto issue = [&](int chunk) {
        const int s0 = chunk * Sw, R = min(Sw, n - s0) * HW;
        const f32x4 *a4 = reinterpret_cast<const f32x4 *>(act) + (size_t)s0 * HW * 16;
        const f32x4 *d4 = reinterpret_cast<const f32x4 *>(dy) + (size_t)s0 * HW * 16;
#pragma unroll
        for (int j = 0; j < WG_MAXLD; j++) {
            const int i = tid + j * TT;
            const int which = i >= img4, ii = i - which * img4;
            pf[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (i < 2 * img4 && (ii >> 4) < R) pf[j] = which ? d4[ii] : a4[ii];
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < WG_MAXLD; j++) {
            const int i = tid + j * TT;
            const int which = i >= img4, ii = i - which * img4;
            if (i < 2 * img4) {
                const int row = ii >> 4, c4 = ii & 15;
                if (which) *reinterpret_cast<f32x4 *>(Di + (size_t)row * WG_STRIDE + c4 * 4) = pf[j];
                else if (row < Sw * HW) // (rows of samples past the batch's end arrive as zeros; their dY rows are zero as well)
                    *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(Ai) + tab[row] + (PW + 1) * WG_STRIDE * 4 + c4 * 16) = pf[j];
            }
        }
    };
    __syncthreads(); // tab, zeroed image
    if ((int)blockIdx.x < nchunks) issue(blockIdx.x);
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; t++) toff[t] = ((t / 3) * PW + (t % 3)) * WG_STRIDE * 4;
    const char *abase = reinterpret_cast<const char *>(Ai) + (cit * 16 + m16) * 4;
    const float *dbase = Di + ch * 32 + m16;
    const int NK = RW / 4;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        __syncthreads(); // the previous chunk's reads are done
        commit();
        __syncthreads();
        if (chunk + (int)gridDim.x < nchunks) issue(chunk + gridDim.x);
        float a0[9], a1[9], b0[2], b1[2];
        auto load = [&](float (&av)[9], float (&bv)[2], int k, int tb) {
            const int r = 4 * k + gq;
            bv[0] = dbase[(size_t)r * WG_STRIDE];
            bv[1] = dbase[(size_t)r * WG_STRIDE + 16];
#pragma unroll
            for (int t = 0; t < 9; t++) av[t] = *reinterpret_cast<const float *>(abase + tb + toff[t]);
        };
        auto mma = [&](const float (&av)[9], const float (&bv)[2]) {
#pragma unroll
            for (int t = 0; t < 9; t++) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[0], acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[1], acc[t][1], 0, 0, 0);
            }
        };
        int tbA = tab[gq], tbB = tab[min(4 + gq, RW - 1)], tbC = 0, tbD = 0;
        load(a0, b0, 0, tbA);
        for (int k = 0; k < NK; k += 2) {
            tbC = tab[min(4 * (k + 2) + gq, RW - 1)];
            tbD = tab[min(4 * (k + 3) + gq, RW - 1)];
            if (k + 1 < NK) load(a1, b1, k + 1, tbB);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (k + 2 < NK) load(a0, b0, k + 2, tbC);
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < NK) mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            tbB = tbD;
        }
    }
    // lane holds dW[tap][cin = cit*16 + 4 gq + i][cout = (2 ch + j)*16 + m16]
    float *o = part + (size_t)blockIdx.x * 9 * TC * TC;
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                o[((size_t)tap * TC + cit * 16 + 4 * gq + i) * TC + (2 * ch + j) * 16 + m16] = acc[tap][j][i];
}

// ------------------------------------------------------------------------------------
// The same weight gradient on the f16 MFMA pipe (f16x3, as k_conv_t): K = 32 rows per v_mfma_f32_16x16x32_f16 step, both
// operands (hi, lo) pairs of halves scaled by the tensors' own maxima.  The MFMA wants 8 consecutive K values (rows) of one
// channel per lane while the images are row-major [row][channel]: gfx950's transposing LDS read (ds_read_b64_tr_b16) hands
// every lane column i of four rows whose addresses the lanes supply, so rows need not even be contiguous -- K slot 4h + q of
// lane group g is row 16h + 4g + q of the step for BOTH operands (any bijection is a valid K order).
// Images: rows of [64 hi | 64 lo | 16 pad] halves (288 B: eight consecutive rows cover the 64 banks once).  dY is stored with
// one zero pad column per line (K runs over H x (W+1) cells per sample: a K step is whole lines), A zero-padded as in
// k_wgrad; tabA[row] = byte offset of a dY row's 3x3 window in the A image.  Wave w owns cin tile w & 3 and cout tiles
// 2 (w >> 2), +1 for all 9 taps; A fragments run two taps ahead in a 3-deep ring.
// ------------------------------------------------------------------------------------
#define WH_SB 288 // bytes per image row
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
union FragH { s4v s[2]; f16x8 h; };

struct WhGeo { int RK, RA, PW, G, NK; };
__host__ __device__ inline WhGeo wh_geo(int Sw, int H, int W)
{
    WhGeo g;
    g.PW = W + 1;
    g.G = g.PW + 1;
    g.NK = (Sw * H * g.PW + 31) / 32;              // K steps per chunk
    g.RK = g.NK * 32;                              // rows of the dY image
    g.RA = 2 * g.G + Sw * (H + 1) * g.PW;          // rows of the padded A image (the last window ends at row Sw*(H+1)*PW + G)
    return g;
}
static size_t wh_lds_bytes(int Sw, int H, int W)
{
    const WhGeo g = wh_geo(Sw, H, W);
    return (size_t)(g.RA + g.RK) * WH_SB + (size_t)g.RK * 4 + (size_t)Sw * H * W * 4;
}

typedef s4v __attribute__((address_space(3))) *lds_s4v_ptr;
// transposing read at a 32-bit LDS byte address
__device__ __forceinline__ s4v tr_read(unsigned a) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(size_t)a); }

template <int PWC> // W + 1 at compile time (tap offsets become immediates of the reads), 0 = any board
__global__ void __launch_bounds__(TT, 2) k_wgrad_h3(const float *__restrict__ act, const float *__restrict__ dy, const unsigned *act_max,
                                                    const unsigned *dy_max, int n, int Sw, int H, int W, float *__restrict__ part
#ifdef DBAZ_STAMP
                                                    , unsigned long long *stamp_out
#endif
)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef DBAZ_STAMP
    unsigned long long ws0, ws1, wsa, wsb, wsc, ws_commit = 0, ws_loop = 0, wr0, wr1;
    TSTAMP_RT(wr0);
    TSTAMP(ws0);
#endif
    const int HW = H * W;
    const WhGeo geo = wh_geo(Sw, H, W);
    const int PW = PWC ? PWC : geo.PW, NK = geo.NK;
    char *Ai = reinterpret_cast<char *>(lds);
    char *Di = Ai + (size_t)geo.RA * WH_SB;
    int *tabA = reinterpret_cast<int *>(Di + (size_t)geo.RK * WH_SB);   // [RK] window start of dY row j in the A image (bytes)
    int *rowmap = tabA + geo.RK;                                        // [Sw*HW] dY row | A row << 16 of a real position
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cit = wave & 3, ch = wave >> 2;
    const int m16 = lane & 15, gq = lane >> 4;
    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; t++) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const float sA = scale_from_max(*act_max), sD = scale_from_max(*dy_max);
    const int nchunks = (n + Sw - 1) / Sw;
    f32x4 pf[WG_MAXLD];
    const int img4 = Sw * HW * 16; // float4 per tensor and chunk
    // (tz: an opaque zero -- without it the compiler keeps the 13 global offsets and 13 LDS destinations of the staging, which
    // do not depend on the chunk, in 39 registers across the MFMA loop and parks the prefetched rows in scratch instead)
    auto issue = [&](int chunk, int tz) {
        const int s0 = chunk * Sw, R = min(Sw, n - s0) * HW;
        const f32x4 *a4 = reinterpret_cast<const f32x4 *>(act) + (size_t)s0 * HW * 16;
        const f32x4 *d4 = reinterpret_cast<const f32x4 *>(dy) + (size_t)s0 * HW * 16;
#pragma unroll
        for (int j = 0; j < WG_MAXLD; j++) {
            const int i = tid + tz + j * TT;
            const int which = i >= img4, ii = i - which * img4;
            pf[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (i < 2 * img4 && (ii >> 4) < R) pf[j] = which ? d4[ii] : a4[ii];
        }
    };
    auto commit = [&](int tz) {
#pragma unroll
        for (int j = 0; j < WG_MAXLD; j++) {
            const int i = tid + tz + j * TT;
            const int which = i >= img4, ii = i - which * img4;
            if (i < 2 * img4) {
                const int rm = rowmap[ii >> 4], c4 = ii & 15;
                const f32x4 v = pf[j] * (which ? sD : sA);
                union { h2v h[2]; u32x2 u; } oh, ol;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const f2v x = {v[2 * q], v[2 * q + 1]};
                    const h2v h = __builtin_convertvector(x, h2v);
                    oh.h[q] = h;
                    ol.h[q] = __builtin_convertvector(x - __builtin_convertvector(h, f2v), h2v);
                }
                char *dst = (which ? Di + (size_t)(rm & 0xffff) * WH_SB : Ai + (size_t)(rm >> 16) * WH_SB) + c4 * 8;
                *reinterpret_cast<u32x2 *>(dst) = oh.u;
                *reinterpret_cast<u32x2 *>(dst + 128) = ol.u;
            }
        }
    };
    // the first chunk's rows are requested BEFORE the images are zeroed and the tables built (2 us of HBM latency that used to
    // follow them: the prologue was 9 % of the workgroup)
    if ((int)blockIdx.x < nchunks) issue(blockIdx.x, 0);
    for (int i = tid; i < (geo.RA + geo.RK) * (WH_SB / 16); i += TT) reinterpret_cast<f32x4 *>(Ai)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int per_s = H * PW; // dY rows per sample
    for (int j = tid; j < geo.RK; j += TT) tabA[j] = j < Sw * per_s ? (j + (j / per_s) * PW) * WH_SB : 0;
    for (int r = tid; r < Sw * HW; r += TT) {
        const int sidx = r / HW, pos = r - sidx * HW, y = pos / W, x = pos - y * W;
        const int j = sidx * per_s + y * PW + x;
        rowmap[r] = j | ((geo.G + sidx * (H + 1) * PW + y * PW + x) << 16);
    }
    __syncthreads(); // tables, zeroed images
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; t++) toff[t] = ((t / 3) * PW + (t % 3)) * WH_SB;
    // this lane's part of a transposed read: row 4 gq + (lane >> 2 & 3) (+ 16 for the second half of the K slots), columns 4 (lane & 3) ..
    const int lrow = 4 * gq + ((lane >> 2) & 3), lcol = (lane & 3) * 8;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)Ai; // 32-bit LDS addresses from here on
    const unsigned abase = lds0 + cit * 32 + lcol;
    const unsigned dbase = lds0 + (unsigned)geo.RA * WH_SB + lrow * WH_SB + ch * 64 + lcol;
    TSTAMP(ws1);
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        int tz = 0;
        asm volatile("" : "+v"(tz));
        TSTAMP(wsa);
        __syncthreads(); // the previous chunk's reads are done
        commit(tz);
        __syncthreads();
        TSTAMP(wsb);
        asm volatile("" : "+v"(tz));
        if (chunk + (int)gridDim.x < nchunks) issue(chunk + gridDim.x, tz);
        FragH ah[3], al[3];        // A fragments of three consecutive taps
        FragH bh[2], bl[2];        // dY fragments (two cout tiles) of the K step
        int tA0 = tabA[lrow], tA1 = tabA[lrow + 16], nA0 = 0, nA1 = 0;
        auto loadA = [&](int slot, int t0, int t1, int t) {
            const unsigned p0 = abase + t0 + toff[t], p1 = abase + t1 + toff[t];
            ah[slot].s[0] = tr_read(p0); ah[slot].s[1] = tr_read(p1);
            al[slot].s[0] = tr_read(p0 + 128); al[slot].s[1] = tr_read(p1 + 128);
        };
        auto loadB = [&](FragH (&h)[2], FragH (&l)[2], int k) {
            const unsigned p = dbase + k * 32 * WH_SB;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                h[c].s[0] = tr_read(p + c * 32); h[c].s[1] = tr_read(p + c * 32 + 16 * WH_SB);
                l[c].s[0] = tr_read(p + c * 32 + 128); l[c].s[1] = tr_read(p + c * 32 + 16 * WH_SB + 128);
            }
        };
        loadA(0, tA0, tA1, 0);
        loadA(1, tA0, tA1, 1);
        for (int k = 0; k < NK; k++) {
            const bool more = k + 1 < NK;
            // (the dY fragments are single-buffered -- registers: the 52 of the HBM prefetch must stay out of scratch -- and
            // their read latency shows once per 54 MFMAs, where the SIMD's other wave covers it)
            loadB(bh, bl, k);
            if (more) { nA0 = tabA[(k + 1) * 32 + lrow]; nA1 = tabA[(k + 1) * 32 + lrow + 16]; }
#pragma unroll
            for (int t = 0; t < 9; t++) {
                if (t + 2 < 9) loadA((t + 2) % 3, tA0, tA1, t + 2);
                else if (more) loadA((t + 2) % 3, nA0, nA1, t + 2 - 9);
                __builtin_amdgcn_sched_barrier(0);
                const int sl = t % 3;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[sl].h, bh[c].h, acc[t][c], 0, 0, 0);
                    acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[sl].h, bh[c].h, acc[t][c], 0, 0, 0);
                    acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[sl].h, bl[c].h, acc[t][c], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            tA0 = nA0; tA1 = nA1;
        }
#ifdef DBAZ_STAMP
        TSTAMP(wsc);
        ws_commit += wsb - wsa;
        ws_loop += wsc - wsb;
#endif
    }
#ifdef DBAZ_STAMP
    TSTAMP(wsa);
#endif
    // lane holds dW[tap][cin = cit*16 + 4 gq + i][cout = (2 ch + j)*16 + m16], scaled by sA * sD
    const float inv = 1.0f / (sA * sD);
    float *o = part + (size_t)blockIdx.x * 9 * TC * TC;
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                o[((size_t)tap * TC + cit * 16 + 4 * gq + i) * TC + (2 * ch + j) * 16 + m16] = acc[tap][j][i] * inv;
#ifdef DBAZ_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TSTAMP(wsb);
    TSTAMP_RT(wr1);
    if (stamp_out && lane == 0) {
        unsigned long long *so = stamp_out + ((size_t)blockIdx.x * 8 + wave) * 8;
        so[0] = ws1 - ws0; so[1] = ws_commit; so[2] = ws_loop; so[3] = wsb - wsa; so[4] = wsb - ws0; so[5] = wr0; so[6] = wr1; so[7] = 0;
    }
#endif
}

// sums the workgroups' partial gradients (f64) and writes torch's [cout][cin][3][3]: 64 outputs x 4 partial lanes per block
struct BnFinArgs { // the BatchNorm-backward totals of the NEXT layer down, finished by this launch's last C workgroups (or nparts = 0)
    const double *part; int nparts; double *sums; float *dbeta, *dgamma; unsigned *zero;
};
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float *__restrict__ part, int nparts, float *__restrict__ dw,
                                                      const double *__restrict__ bias_part, int bias_nparts, float *__restrict__ dbias,
                                                      BnFinArgs fin)
{
    __shared__ double red[4][64];
    const int o = threadIdx.x & 63, j = threadIdx.x >> 6;
    if (blockIdx.x >= 9 * TC * TC / 64 + TC) {
        // sum(g), sum(g * yhat) of the layer below (partial rows left by the input-gradient conv that ran before this launch) ->
        // sums for k_bn_bwd_apply, dbeta, dgamma; clears the max|dY| word (a launch of its own before: 5 us per layer)
        const int c = blockIdx.x - (9 * TC * TC / 64 + TC);
        double v0 = 0.0, v1 = 0.0;
        for (int bb = threadIdx.x; bb < fin.nparts; bb += 256) {
            v0 += fin.part[((size_t)bb * 2) * TC + c];
            v1 += fin.part[((size_t)bb * 2 + 1) * TC + c];
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) { v0 += __shfl_xor(v0, sh); v1 += __shfl_xor(v1, sh); }
        if (o == 0) { red[j][0] = v0; red[j][1] = v1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double t0 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]), t1 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
            fin.sums[c] = t0;
            fin.sums[TC + c] = t1;
            fin.dbeta[c] = (float)t0;
            fin.dgamma[c] = (float)t1;
            if (c == 0) *fin.zero = 0u;
        }
        return;
    }
    if (blockIdx.x >= 9 * TC * TC / 64) {
        // C workgroups: the conv bias gradient of channel c = the total of k_bn_bwd_apply's partial sums of dY (a launch
        // of its own before: 4.8 us per layer)
        const int c = blockIdx.x - 9 * TC * TC / 64;
        double v = 0.0;
        for (int bb = threadIdx.x; bb < bias_nparts; bb += 256) v += bias_part[(size_t)bb * TC + c];
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_xor(v, sh);
        if (o == 0) red[j][0] = v;
        __syncthreads();
        if (threadIdx.x == 0) dbias[c] = (float)((red[0][0] + red[1][0]) + (red[2][0] + red[3][0]));
        return;
    }
    const int i = blockIdx.x * 64 + o; // (tap * C + cin) * C + cout
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = j;
    // sixteen partials requested before the first is added (four in flight left the launch latency-bound: 10.8 us for 38 MB)
    for (; b + 60 < nparts; b += 64) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = part[(size_t)(b + 4 * u) * 9 * TC * TC + i];
#pragma unroll
        for (int u = 0; u < 16; u += 4) { s0 += (double)v[u]; s1 += (double)v[u + 1]; s2 += (double)v[u + 2]; s3 += (double)v[u + 3]; }
    }
    for (; b + 12 < nparts; b += 16) {
        s0 += (double)part[(size_t)b * 9 * TC * TC + i];
        s1 += (double)part[(size_t)(b + 4) * 9 * TC * TC + i];
        s2 += (double)part[(size_t)(b + 8) * 9 * TC * TC + i];
        s3 += (double)part[(size_t)(b + 12) * 9 * TC * TC + i];
    }
    for (; b < nparts; b += 4) s0 += (double)part[(size_t)b * 9 * TC * TC + i];
    red[j][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (j == 0) {
        const double s = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
        const int co = i & 63, ci = (i >> 6) & 63, tap = i >> 12;
        dw[((size_t)co * TC + ci) * 9 + tap] = (float)s;
    }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------

extern "C" const char *dbaz_trainer_last_error(const dbaz_trainer *t) { return t ? t->err.c_str() : g_train_error.c_str(); }

extern "C" void dbaz_trainer_destroy(dbaz_trainer *t)
{
    if (!t) return;
    (void)hipSetDevice(t->dev);
    net_free(t);
    void *ptrs[] = {t->A, t->Y, t->G, t->dA[0], t->dA[1], t->dY, t->wpk, t->wsc, t->amax, t->mean, t->invstd, t->part, t->part_bs, t->sums, t->wg_part, t->relu_mask};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete t;
}

extern "C" int dbaz_trainer_create(int32_t rows, int32_t cols, int32_t channels, int32_t blocks, int32_t max_batch, int32_t device,
                                   dbaz_trainer **out)
{
    if (!out) return terr(nullptr, DBAZ_EINVAL, "null argument");
    *out = nullptr;
    if (channels != TC) return terr(nullptr, DBAZ_EINVAL, "the training tower is built for %d channels (got %d)", TC, channels);
    if (rows < 1 || cols < 1 || (rows + 1) * (cols + 1) > 196) return terr(nullptr, DBAZ_EINVAL, "board %dx%d unsupported", rows, cols);
    if (blocks < 1 || 2 * blocks > TL_MAX) return terr(nullptr, DBAZ_EINVAL, "blocks must be in 1..%d", TL_MAX / 2);
    if (max_batch < 1) return terr(nullptr, DBAZ_EINVAL, "max_batch must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return terr(nullptr, DBAZ_EDEVICE, "no HIP device %d", device);
    dbaz_trainer *t = new dbaz_trainer();
    t->dev = device; t->H = rows + 1; t->W = cols + 1; t->HW = t->H * t->W; t->L = 2 * blocks; t->maxN = max_batch;
    if (hipSetDevice(device) != hipSuccess) { delete t; return terr(nullptr, DBAZ_EDEVICE, "hipSetDevice(%d) failed", device); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) t->cus = prop.multiProcessorCount;
    t->S = 256 / t->HW;
    t->Sw = 1; // samples per k_wgrad chunk: as many as the prefetch registers (208 rows) and 150 KB of LDS hold
    while ((t->Sw + 1) * t->HW <= 208 && wg_lds_bytes(t->Sw + 1, t->H, t->W) <= 150 * 1024) t->Sw++;
    {
        const int S4 = (TC + 8) / 4;
        const int zu = (t->S * t->HW * S4 + 15) & ~15;
        t->conv_lds = (size_t)(zu + 3 * S4) * 16 + (size_t)(TT / 64) * 2 * TC * 8 + 16; // image + zero rows + the epilogue's column-sum slots
        t->wgrad_lds = wg_lds_bytes(t->Sw, t->H, t->W);
        t->Swh = 1;
        while ((t->Swh + 1) * t->HW <= 208 && wh_lds_bytes(t->Swh + 1, t->H, t->W) <= 150 * 1024) t->Swh++;
        t->wgrad_h3_lds = wh_lds_bytes(t->Swh, t->H, t->W);
        t->wgrad_h3 = 1;
#ifdef DBAZ_DEBUG
        if (getenv("DBAZ_TRAIN_WGRAD_F32")) t->wgrad_h3 = 0; // debug build: the exact-f32 weight gradient kernel (A/B reference)
#endif
    }
    const size_t ae = act_elems(t);
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc((void **)&t->A, ae * (t->L + 1) * 4);
    alloc((void **)&t->Y, ae * t->L * 4);
    alloc((void **)&t->G, ae * 4);
    alloc((void **)&t->dA[0], ae * 4);
    alloc((void **)&t->dA[1], ae * 4);
    alloc((void **)&t->dY, ae * 4);
    alloc((void **)&t->wpk, (size_t)2 * t->L * TC * TC * 9 * 2 * sizeof(_Float16));
    alloc((void **)&t->wsc, (size_t)2 * t->L * 4);
    alloc((void **)&t->amax, (size_t)(t->L + 2) * 4);
    alloc((void **)&t->mean, (size_t)t->L * TC * 4);
    alloc((void **)&t->invstd, (size_t)t->L * TC * 4);
    alloc((void **)&t->part, std::max((size_t)RED_BLOCKS * 4, (size_t)(t->maxN / t->S + 1) * 2) * TC * 8);
    alloc((void **)&t->part_bs, std::max((size_t)RED_BLOCKS, (size_t)(t->maxN / t->S + 1)) * 2 * TC * 8);
    alloc((void **)&t->sums, (size_t)4 * TC * 8);
    alloc((void **)&t->wg_part, (size_t)t->cus * 9 * TC * TC * 4);
    alloc((void **)&t->relu_mask, (size_t)t->L * t->maxN * t->HW * 8);
#ifdef DBAZ_STAMP
    alloc((void **)&t->stamps, (size_t)(t->maxN / t->S + 1) * 64 * 8);
    alloc((void **)&t->stamps_wg, (size_t)(t->cus + 1) * 64 * 8);
#endif
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_conv_t, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->conv_lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->wgrad_lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_wgrad_h3<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->wgrad_h3_lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_wgrad_h3<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->wgrad_h3_lds);
    if (e != hipSuccess) {
        const std::string msg = hipGetErrorString(e);
        dbaz_trainer_destroy(t);
        return terr(nullptr, DBAZ_EDEVICE, "trainer allocation failed: %s", msg.c_str());
    }
    *out = t;
    return DBAZ_OK;
}

#ifdef DBAZ_STAMP
extern "C" int dbaz_debug_trainer_wgrad_stamps(dbaz_trainer *t, unsigned long long *out, int n_wg)
{
    if (!t || !out) return DBAZ_EINVAL;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out, t->stamps_wg, (size_t)std::min(n_wg, t->cus) * 64 * 8, hipMemcpyDeviceToHost) == hipSuccess ? DBAZ_OK : DBAZ_ESTATE;
}
extern "C" int dbaz_debug_trainer_stamps(dbaz_trainer *t, unsigned long long *out, int n_wg)
{
    if (!t || !out) return DBAZ_EINVAL;
    (void)hipDeviceSynchronize();
    const int have = t->maxN / t->S + 1;
    return hipMemcpy(out, t->stamps, (size_t)std::min(n_wg, have) * 64 * 8, hipMemcpyDeviceToHost) == hipSuccess ? DBAZ_OK : DBAZ_ESTATE;
}
#endif

// k_bn_apply: workgroups of 256 threads x 4 quads per pass; as few passes as 1 024 workgroups allow, and a grid that
// divides the tensor into WHOLE passes (1 024 workgroups left 3.06 passes at batch 4 096: a fourth latency round for 6 % of the rows)
static int bn_apply_grid(long long n4)
{
    const long long per = 256 * 4;
    const long long passes = std::max<long long>(1, (n4 + 1024 * per - 1) / (1024 * per));
    return (int)std::max<long long>(1, (n4 + per * passes - 1) / (per * passes));
}


void tower_forward_rows(dbaz_trainer *t, int n, const float *const *conv_w, const float *const *conv_b, const float *const *bn_w,
                        const float *const *bn_b, float *const *run_mean, float *const *run_var, hipStream_t s)
{
    const int L = t->L, HW = t->HW;
    const size_t ae = act_elems(t);
    const long long M = (long long)n * HW;
    PackArgs pa;
    for (int l = 0; l < L; l++) pa.w[l] = conv_w[l];
    hipLaunchKernelGGL(k_pack_w, dim3(L, 2, 4), dim3(TT), 0, s, pa, t->wpk, t->wsc, L);
    const int grid = (n + t->S - 1) / t->S;
    const long long n4 = M * 16;
    const int ab = bn_apply_grid(n4);
    for (int l = 0; l < L; l++) {
        ConvArgs ca = {};
        ca.in = t->A + ae * l; ca.in_max = t->amax + l;
        ca.wpk = t->wpk + (size_t)l * TC * TC * 9 * 2; ca.wsc = t->wsc + l;
        ca.bias = conv_b[l]; ca.add = nullptr; ca.out = t->Y + ae * l; ca.stat_part = t->part;
        ca.n = n; ca.S = t->S; ca.H = t->H; ca.W = t->W;
#ifdef DBAZ_STAMP
        ca.stamp_out = t->stamps;
#endif
        hipLaunchKernelGGL(k_conv_t, dim3(grid), dim3(TT), t->conv_lds, s, ca);
        hipLaunchKernelGGL(k_bn_stats_fin, dim3(FIN_BLOCKS), dim3(TT), 0, s, t->part, grid, M, t->eps, t->momentum, t->mean + l * TC,
                           t->invstd + l * TC, run_mean ? run_mean[l] : nullptr, run_var ? run_var[l] : nullptr);
        hipLaunchKernelGGL(k_bn_apply, dim3(ab), dim3(256), 0, s, reinterpret_cast<const f32x4 *>(t->Y + ae * l),
                           (l & 1) ? reinterpret_cast<const f32x4 *>(t->A + ae * (l - 1)) : nullptr,
                           reinterpret_cast<f32x4 *>(t->A + ae * (l + 1)), n4, t->mean + l * TC, t->invstd + l * TC, bn_w[l], bn_b[l],
                           t->amax + l + 1, t->relu_mask + (size_t)l * t->maxN * HW);
    }
}

int tower_backward_rows(dbaz_trainer *t, const float *const *bn_w, float *const *g_conv_w, float *const *g_conv_b, float *const *g_bn_w,
                               float *const *g_bn_b, const BelowTower &below, hipStream_t s)
{
    const int L = t->L, HW = t->HW, n = t->n;
    const size_t ae = act_elems(t);
    const long long M = (long long)n * HW;
    const int grid = (n + t->S - 1) / t->S;
    const int rb = red_blocks(M);
    const int Sw = t->wgrad_h3 ? t->Swh : t->Sw;
    const int nchunks = (n + Sw - 1) / Sw;
    const int wg = std::min(t->cus, nchunks);
    unsigned *dymax = t->amax + L + 1;
    int cur = 0;
    for (int l = L - 1; l >= 0; l--) {
        const f32x4 *dA4 = reinterpret_cast<const f32x4 *>(t->dA[cur]);
        const unsigned long long *ao4 = t->relu_mask + (size_t)l * t->maxN * HW; // sign bits of A[l + 1]
        const f32x4 *y4 = reinterpret_cast<const f32x4 *>(t->Y + ae * l);
        // sum(g), sum(g * yhat) -> t->sums, dbeta, dgamma: partial rows left in t->part_bs by the conv that produced dA (the layer
        // above's input-gradient conv) and totalled by that layer's k_wgrad_reduce launch; the top layer's dA comes from the caller
        if (l == L - 1) {
            hipLaunchKernelGGL(k_bn_bwd_sums, dim3(rb), dim3(TT), 0, s, dA4, ao4, y4, M, t->mean + l * TC, t->invstd + l * TC, t->part_bs);
            hipLaunchKernelGGL(k_bn_bwd_sums_fin, dim3(FIN_BLOCKS), dim3(TT), 0, s, t->part_bs, rb, t->sums, g_bn_b[l], g_bn_w[l], dymax);
        }
        hipLaunchKernelGGL(k_bn_bwd_apply, dim3(rb), dim3(TT), 0, s, dA4, ao4, y4, M, t->mean + l * TC, t->invstd + l * TC, bn_w[l],
                           t->sums, reinterpret_cast<f32x4 *>(t->dY), (l & 1) ? reinterpret_cast<f32x4 *>(t->G) : (f32x4 *)nullptr,
                           dymax, t->part);
        ConvArgs ca = {};
        ca.in = t->dY; ca.in_max = dymax;
        ca.wpk = t->wpk + ((size_t)L + l) * TC * TC * 9 * 2; ca.wsc = t->wsc + L + l;
        ca.bias = nullptr; ca.out = t->dA[1 - cur]; ca.stat_part = nullptr;
        BnFinArgs fin = {};
        if (l > 0) {
            ca.bs_part = t->part_bs; ca.bs_mask = t->relu_mask + (size_t)(l - 1) * t->maxN * HW; ca.bs_y = t->Y + ae * (l - 1);
            ca.bs_mean = t->mean + (l - 1) * TC; ca.bs_invstd = t->invstd + (l - 1) * TC;
            fin.part = t->part_bs; fin.nparts = grid; fin.sums = t->sums; fin.dbeta = g_bn_b[l - 1]; fin.dgamma = g_bn_w[l - 1]; fin.zero = dymax;
        } else if (below.y) {
            ca.bs_part = t->part_bs; ca.bs_mask = below.mask; ca.bs_y = below.y; ca.bs_mean = below.mean; ca.bs_invstd = below.invstd;
            fin.part = t->part_bs; fin.nparts = grid; fin.sums = t->sums; fin.dbeta = below.g_b; fin.dgamma = below.g_w; fin.zero = dymax;
        }
        ca.add = (l & 1) ? nullptr : t->G; // the input of a block's first conv is also the block's skip input: + g of its end
        ca.n = n; ca.S = t->S; ca.H = t->H; ca.W = t->W;
#ifdef DBAZ_STAMP
        ca.stamp_out = t->stamps;
#endif
        hipLaunchKernelGGL(k_conv_t, dim3(grid), dim3(TT), t->conv_lds, s, ca);
        if (t->wgrad_h3 && t->W == 7)
            hipLaunchKernelGGL((k_wgrad_h3<8>), dim3(wg), dim3(TT), t->wgrad_h3_lds, s, t->A + ae * l, t->dY, t->amax + l, dymax, n, Sw, t->H,
                               t->W, t->wg_part
#ifdef DBAZ_STAMP
                               , t->stamps_wg
#endif
                               );
        else if (t->wgrad_h3)
            hipLaunchKernelGGL((k_wgrad_h3<0>), dim3(wg), dim3(TT), t->wgrad_h3_lds, s, t->A + ae * l, t->dY, t->amax + l, dymax, n, Sw, t->H,
                               t->W, t->wg_part
#ifdef DBAZ_STAMP
                               , t->stamps_wg
#endif
                               );
        else
            hipLaunchKernelGGL(k_wgrad, dim3(wg), dim3(TT), t->wgrad_lds, s, t->A + ae * l, t->dY, n, Sw, t->H, t->W, t->wg_part);
        // weight gradient totals + conv bias gradient + (last C workgroups) the BatchNorm-backward totals of the layer below
        hipLaunchKernelGGL(k_wgrad_reduce, dim3(9 * TC * TC / 64 + TC + (fin.nparts ? TC : 0)), dim3(256), 0, s, t->wg_part, wg, g_conv_w[l], t->part,
                           rb, g_conv_b[l], fin);
        cur = 1 - cur;
    }
    return cur;
}

// Training-mode forward of the tower.  x / out: DEVICE float32 [n][64][H][W] (torch NCHW); conv_w[l] [64][64][3][3],
// conv_b[l], bn_w[l], bn_b[l], run_mean[l], run_var[l] [64]: DEVICE pointers of layer l = 2*block + (0: conv1/bn1, 1:
// conv2/bn2); the running statistics are updated in place.  Asynchronous on `stream`.
extern "C" int dbaz_trainer_forward(dbaz_trainer *t, int32_t n, const float *x, const float *const *conv_w, const float *const *conv_b,
                                    const float *const *bn_w, const float *const *bn_b, float *const *run_mean, float *const *run_var,
                                    float *out, void *stream)
{
    if (!t) return DBAZ_EINVAL;
    if (n < 1 || n > t->maxN) return terr(t, DBAZ_EINVAL, "batch %d outside 1..%d", n, t->maxN);
    if (!x || !out || !conv_w || !conv_b || !bn_w || !bn_b) return terr(t, DBAZ_EINVAL, "null argument");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(t, hipSetDevice(t->dev));
    const int L = t->L, HW = t->HW;
    t->have_fwd = false;
    t->net_fwd = false;
    HIPCHK(t, hipMemsetAsync(t->amax, 0, (size_t)(L + 2) * 4, s));
    hipLaunchKernelGGL(k_nchw_to_rows, dim3(n), dim3(256), (size_t)TC * (HW + 1) * 4, s, x, t->A, HW, t->amax);
    tower_forward_rows(t, n, conv_w, conv_b, bn_w, bn_b, run_mean, run_var, s);
    hipLaunchKernelGGL(k_rows_to_nchw, dim3(n), dim3(256), (size_t)HW * (TC + 1) * 4, s, t->A + act_elems(t) * L, out, HW);
    HIPCHK(t, hipGetLastError());
    t->n = n;
    t->have_fwd = true;
    return DBAZ_OK;
}

// Backward of the forward pass still held by the handle.  grad_out / grad_x: DEVICE [n][64][H][W]; the parameter gradients
// are written (not accumulated) to g_conv_w[l] [64][64][3][3], g_conv_b[l], g_bn_w[l], g_bn_b[l] [64].  conv_w / bn_w: the
// same pointers as in the forward call.  Asynchronous on `stream`.
extern "C" int dbaz_trainer_backward(dbaz_trainer *t, const float *grad_out, const float *const *bn_w, float *grad_x,
                                     float *const *g_conv_w, float *const *g_conv_b, float *const *g_bn_w, float *const *g_bn_b,
                                     void *stream)
{
    if (!t) return DBAZ_EINVAL;
    if (!t->have_fwd || t->net_fwd) return terr(t, DBAZ_ESTATE, "dbaz_trainer_backward without a dbaz_trainer_forward pass");
    if (!grad_out || !grad_x || !bn_w || !g_conv_w || !g_conv_b || !g_bn_w || !g_bn_b) return terr(t, DBAZ_EINVAL, "null argument");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(t, hipSetDevice(t->dev));
    const int HW = t->HW, n = t->n;
    hipLaunchKernelGGL(k_nchw_to_rows, dim3(n), dim3(256), (size_t)TC * (HW + 1) * 4, s, grad_out, t->dA[0], HW, (unsigned *)nullptr);
    const int cur = tower_backward_rows(t, bn_w, g_conv_w, g_conv_b, g_bn_w, g_bn_b, BelowTower(), s);
    hipLaunchKernelGGL(k_rows_to_nchw, dim3(n), dim3(256), (size_t)HW * (TC + 1) * 4, s, t->dA[cur], grad_x, HW);
    HIPCHK(t, hipGetLastError());
    t->have_fwd = false;
    return DBAZ_OK;
}

// ------------------------------------------------------------------------------------
// Training-mode BatchNorm2d (+ optional ReLU) on torch's NCHW tensors with any channel count: bn_input (3 channels), bn0 (64)
// and the heads' bn0 (16) of ResNetZero (nn.py:19-21,81-83,98-100,114).  MIOpen's kernels for these shapes take 114 us forward
// and 157 us backward per layer (one workgroup per channel); these are 4 launches of many small workgroups.
// Element (n, c, p) lives at (n * C + c) * HW + p; a workgroup of 64 x 8 threads handles channel blockIdx.x and the samples
// n = blockIdx.y * 8 + threadIdx.y (+ gridDim.y * 8 ...), lane x = position p (+ 64 ...).  Partial sums in f64: part[c][blockIdx.y][K].
// ------------------------------------------------------------------------------------

template <int K>
__device__ __forceinline__ void bn2d_block_store(double (&s)[K], double *part, int C)
{
    __shared__ double red[8][K];
    const int lane = threadIdx.x, w = threadIdx.y;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double v = s[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[w][k] = v;
    }
    __syncthreads();
    if (w == 0 && lane < K) {
        double v = 0.0;
        for (int i = 0; i < 8; i++) v += red[i][lane];
        part[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * K + lane] = v;
    }
}

__global__ void __launch_bounds__(512) k_bn2d_stats(const float *__restrict__ x, int n, int C, int HW, double *part)
{
    const int c = blockIdx.x;
    double s[2] = {0.0, 0.0};
    for (int i = blockIdx.y * 8 + threadIdx.y; i < n; i += gridDim.y * 8)
        for (int p = threadIdx.x; p < HW; p += 64) {
            const double v = x[((size_t)i * C + c) * HW + p];
            s[0] += v;
            s[1] += v * v;
        }
    bn2d_block_store<2>(s, part, C);
}

// one workgroup of 64 threads per channel: totals of the K partial sums -> tot[k] (all lanes)
template <int K>
__device__ __forceinline__ void bn2d_total(const double *part, int nb, double (&tot)[K])
{
    const int c = blockIdx.x, lane = threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double v = 0.0;
        for (int b = lane; b < nb; b += 64) v += part[((size_t)c * nb + b) * K + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        tot[k] = v;
    }
}

__global__ void __launch_bounds__(64) k_bn2d_stats_fin(const double *part, int nb, long long M, float eps, float momentum, float *mean,
                                                       float *invstd, float *run_mean, float *run_var)
{
    double tot[2];
    bn2d_total<2>(part, nb, tot);
    if (threadIdx.x == 0) {
        const int c = blockIdx.x;
        const double m = tot[0] / (double)M;
        double var = tot[1] / (double)M - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean) run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + (double)momentum * m);
        if (run_var) {
            const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
            run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + (double)momentum * unb);
        }
    }
}

// out = gamma * (x - mean) * invstd + beta, ReLU if asked; rows r = n * C + c of HW floats, 8 rows per workgroup pass
__global__ void __launch_bounds__(512) k_bn2d_apply(const float *__restrict__ x, float *__restrict__ out, long long rows, int C, int HW,
                                                    const float *mean, const float *invstd, const float *gamma, const float *beta, int relu)
{
    for (long long r = (long long)blockIdx.x * 8 + threadIdx.y; r < rows; r += (long long)gridDim.x * 8) {
        const int c = (int)(r % C);
        const float mu = mean[c], sc = invstd[c] * gamma[c], be = beta[c];
        for (int p = threadIdx.x; p < HW; p += 64) {
            float v = (x[r * HW + p] - mu) * sc + be;
            if (relu) v = fmaxf(v, 0.0f);
            out[r * HW + p] = v;
        }
    }
}

// backward sums: g = dout * (out > 0 if relu); sum(g), sum(g * xhat)
__global__ void __launch_bounds__(512) k_bn2d_bwd_sums(const float *__restrict__ dout, const float *__restrict__ out, const float *__restrict__ x,
                                                       int n, int C, int HW, const float *mean, const float *invstd, int relu, double *part)
{
    const int c = blockIdx.x;
    double s[2] = {0.0, 0.0};
    const float mu = mean[c], is = invstd[c];
    for (int i = blockIdx.y * 8 + threadIdx.y; i < n; i += gridDim.y * 8)
        for (int p = threadIdx.x; p < HW; p += 64) {
            const size_t o = ((size_t)i * C + c) * HW + p;
            float g = dout[o];
            if (relu && !(out[o] > 0.0f)) g = 0.0f;
            s[0] += (double)g;
            s[1] += (double)g * (double)((x[o] - mu) * is);
        }
    bn2d_block_store<2>(s, part, C);
}

__global__ void __launch_bounds__(64) k_bn2d_bwd_fin(const double *part, int nb, double *sums /*[C][2]*/, float *dgamma, float *dbeta)
{
    double tot[2];
    bn2d_total<2>(part, nb, tot);
    if (threadIdx.x == 0) {
        const int c = blockIdx.x;
        sums[c * 2] = tot[0];
        sums[c * 2 + 1] = tot[1];
        dbeta[c] = (float)tot[0];
        dgamma[c] = (float)tot[1];
    }
}

__global__ void __launch_bounds__(512) k_bn2d_bwd_apply(const float *__restrict__ dout, const float *__restrict__ out, const float *__restrict__ x,
                                                        float *__restrict__ dx, long long rows, long long M, int C, int HW, const float *mean,
                                                        const float *invstd, const float *gamma, const double *sums, int relu)
{
    for (long long r = (long long)blockIdx.x * 8 + threadIdx.y; r < rows; r += (long long)gridDim.x * 8) {
        const int c = (int)(r % C);
        const float mu = mean[c], is = invstd[c], gi = gamma[c] * is;
        const float mg = (float)(sums[c * 2] / (double)M), mgy = (float)(sums[c * 2 + 1] / (double)M);
        for (int p = threadIdx.x; p < HW; p += 64) {
            float g = dout[r * HW + p];
            if (relu && !(out[r * HW + p] > 0.0f)) g = 0.0f;
            const float xh = (x[r * HW + p] - mu) * is;
            dx[r * HW + p] = gi * (g - mg - xh * mgy);
        }
    }
}

// workspace (caller-owned DEVICE memory, 8-byte aligned): C * BN_NB * 2 doubles of partials + C * 2 doubles of sums
extern "C" int64_t dbaz_bn2d_workspace_bytes(int32_t channels) { return (int64_t)channels * (BN_NB * 2 + 2) * 8; }

// Training-mode BatchNorm2d forward on x [n][C][H*W] (NCHW, contiguous): out = relu?(gamma * xhat + beta); saves the batch mean /
// invstd ([C] each) for the backward call, updates run_mean / run_var in place (NULL: skip).  Asynchronous on `stream`.
extern "C" int dbaz_bn2d_forward(const float *x, int32_t n, int32_t channels, int32_t hw, const float *gamma, const float *beta,
                                 float *run_mean, float *run_var, float eps, float momentum, int32_t relu, float *out, float *save_mean,
                                 float *save_invstd, void *workspace, void *stream)
{
    if (!x || !gamma || !beta || !out || !save_mean || !save_invstd || !workspace) return terr(nullptr, DBAZ_EINVAL, "null argument");
    if (n < 1 || channels < 1 || hw < 1) return terr(nullptr, DBAZ_EINVAL, "bn2d: n, channels and H*W must be >= 1");
    hipStream_t s = (hipStream_t)stream;
    double *part = reinterpret_cast<double *>(workspace);
    const int nb = std::min(BN_NB, (n + 7) / 8);
    const long long rows = (long long)n * channels, M = (long long)n * hw;
    hipLaunchKernelGGL(k_bn2d_stats, dim3(channels, nb), dim3(64, 8), 0, s, x, n, channels, hw, part);
    hipLaunchKernelGGL(k_bn2d_stats_fin, dim3(channels), dim3(64), 0, s, part, nb, M, eps, momentum, save_mean, save_invstd, run_mean, run_var);
    hipLaunchKernelGGL(k_bn2d_apply, dim3((unsigned)std::min<long long>((rows + 7) / 8, 4096)), dim3(64, 8), 0, s, x, out, rows, channels, hw,
                       save_mean, save_invstd, gamma, beta, relu);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DBAZ_OK : terr(nullptr, DBAZ_EDEVICE, "bn2d forward: %s", hipGetErrorString(e));
}

// backward: dout / out / x as in the forward call -> dx [n][C][H*W], dgamma, dbeta [C] (written)
extern "C" int dbaz_bn2d_backward(const float *dout, const float *out, const float *x, int32_t n, int32_t channels, int32_t hw,
                                  const float *gamma, const float *save_mean, const float *save_invstd, int32_t relu, float *dx,
                                  float *dgamma, float *dbeta, void *workspace, void *stream)
{
    if (!dout || !out || !x || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace)
        return terr(nullptr, DBAZ_EINVAL, "null argument");
    if (n < 1 || channels < 1 || hw < 1) return terr(nullptr, DBAZ_EINVAL, "bn2d: n, channels and H*W must be >= 1");
    hipStream_t s = (hipStream_t)stream;
    double *part = reinterpret_cast<double *>(workspace);
    double *sums = part + (size_t)channels * BN_NB * 2;
    const int nb = std::min(BN_NB, (n + 7) / 8);
    const long long rows = (long long)n * channels, M = (long long)n * hw;
    hipLaunchKernelGGL(k_bn2d_bwd_sums, dim3(channels, nb), dim3(64, 8), 0, s, dout, out, x, n, channels, hw, save_mean, save_invstd, relu, part);
    hipLaunchKernelGGL(k_bn2d_bwd_fin, dim3(channels), dim3(64), 0, s, part, nb, sums, dgamma, dbeta);
    hipLaunchKernelGGL(k_bn2d_bwd_apply, dim3((unsigned)std::min<long long>((rows + 7) / 8, 4096)), dim3(64, 8), 0, s, dout, out, x, dx, rows, M,
                       channels, hw, save_mean, save_invstd, gamma, sums, relu);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DBAZ_OK : terr(nullptr, DBAZ_EDEVICE, "bn2d backward: %s", hipGetErrorString(e));
}

// ---- launches for train_net.hip (declared in train.h)
void train_bn_forward_rows(dbaz_trainer *t, hipStream_t s, const double *part, int nparts, long long M, float *mean, float *invstd, float *run_mean,
                           float *run_var, const float *y, const float *res, float *out, const float *gamma, const float *beta, unsigned *amax,
                           unsigned long long *mask)
{
    hipLaunchKernelGGL(k_bn_stats_fin, dim3(FIN_BLOCKS), dim3(TT), 0, s, part, nparts, M, t->eps, t->momentum, mean, invstd, run_mean, run_var);
    const long long n4 = M * 16;
    hipLaunchKernelGGL(k_bn_apply, dim3(bn_apply_grid(n4)), dim3(256), 0, s, reinterpret_cast<const f32x4 *>(y), reinterpret_cast<const f32x4 *>(res),
                       reinterpret_cast<f32x4 *>(out), n4, mean, invstd, gamma, beta, amax, mask);
}

void train_bn_backward_apply_rows(dbaz_trainer *t, hipStream_t s, const float *dA, const unsigned long long *mask, const float *y, long long M,
                                  const float *mean, const float *invstd, const float *gamma, const double *sums, float *dY, unsigned *dymax,
                                  double *part)
{
    (void)t;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(red_blocks(M)), dim3(TT), 0, s, reinterpret_cast<const f32x4 *>(dA), mask, reinterpret_cast<const f32x4 *>(y),
                       M, mean, invstd, gamma, sums, reinterpret_cast<f32x4 *>(dY), (f32x4 *)nullptr, dymax, part);
}

void train_bn2d_statistics(dbaz_trainer *t, hipStream_t s, const float *x, int n, int C, int HW, double *ws, float *mean, float *invstd,
                           float *run_mean, float *run_var)
{
    const int nb = std::min(BN_NB, (n + 7) / 8);
    hipLaunchKernelGGL(k_bn2d_stats, dim3(C, nb), dim3(64, 8), 0, s, x, n, C, HW, ws);
    hipLaunchKernelGGL(k_bn2d_stats_fin, dim3(C), dim3(64), 0, s, ws, nb, (long long)n * HW, t->eps, t->momentum, mean, invstd, run_mean, run_var);
}

// ------------------------------------------------------------------------------------
// AlphaZeroLoss (nn.py:131-138) forward + backward and torch.optim.SGD's update (nn.py:179, 203-221: momentum, weight
// decay) as HIP kernels -- the parts of the optimizer step around the network that are pure elementwise / reduction work.
// Stateless C calls on DEVICE pointers, asynchronous on `stream`; errors: dbaz_trainer_last_error(NULL).
// ------------------------------------------------------------------------------------
#define LOSS_NB 128 // partial-sum workgroups of the loss

// loss_v = mean((z - v)^2), loss_pi = -mean_n(sum_a pi * logp); gradients of (loss_v + loss_pi) * gscale.
// One wave per sample row; partial sums in f64 per workgroup, added in workgroup order by k_az_loss_fin (deterministic).
__global__ void __launch_bounds__(256) k_az_loss(const float *__restrict__ logp, const float *__restrict__ v, const float *__restrict__ pi,
                                                 const float *__restrict__ z, int n, int A, float gscale, float *__restrict__ d_logp,
                                                 float *__restrict__ d_v, double *__restrict__ part)
{
    __shared__ double sh[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_n = 1.0f / (float)n;
    double s_pi = 0.0, s_v = 0.0;
    for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {
        const float *lp = logp + (size_t)row * A, *pr = pi + (size_t)row * A;
        float acc = 0.0f;
        for (int a = lane; a < A; a += 64) {
            const float p = pr[a];
            acc += p * lp[a];
            if (d_logp) d_logp[(size_t)row * A + a] = -p * inv_n * gscale;
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) {
            const float d = v[row] - z[row];
            s_pi += (double)acc;
            s_v += (double)(d * d);
            if (d_v) d_v[row] = 2.0f * d * inv_n * gscale;
        }
    }
    if (lane == 0) { sh[wave][0] = s_pi; sh[wave][1] = s_v; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]);
        part[blockIdx.x * 2 + 1] = (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]);
    }
}

__global__ void __launch_bounds__(64) k_az_loss_fin(const double *__restrict__ part, int nb, int n, float *__restrict__ loss3)
{
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) { a += part[i * 2]; b += part[i * 2 + 1]; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (threadIdx.x == 0) {
        const float lpi = (float)(-a / (double)n), lv = (float)(b / (double)n);
        loss3[0] = lv + lpi; // AlphaZeroLoss.forward: loss_v + loss_pi
        loss3[1] = lpi;
        loss3[2] = lv;
    }
}

extern "C" int64_t dbaz_az_loss_workspace_bytes(void) { return (int64_t)LOSS_NB * 2 * 8; }

extern "C" int dbaz_az_loss(const float *logp, const float *v, const float *pi, const float *z, int32_t n, int32_t n_actions, float grad_scale,
                            float *loss3, float *d_logp, float *d_v, void *workspace, void *stream)
{
    if (!logp || !v || !pi || !z || !loss3 || !workspace || n < 1 || n_actions < 1) return terr(nullptr, DBAZ_EINVAL, "dbaz_az_loss: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int nb = std::min(LOSS_NB, (n + 3) / 4);
    hipLaunchKernelGGL(k_az_loss, dim3(nb), dim3(256), 0, s, logp, v, pi, z, n, n_actions, grad_scale, d_logp, d_v, (double *)workspace);
    hipLaunchKernelGGL(k_az_loss_fin, dim3(1), dim3(64), 0, s, (const double *)workspace, nb, n, loss3);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DBAZ_OK : terr(nullptr, DBAZ_EDEVICE, "az loss: %s", hipGetErrorString(e));
}

// torch.optim.SGD (dampening 0, no nesterov): d = g + wd p; buf = momentum buf + d; p -= lr buf -- every parameter tensor of the
// model in ONE launch.  table[t] = {p, g, buf, numel}; chunk c of SGD_CHUNK elements belongs to tensor chunk_tensor[c] at
// element offset chunk_off[c].  The operation order (and its roundings) is torch's _foreach implementation's:
// g + wd*p as one fma, buf*momentum then + d, p + (-lr)*buf as one fma.
#define SGD_CHUNK 2048
struct SgdEntry { float *p; const float *g; float *buf; long long numel; };

__global__ void __launch_bounds__(256) k_sgd(const SgdEntry *__restrict__ table, const int32_t *__restrict__ chunk_tensor,
                                             const int32_t *__restrict__ chunk_off, float lr, float momentum, float wd)
{
    const SgdEntry e = table[chunk_tensor[blockIdx.x]];
    const long long base = (long long)chunk_off[blockIdx.x] * SGD_CHUNK;
    for (int k = 0; k < SGD_CHUNK / 256; k++) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < e.numel) {
            float p = e.p[i];
            float d = wd != 0.0f ? fmaf(wd, p, e.g[i]) : e.g[i];
            if (momentum != 0.0f) {
                float b = e.buf[i] * momentum;
                b = b + d;
                e.buf[i] = b;
                d = b;
            }
            e.p[i] = fmaf(-lr, d, p);
        }
    }
}

// the pointer table travels to the device as KERNEL ARGUMENTS (96 entries per launch): the gradient tensors are new ones every
// step, and a host -> device copy of 10 KB on the compute stream makes the host wait for the device to get there (measured:
// +2 ms per training step, the host loses its run-ahead); kernel arguments are copied by the launch itself
#define SGD_ARGS 96
struct SgdArgs { SgdEntry e[SGD_ARGS]; };
__global__ void __launch_bounds__(128) k_sgd_table(SgdArgs a, int n, SgdEntry *__restrict__ table)
{
    if ((int)threadIdx.x < n) table[threadIdx.x] = a.e[threadIdx.x];
}

extern "C" int dbaz_sgd_step(int32_t n_tensors, const void *const *params, const void *const *grads, const void *const *bufs,
                             const int64_t *numels, void *table_dev, const int32_t *chunk_tensor_dev, const int32_t *chunk_off_dev,
                             int32_t n_chunks, float lr, float momentum, float weight_decay, void *stream)
{
    if (n_tensors < 0 || !params || !grads || !numels || !table_dev || !chunk_tensor_dev || !chunk_off_dev || n_chunks < 0)
        return terr(nullptr, DBAZ_EINVAL, "dbaz_sgd_step: bad argument");
    if (n_chunks == 0 || n_tensors == 0) return DBAZ_OK;
    SgdEntry *table = (SgdEntry *)table_dev;
    for (int base = 0; base < n_tensors; base += SGD_ARGS) {
        SgdArgs a;
        const int n = std::min(SGD_ARGS, n_tensors - base);
        for (int i = 0; i < n; i++) {
            a.e[i].p = (float *)params[base + i];
            a.e[i].g = (const float *)grads[base + i];
            a.e[i].buf = bufs ? (float *)bufs[base + i] : nullptr;
            a.e[i].numel = numels[base + i];
        }
        hipLaunchKernelGGL(k_sgd_table, dim3(1), dim3(128), 0, (hipStream_t)stream, a, n, table + base);
    }
    hipLaunchKernelGGL(k_sgd, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const SgdEntry *)table_dev, chunk_tensor_dev, chunk_off_dev, lr,
                       momentum, weight_decay);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DBAZ_OK : terr(nullptr, DBAZ_EDEVICE, "sgd step: %s", hipGetErrorString(e));
}

