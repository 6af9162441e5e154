// nn.hip -- policy/value network forward on gfx950 (CDNA4).
//
// ResNetZero (reference nn.py:108-122): bn_input -> conv3x3(3->C)+BN+ReLU ->
//   blocks x [conv3x3+BN+ReLU, conv3x3+BN, +x, ReLU] -> policy head (conv1x1+BN+ReLU,
//   FC, softmax) / value head (conv1x1+BN+ReLU, FC, ReLU, FC, tanh); predict contract
//   NeuralNetWrapper.predict_sync (nn.py:155-160): eval-mode BN, p = exp(log_softmax).
// SimpleNN (reference dots_boxes_nn.py:61-98, 3x3 boards): BN follows the ReLU.
//
// Kernels
//   k_tower<C,NTA,NTB,PREC>  the whole ResNetZero trunk in ONE launch: conv0, the 2*blocks
//                            conv3x3 layers and both 1x1 head convs; the S samples of a
//                            workgroup stay in two ping-pong LDS images, weights stream from L2
//   k_simple_trunk<PREC>     the same for SimpleNN's conv0..conv4 (256 channels)
//   k_dense                  SimpleNN's FC layers (f32 MFMA GEMM, 16 samples per workgroup)
//   k_head_fc                head FCs + softmax / tanh (f32 MFMA GEMM, 16 samples per workgroup)
// conv_lds_f32 / conv_lds_h3 are the per-layer device functions (LDS -> LDS).
//
// The conv layer is an implicit GEMM  Out^T[cout][pos] = W[cout][tap,cin] * In[tap,cin][pos].
// A operand = weights, pre-packed on the host in fragment order and streamed from L2 straight
// into registers (a wave owns one 16-cout tile and half of the 16-row position tiles);
// B operand = activations read with ds_read_b128 (row stride C+8 dwords => conflict-free);
// out-of-image taps read a zero region at the lane's own bank slot.  The accumulator layout
// puts 4 consecutive couts of one position in each lane, so the epilogue (bias, residual,
// ReLU) is vector code.  Eval-mode BatchNorms that FOLLOW a conv are folded into its
// weights/bias on the host (double precision); bn_input is applied to the in-bounds pixels when
// conv0 stages its input (zero padding happens after bn_input in the reference).
//
// PREC 0: exact f32 on v_mfma_f32_16x16x4_f32 (bitwise a k-ordered fmaf chain).
// PREC 1: "f16x3" -- every f32 operand is an error-compensated (hi, lo) pair of halves on
//         v_mfma_f32_16x16x32_f16 with f32 accumulation (see conv_lds_h3).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "nn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CONV_THREADS 512 // 8 waves: two per SIMD, wave = (cout tile, half of the position tiles)
#define MAXT 13 // position tiles (16 rows each) per workgroup (16x16x32 tiling); the 32x32x16 tiling holds 8 tiles of 32 rows
#define MAXROWS 256

struct NNState {
    Geo g;
    int max_batch = 0, precision = 0;
    int kind = 0, C = 0 /* padded to 16/32/64/128 */, Craw = 0 /* state_dict channels */, blocks = 0, hc = 0, vf = 0;
    bool ready = false;
    std::map<std::string, std::vector<float>> sd;
    std::vector<float> osc_host;
    // device
    std::vector<void *> allocs;
    float *in_s = nullptr, *in_t = nullptr;     // bn_input affine [3]
    float *w0 = nullptr, *b0 = nullptr;         // conv0 [9][3][C], [C]
    float *tw = nullptr, *tb = nullptr;         // tower: packed weights [2*blocks][C*C*9], bias [2*blocks][C]
    float *tosc = nullptr;                      // f16x3: per-layer output scale 2^-(sw+ACT_SHIFT)
    unsigned long long *stamp_out = nullptr;    // diagnostic build (-DDBAZ_STAMP) only
    int *overflow = nullptr;                    // f16x3: [0] set when an activation left f16's range, [1] samples re-evaluated in exact f32
    int *ovf_flags = nullptr;                   // f16x3: per sample of the batch, 1 = its workgroup saw an out-of-range activation
    float *tw32 = nullptr, *tb32 = nullptr;     // f16x3: the tower in exact-f32 operand format as well (fallback launch)
    float *hw = nullptr, *hb = nullptr;         // head conv1x1: [2*hc][C], [2*hc]
    float *w0p = nullptr;                       // f16x3: conv0 as a K=32 GEMM on im2col rows, fragments [ct][hi|lo][lane][8 halves]
    float osc0 = 1.0f;                          // f16x3: 2^-sw0
    float *hwp = nullptr;                       // f16x3: head conv weights packed as MFMA fragments [ct][ks][hi|lo][lane][8 halves]
    float hosc = 1.0f;                          // f16x3: 2^-(sw_h + ACT_SHIFT)
    float *hact = nullptr;                      // [batch][2][hc*HW]
    float *wfc = nullptr, *bfc = nullptr;       // head FC GEMM: packed weights [ntp+ntv][KP/16][64][4], bias [(ntp+ntv)*16]
    int KP = 0, RS4 = 0, ntp = 0, ntv = 0;
    size_t fc_lds = 0;
    float *wv1 = nullptr, *bv1 = nullptr;       // [vf], [1]
    // SimpleNN
    float *sn_s0 = nullptr, *sn_t0 = nullptr, *sn_ts = nullptr, *sn_tt = nullptr;
    float *sn_flat = nullptr, *sn_h1 = nullptr;
    float *sn_w0 = nullptr, *sn_b0 = nullptr, *sn_ps0 = nullptr, *sn_pt0 = nullptr; // fc0
    float *sn_w1 = nullptr, *sn_b1 = nullptr, *sn_ps1 = nullptr, *sn_pt1 = nullptr; // fc1
    size_t sn_lds = 0;
    int S = 1, NT = 1, NTT = 7;                 // samples / position tiles per conv workgroup (NTT: compiled tile count)
    int S_small = 0, S_mid = 0, S_big = 0, cus = 256; // tail launches: samples per workgroup of the <2,2> / <4,4> / <5,5> variants (0: unused)
    size_t conv_lds = 0;
    // f16x3 on the 32x32x16 tiling (k_tower<..., MF = 1>): its own workgroup geometry; S / NTT / conv_lds above then describe the
    // 16x16 kernels, which remain in use for the exact-f32 fallback launch
    int want_mf32 = 0, mf32 = 0, S_mf = 0, NT2 = 0, S_mf_tail = 0;
    size_t conv_lds_mf = 0;
    // f16x3 on 16x16x32 with two cout tiles per wave (k_tower<64, NT, 0, 1, 2>, 64 channels): 4 tile groups of NT_c2 tiles
    int want_c2 = 0, c2 = 0, S_c2 = 0, NT_c2 = 0; // (its remainder goes to the one-cout-tile kernels)
    int use_rem = 0; // f16x3, NTT == 7: the remainder sizes live in ONE launch (k_tower_rem)
    bool no_fallback = false; // dbaz_config.debug_flags & DBAZ_DBG_NO_FALLBACK (timing runs only)
    int variant = 5;          // VAR of the main two-cout-tile launch (conv_lds_h3_c2): 5 = shipped; the debug build selects others
    size_t conv_lds_c2 = 0;
};

// ------------------------------------------------------------------------------------
// conv3x3 C -> C on MFMA (f32 exact)
// ------------------------------------------------------------------------------------
// One 3x3 conv layer, LDS -> LDS, for the S samples a workgroup owns (device function of the
// fused tower kernel).  NTT = compile-time number of 16-row position tiles (rows >= R read the
// zero row for every tap and are never written).  The (tap, 16-cin chunk) loop is software
// pipelined by hand: the weight fragment (L2 -> registers) and the NTT activation fragments
// (ds_read_b128) of step i+1 are in flight while the 4*NTT MFMAs of step i issue; a
// sched_barrier keeps hipcc from sinking the prefetch to its first use.
// residual != 0: dst already holds the block input x; the layer writes relu(conv + bias + x)
// in place (every element is read and written by the same lane).
template <int C, int NTT>
__device__ __forceinline__ void conv_lds_f32(const f32x4 *__restrict__ src4, f32x4 *dst4, const float *__restrict__ wpk,
                                             const float *__restrict__ bias, const int *vm, int rowbase, int zbase,
                                             int W, int R, int wave, int lane, int residual, int tbase,
                                             const float *post_s = nullptr, const float *post_t = nullptr)
{
    constexpr int S4 = (C + 8) / 4;  // float4 per LDS row
    constexpr int KC = C / 16;       // 16-cin chunks per tap
    const int jrow = lane & 15, gq = lane >> 4;
    for (int ct = wave & 3; ct < C / 16; ct += 4) {
        f32x4 acc[NTT];
#pragma unroll
        for (int t = 0; t < NTT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 *wbase = reinterpret_cast<const f32x4 *>(wpk) + (size_t)ct * 9 * KC * 64 + lane;
        int addr[NTT];
#pragma unroll
        for (int t = 0; t < NTT; t++) addr[t] = (vm[t] & 1) ? rowbase + (-W - 1) * S4 : zbase + ((rowbase + (-W - 1) * S4) & 15) - t * 16 * S4;
        f32x4 a_cur = wbase[0];
        f32x4 b_cur[NTT];
#pragma unroll
        for (int t = 0; t < NTT; t++) b_cur[t] = src4[addr[t] + t * 16 * S4];
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++) {
#pragma unroll
            for (int kc = 0; kc < KC; kc++) {
                // ---- prefetch step i+1 (the first chunk of the next tap after the last chunk)
                f32x4 a_nxt, b_nxt[NTT];
                if (kc == KC - 1) {
                    const int tn = tap + 1;
                    const int off = ((tn / 3 - 1) * W + (tn % 3 - 1)) * S4;
#pragma unroll
                    for (int t = 0; t < NTT; t++) addr[t] = ((vm[t] >> tn) & 1) ? rowbase + off : zbase + ((rowbase + off) & 15) - t * 16 * S4;
                    a_nxt = wbase[(size_t)(tn < 9 ? tn * KC : 0) * 64];
#pragma unroll
                    for (int t = 0; t < NTT; t++) b_nxt[t] = src4[addr[t] + t * 16 * S4];
                } else {
                    a_nxt = wbase[(size_t)(tap * KC + kc + 1) * 64];
#pragma unroll
                    for (int t = 0; t < NTT; t++) b_nxt[t] = src4[addr[t] + t * 16 * S4 + (kc + 1) * 4];
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- 4 * NTT MFMAs of step i
#pragma unroll
                for (int t = 0; t < NTT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[0], b_cur[t][0], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NTT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[1], b_cur[t][1], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NTT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[2], b_cur[t][2], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NTT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[3], b_cur[t][3], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a_cur = a_nxt;
#pragma unroll
                for (int t = 0; t < NTT; t++) b_cur[t] = b_nxt[t];
            }
        }
        // ---- epilogue: lane holds couts ct*16 + 4*gq .. +3 of position row t*16 + jrow
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + ct * 16 + gq * 4);
#pragma unroll
        for (int t = 0; t < NTT; t++) {
            const int row = (tbase + t) * 16 + jrow;
            if (row < R) {
                const int o4 = row * S4 + ct * 4 + gq;
                f32x4 v = acc[t] + bv;
                if (residual) v += dst4[o4];
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                if (post_s) // SimpleNN: BatchNorm FOLLOWS the ReLU (dots_boxes_nn.py:86-90)
                    v = v * *reinterpret_cast<const f32x4 *>(post_s + ct * 16 + gq * 4) +
                        *reinterpret_cast<const f32x4 *>(post_t + ct * 16 + gq * 4);
                dst4[o4] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// f16x3 variant of the same tower (nn_precision = 1): f32-grade results on the f16 MFMA pipe.
// Every f32 operand is carried as an error-compensated pair of halves  v = hi + lo
// (hi = rn_f16(v), lo = rn_f16(v - hi)); a product is evaluated as hi*hi + hi*lo + lo*hi with
// f32 accumulation (v_mfma_f32_16x16x32_f16, three instructions per K=32 step); the dropped
// lo*lo term is 2^-22 relative.  Weights are pre-scaled by a per-layer power of two (and
// activations by 2^ACT_SHIFT) so that the lo halves stay in f16's normal range; the scales are
// removed exactly in the epilogue.  Same LDS image size as the f32 tower: a row holds
// [C halves hi | C halves lo | 32 B pad] = (C+8) dwords.
// ------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define ACT_SHIFT 5
#define ACT_SCALE 32.0f
#define F16_GUARD 60000.0f

union u128h { f32x4 f; f16x8 h; };

#ifdef DBAZ_STAMP
// diagnostic build only (never shipped): per-wave cycle sums of the layer phases
#define STAMP(var)                                                                      \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

// first two weight steps of a layer, fetched BEFORE the previous layer's epilogue and barrier so
// that the L2 latency of a layer's restart is hidden
struct WPre { f32x4 h0, l0, h1, l1; };

template <int C>
__device__ __forceinline__ void wpre_load(WPre &pre, const f32x4 *wpk_layer, int wave, int lane)
{
    constexpr int N = 9 * (C / 32);
    const f32x4 *wb = wpk_layer + (size_t)(wave & 3) * N * 2 * 64 + lane;
    pre.h0 = wb[0];
    pre.l0 = wb[64];
    pre.h1 = wb[128];
    pre.l1 = wb[192];
}

template <int C, int NTT, bool RR = false>
__device__ __forceinline__ void conv_lds_h3(const f32x4 *__restrict__ src4, f32x4 *dst4, const f32x4 *__restrict__ wpk /*layer*/,
                                            const float *__restrict__ bias, float oscale, const int *vm, int rowbase,
                                            int zbase, int W, int R, int wave, int lane, int residual, bool &ovf_out, int tbase,
                                            const float *post_s, const float *post_t, WPre &pre, const f32x4 *next_wpk,
                                            f32x4 (&res)[NTT], unsigned long long *stamps = nullptr)
{
    // RR (the remainder bodies that run beside conv_lds_h3_c2's main launch, 64 channels): a wave owns ONE cout tile, the same
    // outputs in every layer, so the block's residual input stays in its f32 registers (res, activation-scaled) exactly as in
    // conv_lds_h3_c2 -- the two kernels must round identically: a sample's (p, v) may not depend on which of them evaluated it.
    // Otherwise (!RR: geometries whose main launch is this kernel, e.g. 9x9 with its 7-tile waves, which have no 28 registers
    // to spare) the residual is decoded from the (hi, lo) image in LDS, in every body of that geometry alike.
    static_assert(!RR || C <= 64, "one cout tile per wave");

    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    (void)t0; (void)t1; (void)t2; (void)t3; (void)stamps;
    constexpr int S4 = (C + 8) / 4;  // 16-byte units per LDS row
    constexpr int KS = C / 32;       // K=32 steps per tap
    constexpr int LO = C / 8;        // unit offset of the lo halves inside a row
    constexpr int N = 9 * KS;        // pipeline steps per cout tile
    const int jrow = lane & 15, gq = lane >> 4;
    bool ovf = false;
    for (int ct = wave & 3; ct < C / 16; ct += 4) {
        f32x4 acc[NTT];
#pragma unroll
        for (int t = 0; t < NTT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // packed [ct][step = tap*KS + ks][hi|lo][lane] 16-byte fragments
        const f32x4 *wbase = wpk + (size_t)ct * N * 2 * 64 + lane;
        // Schedule of one step (all fragments SINGLE-buffered, reloaded right after their last use,
        // so that at most NTT LDS reads are in flight at any wait -- lgkmcnt is a 4-bit counter):
        //   G1 hi*hi
        //   G3 lo*hi, each MFMA followed by the ds_read that reloads its (now dead) bh[t]
        //   G2 hi*lo, each MFMA followed by the ds_read that reloads its bl[t]
        // i.e. loads and address math issue in the MFMAs' shadow instead of as separate blocks.
        // Weight fragments (L2 -> registers) run two steps ahead in a 3-deep ring.
        u128h a_h[3], a_l[3];
        u128h bh[NTT], bl[NTT];
        const char *sb = reinterpret_cast<const char *>(src4);
        int ab[NTT]; // BYTE address of this lane's fragment for the current tap (tile constant folded out)
#pragma unroll
        for (int t = 0; t < NTT; t++)
            ab[t] = ((vm[t] & 1) ? rowbase + (-W - 1) * S4 : zbase + ((rowbase + (-W - 1) * S4) & 15) - t * 16 * S4) * 16;
        STAMP(t0);
        if (ct == (wave & 3)) { // first cout tile of the layer: fragments were prefetched across the barrier
            a_h[0].f = pre.h0; a_l[0].f = pre.l0; a_h[1].f = pre.h1; a_l[1].f = pre.l1;
        } else {
            a_h[0].f = wbase[0];
            a_l[0].f = wbase[64];
            a_h[1].f = wbase[128];
            a_l[1].f = wbase[192];
        }
#pragma unroll
        for (int t = 0; t < NTT; t++) bh[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4);
#pragma unroll
        for (int t = 0; t < NTT; t++) bl[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4 + LO * 16);
        STAMP(t1);
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int cur = i % 3, pre = (i + 2) % 3;
            const int ni = i + 1, ntap = ni / KS, nks = ni % KS;
            if (i + 2 < N) {
                a_h[pre].f = wbase[(size_t)(i + 2) * 128];
                a_l[pre].f = wbase[(size_t)(i + 2) * 128 + 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            // G1: hi*hi (the next tap's addresses are computed in its shadow)
#pragma unroll
            for (int t = 0; t < NTT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[cur].h, bh[t].h, acc[t], 0, 0, 0);
            if (ni < N && nks == 0) {
                const int off = ((ntap / 3 - 1) * W + (ntap % 3 - 1)) * S4;
                const int zt = zbase + ((rowbase + off) & 15);
#pragma unroll
                for (int t = 0; t < NTT; t++) ab[t] = (((vm[t] >> ntap) & 1) ? rowbase + off : zt - t * 16 * S4) * 16;
            }
            __builtin_amdgcn_sched_barrier(0);
            // G3: lo*hi; bh[t] is dead after its MFMA -> reload it for the next step right there
#pragma unroll
            for (int t = 0; t < NTT; t++) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l[cur].h, bh[t].h, acc[t], 0, 0, 0);
                if (ni < N) bh[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4 + nks * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
            // G2: hi*lo; same for bl[t]
#pragma unroll
            for (int t = 0; t < NTT; t++) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[cur].h, bl[t].h, acc[t], 0, 0, 0);
                if (ni < N) bl[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4 + nks * 64 + LO * 16);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP(t2);
        // the bias is fetched BEFORE the cross-barrier weight prefetch: loads return in order, so a
        // bias load issued after it would make the epilogue wait for the prefetch
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + ct * 16 + gq * 4);
        asm volatile("" ::"v"(bv));
        __builtin_amdgcn_sched_barrier(0);
        if (next_wpk && ct + 4 >= C / 16) wpre_load<C>(pre, next_wpk, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue: scale back, bias, residual, ReLU, split into halves.
        // All residual reads are issued first (one LDS round trip for the whole wave, not one per
        // tile); the (hi, lo) split uses gfx950's packed round-to-nearest converts; one running max
        // replaces per-value range checks.
        _Float16 *dsth = reinterpret_cast<_Float16 *>(dst4);
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        u32x2 rh[NTT], rl[NTT];
        if constexpr (!RR) {
            if (residual) {
#pragma unroll
                for (int t = 0; t < NTT; t++) {
                    const int row = min((tbase + t) * 16 + jrow, R - 1);
                    const _Float16 *ph = dsth + (size_t)row * (S4 * 8) + ct * 16 + gq * 4;
                    rh[t] = *reinterpret_cast<const u32x2 *>(ph);
                    rl[t] = *reinterpret_cast<const u32x2 *>(ph + C);
                }
            }
        }
        float vmax = 0.0f;
#pragma unroll
        for (int t = 0; t < NTT; t++) {
            const int row = (tbase + t) * 16 + jrow;
            f32x4 v = acc[t] * oscale + bv; // activation-scaled: value * 2^ACT_SHIFT
            if (residual) {
                if constexpr (RR) {
                    v += res[t];
                } else {
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    union { unsigned int u; h2 h; } c0, c1, d0, d1;
                    c0.u = rh[t][0]; c1.u = rh[t][1]; d0.u = rl[t][0]; d1.u = rl[t][1];
                    v[0] += (float)c0.h[0] + (float)d0.h[0];
                    v[1] += (float)c0.h[1] + (float)d0.h[1];
                    v[2] += (float)c1.h[0] + (float)d1.h[0];
                    v[3] += (float)c1.h[1] + (float)d1.h[1];
                }
            }
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            if constexpr (RR) {
                if (residual) res[t] = v; // the block's output = the next block's residual input
            }
            if (post_s) { // SimpleNN: BatchNorm follows the ReLU; post_t is pre-scaled
                v = v * *reinterpret_cast<const f32x4 *>(post_s + ct * 16 + gq * 4) +
                    *reinterpret_cast<const f32x4 *>(post_t + ct * 16 + gq * 4);
                vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            } else {
                vmax = fmaxf(vmax, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
            }
            // hi = rn_f16(v), lo = rn_f16(v - hi) on packed pairs: v_cvt_pk_f16_f32 / v_pk_add_f32
            typedef _Float16 h2v __attribute__((ext_vector_type(2)));
            typedef float f2v __attribute__((ext_vector_type(2)));
            union { h2v h[2]; u32x2 u; } oh, ol;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const f2v x = {v[2 * q], v[2 * q + 1]};
                const h2v h = __builtin_convertvector(x, h2v);
                oh.h[q] = h;
                ol.h[q] = __builtin_convertvector(x - __builtin_convertvector(h, f2v), h2v);
            }
            if (row < R) {
                _Float16 *ph = dsth + (size_t)row * (S4 * 8) + ct * 16 + gq * 4;
                *reinterpret_cast<u32x2 *>(ph) = oh.u;
                *reinterpret_cast<u32x2 *>(ph + C) = ol.u;
            }
        }
        ovf |= vmax > F16_GUARD;
        STAMP(t3);
#ifdef DBAZ_STAMP
        if (stamps) { stamps[0] += t1 - t0; stamps[1] += t2 - t1; stamps[2] += t3 - t2; }
#endif
    }
    ovf_out |= ovf;
}

// ------------------------------------------------------------------------------------
// conv_lds_h3 with TWO 16-cout tiles per wave (MF = 2): a wave owns couts [32 (wave & 1), +32) and NTT position tiles of
// the tile group wave >> 1, so every activation fragment it reads from LDS feeds 6 MFMAs instead of 3 -- half the LDS read
// bytes per FLOP, fewer issue slots -- at the price of twice the weight stream per wave (four fragments per K-step).
// The kernel is power-bound on real data (bench --zero-weights): not re-reading every other activation fragment
// (timing experiment, wrong results) bought 6 %.
// ------------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ void wpre_load_c2(WPre (&pre)[2], const f32x4 *wpk_layer, int wave, int lane)
{
    constexpr int N = 9 * (C / 32);
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const f32x4 *wb = wpk_layer + (size_t)((wave & 1) * 2 + c) * N * 2 * 64 + lane;
        pre[c].h0 = wb[0];
        pre[c].l0 = wb[64];
        pre[c].h1 = wb[128];
        pre[c].l1 = wb[192];
    }
}

// VAR (A/B variants, EXPERIMENTS.md "k_tower, round 3"; the shipped kernel is VAR = 0):
//   1  the block's residual input stays in f32 registers of the wave that owns those outputs in every layer (no LDS decode)
//   2  the 16-byte column chunks of an LDS row are XOR-swizzled by (row >> 2) & 3: the epilogue's ds_write_b64 then conflict
//      2-way instead of 4-way, the fragment reads stay conflict-free
//   8  waves 4-7 (the younger wave of every SIMD, which loses every arbitration) run at s_setprio 1
//   4  the layer's weight fragments reach the workgroup ONCE per K-step, by LDS-DMA into a two-slot ring behind the activation
//      images (each wave fetches one of the step's eight 1-KB fragments), instead of four times into registers: a quarter of
//      the L2 -> CU weight stream, four more LDS fragment reads and one workgroup barrier per K-step
#define VAR_RESREG 1
#define VAR_SWZ 2
#define VAR_WLDS 4
#define VAR_PRIO 8
#define VAR_LO0 16 // timing bound only (wrong results): every lo half zero -- what operand toggling in the two cross-term MFMAs costs
#define VAR_LO8 32 // the lo halves carry 8 significant bits instead of 11 (3 trailing zero mantissa bits; 19-bit products)
#define WRING_UNITS 512 // 16-byte units per ring slot: 4 cout tiles x (hi, lo) x 64 lanes

// LDS-DMA of 16 bytes per lane: lane's global source -> lds_dst (wave-uniform LDS byte address) + 16 * lane
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p;
}
template <int C, int NTT, int VAR = 0>
__device__ __forceinline__ void conv_lds_h3_c2(const f32x4 *__restrict__ src4, f32x4 *dst4, const f32x4 *__restrict__ wpk /*layer*/,
                                               const float *__restrict__ bias, float oscale, const int *vm, int rowbase,
                                               int zbase, int W, int R, int wave, int lane, int residual, bool &ovf_out, int tbase,
                                               WPre (&pre)[2], const f32x4 *next_wpk, f32x4 (&res)[2][NTT],
                                               unsigned long long *stamps = nullptr, f32x4 *wring = nullptr)
{
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    (void)t0; (void)t1; (void)t2; (void)t3; (void)stamps;
    static_assert(C == 64, "two cout tiles per wave x two wave parities = 64 channels");
    constexpr int S4 = (C + 8) / 4;  // 16-byte units per LDS row
    constexpr int KS = C / 32;       // K=32 steps per tap
    constexpr int LO = C / 8;        // unit offset of the lo halves inside a row
    constexpr int N = 9 * KS;        // pipeline steps
    const int jrow = lane & 15, gq = lane >> 4;
    const int ct0 = (wave & 1) * 2;
    f32x4 acc[2][NTT];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int t = 0; t < NTT; t++) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 *wb0 = wpk + (size_t)ct0 * N * 2 * 64 + lane;       // packed [ct][step][hi|lo][lane] 16-byte fragments
    const f32x4 *wb1 = wb0 + (size_t)N * 2 * 64;
    u128h a_h[2][3], a_l[2][3];
    u128h bh[NTT], bl[NTT];
    const char *sb = reinterpret_cast<const char *>(src4);
    int ab[NTT];
    // 16-byte unit of this lane's fragment of tile 0 for a tap at row offset offr (the tile constant t*16*S4 is an immediate;
    // with VAR_SWZ the chunk index gq is XORed with (source row >> 2) & 3, which does not depend on the tile: 16 | tile rows)
    const int row0 = tbase * 16 + jrow;
    auto tap_unit = [&](int offr) -> int {
        if constexpr (VAR & VAR_SWZ) {
            const int su = row0 + offr;
            return su * S4 + (gq ^ ((su >> 2) & 3));
        } else {
            return rowbase + offr * S4;
        }
    };
    {
        const int u0 = tap_unit(-W - 1);
#pragma unroll
        for (int t = 0; t < NTT; t++) ab[t] = ((vm[t] & 1) ? u0 : zbase + (u0 & 15) - t * 16 * S4) * 16;
    }
    STAMP(t0);
    // VAR_WLDS: this wave's DMA piece of a step = fragment (cout tile wave >> 1, hi | lo = wave & 1); the fragments it consumes are
    // those of cout tiles ct0, ct0 + 1: ring units (ct * 2 + hl) * 64 + lane of slot (step & 1)
    const f32x4 *dsrc = wpk + ((size_t)(wave >> 1) * N * 2 + (wave & 1)) * 64 + lane;
    const f32x4 *dnext = next_wpk ? next_wpk + ((size_t)(wave >> 1) * N * 2 + (wave & 1)) * 64 + lane : nullptr;
    const f32x4 *rsrc = wring + (size_t)ct0 * 2 * 64 + lane;
    unsigned ring_dst = 0;
    if constexpr (VAR & VAR_WLDS) {
        static_assert(!(VAR & VAR_WLDS) || N % 2 == 0, "the next layer's step 0 must land in slot 0");
        ring_dst = __builtin_amdgcn_readfirstlane(lds_addr(wring + (size_t)wave * 64));
        // step 0's fragments were taken out of slot 0 BEFORE the layer barrier (below / tower_group's prologue): step 0 refills
        // that slot at once, and a wave late out of the barrier must not find another wave's DMA there
#pragma unroll
        for (int c = 0; c < 2; c++) { a_h[c][0].f = pre[c].h0; a_l[c][0].f = pre[c].l0; }
    } else {
#pragma unroll
        for (int c = 0; c < 2; c++) { a_h[c][0].f = pre[c].h0; a_l[c][0].f = pre[c].l0; a_h[c][1].f = pre[c].h1; a_l[c][1].f = pre[c].l1; }
    }
#pragma unroll
    for (int t = 0; t < NTT; t++) bh[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4);
#pragma unroll
    for (int t = 0; t < NTT; t++) bl[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4 + LO * 16);
    STAMP(t1);
#pragma unroll
    for (int i = 0; i < N; i++) {
        // register set of step i's weight fragments: a 3-deep ring fed from L2, or (VAR_WLDS) two sets fed from the LDS ring
        const int cur = (VAR & VAR_WLDS) ? (i & 1) : i % 3, nxt = (VAR & VAR_WLDS) ? ((i + 1) & 1) : (i + 2) % 3;
        const int ni = i + 1, ntap = ni / KS, nks = ni % KS;
        if constexpr (VAR & VAR_WLDS) {
            // step i + 2's piece -> slot i & 1 (every wave read step i's fragments out of it before the last barrier); behind the
            // layer's last steps: the first two steps of the next layer
            if (i + 2 < N) glds16(dsrc + (size_t)(i + 2) * 128, ring_dst + (i & 1) * (WRING_UNITS * 16));
            else if (dnext) glds16(dnext + (size_t)(i + 2 - N) * 128, ring_dst + (i & 1) * (WRING_UNITS * 16));
            if (ni < N) { // step i + 1's fragments out of slot (i + 1) & 1 (landed and fenced by the barrier that ended step i - 1)
                const f32x4 *r1 = rsrc + (size_t)(ni & 1) * WRING_UNITS;
                a_h[0][nxt].f = r1[0]; a_l[0][nxt].f = r1[64]; a_h[1][nxt].f = r1[128]; a_l[1][nxt].f = r1[192];
            }
        } else if (i + 2 < N) {
            a_h[0][nxt].f = wb0[(size_t)(i + 2) * 128];
            a_l[0][nxt].f = wb0[(size_t)(i + 2) * 128 + 64];
            a_h[1][nxt].f = wb1[(size_t)(i + 2) * 128];
            a_l[1][nxt].f = wb1[(size_t)(i + 2) * 128 + 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        // hi*hi for both cout tiles (the next tap's addresses are computed in their shadow)
#pragma unroll
        for (int t = 0; t < NTT; t++) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[0][cur].h, bh[t].h, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[1][cur].h, bh[t].h, acc[1][t], 0, 0, 0);
        }
        if (ni < N && nks == 0) {
            const int un = tap_unit((ntap / 3 - 1) * W + (ntap % 3 - 1));
            const int zt = zbase + (un & 15);
#pragma unroll
            for (int t = 0; t < NTT; t++) ab[t] = (((vm[t] >> ntap) & 1) ? un : zt - t * 16 * S4) * 16;
        }
        __builtin_amdgcn_sched_barrier(0);
        // lo*hi; bh[t] is dead after its second MFMA -> reload it for the next step right there
#pragma unroll
        for (int t = 0; t < NTT; t++) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l[0][cur].h, bh[t].h, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l[1][cur].h, bh[t].h, acc[1][t], 0, 0, 0);
            if (ni < N) bh[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4 + nks * 64);
            __builtin_amdgcn_sched_barrier(0);
        }
        // hi*lo; same for bl[t]
#pragma unroll
        for (int t = 0; t < NTT; t++) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[0][cur].h, bl[t].h, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[1][cur].h, bl[t].h, acc[1][t], 0, 0, 0);
            if (ni < N) bl[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4 + nks * 64 + LO * 16);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (VAR & VAR_WLDS) {
            // end of step i: this wave's DMA piece has landed (vmcnt) and its ring reads of step i + 1 have returned (LDS returns in
            // order; only the 2 * NTT activation reads issued after them may still be out) -- then every wave's have
            if (ni < N) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(%0)" ::"n"(2 * NTT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ni < N) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    STAMP(t2);
    f32x4 bv[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        bv[c] = *reinterpret_cast<const f32x4 *>(bias + (ct0 + c) * 16 + gq * 4);
        asm volatile("" ::"v"(bv[c]));
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(VAR & VAR_WLDS)) {
        if (next_wpk) wpre_load_c2<C>(pre, next_wpk, wave, lane);
    } else if (next_wpk) {
        // the next layer's step 0 (slot 0: DMAed in step N - 2, fenced by that step's barrier) into registers before the layer barrier
        pre[0].h0 = rsrc[0]; pre[0].l0 = rsrc[64]; pre[1].h0 = rsrc[128]; pre[1].l0 = rsrc[192];
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue (as conv_lds_h3): lane holds couts (ct0 + c) * 16 + 4 gq .. +3 of position row (tbase + t) * 16 + jrow
    _Float16 *dsth = reinterpret_cast<_Float16 *>(dst4);
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    float vmax = 0.0f;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        // column (in halves) of this lane's 4 couts inside a row
        int col = (ct0 + c) * 16 + gq * 4;
        if constexpr (VAR & VAR_SWZ) col = ((((ct0 + c) * 2 + (gq >> 1)) ^ ((jrow >> 2) & 3)) * 8) + (gq & 1) * 4;
        u32x2 rh[NTT], rl[NTT];
        if constexpr (!(VAR & VAR_RESREG)) {
            if (residual) {
#pragma unroll
                for (int t = 0; t < NTT; t++) {
                    const int row = min((tbase + t) * 16 + jrow, R - 1);
                    const _Float16 *ph = dsth + (size_t)row * (S4 * 8) + col;
                    rh[t] = *reinterpret_cast<const u32x2 *>(ph);
                    rl[t] = *reinterpret_cast<const u32x2 *>(ph + C);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NTT; t++) {
            const int row = (tbase + t) * 16 + jrow;
            f32x4 v = acc[c][t] * oscale + bv[c];
            if (residual) {
                if constexpr (VAR & VAR_RESREG) {
                    v += res[c][t];
                } else {
                    union { unsigned int u; h2v h; } c0, c1, d0, d1;
                    c0.u = rh[t][0]; c1.u = rh[t][1]; d0.u = rl[t][0]; d1.u = rl[t][1];
                    v[0] += (float)c0.h[0] + (float)d0.h[0];
                    v[1] += (float)c0.h[1] + (float)d0.h[1];
                    v[2] += (float)c1.h[0] + (float)d1.h[0];
                    v[3] += (float)c1.h[1] + (float)d1.h[1];
                }
            }
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            if constexpr (VAR & VAR_RESREG) {
                if (residual) res[c][t] = v; // the block's output = the next block's residual input
            }
            vmax = fmaxf(vmax, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
            union { h2v h[2]; u32x2 u; } oh, ol;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const f2v x = {v[2 * q], v[2 * q + 1]};
                const h2v h = __builtin_convertvector(x, h2v);
                oh.h[q] = h;
                ol.h[q] = __builtin_convertvector(x - __builtin_convertvector(h, f2v), h2v);
            }
            if constexpr (VAR & VAR_LO0) ol.u = (u32x2){0u, 0u};
            if constexpr (VAR & VAR_LO8) ol.u &= (u32x2){0xFFF8FFF8u, 0xFFF8FFF8u};
            if (row < R) {
                _Float16 *ph = dsth + (size_t)row * (S4 * 8) + col;
                *reinterpret_cast<u32x2 *>(ph) = oh.u;
                *reinterpret_cast<u32x2 *>(ph + C) = ol.u;
            }
        }
    }
    ovf_out |= vmax > F16_GUARD;
    STAMP(t3);
#ifdef DBAZ_STAMP
    if (stamps) { stamps[0] += t1 - t0; stamps[1] += t2 - t1; stamps[2] += t3 - t2; }
#endif
}


#ifdef DBAZ_DEBUG // A/B record (EXPERIMENTS.md): measured 5.6 % slower per evaluation than the shipped 8-wave kernel
// ------------------------------------------------------------------------------------
// MF = 3: FOUR waves per workgroup, one per SIMD with the whole register file; a wave owns all four cout tiles of a quarter of
// the position tiles (NTT tiles): 16 accumulator tiles, 48 MFMAs per K-step.  An activation fragment then feeds 12 MFMAs and a
// weight fragment (out of the LDS ring) 12 as well: 16 ds_read_b128 per 48 MFMAs against 12 per 24 in conv_lds_h3_c2 -- a third
// fewer LDS fragment reads per FLOP (the kernel is power-bound: EXPERIMENTS.md).  The weight ring, the register-resident residual
// stream and the rounding are conv_lds_h3_c2's (bit-identical results).
// ------------------------------------------------------------------------------------
template <int C, int NTT>
__device__ __forceinline__ void conv_lds_h3_w4(const f32x4 *__restrict__ src4, f32x4 *dst4, const f32x4 *__restrict__ wpk /*layer*/,
                                               const float *__restrict__ bias, float oscale, const int *vm, int zbase, int W, int R,
                                               int wave, int lane, int residual, bool &ovf_out, int tbase, f32x4 (&pre)[8],
                                               const f32x4 *next_wpk, f32x4 (&res)[4][NTT], f32x4 *wring)
{
    static_assert(C == 64, "four cout tiles per wave = 64 channels");
    constexpr int S4 = (C + 8) / 4, KS = C / 32, LO = C / 8, N = 9 * KS;
    static_assert(N % 2 == 0, "the next layer's step 0 must land in slot 0");
    const int jrow = lane & 15, gq = lane >> 4;
    f32x4 acc[4][NTT];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int t = 0; t < NTT; t++) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u128h a[2][8]; // [register set][cout tile * 2 + (hi | lo)]
    u128h bh[NTT], bl[NTT];
    const char *sb = reinterpret_cast<const char *>(src4);
    int ab[NTT];
    const int rowbase = (tbase * 16 + jrow) * S4 + gq;
    {
        const int u0 = rowbase + (-W - 1) * S4;
#pragma unroll
        for (int t = 0; t < NTT; t++) ab[t] = ((vm[t] & 1) ? u0 : zbase + (u0 & 15) - t * 16 * S4) * 16;
    }
    // this wave's two DMA pieces of a step: the (hi, lo) fragments of cout tile `wave` (ring units (2 wave + hl) * 64 + lane)
    const f32x4 *dsrc = wpk + (size_t)wave * N * 2 * 64 + lane;
    const f32x4 *dnext = next_wpk ? next_wpk + (size_t)wave * N * 2 * 64 + lane : nullptr;
    const unsigned ring_dst = __builtin_amdgcn_readfirstlane(lds_addr(wring + (size_t)wave * 2 * 64));
    const f32x4 *rsrc = wring + lane;
#pragma unroll
    for (int c = 0; c < 8; c++) a[0][c].f = pre[c]; // step 0: taken out of slot 0 before the layer barrier
#pragma unroll
    for (int t = 0; t < NTT; t++) bh[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4);
#pragma unroll
    for (int t = 0; t < NTT; t++) bl[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4 + LO * 16);
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int cur = i & 1, nxt = cur ^ 1;
        const int ni = i + 1, ntap = ni / KS, nks = ni % KS;
        // step i + 2's pieces -> slot i & 1 (every wave took step i's fragments out of it before the last barrier)
        if (i + 2 < N) {
            glds16(dsrc + (size_t)(i + 2) * 128, ring_dst + (i & 1) * (WRING_UNITS * 16));
            glds16(dsrc + (size_t)(i + 2) * 128 + 64, ring_dst + (i & 1) * (WRING_UNITS * 16) + 1024);
        } else if (dnext) {
            glds16(dnext + (size_t)(i + 2 - N) * 128, ring_dst + (i & 1) * (WRING_UNITS * 16));
            glds16(dnext + (size_t)(i + 2 - N) * 128 + 64, ring_dst + (i & 1) * (WRING_UNITS * 16) + 1024);
        }
        if (ni < N) { // step i + 1's eight fragments out of slot (i + 1) & 1: at the head of the step, long back when the
                      // activation reloads of this step are issued (lgkmcnt is a 4-bit counter)
            const f32x4 *r1 = rsrc + (size_t)(ni & 1) * WRING_UNITS;
#pragma unroll
            for (int c = 0; c < 8; c++) a[nxt][c].f = r1[c * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        // hi*hi
#pragma unroll
        for (int t = 0; t < NTT; t++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][2 * c].h, bh[t].h, acc[c][t], 0, 0, 0);
        if (ni < N && nks == 0) {
            const int un = rowbase + ((ntap / 3 - 1) * W + (ntap % 3 - 1)) * S4;
            const int zt = zbase + (un & 15);
#pragma unroll
            for (int t = 0; t < NTT; t++) ab[t] = (((vm[t] >> ntap) & 1) ? un : zt - t * 16 * S4) * 16;
        }
        __builtin_amdgcn_sched_barrier(0);
        // lo*hi; bh[t] is dead after its fourth MFMA -> reload it for the next step right there
#pragma unroll
        for (int t = 0; t < NTT; t++) {
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][2 * c + 1].h, bh[t].h, acc[c][t], 0, 0, 0);
            if (ni < N) bh[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4 + nks * 64);
            __builtin_amdgcn_sched_barrier(0);
        }
        // hi*lo; same for bl[t]
#pragma unroll
        for (int t = 0; t < NTT; t++) {
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][2 * c].h, bl[t].h, acc[c][t], 0, 0, 0);
            if (ni < N) bl[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 256 * S4 + nks * 64 + LO * 16);
            __builtin_amdgcn_sched_barrier(0);
        }
        // end of step: this wave's DMA pieces have landed, its ring reads have returned (only the 2 NTT activation reloads issued
        // after them may still be out) -- then every wave's have
        if (ni < N) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(%0)" ::"n"(2 * NTT) : "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 bv[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        bv[c] = *reinterpret_cast<const f32x4 *>(bias + c * 16 + gq * 4);
        asm volatile("" ::"v"(bv[c]));
    }
    if (next_wpk) { // the next layer's step 0 (slot 0: DMAed in step N - 2, fenced by that step's barrier) before the layer barrier
#pragma unroll
        for (int c = 0; c < 8; c++) pre[c] = rsrc[c * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    _Float16 *dsth = reinterpret_cast<_Float16 *>(dst4);
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    float vmax = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int col = c * 16 + gq * 4;
#pragma unroll
        for (int t = 0; t < NTT; t++) {
            const int row = (tbase + t) * 16 + jrow;
            f32x4 v = acc[c][t] * oscale + bv[c];
            if (residual) v += res[c][t];
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            if (residual) res[c][t] = v;
            vmax = fmaxf(vmax, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
            union { h2v h[2]; u32x2 u; } oh, ol;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const f2v x = {v[2 * q], v[2 * q + 1]};
                const h2v h = __builtin_convertvector(x, h2v);
                oh.h[q] = h;
                ol.h[q] = __builtin_convertvector(x - __builtin_convertvector(h, f2v), h2v);
            }
            if (row < R) {
                _Float16 *ph = dsth + (size_t)row * (S4 * 8) + col;
                *reinterpret_cast<u32x2 *>(ph) = oh.u;
                *reinterpret_cast<u32x2 *>(ph + C) = ol.u;
            }
        }
    }
    ovf_out |= vmax > F16_GUARD;
}

#endif // DBAZ_DEBUG (conv_lds_h3_w4)

#ifdef DBAZ_DEBUG // A/B tiling of the debug build (tools/ab_tilings.sh); measured 2.4 % slower per evaluation (EXPERIMENTS.md)
// ------------------------------------------------------------------------------------
// The same f16x3 layer on v_mfma_f32_32x32x16_f16 (MF = 1).  Output tile = 32 couts x 32 positions: a wave owns one
// 32-cout tile (wave & 1) and NTT position tiles of 32 rows (tile group wave >> 1); per K=16 step it needs ONE weight
// fragment pair for 3 * NTT MFMAs of 32 cycles each, and every activation fragment feeds twice the MACs of the 16x16x32
// tiling -- half the LDS read bytes and half the vector-issue slots per FLOP (the 16x16x32 form holds the issue port
// for 8 of its 16 cycles, this one for 8 of 32).  LDS rows are [C hi | C lo | 16 B pad] = C + 4 dwords: with 17 units
// of 16 B per row the 32 rows of a fragment fall on distinct bank slots inside every 16-lane group of ds_read_b128.
// Lane map (cdna_hip_programming.md 3): lane l, r = l & 31, h = l >> 5: A[cout r][k = 8h + j], B[k = 8h + j][pos r];
// D: pos = l & 31, cout = (reg & 3) + 8 (reg >> 2) + 4 h.
// ------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int C>
__device__ __forceinline__ void wpre_load32(WPre &pre, const f32x4 *wpk_layer, int wave, int lane)
{
    constexpr int N = 9 * (C / 16);
    const f32x4 *wb = wpk_layer + (size_t)(wave & 1) * N * 2 * 64 + lane;
    pre.h0 = wb[0];
    pre.l0 = wb[64];
    pre.h1 = wb[128];
    pre.l1 = wb[192];
}

template <int C, int NTT>
__device__ __forceinline__ void conv_lds_h3_32(const f32x4 *__restrict__ src4, f32x4 *dst4, const f32x4 *__restrict__ wpk /*layer*/,
                                               const float *__restrict__ bias, float oscale, const int *vm, int rowbase,
                                               int zbase, int W, int R, int wave, int lane, int residual, bool &ovf_out, int tbase,
                                               WPre &pre, const f32x4 *next_wpk, unsigned long long *stamps = nullptr)
{
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    (void)t0; (void)t1; (void)t2; (void)t3; (void)stamps;
    constexpr int S4 = (C + 4) / 4;  // 16-byte units per LDS row
    constexpr int KS = C / 16;       // K=16 steps per tap
    constexpr int LO = C / 8;        // unit offset of the lo halves inside a row
    constexpr int N = 9 * KS;        // pipeline steps per cout tile
    const int jrow = lane & 31, gq = lane >> 5;
    bool ovf = false;
    for (int ct = wave & 1; ct < C / 32; ct += 2) {
        f32x16 acc[NTT];
#pragma unroll
        for (int t = 0; t < NTT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
        const f32x4 *wbase = wpk + (size_t)ct * N * 2 * 64 + lane; // packed [ct][step][hi|lo][lane] 16-byte fragments
        u128h a_h[3], a_l[3];
        u128h bh[NTT], bl[NTT];
        const char *sb = reinterpret_cast<const char *>(src4);
        int ab[NTT]; // BYTE address of this lane's fragment for the current tap (tile constant folded out)
#pragma unroll
        for (int t = 0; t < NTT; t++)
            ab[t] = ((vm[t] & 1) ? rowbase + (-W - 1) * S4 : zbase + ((rowbase + (-W - 1) * S4) & 15) - t * 32 * S4) * 16;
        STAMP(t0);
        if (ct == (wave & 1)) { // first cout tile of the layer: fragments were prefetched across the barrier
            a_h[0].f = pre.h0; a_l[0].f = pre.l0; a_h[1].f = pre.h1; a_l[1].f = pre.l1;
        } else {
            a_h[0].f = wbase[0];
            a_l[0].f = wbase[64];
            a_h[1].f = wbase[128];
            a_l[1].f = wbase[192];
        }
#pragma unroll
        for (int t = 0; t < NTT; t++) bh[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 512 * S4);
#pragma unroll
        for (int t = 0; t < NTT; t++) bl[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 512 * S4 + LO * 16);
        STAMP(t1);
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int cur = i % 3, nxt = (i + 2) % 3;
            const int ni = i + 1, ntap = ni / KS, nks = ni % KS;
            if (i + 2 < N) {
                a_h[nxt].f = wbase[(size_t)(i + 2) * 128];
                a_l[nxt].f = wbase[(size_t)(i + 2) * 128 + 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            // G1: hi*hi (the next tap's addresses are computed in its shadow)
#pragma unroll
            for (int t = 0; t < NTT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[cur].h, bh[t].h, acc[t], 0, 0, 0);
            if (ni < N && nks == 0) {
                const int off = ((ntap / 3 - 1) * W + (ntap % 3 - 1)) * S4;
                const int zt = zbase + ((rowbase + off) & 15);
#pragma unroll
                for (int t = 0; t < NTT; t++) ab[t] = (((vm[t] >> ntap) & 1) ? rowbase + off : zt - t * 32 * S4) * 16;
            }
            __builtin_amdgcn_sched_barrier(0);
            // G3: lo*hi; bh[t] is dead after its MFMA -> reload it for the next step right there
#pragma unroll
            for (int t = 0; t < NTT; t++) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[cur].h, bh[t].h, acc[t], 0, 0, 0);
                if (ni < N) bh[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 512 * S4 + nks * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
            // G2: hi*lo; same for bl[t]
#pragma unroll
            for (int t = 0; t < NTT; t++) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[cur].h, bl[t].h, acc[t], 0, 0, 0);
                if (ni < N) bl[t].f = *reinterpret_cast<const f32x4 *>(sb + ab[t] + t * 512 * S4 + nks * 32 + LO * 16);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP(t2);
        // bias before the cross-barrier weight prefetch (loads return in order)
        f32x4 bv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            bv[q] = *reinterpret_cast<const f32x4 *>(bias + ct * 32 + 8 * q + 4 * gq);
            asm volatile("" ::"v"(bv[q]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (next_wpk && ct + 2 >= C / 32) wpre_load32<C>(pre, next_wpk, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue: this lane holds, of position row (tbase + t) * 32 + jrow, the couts ct*32 + 8q + 4gq .. +3 (q = 0..3)
        _Float16 *dsth = reinterpret_cast<_Float16 *>(dst4);
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        typedef float f2v __attribute__((ext_vector_type(2)));
        u32x2 rh[NTT][4], rl[NTT][4];
        if (residual) {
#pragma unroll
            for (int t = 0; t < NTT; t++) {
                const int row = min((tbase + t) * 32 + jrow, R - 1);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const _Float16 *ph = dsth + (size_t)row * (S4 * 8) + ct * 32 + 8 * q + 4 * gq;
                    rh[t][q] = *reinterpret_cast<const u32x2 *>(ph);
                    rl[t][q] = *reinterpret_cast<const u32x2 *>(ph + C);
                }
            }
        }
        float vmax = 0.0f;
#pragma unroll
        for (int t = 0; t < NTT; t++) {
            const int row = (tbase + t) * 32 + jrow;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 v = (f32x4){acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]} * oscale + bv[q];
                if (residual) {
                    union { unsigned int u; h2v h; } c0, c1, d0, d1;
                    c0.u = rh[t][q][0]; c1.u = rh[t][q][1]; d0.u = rl[t][q][0]; d1.u = rl[t][q][1];
                    v[0] += (float)c0.h[0] + (float)d0.h[0];
                    v[1] += (float)c0.h[1] + (float)d0.h[1];
                    v[2] += (float)c1.h[0] + (float)d1.h[0];
                    v[3] += (float)c1.h[1] + (float)d1.h[1];
                }
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                vmax = fmaxf(vmax, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
                union { h2v h[2]; u32x2 u; } oh, ol;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const f2v x = {v[2 * e], v[2 * e + 1]};
                    const h2v h = __builtin_convertvector(x, h2v);
                    oh.h[e] = h;
                    ol.h[e] = __builtin_convertvector(x - __builtin_convertvector(h, f2v), h2v);
                }
                if (row < R) {
                    _Float16 *ph = dsth + (size_t)row * (S4 * 8) + ct * 32 + 8 * q + 4 * gq;
                    *reinterpret_cast<u32x2 *>(ph) = oh.u;
                    *reinterpret_cast<u32x2 *>(ph + C) = ol.u;
                }
            }
        }
        ovf |= vmax > F16_GUARD;
        STAMP(t3);
#ifdef DBAZ_STAMP
        if (stamps) { stamps[0] += t1 - t0; stamps[1] += t2 - t1; stamps[2] += t3 - t2; }
#endif
    }
    ovf_out |= ovf;
}

#endif // DBAZ_DEBUG

// ------------------------------------------------------------------------------------
// The whole convolutional trunk in ONE launch per step.  A workgroup owns S samples; their
// activations live in two ping-pong LDS images of (S*HW+1) rows x (C+8) dwords for
//   conv0 (3 -> C, VALU, bn_input fused on load, BN0 folded, ReLU)
//   2*blocks conv3x3 layers on MFMA (conv_lds_f32 or conv_lds_h3)
//   the two 1x1 head convs (C -> 2*hc, VALU, BN folded, ReLU)
// Only the leaf feature planes (588 B/sample) come in and the head activations
// (2*hc*HW floats/sample, in the reference's x.view(n,-1) flatten order) go out; weights stream
// from L2.  PREC 0: exact f32 (rows hold C floats); PREC 1: f16x3 (rows hold C hi + C lo halves
// of the activation scaled by 2^ACT_SHIFT).
// ------------------------------------------------------------------------------------
// index (in halves) of channel c inside the hi part of LDS row `row` (VAR_SWZ: 16-byte chunks XORed with (row >> 2) & 3)
template <int SWZ>
__device__ __forceinline__ int act_col(int row, int c)
{
    if constexpr (SWZ) return (((c >> 3) ^ ((row >> 2) & 3)) << 3) | (c & 7);
    else return c;
}
template <int C, int PREC, int STRIDE = C + 8, int SWZ = 0>
__device__ __forceinline__ void act_store(float *lds, int row, int c, float v, bool &ovf)
{
    if constexpr (PREC == 0) {
        lds[row * STRIDE + c] = v;
    } else {
        _Float16 *h = reinterpret_cast<_Float16 *>(lds) + (size_t)row * (STRIDE * 2);
        const float x = v * ACT_SCALE;
        ovf |= fabsf(x) > F16_GUARD;
        const _Float16 hi = (_Float16)x;
        const int cc = act_col<SWZ>(row, c);
        h[cc] = hi;
        h[C + cc] = (_Float16)(x - (float)hi);
    }
}
template <int C, int PREC, int STRIDE = C + 8, int SWZ = 0>
__device__ __forceinline__ float act_load(const float *lds, int row, int c)
{
    if constexpr (PREC == 0) {
        return lds[row * STRIDE + c];
    } else {
        const _Float16 *h = reinterpret_cast<const _Float16 *>(lds) + (size_t)row * (STRIDE * 2);
        const int cc = act_col<SWZ>(row, c);
        return ((float)h[cc] + (float)h[C + cc]) * (1.0f / ACT_SCALE);
    }
}

// Full rounds only (engine: self-play stepping): the leaves the network takes from a list of n -- every kernel of a step applies
// the same rule to the same count, so no launch of its own is needed for it (see "Full rounds only" in tree.hip)
__device__ __forceinline__ int cut_n(int n, int round, int defer_max)
{
    if (round <= 0) return n;
    const int n_full = (n / round) * round;
    return (n_full > 0 && n - n_full <= defer_max) ? n_full : n;
}

struct TowerArgs {
    const float *feat;       // [slot][3][HW] leaf feature planes
    const int32_t *list;     // compacted slot list (nullptr: identity)
    const int32_t *n_dev;    // number of samples
    const float *in_s, *in_t; // bn_input affine
    const float *w0, *b0;    // conv0 [27][C], [C]
    const float *tw;         // tower weights, packed per layer
    const float *tb;         // [2*blocks][C]
    const float *tosc;       // f16x3 per-layer output scale
    const float *hw, *hb;    // head conv1x1 [2*hc][C], [2*hc]
    const float *hwp;        // f16x3: packed (hi, lo) fragments of hw (nullptr: VALU head conv)
    float hosc;
    const float *w0p;        // f16x3: packed (hi, lo) fragments of w0 over k = tap*3 + c, padded to 32 (nullptr: VALU conv0)
    float osc0;
    float *hact;             // out: [sample][2*hc*HW] (unused since the head FCs run inside the tower, kept for diagnostics)
    // head FCs + softmax / tanh inside the tower workgroup (head_fc_fused)
    const float *wfc, *bfc;  // packed FC weights [ntp+ntv][KP/16][64][4], bias [(ntp+ntv)*16]
    const float *wv1, *bv1;  // value FC1 [vf], [1]
    float *P, *V;            // out: softmax policy [slot][AS], tanh value [slot]
    int KP, ntp, ntv, vf, AS;
    int32_t *n_used;         // (optional) the count the step's network took (k_expand_backup: list positions >= n ask again next step)
    int *overflow;           // [0] sticky "an activation left f16's range", [1] samples re-evaluated in f32
    int *ovf_flags;          // per sample: its f16x3 workgroup overflowed (set by PREC 1, consumed by the fallback launch)
    int fallback;            // PREC 0 launch behind a PREC 1 one: only workgroups holding a flagged sample run
    int S, nblocks, hc;
    // tail handling (see nn_forward): role 0 = main launch, 1 / 2 = tail launches with fewer samples per workgroup
    int role, S_main, S_small, S_mid, S_big, S_huge, cus;
    int cut_round, cut_defer; // cut_n's rule for this step's list (0: every leaf)
    unsigned long long *stamp_out; // diagnostic build only
};

// ------------------------------------------------------------------------------------
// Head FCs + softmax / tanh for the S samples of a tower workgroup, straight from the head activations it has just staged in LDS
// (stage[sample][2 hc][HW], the reference's x.view(n, -1) flatten order: policy rows first, value rows behind them).
// Policy FC (A outputs) and value FC0 (vf outputs) are ONE GEMM  Out^T[out][sample] = Wfc . hact^T  on v_mfma_f32_16x16x4_f32:
// wave w owns the 16-output tiles w, w + #waves, ...; the samples are the MFMA's 16 columns (S <= 16); K runs in order, so a
// sample's logits do not depend on the batch or the workgroup size it is evaluated in.  Weight fragments come from L2 in operand
// layout, eight K-chunks per round trip.  (Until round 3 this was a kernel of its own, k_head_fc: 33 us per 6x6 step, 14.5 us per
// 3x3 step, plus a kernel boundary and the round trip of the head activations through HBM; SimpleNN still uses it.)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void head_fc_fused(const Geo &g, const TowerArgs &a, const float *stage, float *lg, int ns, int s0,
                                              int tid, int nthr)
{
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    const int K = a.hc * g.HW, KC = a.KP / 16, NJ = a.ntp + a.ntv, LGS = NJ * 16 + 1, A = g.A;
    const int jrow = lane & 15, gq = lane >> 4;
    const f32x4 *wfc4 = reinterpret_cast<const f32x4 *>(a.wfc);
    const float *rowp = stage + (size_t)min(jrow, ns - 1) * (2 * K) + gq * 4;
    constexpr int G = 8; // K-chunks per weight round trip
    for (int job = wave; job < NJ; job += nw) {
        const float *brow = rowp + (job < a.ntp ? 0 : K);
        const f32x4 *wp = wfc4 + (size_t)job * KC * 64 + lane;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 wc[G], wn[G];
#pragma unroll
        for (int j = 0; j < G; j++) wc[j] = j < KC ? wp[(size_t)j * 64] : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int base = 0; base < KC; base += G) {
#pragma unroll
            for (int j = 0; j < G; j++) wn[j] = base + G + j < KC ? wp[(size_t)(base + G + j) * 64] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < G; j++)
                if (base + j < KC) {
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(brow + (base + j) * 16);
#pragma unroll
                    for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[j][e], b[e], acc, 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < G; j++) wc[j] = wn[j];
        }
        if (jrow < ns) {
#pragma unroll
            for (int e = 0; e < 4; e++) lg[jrow * LGS + job * 16 + gq * 4 + e] = acc[e] + a.bfc[job * 16 + gq * 4 + e];
        }
    }
    __syncthreads();
    for (int sidx = wave; sidx < ns; sidx += nw) {
        const int dst = a.list ? a.list[s0 + sidx] : s0 + sidx;
        const float *l = lg + sidx * LGS;
        float mx = -INFINITY;
        for (int o = lane; o < A; o += 64) mx = fmaxf(mx, l[o]);
        for (int sh = 32; sh > 0; sh >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
        float sum = 0.0f;
        for (int o = lane; o < A; o += 64) sum += expf(l[o] - mx);
        for (int sh = 32; sh > 0; sh >>= 1) sum += __shfl_xor(sum, sh);
        for (int o = lane; o < A; o += 64) a.P[(size_t)dst * a.AS + o] = expf(l[o] - mx) / sum;
        float hv = 0.0f;
        for (int u = lane; u < a.vf; u += 64) hv += fmaxf(l[a.ntp * 16 + u], 0.0f) * a.wv1[u];
        for (int sh = 32; sh > 0; sh >>= 1) hv += __shfl_xor(hv, sh);
        if (lane == 0) a.V[dst] = tanhf(hv + a.bv1[0]);
    }
}

// the S samples [s0, s0 + ns) of one workgroup through the whole trunk
template <int C, int NTA, int NTB, int PREC, int MF, int VAR = 0>
__device__ __forceinline__ void tower_group(const Geo &g, const TowerArgs &a, float *lds, const int S, const int s0, const int ns)
{
    static_assert(MF == 0 || PREC == 1, "the alternative tilings exist for the f16x3 mode only");
    static_assert(VAR == 0 || MF >= 2 || (MF == 0 && VAR == VAR_RESREG), "the A/B variants belong to the two-cout-tile kernel");
    constexpr int SWZ = (VAR & VAR_SWZ) ? 1 : 0;
    // MF: 0 = 16x16x32, a wave = one cout tile x half of the position tiles; 1 = 32x32x16; 2 = 16x16x32, a wave = two cout
    // tiles x a quarter of the position tiles (conv_lds_h3_c2)
    constexpr int STRIDE = MF == 1 ? C + 4 : C + 8; // dwords per LDS row (see conv_lds_h3_32 for the 32x32x16 tiling's choice)
    constexpr int S4 = STRIDE / 4;
    const int HW = g.HW, W = g.W, H = g.H;
    const int R = ns * HW;           // valid rows in this workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NTHR = blockDim.x;
    // Zero REGION (3 rows, starting at a multiple of 16 float4 units) behind the S*HW rows of each
    // image: a lane whose tap falls outside the board reads the zero whose bank slot equals the
    // slot of its natural address, so out-of-image lanes never collide with in-image lanes of
    // their ds_read_b128 group (one shared zero row cost +1 LDS cycle per group: 44 % of LDS time).
    const int zu = (S * HW * S4 + 15) & ~15;  // unit index of the zero region
    const int img_units = zu + 3 * S4;
    float *X = lds;
    float *Y = lds + (size_t)img_units * 4;
    f32x4 *X4 = reinterpret_cast<f32x4 *>(X);
    f32x4 *Y4 = reinterpret_cast<f32x4 *>(Y);
    bool ovf = false;
    unsigned long long tE0 = 0, tE1 = 0, tL0 = 0, tL1 = 0, tR0 = 0;
    (void)tE0; (void)tE1; (void)tL0; (void)tL1; (void)tR0;
    STAMP(tE0);
#ifdef DBAZ_STAMP
    tR0 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- conv0: stage zero-padded bn_input(planes) and the 27*C weights in the (idle) Y image
    {
        const int PW = W + 2, PH = H + 2, PP = 3 * PH * PW;
        float *pad = Y;               // [ns][3][PH][PW]
        float *wl = Y + S * PP;       // [27][C]   (VALU path only)
        bool mfma0 = false; // the im2col image (10 units per row) must fit behind the padded planes in the idle image
        if constexpr (PREC == 1) mfma0 = a.w0p != nullptr && ((((S * PP + 3) / 4 + 1) & ~1) + S * HW * 10 <= img_units);
        __shared__ int slot_s[16];    // sample -> slot (S <= 13)
        __shared__ int rowbase_s[MAXROWS]; // position row -> offset of its 3x3 window in the padded planes
        for (int i = tid; i < ns * PP; i += NTHR) pad[i] = 0.0f;
        if (tid < ns) slot_s[tid] = a.list ? a.list[s0 + tid] : s0 + tid;
        if (tid < R) {
            const int sidx = tid / HW, p = tid - sidx * HW, y = p / W;
            rowbase_s[tid] = sidx * PP + y * PW + (p - y * W);
        }
        if (!mfma0)
            for (int i = tid; i < 27 * C; i += NTHR) wl[i] = a.w0[i];
        __syncthreads();
        {
            // element r of a sample's planes is handled by thread r (its plane / row / column are computed once);
            // the loads of all samples are independent
            const int F3 = 3 * HW;
            for (int r = tid; r < F3; r += NTHR) {
                const int c = r / HW, p = r - c * HW, y = p / W, x = p - y * W;
                const float sc = a.in_s[c], tc = a.in_t[c];
                const int dst = (c * PH + y + 1) * PW + x + 1;
                for (int sidx = 0; sidx < ns; sidx++)
                    pad[sidx * PP + dst] = a.feat[(size_t)slot_s[sidx] * F3 + r] * sc + tc;
            }
        }
        __syncthreads();
        if constexpr (PREC == 1) {
            if (mfma0) {
                // conv0 as a GEMM on MFMA: the 27 inputs of a position (k = tap*3 + c, padded to 32) form a row of an
                // im2col image in the tower's (hi, lo) operand format for 32 channels -- [32 halves hi | 32 halves lo |
                // 16 B pad] = 10 units of 16 B, activation-scaled -- and the layer is one K=32 step of the f16x3 scheme
                constexpr int MU = 10;
                const int m_off = ((S * PP + 3) / 4 + 1) & ~1;            // units; pad image first
                f32x4 *M4 = Y4 + m_off;
                _Float16 *Mh = reinterpret_cast<_Float16 *>(M4);
                for (int i = tid; i < R * 16; i += NTHR) {
                    const int row = i >> 4, kk = (i & 15) * 2;
                    const int rb = rowbase_s[row];
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    h2 hi, lo;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int k = kk + q;
                        float v = 0.0f;
                        if (k < 27) {
                            const int tap = k / 3, c = k - tap * 3, ty = tap / 3, tx = tap - ty * 3; // constant divisors
                            v = pad[rb + (c * PH + ty) * PW + tx] * ACT_SCALE;
                        }
                        ovf |= fabsf(v) > F16_GUARD;
                        hi[q] = (_Float16)v;
                        lo[q] = (_Float16)(v - (float)hi[q]);
                    }
                    *reinterpret_cast<h2 *>(Mh + (size_t)row * (MU * 8) + kk) = hi;
                    *reinterpret_cast<h2 *>(Mh + (size_t)row * (MU * 8) + 32 + kk) = lo;
                }
                __syncthreads();
                const int jr = lane & 15, gg = lane >> 4;
                const int ct = wave & 3, half = wave >> 2, nhalf = NTHR >> 8;
                const int NT = (R + 15) / 16;
                for (int cto = ct; cto < C / 16; cto += 4) {
                    const f32x4 *wp = reinterpret_cast<const f32x4 *>(a.w0p) + (size_t)cto * 2 * 64 + lane;
                    u128h ah, al;
                    ah.f = wp[0];
                    al.f = wp[64];
                    const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.b0 + cto * 16 + gg * 4) * ACT_SCALE;
                    _Float16 *dsth = reinterpret_cast<_Float16 *>(X4);
                    float vmax = 0.0f;
                    for (int t = half; t < NT; t += nhalf) {
                        const int row = t * 16 + jr;
                        const int rr = min(row, R - 1);
                        u128h bh, bl;
                        bh.f = M4[(size_t)rr * MU + gg];
                        bl.f = M4[(size_t)rr * MU + 4 + gg];
                        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bh.h, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.h, bh.h, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bl.h, acc, 0, 0, 0);
                        f32x4 v = acc * a.osc0 + bv; // activation-scaled
                        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                        vmax = fmaxf(vmax, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
                        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                        typedef float f2v __attribute__((ext_vector_type(2)));
                        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                        union { h2v h[2]; u32x2 u; } oh, ol;
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            const f2v xx = {v[2 * q], v[2 * q + 1]};
                            const h2v hh = __builtin_convertvector(xx, h2v);
                            oh.h[q] = hh;
                            ol.h[q] = __builtin_convertvector(xx - __builtin_convertvector(hh, f2v), h2v);
                        }
                        if (row < R) {
                            int col = cto * 16 + gg * 4;
                            if constexpr (SWZ) col = (((cto * 2 + (gg >> 1)) ^ ((jr >> 2) & 3)) * 8) + (gg & 1) * 4;
                            _Float16 *ph = dsth + (size_t)row * (S4 * 8) + col;
                            *reinterpret_cast<u32x2 *>(ph) = oh.u;
                            *reinterpret_cast<u32x2 *>(ph + C) = ol.u;
                        }
                    }
                    ovf |= vmax > F16_GUARD;
                }
            }
        }
        if (!mfma0) {
        // one work item = (row, 16 couts): its 27 inputs are read once, the weights are
        // broadcast reads (16 lanes share an address)
        for (int i = tid; i < R * (C / 16); i += NTHR) {
            const int row = i % R, cq = i / R;
            const float *pp = pad + rowbase_s[row];
            float in27[27];
#pragma unroll
            for (int tap = 0; tap < 9; tap++)
#pragma unroll
                for (int c = 0; c < 3; c++) in27[tap * 3 + c] = pp[(c * PH + tap / 3) * PW + tap % 3];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = cq * 16 + q * 4;
                f32x4 acc = *reinterpret_cast<const f32x4 *>(a.b0 + co);
#pragma unroll
                for (int k = 0; k < 27; k++) acc += in27[k] * *reinterpret_cast<const f32x4 *>(wl + k * C + co);
#pragma unroll
                for (int e = 0; e < 4; e++) act_store<C, PREC, STRIDE, SWZ>(X, row, co + e, fmaxf(acc[e], 0.0f), ovf);
            }
        }
        }
        __syncthreads();
        if (tid < 3 * S4) {
            X4[zu + tid] = (f32x4){0.f, 0.f, 0.f, 0.f};
            Y4[zu + tid] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    // position-tile rows / lane map of the layer MFMA: 16x16x32 (row = lane & 15, k quarter = lane >> 4) or
    // 32x32x16 (row = lane & 31, k half = lane >> 5)
    constexpr int TR = MF == 1 ? 32 : 16;
    const int jrow = lane & (TR - 1), gq = MF == 1 ? (lane >> 5) : (lane >> 4);
    // per position tile: 9-bit mask of the taps whose source pixel lies inside the image
    // 16x16x32: waves 0-3 own position tiles [0, NTA), waves 4-7 tiles [NTA, NTA+NTB)
    // 32x32x16: the wave pair (wave >> 1) owns tiles [(wave >> 1) * NTA, +NTA), one 32-cout tile each
    //           (NTB > 0: the older wave half takes NTA tiles per wave, the younger NTB -- tools/ab_mf32.sh)
    const bool first = wave < 4;
    const int tbase = MF == 3 ? wave * NTA // (four waves, a quarter of the position tiles each)
                    : MF ? (NTB > 0 ? (first ? ((wave >> 1) & 1) * NTA : 2 * NTA + ((wave >> 1) & 1) * NTB) : (wave >> 1) * NTA)
                         : (first ? 0 : NTA);
    int vm[NTA];
#pragma unroll
    for (int t = 0; t < NTA; t++) {
        int row = (tbase + t) * TR + jrow;
        int pos = row % HW, y = pos / W, x = pos - y * W;
        int m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            m |= ((yy >= 0) && (yy < H) && (xx >= 0) && (xx < W)) ? (1 << tap) : 0;
        }
        vm[t] = row < R ? m : 0;
    }
    const int rowbase = (tbase * TR + jrow) * S4 + gq;
    const int zbase = zu; // multiple of 16 units; the per-lane slot is added per tap
    if constexpr (PREC == 0) {
        const size_t wl = (size_t)C * C * 9;
        for (int l = 0; l < 2 * a.nblocks; l++) {
            const f32x4 *src = (l & 1) ? Y4 : X4;
            f32x4 *dst = (l & 1) ? X4 : Y4;
            if (first) conv_lds_f32<C, NTA>(src, dst, a.tw + (size_t)l * wl, a.tb + l * C, vm, rowbase, zbase, W, R, wave, lane, l & 1, tbase);
            else if constexpr (NTB > 0) conv_lds_f32<C, NTB>(src, dst, a.tw + (size_t)l * wl, a.tb + l * C, vm, rowbase, zbase, W, R, wave, lane, l & 1, tbase);
            __syncthreads();
        }
    } else {
        const f32x4 *tw4 = reinterpret_cast<const f32x4 *>(a.tw);
        const size_t wl = (size_t)C * C * 9 * 2 * 2 / 16; // 16-byte units per layer (hi + lo halves)
        const int NL = 2 * a.nblocks;
        WPre pre;
        WPre pre2[2];
        (void)pre2;
        if (NL > 0) {
            if constexpr (MF == 3) { }
            else if constexpr (MF == 2) { if constexpr (!(VAR & VAR_WLDS)) wpre_load_c2<C>(pre2, tw4, wave, lane); }
#ifdef DBAZ_DEBUG
            else if constexpr (MF == 1) wpre_load32<C>(pre, tw4, wave, lane);
#endif
            else wpre_load<C>(pre, tw4, wave, lane);
        }
        unsigned long long stamps[4] = {0, 0, 0, 0};
        unsigned long long tb0 = 0, tb1 = 0, tk0 = 0, tk1 = 0;
        (void)tb0; (void)tb1; (void)tk0; (void)tk1;
        f32x4 res[2][NTA]; // the residual stream of this wave's outputs (activation-scaled f32): conv_lds_h3_c2 / conv_lds_h3
        (void)res;
        if constexpr (MF == 0 && PREC == 1 && (VAR & VAR_RESREG) != 0) {
            // block 0's input = conv0's output, which other waves wrote: decode this wave's share once (one cout tile, wave & 3)
            const _Float16 *xh = reinterpret_cast<const _Float16 *>(X4);
            const int ctc = wave & 3;
            if (ctc < C / 16) {
#pragma unroll
                for (int t = 0; t < NTA; t++) {
                    const int row = min((tbase + t) * 16 + jrow, R - 1);
                    const _Float16 *ph = xh + (size_t)row * (S4 * 8) + ctc * 16 + gq * 4;
#pragma unroll
                    for (int e = 0; e < 4; e++) res[0][t][e] = (float)ph[e] + (float)ph[C + e];
                }
            }
        }
        if constexpr (MF == 2 && (VAR & VAR_RESREG) != 0) {
            // block 0's input = conv0's output, which other waves wrote: decode this wave's share once
            const _Float16 *xh = reinterpret_cast<const _Float16 *>(X4);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int ctc = (wave & 1) * 2 + c;
                int col = ctc * 16 + gq * 4;
                if constexpr (SWZ) col = (((ctc * 2 + (gq >> 1)) ^ ((jrow >> 2) & 3)) * 8) + (gq & 1) * 4;
#pragma unroll
                for (int t = 0; t < NTA; t++) {
                    const int row = min((tbase + t) * 16 + jrow, R - 1);
                    const _Float16 *ph = xh + (size_t)row * (S4 * 8) + col;
#pragma unroll
                    for (int e = 0; e < 4; e++) res[c][t][e] = (float)ph[e] + (float)ph[C + e];
                }
            }
        }
        f32x4 *wring = nullptr; // VAR_WLDS: two-slot weight ring behind the two activation images
        f32x4 pre8[8];          // MF = 3: the eight weight fragments of a layer's step 0
        f32x4 res4[4][NTA];     // MF = 3: residual stream of the wave's 4 x NTA output tiles
        (void)pre8; (void)res4;
        if constexpr (MF == 3) {
            wring = Y4 + img_units;
            if (NL > 0) {
                const f32x4 *d0 = tw4 + (size_t)wave * (9 * (C / 32)) * 2 * 64 + lane;
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(wring + (size_t)wave * 2 * 64));
                glds16(d0, dst);
                glds16(d0 + 64, dst + 1024);
                glds16(d0 + 128, dst + WRING_UNITS * 16);
                glds16(d0 + 128 + 64, dst + WRING_UNITS * 16 + 1024);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 8; c++) pre8[c] = wring[c * 64 + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            const _Float16 *xh = reinterpret_cast<const _Float16 *>(X4);
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int t = 0; t < NTA; t++) {
                    const int row = min((tbase + t) * 16 + jrow, R - 1);
                    const _Float16 *ph = xh + (size_t)row * (S4 * 8) + c * 16 + gq * 4;
#pragma unroll
                    for (int e = 0; e < 4; e++) res4[c][t][e] = (float)ph[e] + (float)ph[C + e];
                }
        }
        if constexpr (MF == 2 && (VAR & VAR_WLDS) != 0) {
            wring = Y4 + img_units;
            if (NL > 0) {
                const f32x4 *d0 = tw4 + ((size_t)(wave >> 1) * (9 * (C / 32)) * 2 + (wave & 1)) * 64 + lane;
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(wring + (size_t)wave * 64));
                glds16(d0, dst);
                glds16(d0 + 128, dst + WRING_UNITS * 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            {   // layer 0's step 0 out of slot 0, by every wave, before any wave's step 0 refills the slot
                const f32x4 *r0 = wring + (size_t)((wave & 1) * 2) * 2 * 64 + lane;
                pre2[0].h0 = r0[0]; pre2[0].l0 = r0[64]; pre2[1].h0 = r0[128]; pre2[1].l0 = r0[192];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
        if constexpr ((VAR & VAR_PRIO) != 0) {
            if (wave >= 4) __builtin_amdgcn_s_setprio(1);
        }
        STAMP(tk0);
        tL0 = tk0;
        for (int l = 0; l < NL; l++) {
            const f32x4 *src = (l & 1) ? Y4 : X4;
            f32x4 *dst = (l & 1) ? X4 : Y4;
            const f32x4 *nxt = l + 1 < NL ? tw4 + (size_t)(l + 1) * wl : nullptr;
#ifdef DBAZ_DEBUG
            if constexpr (MF == 3) {
                conv_lds_h3_w4<C, NTA>(src, dst, tw4 + (size_t)l * wl, a.tb + l * C, a.tosc[l], vm, zbase, W, R, wave, lane, l & 1, ovf, tbase, pre8, nxt, res4, wring);
            } else
#endif
            if constexpr (MF == 2) {
                conv_lds_h3_c2<C, NTA, VAR>(src, dst, tw4 + (size_t)l * wl, a.tb + l * C, a.tosc[l], vm, rowbase, zbase, W, R, wave, lane, l & 1, ovf, tbase, pre2, nxt, res, stamps, wring);
#ifdef DBAZ_DEBUG
            } else if constexpr (MF && NTB > 0) {
                if (first) conv_lds_h3_32<C, NTA>(src, dst, tw4 + (size_t)l * wl, a.tb + l * C, a.tosc[l], vm, rowbase, zbase, W, R, wave, lane, l & 1, ovf, tbase, pre, nxt, stamps);
                else conv_lds_h3_32<C, NTB>(src, dst, tw4 + (size_t)l * wl, a.tb + l * C, a.tosc[l], vm, rowbase, zbase, W, R, wave, lane, l & 1, ovf, tbase, pre, nxt, stamps);
            } else if constexpr (MF) {
                conv_lds_h3_32<C, NTA>(src, dst, tw4 + (size_t)l * wl, a.tb + l * C, a.tosc[l], vm, rowbase, zbase, W, R, wave, lane, l & 1, ovf, tbase, pre, nxt, stamps);
#endif
            } else {
                constexpr bool RR = (VAR & VAR_RESREG) != 0;
                if (first) conv_lds_h3<C, NTA, RR>(src, dst, tw4 + (size_t)l * wl, a.tb + l * C, a.tosc[l], vm, rowbase, zbase, W, R, wave, lane, l & 1, ovf, tbase, nullptr, nullptr, pre, nxt, res[0], stamps);
                else if constexpr (NTB > 0) conv_lds_h3<C, NTB, RR>(src, dst, tw4 + (size_t)l * wl, a.tb + l * C, a.tosc[l], vm, rowbase, zbase, W, R, wave, lane, l & 1, ovf, tbase, nullptr, nullptr, pre, nxt, reinterpret_cast<f32x4(&)[NTB]>(res[0]), stamps);
            }
            STAMP(tb0);
            __syncthreads();
            STAMP(tb1);
#ifdef DBAZ_STAMP
            stamps[3] += tb1 - tb0;
#endif
        }
        if constexpr ((VAR & VAR_PRIO) != 0) __builtin_amdgcn_s_setprio(0);
#ifdef DBAZ_STAMP
        STAMP(tk1);
        tL1 = tk1;
        if (a.stamp_out && lane == 0) {
            unsigned long long *o = a.stamp_out + ((size_t)blockIdx.x * 8 + wave) * 10;
            o[0] = stamps[0]; o[1] = stamps[1]; o[2] = stamps[2]; o[3] = stamps[3]; o[4] = tk1 - tk0;
        }
#endif
    }
    // ---- head conv1x1 (both heads), results staged in Y as [sample][oc][pos] and written out coalesced
    bool heads_done = false;
    if constexpr (PREC == 1) {
        if (a.hwp) {
            // f16x3 on MFMA, the tower's operand format with a single (centre) tap: wave = (16-output tile,
            // group of position tiles); A = packed weight fragments from L2, B = this tile's own rows
            constexpr int KS = C / 32, LO = C / 8;
            const int OC = 2 * a.hc, n_ct = ((OC + 15) & ~15) / 16, ngrp = (NTHR >> 6) / n_ct;
            const int ct = wave % n_ct, grp = wave / n_ct;
            const int NT = (R + 15) / 16;
            float *stage = Y; // [ns][OC][HW]
            const f32x4 *wp = reinterpret_cast<const f32x4 *>(a.hwp) + (size_t)ct * KS * 2 * 64 + lane;
            u128h ah[KS], al[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) { ah[ks].f = wp[(size_t)ks * 128]; al[ks].f = wp[(size_t)ks * 128 + 64]; }
            const int hj = lane & 15, hq = lane >> 4; // this GEMM stays on the 16x16x32 form
            const int oc0 = ct * 16 + hq * 4;
            f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; r++) if (oc0 + r < OC) bv[r] = a.hb[oc0 + r];
            for (int t = grp; t < NT; t += ngrp) {
                const int row = t * 16 + hj;
                const int rr = min(row, R - 1);
                const f32x4 *bp = X4 + (size_t)rr * S4 + (SWZ ? (hq ^ ((rr >> 2) & 3)) : hq);
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    u128h bh, bl;
                    bh.f = bp[ks * 4];
                    bl.f = bp[ks * 4 + LO];
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks].h, bh.h, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks].h, bh.h, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks].h, bl.h, acc, 0, 0, 0);
                }
                if (row < R) {
                    const int sidx = row / HW, pp = row - sidx * HW;
                    const f32x4 v = acc * a.hosc + bv;
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (oc0 + r < OC) stage[(sidx * OC + oc0 + r) * HW + pp] = fmaxf(v[r], 0.0f);
                }
            }
            // K may not be a multiple of the GEMM's 16-wide chunks (head_channels * HW): the last chunk reads up to 15 floats
            // behind the last sample's rows -- their weights are zero, the values only have to be finite
            if (tid < 16) stage[(size_t)ns * OC * HW + tid] = 0.0f;
            __syncthreads();
            head_fc_fused(g, a, stage, stage + (size_t)S * OC * HW + 16, ns, s0, tid, NTHR);
            heads_done = true;
        }
    }
    if (!heads_done) {
        const int OC = 2 * a.hc;
        float *wl = Y;                         // [OC][C+4]
        float *stage = Y + OC * (C + 4);       // [ns][OC][HW]
        for (int i = tid; i < OC * C; i += NTHR) {
            int o = i / C, c = i - o * C;
            wl[o * (C + 4) + c] = a.hw[i];
        }
        __syncthreads();
        // one thread = one row: the row's C activations are read once into registers, the
        // weights are wave-uniform (broadcast) float4 reads
        if (tid < R) {
            const int row = tid;
            float xr[C];
#pragma unroll
            for (int c = 0; c < C; c++) xr[c] = act_load<C, PREC, STRIDE, SWZ>(X, row, c);
            const int sidx = row / HW, p = row - sidx * HW;
            for (int oc = 0; oc < OC; oc++) {
                float acc = a.hb[oc];
                const float *wr = wl + oc * (C + 4);
#pragma unroll
                for (int c = 0; c < C; c += 4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wr + c);
                    acc += xr[c] * w4[0] + xr[c + 1] * w4[1] + xr[c + 2] * w4[2] + xr[c + 3] * w4[3];
                }
                stage[(sidx * OC + oc) * HW + p] = fmaxf(acc, 0.0f);
            }
        }
        if (tid < 16) stage[(size_t)ns * OC * HW + tid] = 0.0f;
        __syncthreads();
        head_fc_fused(g, a, stage, stage + (size_t)S * OC * HW + 16, ns, s0, tid, NTHR);
    }
    if (PREC == 1 && ovf) {
        // (every thread that saw it says so; the stores are idempotent)
        for (int i = 0; i < ns; i++) a.ovf_flags[s0 + i] = 1;
        atomicOr(a.overflow, 1);
    }
#ifdef DBAZ_STAMP
    STAMP(tE1);
    if (PREC == 1 && a.stamp_out && lane == 0) {
        unsigned long long *o = a.stamp_out + ((size_t)blockIdx.x * 8 + wave) * 10;
        o[5] = tL0 - tE0; // conv0 phase (staging, VALU conv, zero regions)
        o[6] = tE1 - tL1; // head conv1x1 phase + output
        o[7] = tE1 - tE0; // whole workgroup
        o[8] = __builtin_amdgcn_s_memrealtime() - tR0; // the same interval in 100 MHz ticks: clock = o[7] / o[8] * 100 MHz
    }
#endif
}

// which launch takes the samples behind the last full round of the main launch: 0 = the main launch itself, 1..4 = one
// round of workgroups with S_small / S_mid / S_big / S_huge samples each (every launch derives this from n on the device)
__device__ __forceinline__ int tower_split(const TowerArgs &a, int n, int &n_full)
{
    const int per_round = a.cus * a.S_main;
    n_full = per_round > 0 ? (n / per_round) * per_round : 0;
    const int tail = n - n_full;
    if (tail <= 0) return 0;
    if (a.S_small > 0 && tail <= a.cus * a.S_small) return 1;
    if (a.S_mid > 0 && tail <= a.cus * a.S_mid) return 2;
    if (a.S_big > 0 && tail <= a.cus * a.S_big) return 3;
    if (a.S_huge > 0 && tail <= a.cus * a.S_huge) return 4; // (main = two cout tiles per wave: 4 = one round of the one-cout-tile kernel)
    return 0;
}

template <int C, int NTA, int NTB, int PREC, int MF = 0, int VAR = 0>
__global__ void __launch_bounds__(MF == 3 ? 256 : CONV_THREADS, 1) k_tower(Geo g, TowerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = cut_n(*a.n_dev, a.cut_round, a.cut_defer);
    const int S = a.S;
    // Every workgroup of a launch takes the same time, so the launch costs ceil(workgroups / CUs) rounds and a
    // nearly empty last round costs a full one.  The samples beyond the last FULL round of the main launch are
    // therefore left to a tail launch with fewer samples (and position tiles) per workgroup whenever one round of
    // those smaller, faster workgroups can take them; every launch derives the split from n on the device.
    int first_sample = 0, limit = n;
    {
        int n_full;
        const int mode = tower_split(a, n, n_full);
        if (a.role == 0) {
            if (mode) limit = n_full;
        } else {
            if (mode != a.role) return;
            first_sample = n_full;
        }
    }
    if constexpr (PREC == 0) {
        if (a.fallback) {
            // f16x3 safety net: this exact-f32 launch follows the f16x3 launches of the same batch and redoes the sample
            // groups in which a workgroup saw an activation beyond f16's range (none, normally).  It is ONE round of
            // workgroups; thread t of workgroup b looks at group b + t * gridDim.x, the flagged ones are then redone one
            // after the other (a full grid of idle 114-KB-LDS workgroups cost 17 us per step just to be dispatched).
            __shared__ int n_redo;
            __shared__ int redo_grp[CONV_THREADS];
            if (threadIdx.x == 0) n_redo = 0;
            __syncthreads();
            {
                const int gt = blockIdx.x + threadIdx.x * gridDim.x;
                const long long st = (long long)gt * S;
                if (st < n) {
                    int any = 0;
                    const int m = min(S, n - (int)st);
                    for (int i = 0; i < m; i++) any |= a.ovf_flags[st + i];
                    if (any) {
                        for (int i = 0; i < m; i++) a.ovf_flags[st + i] = 0;
                        redo_grp[atomicAdd(&n_redo, 1)] = gt;
                        atomicAdd(a.overflow + 1, m);
                    }
                }
            }
            __syncthreads();
            const int nr = n_redo;
            for (int r = 0; r < nr; r++) {
                const int s0 = redo_grp[r] * S;
                tower_group<C, NTA, NTB, PREC, MF, VAR>(g, a, lds, S, s0, min(S, n - s0));
                __syncthreads();
            }
            return;
        }
    }
    if (a.n_used && a.role == 0 && !a.fallback && blockIdx.x == 0 && threadIdx.x == 0) *a.n_used = n;
    const int s0 = first_sample + blockIdx.x * S;
    if (s0 >= limit) return;
    tower_group<C, NTA, NTB, PREC, MF, VAR>(g, a, lds, S, s0, min(S, limit - s0));
}

// The remainder of a batch in ONE launch (f16x3, one-cout-tile tiling): the workgroups pick the size the split asks for --
// <2,2> / <4,4> / <5,5> / <7,6> tiles per wave half, S_small / S_mid / S_big / S_huge samples -- instead of four launches of
// which three leave at once (5 us each: 1 % of a 6x6 step, 4 % of a 3x3 step).
template <int C, int RR = 0>
__global__ void __launch_bounds__(CONV_THREADS, 1) k_tower_rem(Geo g, TowerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = cut_n(*a.n_dev, a.cut_round, a.cut_defer);
    int n_full;
    const int mode = tower_split(a, n, n_full);
    if (mode == 0) return;
    const int S = mode == 1 ? a.S_small : mode == 2 ? a.S_mid : mode == 3 ? a.S_big : a.S_huge;
    const int s0 = n_full + blockIdx.x * S;
    if (s0 >= n) return;
    const int ns = min(S, n - s0);
    // RR = 1: the bodies beside a two-cout-tile main launch keep the residual stream in registers like it (the 7-tile body then
    // spills 28 registers around its main loop: +2 % on that body, which as a remainder round still beats a partial round of the
    // main launch by 8 %; where the 7-tile body IS the main launch -- 9x9 -- every body of the geometry stays RR = 0)
    if (mode == 1) tower_group<C, 2, 2, 1, 0, RR>(g, a, lds, S, s0, ns);
    else if (mode == 2) tower_group<C, 4, 4, 1, 0, RR>(g, a, lds, S, s0, ns);
    else if (mode == 3) tower_group<C, 5, 5, 1, 0, RR>(g, a, lds, S, s0, ns);
    else tower_group<C, 7, 6, 1, 0, RR>(g, a, lds, S, s0, ns);
}

// ------------------------------------------------------------------------------------
// SimpleNN trunk (dots_boxes_nn.py:85-91, 3x3 boards): bn_i(relu(conv_i(x))) for conv0 (3->256)
// and conv1..conv4 (256->256), all LDS-resident like k_tower.  conv4 is unpadded in the
// reference (4x4 -> 2x2); it is evaluated as the padded conv and only the four inner positions
// -- whose taps all lie inside the image, so the two agree exactly -- are gathered into the
// flatten order x.view(n, -1) = [c*4 + i*2 + j].
// ------------------------------------------------------------------------------------
struct SimpleArgs {
    const float *feat;
    const int32_t *list, *n_dev;
    const float *w0, *b0, *s0, *t0; // conv0 [27][C], bias, post affine (t0 unscaled)
    const float *tw, *tb, *ts, *tt; // conv1..4 packed weights, bias, post scale, post shift
    const float *tosc;              // f16x3 per-layer output scale
    float *flat;                    // out [sample][1024]
    int *overflow;
    int S;
};

template <int PREC>
__global__ void __launch_bounds__(CONV_THREADS, 1) k_simple_trunk(Geo g, SimpleArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int C = 256, NTH = 2;
    constexpr int STRIDE = C + 8, S4 = STRIDE / 4;
    const int n = *a.n_dev;
    const int S = a.S;
    const int s0 = blockIdx.x * S;
    if (s0 >= n) return;
    const int HW = g.HW, W = g.W, H = g.H; // 16, 4, 4
    const int ns = min(S, n - s0);
    const int R = ns * HW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int zu = (S * HW * S4 + 15) & ~15;
    const int img_units = zu + 3 * S4;
    float *X = lds;
    float *Y = lds + (size_t)img_units * 4;
    f32x4 *X4 = reinterpret_cast<f32x4 *>(X);
    f32x4 *Y4 = reinterpret_cast<f32x4 *>(Y);
    bool ovf = false;
    unsigned long long tE0 = 0, tE1 = 0, tL0 = 0, tL1 = 0, tR0 = 0;
    (void)tE0; (void)tE1; (void)tL0; (void)tL1; (void)tR0;
    STAMP(tE0);
#ifdef DBAZ_STAMP
    tR0 = __builtin_amdgcn_s_memrealtime();
#endif
    {
        const int PW = W + 2, PH = H + 2, PP = 3 * PH * PW;
        float *pad = Y;
        float *wl = Y + S * PP;
        for (int i = tid; i < ns * PP; i += CONV_THREADS) pad[i] = 0.0f;
        for (int i = tid; i < 27 * C; i += CONV_THREADS) wl[i] = a.w0[i];
        __syncthreads();
        for (int i = tid; i < ns * 3 * HW; i += CONV_THREADS) {
            int sidx = i / (3 * HW), r = i - sidx * 3 * HW;
            int c = r / HW, p = r - c * HW, y = p / W, x = p - y * W;
            const int slot = a.list ? a.list[s0 + sidx] : s0 + sidx;
            pad[sidx * PP + (c * PH + y + 1) * PW + x + 1] = a.feat[(size_t)slot * 3 * HW + r]; // no bn_input in SimpleNN
        }
        __syncthreads();
        for (int i = tid; i < R * (C / 16); i += CONV_THREADS) {
            const int row = i % R, cq = i / R;
            const int sidx = row / HW, p = row - sidx * HW, y = p / W, x = p - y * W;
            const float *pp = pad + sidx * PP;
            float in27[27];
#pragma unroll
            for (int tap = 0; tap < 9; tap++)
#pragma unroll
                for (int c = 0; c < 3; c++) in27[tap * 3 + c] = pp[(c * PH + y + tap / 3) * PW + x + tap % 3];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = cq * 16 + q * 4;
                f32x4 acc = *reinterpret_cast<const f32x4 *>(a.b0 + co);
#pragma unroll
                for (int k = 0; k < 27; k++) acc += in27[k] * *reinterpret_cast<const f32x4 *>(wl + k * C + co);
#pragma unroll
                for (int e = 0; e < 4; e++)
                    act_store<C, PREC>(X, row, co + e, fmaxf(acc[e], 0.0f) * a.s0[co + e] + a.t0[co + e], ovf);
            }
        }
        __syncthreads();
        if (tid < 3 * S4) {
            X4[zu + tid] = (f32x4){0.f, 0.f, 0.f, 0.f};
            Y4[zu + tid] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    const int jrow = lane & 15, gq = lane >> 4;
    const int tbase = (wave < 4) ? 0 : NTH;
    int vm[NTH];
#pragma unroll
    for (int t = 0; t < NTH; t++) {
        int row = (tbase + t) * 16 + jrow;
        int pos = row % HW, y = pos / W, x = pos - y * W;
        int m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            m |= ((yy >= 0) && (yy < H) && (xx >= 0) && (xx < W)) ? (1 << tap) : 0;
        }
        vm[t] = row < R ? m : 0;
    }
    const int rowbase = (tbase * 16 + jrow) * S4 + gq;
    for (int l = 0; l < 4; l++) {
        const f32x4 *src = (l & 1) ? Y4 : X4;
        f32x4 *dst = (l & 1) ? X4 : Y4;
        if constexpr (PREC == 0) {
            conv_lds_f32<C, NTH>(src, dst, a.tw + (size_t)l * C * C * 9, a.tb + l * C, vm, rowbase, zu, W, R, wave, lane, 0, tbase,
                                 a.ts + l * C, a.tt + l * C);
        } else {
            const size_t wl16 = (size_t)C * C * 9 * 2 * 2 / 16;
            const f32x4 *tw4 = reinterpret_cast<const f32x4 *>(a.tw);
            WPre pre;
            wpre_load<C>(pre, tw4 + (size_t)l * wl16, wave, lane);
            f32x4 res_unused[NTH]; // (no residual connections in SimpleNN)
            conv_lds_h3<C, NTH>(src, dst, tw4 + (size_t)l * wl16, a.tb + l * C, a.tosc[l], vm, rowbase, zu, W, R, wave, lane, 0,
                                ovf, tbase, a.ts + l * C, a.tt + l * C, pre, nullptr, res_unused);
        }
        __syncthreads();
    }
    // after 4 layers the result is back in X; gather the inner 2x2 of every sample
    for (int i = tid; i < ns * 1024; i += CONV_THREADS) {
        const int sidx = i >> 10, f = i & 1023, c = f >> 2, q = f & 3;
        const int row = sidx * HW + (1 + (q >> 1)) * W + 1 + (q & 1);
        a.flat[(size_t)(s0 + sidx) * 1024 + f] = act_load<C, PREC>(X, row, c);
    }
    if (PREC == 1 && ovf) atomicOr(a.overflow, 1);
}

// Dense layer  out[n][O] = post(relu(in[n][K] . W[O][K]^T + b))  on f32 MFMA, 16 samples per
// workgroup (same GEMM core as k_head_fc).  dup = 1 writes the result twice per sample
// ([n][2][O]): the layout k_head_fc expects when both heads read the same vector.
struct DenseArgs {
    const int32_t *n_dev;
    const float *in;        // [n][K]
    const float *w;         // packed [O/16][K/16][64][4]
    const float *b, *ps, *pt;
    float *out;
    int K, O, RS4, dup;
};

__global__ void __launch_bounds__(256) k_dense(DenseArgs h)
{
    extern __shared__ __attribute__((aligned(16))) float ldsd[];
    const int n = *h.n_dev;
    const int j0 = blockIdx.x * 16;
    if (j0 >= n) return;
    const int ns = min(16, n - j0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int RS = h.RS4 * 4, K = h.K, KC = K / 16, NJ = h.O / 16;
    float *act = ldsd;
    for (int i = tid; i < 16 * K; i += 256) {
        int sidx = i / K, k = i - sidx * K;
        act[sidx * RS + k] = sidx < ns ? h.in[(size_t)(j0 + sidx) * K + k] : 0.0f;
    }
    __syncthreads();
    const int jrow = lane & 15, gq = lane >> 4;
    const f32x4 *act4 = reinterpret_cast<const f32x4 *>(act);
    for (int job = wave; job < NJ; job += 4) {
        const f32x4 *wb = reinterpret_cast<const f32x4 *>(h.w) + (size_t)job * KC * 64 + lane;
        const f32x4 *bb = act4 + jrow * h.RS4 + gq;
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        for (int kc = 0; kc < KC; kc += 2) {
            const f32x4 a0 = wb[(size_t)kc * 64], a1 = wb[(size_t)(kc + 1) * 64];
            const f32x4 b0 = bb[kc * 4], b1 = bb[(kc + 1) * 4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b1[e], acc1, 0, 0, 0);
            }
        }
        const f32x4 acc = acc0 + acc1;
        if (jrow < ns) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = job * 16 + gq * 4 + r;
                const float v = fmaxf(acc[r] + h.b[o], 0.0f) * h.ps[o] + h.pt[o];
                if (h.dup) {
                    h.out[((size_t)(j0 + jrow) * 2) * h.O + o] = v;
                    h.out[((size_t)(j0 + jrow) * 2 + 1) * h.O + o] = v;
                } else {
                    h.out[(size_t)(j0 + jrow) * h.O + o] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// heads
// ------------------------------------------------------------------------------------
// Head FCs as one batched GEMM on f32 MFMA: a workgroup takes 16 samples;
//   Out^T[out][sample] = Wfc[out][k] * hact^T[k][sample]
// with A = FC weights pre-packed in fragment order ([job][k/16][lane][4], streamed from L2) and
// B = the 16 samples' head activations staged in LDS (row stride = 2 mod 16 float4 units:
// conflict-free ds_read_b128).  Jobs 0..ntp-1 are 16-output tiles of the policy FC, jobs
// ntp.. of the value FC0.  Then softmax over the A logits (= exp(log_softmax), nn.py:159) and
// tanh(FC1(relu(FC0))) per sample.
struct HeadArgs {
    const int32_t *list, *n_dev;
    const float *hact;      // [sample][2][K]
    const float *wfc;       // packed [ntp+ntv][KC][64][4]
    const float *bfc;       // [(ntp+ntv)*16] bias per GEMM output (0 for padding)
    const float *wv1, *bv1; // value FC1 [vf], [1]
    float *P, *V;
    int K, KP /*K padded to 16*/, RS4 /*LDS row stride, float4 units*/, ntp, ntv, vf, AS;
    int value_direct; // SimpleNN: value = tanh(value_fc(x)), no hidden layer
    int cut_round, cut_defer; // cut_n's rule (0: every leaf); n_used (optional) receives the count the step's network took
    int32_t *n_used;
};

// K is split over the 4 waves (wave w owns the 16-wide k chunks kc = w, w+4, ...).  Nothing is staged:
// a lane reads its activation fragment (16 B of sample row lane&15) and its weight fragments straight
// from global memory / L2 in MFMA operand layout, three chunks ahead of the MFMAs (a register ring
// of depth 3 hides the L2 latency that a staged, one-deep version paid once per chunk); the
// policy tiles and the value tile share one K loop (8 output tiles per pass).  The four partial
// sums meet in LDS and are added in wave order, so a sample's result does not depend on the
// batch it is evaluated in.  Rows >= ns compute on a clamped (valid) row and are discarded.
#define HEAD_JG 8
#define HEAD_MT 1 // sample tiles (16 samples each) per workgroup (2 was measured: 45 vs 36 us -- one wave per SIMD hides less latency)
struct HeadFrag {
    f32x4 a[HEAD_JG];
    f32x4 b0[HEAD_MT], b1[HEAD_MT];
};

__device__ __forceinline__ void head_load(HeadFrag &f, const f32x4 *wfc4, const float *const (&row0)[HEAD_MT], int K, int KC, int kc,
                                          int jg, int nj, int lane, int gq)
{
#pragma unroll
    for (int j = 0; j < HEAD_JG; j++)
        if (j < nj) f.a[j] = wfc4[((size_t)(jg + j) * KC + kc) * 64 + lane];
#pragma unroll
    for (int m = 0; m < HEAD_MT; m++) {
        f.b0[m] = *reinterpret_cast<const f32x4 *>(row0[m] + kc * 16 + gq * 4);
        f.b1[m] = *reinterpret_cast<const f32x4 *>(row0[m] + K + kc * 16 + gq * 4);
    }
}

__device__ __forceinline__ void head_mfma(f32x4 (&acc)[HEAD_MT][HEAD_JG], const HeadFrag &f, int jg, int nj, int ntp)
{
#pragma unroll
    for (int j = 0; j < HEAD_JG; j++)
        if (j < nj) {
#pragma unroll
            for (int m = 0; m < HEAD_MT; m++) {
                const f32x4 b = (jg + j < ntp) ? f.b0[m] : f.b1[m]; // wave-uniform select
#pragma unroll
                for (int e = 0; e < 4; e++) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[j][e], b[e], acc[m][j], 0, 0, 0);
            }
        }
}

__global__ void __launch_bounds__(256) k_head_fc(Geo g, HeadArgs h)
{
    extern __shared__ __attribute__((aligned(16))) float ldsf[];
    constexpr int SPW = 16 * HEAD_MT; // samples per workgroup
    const int n = cut_n(*h.n_dev, h.cut_round, h.cut_defer);
    if (h.n_used && blockIdx.x == 0 && threadIdx.x == 0) *h.n_used = n; // k_expand_backup: list positions >= n ask again next step
    const int j0 = blockIdx.x * SPW;
    if (j0 >= n) return;
    const int ns = min(SPW, n - j0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KP = h.KP, K = h.K, A = g.A;
    const int NJ = h.ntp + h.ntv, LGS = NJ * 16 + 1;
    float *part = ldsf;                          // [4 waves][SPW][NJ*16] partial sums
    float *lg = ldsf + 4 * SPW * NJ * 16;        // [SPW][LGS]
    const int jrow = lane & 15, gq = lane >> 4;
    const int KC = KP / 16;
    const f32x4 *wfc4 = reinterpret_cast<const f32x4 *>(h.wfc);
    const float *row0[HEAD_MT];
#pragma unroll
    for (int m = 0; m < HEAD_MT; m++) row0[m] = h.hact + ((size_t)(j0 + min(m * 16 + jrow, ns - 1)) * 2) * K;
    const int nk = wave < KC ? (KC - wave + 3) / 4 : 0; // chunks of this wave
    for (int jg = 0; jg < NJ; jg += HEAD_JG) {
        const int nj = min(HEAD_JG, NJ - jg);
        f32x4 acc[HEAD_MT][HEAD_JG];
#pragma unroll
        for (int m = 0; m < HEAD_MT; m++)
#pragma unroll
            for (int j = 0; j < HEAD_JG; j++) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        HeadFrag f0, f1, f2;
        if (0 < nk) head_load(f0, wfc4, row0, K, KC, wave, jg, nj, lane, gq);
        if (1 < nk) head_load(f1, wfc4, row0, K, KC, wave + 4, jg, nj, lane, gq);
        for (int i = 0; i < nk; i += 3) {
            if (i + 2 < nk) head_load(f2, wfc4, row0, K, KC, wave + 4 * (i + 2), jg, nj, lane, gq);
            head_mfma(acc, f0, jg, nj, h.ntp);
            if (i + 1 < nk) {
                if (i + 3 < nk) head_load(f0, wfc4, row0, K, KC, wave + 4 * (i + 3), jg, nj, lane, gq);
                head_mfma(acc, f1, jg, nj, h.ntp);
            }
            if (i + 2 < nk) {
                if (i + 4 < nk) head_load(f1, wfc4, row0, K, KC, wave + 4 * (i + 4), jg, nj, lane, gq);
                head_mfma(acc, f2, jg, nj, h.ntp);
            }
        }
#pragma unroll
        for (int m = 0; m < HEAD_MT; m++) {
            float *pw = part + ((size_t)wave * SPW + m * 16 + jrow) * (NJ * 16);
#pragma unroll
            for (int j = 0; j < HEAD_JG; j++)
                if (j < nj) *reinterpret_cast<f32x4 *>(pw + (jg + j) * 16 + gq * 4) = acc[m][j];
        }
    }
    __syncthreads();
    for (int i = tid; i < SPW * NJ * 16; i += 256) {
        const int row = i / (NJ * 16), o = i - row * (NJ * 16);
        const float *p0 = part + i;
        const int st = SPW * NJ * 16;
        lg[row * LGS + o] = ((p0[0] + p0[st]) + (p0[2 * st] + p0[3 * st])) + h.bfc[o];
    }
    __syncthreads();
    for (int sidx = wave; sidx < ns; sidx += 4) {
        const int dst = h.list ? h.list[j0 + sidx] : j0 + sidx;
        const float *l = lg + sidx * LGS;
        float mx = -INFINITY;
        for (int o = lane; o < A; o += 64) mx = fmaxf(mx, l[o]);
        for (int s = 32; s > 0; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s));
        float sum = 0.0f;
        for (int o = lane; o < A; o += 64) sum += expf(l[o] - mx);
        for (int s = 32; s > 0; s >>= 1) sum += __shfl_xor(sum, s);
        for (int o = lane; o < A; o += 64) h.P[(size_t)dst * h.AS + o] = expf(l[o] - mx) / sum;
        float hv = 0.0f;
        for (int u = lane; u < h.vf; u += 64) hv += fmaxf(l[h.ntp * 16 + u], 0.0f) * h.wv1[u];
        for (int s = 32; s > 0; s >>= 1) hv += __shfl_xor(hv, s);
        if (lane == 0) h.V[dst] = h.value_direct ? tanhf(l[h.ntp * 16]) : tanhf(hv + h.bv1[0]);
    }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
template <typename T>
static T *nn_alloc(NNState *nn, size_t count)
{
    void *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16)) != hipSuccess) return nullptr;
    nn->allocs.push_back(p);
    return (T *)p;
}
template <typename T>
static T *nn_upload(NNState *nn, const std::vector<T> &h)
{
    T *d = nn_alloc<T>(nn, h.size());
    if (d && !h.empty()) (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

NNState *nn_create(const Geo &g, int max_batch, int precision, bool no_fallback)
{
    NNState *nn = new NNState();
    nn->g = g;
    nn->max_batch = max_batch;
    nn->no_fallback = no_fallback;
    nn->precision = precision >= 1 ? 1 : 0;
    // 1 = f16x3 on the default tiling (two cout tiles per wave for 64-channel networks, one otherwise)
    nn->want_c2 = precision == 1;
#ifdef DBAZ_DEBUG
    // A/B tilings (debug build): 2 = the arithmetic of 1 on the 32x32x16 MFMA; 3 / 4 = two / one cout tile(s) per wave explicitly
    // (1, 3 and 4 give bit-identical results)
    nn->want_mf32 = precision == 2;
    nn->want_c2 = precision == 1 || precision == 3 || precision >= 5;
    // 5..9: the two-cout-tile kernel with VAR = 1 (register residual), 2 (swizzled columns), 3 (both), 8 (s_setprio), 11 (all)
    // 10, 11: the weight fragments through an LDS ring (VAR 4), and that with the register residual (VAR 5)
    // 12, 13: VAR 5 with every lo half zero (timing bound, wrong results) / with 8-bit lo halves
    // 14: the four-wave kernel (conv_lds_h3_w4)
    static const int var_of[] = {1, 2, 3, 8, 11, 4, 5, 21, 37, 64};
    if (precision == 3) nn->variant = 0; // round 2's kernel: weights L2 -> registers per wave, residual decoded from LDS
    if (precision >= 5 && precision <= 14) nn->variant = var_of[precision - 5];
#endif
    return nn;
}

static void nn_free_device(NNState *nn)
{
    for (void *p : nn->allocs) (void)hipFree(p);
    nn->allocs.clear();
    nn->ready = false;
}

void nn_destroy(NNState *nn)
{
    if (!nn) return;
    nn_free_device(nn);
    delete nn;
}

bool nn_ready(const NNState *nn) { return nn && nn->ready; }

int nn_configure(NNState *nn, int kind, int channels, int blocks, int head_channels, int value_fc, std::string &err)
{
    if (kind == DBAZ_EVAL_SIMPLENN) {
        // hard-wired to 3x3 boards in the reference (fc0 in = 1024, policy_fc out = 32; dots_boxes_nn.py:76,83)
        if (nn->g.rows != 3 || nn->g.cols != 3) { err = "SimpleNN is defined for 3x3 boards only"; return DBAZ_EINVAL; }
        nn_free_device(nn);
        nn->sd.clear();
        nn->kind = kind; nn->C = 256; nn->Craw = 256; nn->blocks = 0; nn->hc = 0; nn->vf = 1;
        return DBAZ_OK;
    }
    if (kind != DBAZ_EVAL_RESNET) { err = "kind must be DBAZ_EVAL_RESNET or DBAZ_EVAL_SIMPLENN"; return DBAZ_EINVAL; }
    if (channels < 1 || channels > 128) { err = "channels must be in 1..128"; return DBAZ_EINVAL; }
    if (blocks < 0 || head_channels < 1 || value_fc < 1 || value_fc > 64) { err = "bad network shape"; return DBAZ_EINVAL; }
    nn_free_device(nn);
    nn->sd.clear();
    // the MFMA tile wants 16 | C: narrower nets run zero-padded (padded channels stay exactly 0)
    int cp = channels <= 16 ? 16 : channels <= 32 ? 32 : channels <= 64 ? 64 : 128;
    if (nn->precision == 1 && cp < 32) cp = 32; // K = 32 per f16 MFMA step
    nn->kind = kind; nn->C = cp; nn->Craw = channels; nn->blocks = blocks; nn->hc = head_channels; nn->vf = value_fc;
    nn->mf32 = (nn->precision == 1 && nn->want_mf32 && (cp == 64 || cp == 128)) ? 1 : 0; // other widths stay on 16x16x32
    nn->c2 = (nn->precision == 1 && nn->want_c2 && !nn->mf32 && cp == 64) ? 1 : 0;
    return DBAZ_OK;
}

int nn_set_tensor(NNState *nn, const char *key, const float *data, int64_t numel, std::string &err)
{
    if (nn->kind == 0) { err = "dbaz_nn_configure not called"; return DBAZ_ESTATE; }
    if (numel < 0) { err = "negative numel"; return DBAZ_EINVAL; }
    nn->sd[key] = std::vector<float>(data, data + numel);
    nn->ready = false;
    return DBAZ_OK;
}

static const std::vector<float> *sd_get(NNState *nn, const std::string &k, size_t numel, std::string &err)
{
    auto it = nn->sd.find(k);
    if (it == nn->sd.end()) { err = "missing state_dict entry '" + k + "'"; return nullptr; }
    if (it->second.size() != numel) {
        err = "state_dict entry '" + k + "' has " + std::to_string(it->second.size()) + " elements, expected " + std::to_string(numel);
        return nullptr;
    }
    return &it->second;
}

// eval-mode BatchNorm as y = x*s + t  (eps = 1e-5, torch default; nn.py:20,45,47)
static bool bn_affine(NNState *nn, const std::string &p, int n, std::vector<double> &s, std::vector<double> &t, std::string &err)
{
    auto w = sd_get(nn, p + ".weight", n, err); if (!w) return false;
    auto b = sd_get(nn, p + ".bias", n, err); if (!b) return false;
    auto m = sd_get(nn, p + ".running_mean", n, err); if (!m) return false;
    auto v = sd_get(nn, p + ".running_var", n, err); if (!v) return false;
    s.resize(n); t.resize(n);
    for (int i = 0; i < n; i++) {
        s[i] = (double)(*w)[i] / sqrt((double)(*v)[i] + 1e-5);
        t[i] = (double)(*b)[i] - (double)(*m)[i] * s[i];
    }
    return true;
}

// conv3x3 [C][C][3][3] + following BN -> packed fragment order [C/16][9][C/16][64][4]
static bool pack_conv(NNState *nn, const std::string &conv, const std::string &bn, int C, std::vector<float> &pk_all,
                      std::vector<float> &bias_all, std::string &err, bool fold = true, std::vector<float> *pk32_all = nullptr,
                      std::vector<float> *bias32_all = nullptr)
{
    const int Cr = nn->Craw;
    auto w = sd_get(nn, conv + ".weight", (size_t)Cr * Cr * 9, err); if (!w) return false;
    auto b = sd_get(nn, conv + ".bias", Cr, err); if (!b) return false;
    std::vector<double> s, t;
    if (fold) {
        if (!bn_affine(nn, bn, Cr, s, t, err)) return false;
    } else {
        s.assign(Cr, 1.0);
        t.assign(Cr, 0.0);
    }
    const int KC = C / 16;
    std::vector<float> pk((size_t)C * C * 9, 0.0f), bias(C, 0.0f);
    for (int ct = 0; ct < C / 16; ct++)
        for (int tap = 0; tap < 9; tap++)
            for (int kc = 0; kc < KC; kc++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 4; e++) {
                        int co = ct * 16 + (lane & 15), ci = kc * 16 + 4 * (lane >> 4) + e;
                        if (co >= Cr || ci >= Cr) continue;
                        double v = (double)(*w)[((size_t)co * Cr + ci) * 9 + tap] * s[co];
                        pk[((((size_t)ct * 9 + tap) * KC + kc) * 64 + lane) * 4 + e] = (float)v;
                    }
    for (int co = 0; co < Cr; co++) bias[co] = (float)((double)(*b)[co] * s[co] + t[co]);
    if (pk32_all) pk32_all->insert(pk32_all->end(), pk.begin(), pk.end());
    if (bias32_all) bias32_all->insert(bias32_all->end(), bias.begin(), bias.end());
    if (nn->precision == 1) {
        // f16x3: folded weights scaled by 2^sw so that max|w| lands in [2^13, 2^14), split into
        // (hi, lo) halves, packed [ct][tap][ks][hi|lo][lane][8]; bias carries the activation scale
        double mx = 0;
        for (int co = 0; co < Cr; co++)
            for (int ci = 0; ci < Cr; ci++)
                for (int tap = 0; tap < 9; tap++) mx = std::max(mx, fabs((double)(*w)[((size_t)co * Cr + ci) * 9 + tap] * s[co]));
        int sw = 0;
        if (mx > 0) { int e; frexp(mx, &e); sw = 14 - e; }
        if (sw > 24) sw = 24;
        if (sw < -24) sw = -24;
        const double wscale = ldexp(1.0, sw);
        // fragment order of the layer MFMA: 16x16x32 (cout = lane & 15, k = 8 (lane >> 4) + e of a 32-wide step) or, on the
        // 32x32x16 tiling, cout = lane & 31, k = 8 (lane >> 5) + e of a 16-wide step; packed [ct][tap][ks][hi|lo][lane][8]
        const bool mf = nn->mf32 && C % 32 == 0;
        const int TM = mf ? 32 : 16, TK = mf ? 16 : 32, LS = mf ? 5 : 4;
        const int KS = C / TK;
        std::vector<_Float16> hp((size_t)C * C * 9 * 2, (_Float16)0.0f);
        for (int ct = 0; ct < C / TM; ct++)
            for (int tap = 0; tap < 9; tap++)
                for (int ks = 0; ks < KS; ks++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int e = 0; e < 8; e++) {
                            int co = ct * TM + (lane & (TM - 1)), ci = ks * TK + 8 * (lane >> LS) + e;
                            if (co >= Cr || ci >= Cr) continue;
                            float v = (float)((double)(*w)[((size_t)co * Cr + ci) * 9 + tap] * s[co] * wscale);
                            _Float16 h = (_Float16)v;
                            _Float16 l = (_Float16)(v - (float)h);
                            if (nn->variant & 16) l = (_Float16)0.0f;            // debug variants VAR_LO0 / VAR_LO8 (see conv_lds_h3_c2)
                            if (nn->variant & 32) { unsigned short u; memcpy(&u, &l, 2); u &= 0xFFF8; memcpy(&l, &u, 2); }
                            size_t base = ((((size_t)ct * 9 + tap) * KS + ks) * 2) * 64 * 8;
                            hp[base + (size_t)lane * 8 + e] = h;
                            hp[base + 64 * 8 + (size_t)lane * 8 + e] = l;
                        }
        const float *as_f = reinterpret_cast<const float *>(hp.data());
        pk_all.insert(pk_all.end(), as_f, as_f + hp.size() / 2);
        for (int co = 0; co < C; co++) bias[co] *= ACT_SCALE;
        bias_all.insert(bias_all.end(), bias.begin(), bias.end());
        nn->osc_host.push_back((float)ldexp(1.0, -sw)); // acc = 2^(sw+ACT_SHIFT) * sum ; keep 2^ACT_SHIFT
        return true;
    }
    pk_all.insert(pk_all.end(), pk.begin(), pk.end());
    bias_all.insert(bias_all.end(), bias.begin(), bias.end());
    return true;
}

// launches (or, with attr_only, raises the dynamic-LDS limit of) the instantiation for (C, NTT, PREC)
template <int C, int NTA, int NTB>
static hipError_t tower_inst(NNState *nn, hipStream_t s, const TowerArgs &ta, int grid, bool attr_only, int prec)
{
    if constexpr (C >= 32) {
        if (prec == 1) {
            if (attr_only)
                return hipFuncSetAttribute((const void *)k_tower<C, NTA, NTB, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nn->conv_lds);
            hipLaunchKernelGGL((k_tower<C, NTA, NTB, 1>), dim3(grid), dim3(CONV_THREADS), nn->conv_lds, s, nn->g, ta);
            return hipSuccess;
        }
    }
    if (attr_only)
        return hipFuncSetAttribute((const void *)k_tower<C, NTA, NTB, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nn->conv_lds);
    hipLaunchKernelGGL((k_tower<C, NTA, NTB, 0>), dim3(grid), dim3(CONV_THREADS), nn->conv_lds, s, nn->g, ta);
    return hipSuccess;
}
#ifdef DBAZ_DEBUG
// the f16x3 tower on the 32x32x16 tiling: nt2 position tiles of 32 rows per wave (8 waves = 2 cout tiles x 4 tile groups)
template <int C>
static hipError_t tower_inst_mf(NNState *nn, hipStream_t s, const TowerArgs &ta, int nt2, int grid, bool attr_only)
{
    if constexpr (C == 64 || C == 128) {
        if (nt2 == 2) {
            if (attr_only) return hipFuncSetAttribute((const void *)k_tower<C, 2, 0, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nn->conv_lds_mf);
            hipLaunchKernelGGL((k_tower<C, 2, 0, 1, 1>), dim3(grid), dim3(CONV_THREADS), nn->conv_lds_mf, s, nn->g, ta);
        } else {
            if (attr_only) return hipFuncSetAttribute((const void *)k_tower<C, 1, 0, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nn->conv_lds_mf);
            hipLaunchKernelGGL((k_tower<C, 1, 0, 1, 1>), dim3(grid), dim3(CONV_THREADS), nn->conv_lds_mf, s, nn->g, ta);
        }
        return hipSuccess;
    } else {
        (void)nn; (void)s; (void)ta; (void)nt2; (void)grid; (void)attr_only;
        return hipErrorInvalidValue;
    }
}
#endif
// the remainder launch (f16x3, geometries whose main one-cout-tile instantiation is <7,6>)
static hipError_t tower_dispatch_rem(NNState *nn, hipStream_t s, const TowerArgs &ta, int grid, bool attr_only)
{
    if (nn->c2) { // beside the two-cout-tile main launch: register-resident residual in every body
        if (attr_only) return hipFuncSetAttribute((const void *)k_tower_rem<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nn->conv_lds);
        hipLaunchKernelGGL((k_tower_rem<64, 1>), dim3(grid), dim3(CONV_THREADS), nn->conv_lds, s, nn->g, ta);
        return hipSuccess;
    }
#define REM_CASE(CC)                                                                                                                  \
    case CC:                                                                                                                          \
        if (attr_only) return hipFuncSetAttribute((const void *)k_tower_rem<CC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nn->conv_lds); \
        hipLaunchKernelGGL((k_tower_rem<CC>), dim3(grid), dim3(CONV_THREADS), nn->conv_lds, s, nn->g, ta);                            \
        return hipSuccess;
    switch (nn->C) {
        REM_CASE(32) REM_CASE(64) REM_CASE(128)
    default: return hipErrorInvalidValue;
    }
#undef REM_CASE
}

// two cout tiles per wave (C = 64): nt position tiles per wave, 4 tile groups
template <int VAR>
static hipError_t tower_dispatch_c2v(NNState *nn, hipStream_t s, const TowerArgs &ta, int nt, int grid, bool attr_only)
{
#define C2_CASE(NT)                                                                                                                   \
    case NT:                                                                                                                          \
        if (attr_only) return hipFuncSetAttribute((const void *)k_tower<64, NT, 0, 1, 2, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nn->conv_lds_c2); \
        hipLaunchKernelGGL((k_tower<64, NT, 0, 1, 2, VAR>), dim3(grid), dim3(CONV_THREADS), nn->conv_lds_c2, s, nn->g, ta);           \
        return hipSuccess;
    switch (nt) {
        C2_CASE(1) C2_CASE(2) C2_CASE(3) C2_CASE(4)
    default: return hipErrorInvalidValue;
    }
#undef C2_CASE
}
#define VAR_SHIPPED (VAR_RESREG | VAR_WLDS) // the main launch of the release library
static hipError_t tower_dispatch_c2(NNState *nn, hipStream_t s, const TowerArgs &ta, int nt, int grid, bool attr_only)
{
#ifdef DBAZ_DEBUG // A/B variants of the main launch (nn_precision 3, 5..13 of the debug build)
    switch (nn->variant) {
    case 0: return tower_dispatch_c2v<0>(nn, s, ta, nt, grid, attr_only);
    case 1: return tower_dispatch_c2v<1>(nn, s, ta, nt, grid, attr_only);
    case 2: return tower_dispatch_c2v<2>(nn, s, ta, nt, grid, attr_only);
    case 3: return tower_dispatch_c2v<3>(nn, s, ta, nt, grid, attr_only);
    case 8: return tower_dispatch_c2v<8>(nn, s, ta, nt, grid, attr_only);
    case 11: return tower_dispatch_c2v<11>(nn, s, ta, nt, grid, attr_only);
    case 4: return tower_dispatch_c2v<4>(nn, s, ta, nt, grid, attr_only);
    case 21: return tower_dispatch_c2v<21>(nn, s, ta, nt, grid, attr_only);
    case 37: return tower_dispatch_c2v<37>(nn, s, ta, nt, grid, attr_only);
    case 64: // MF = 3: four waves, four cout tiles x four position tiles each (NT = 4 geometries; others stay on the 8-wave kernel)
        if (nt == 4) {
            if (attr_only) return hipFuncSetAttribute((const void *)k_tower<64, 4, 0, 1, 3, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nn->conv_lds_c2);
            hipLaunchKernelGGL((k_tower<64, 4, 0, 1, 3, 5>), dim3(grid), dim3(256), nn->conv_lds_c2, s, nn->g, ta);
            return hipSuccess;
        }
        break;
    default: break;
    }
#endif
    return tower_dispatch_c2v<VAR_SHIPPED>(nn, s, ta, nt, grid, attr_only);
}
#ifdef DBAZ_DEBUG
static hipError_t tower_dispatch_mf(NNState *nn, hipStream_t s, const TowerArgs &ta, int nt2, int grid, bool attr_only)
{
    if (nn->C == 64) return tower_inst_mf<64>(nn, s, ta, nt2, grid, attr_only);
    if (nn->C == 128) return tower_inst_mf<128>(nn, s, ta, nt2, grid, attr_only);
    return hipErrorInvalidValue;
}

#else
static hipError_t tower_dispatch_mf(NNState *, hipStream_t, const TowerArgs &, int, int, bool) { return hipErrorInvalidValue; }
#endif

template <int C>
static hipError_t tower_inst_c(NNState *nn, hipStream_t s, const TowerArgs &ta, int ntt, int grid, bool attr_only, int prec)
{
    switch (ntt) {
    case 2: return tower_inst<C, 2, 2>(nn, s, ta, grid, attr_only, prec);
    case 4: return tower_inst<C, 4, 4>(nn, s, ta, grid, attr_only, prec);
    case 5: return tower_inst<C, 5, 5>(nn, s, ta, grid, attr_only, prec);
    default: return tower_inst<C, 7, 6>(nn, s, ta, grid, attr_only, prec);
    }
}
// ntt: tiles per wave of the instantiation (7 -> <7,6>, 5 -> <5,5>, 4 -> <4,4>, 2 -> <2,2>); prec < 0: the handle's
static hipError_t tower_dispatch(NNState *nn, hipStream_t s, const TowerArgs &ta, int ntt, int grid, bool attr_only, int prec = -1)
{
    if (prec < 0) prec = nn->precision;
    switch (nn->C) {
    case 16: return tower_inst_c<16>(nn, s, ta, ntt, grid, attr_only, prec);
    case 32: return tower_inst_c<32>(nn, s, ta, ntt, grid, attr_only, prec);
    case 64: return tower_inst_c<64>(nn, s, ta, ntt, grid, attr_only, prec);
    default: return tower_inst_c<128>(nn, s, ta, ntt, grid, attr_only, prec);
    }
}

// packs a Linear [O][K] for the 16-sample MFMA GEMM: [O/16][K/16][64][4]
static std::vector<float> pack_dense(const std::vector<float> &w, int O, int K)
{
    const int KC = K / 16;
    std::vector<float> pk((size_t)O * K, 0.0f);
    for (int job = 0; job < O / 16; job++)
        for (int kc = 0; kc < KC; kc++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 4; e++) {
                    int o = job * 16 + (lane & 15), k = kc * 16 + 4 * (lane >> 4) + e;
                    pk[(((size_t)job * KC + kc) * 64 + lane) * 4 + e] = w[(size_t)o * K + k];
                }
    return pk;
}

static int dense_rs4(int K)
{
    int rs4 = K / 4;
    return ((rs4 + 15) / 16) * 16 + 2;
}

static int commit_simplenn(NNState *nn, std::string &err)
{
    const Geo &g = nn->g;
    const int C = 256;
    for (void *p : nn->allocs) (void)hipFree(p);
    nn->allocs.clear();
    nn->ready = false;
    auto f32v = [](const std::vector<double> &d) { return std::vector<float>(d.begin(), d.end()); };
    // conv0 (3 -> 256) raw + bn0 applied after the ReLU
    {
        auto w = sd_get(nn, "conv0.weight", (size_t)C * 27, err); if (!w) return DBAZ_EINVAL;
        auto b = sd_get(nn, "conv0.bias", C, err); if (!b) return DBAZ_EINVAL;
        std::vector<double> sc, tc;
        if (!bn_affine(nn, "bn0", C, sc, tc, err)) return DBAZ_EINVAL;
        std::vector<float> pk((size_t)27 * C);
        for (int co = 0; co < C; co++)
            for (int ci = 0; ci < 3; ci++)
                for (int tap = 0; tap < 9; tap++) pk[(size_t)(tap * 3 + ci) * C + co] = (*w)[((size_t)co * 3 + ci) * 9 + tap];
        nn->w0 = nn_upload(nn, pk);
        nn->b0 = nn_upload(nn, *b);
        nn->sn_s0 = nn_upload(nn, f32v(sc));
        nn->sn_t0 = nn_upload(nn, f32v(tc));
    }
    // conv1..conv4 raw, post affine from bn1..bn4
    {
        std::vector<float> pk_all, bias_all, ps_all, pt_all;
        nn->osc_host.clear();
        for (int i = 1; i <= 4; i++) {
            std::string c = "conv" + std::to_string(i), bnn = "bn" + std::to_string(i);
            if (!pack_conv(nn, c, bnn, C, pk_all, bias_all, err, false)) return DBAZ_EINVAL;
            std::vector<double> sc, tc;
            if (!bn_affine(nn, bnn, C, sc, tc, err)) return DBAZ_EINVAL;
            for (int k = 0; k < C; k++) {
                ps_all.push_back((float)sc[k]);
                pt_all.push_back((float)(tc[k] * (nn->precision == 1 ? (double)ACT_SCALE : 1.0)));
            }
        }
        if (nn->osc_host.empty()) nn->osc_host.assign(4, 1.0f);
        nn->tw = nn_upload(nn, pk_all);
        nn->tb = nn_upload(nn, bias_all);
        nn->sn_ts = nn_upload(nn, ps_all);
        nn->sn_tt = nn_upload(nn, pt_all);
        nn->tosc = nn_upload(nn, nn->osc_host);
        nn->overflow = nn_alloc<int>(nn, 4);
        if (!nn->tw || !nn->overflow) { err = "hipMalloc failed (SimpleNN weights)"; return DBAZ_EDEVICE; }
        (void)hipMemset(nn->overflow, 0, 16);
    }
    // fc0 / fc1 with BatchNorm1d after the ReLU
    {
        auto w0 = sd_get(nn, "fc0.weight", (size_t)512 * 1024, err); if (!w0) return DBAZ_EINVAL;
        auto b0 = sd_get(nn, "fc0.bias", 512, err); if (!b0) return DBAZ_EINVAL;
        auto w1 = sd_get(nn, "fc1.weight", (size_t)256 * 512, err); if (!w1) return DBAZ_EINVAL;
        auto b1 = sd_get(nn, "fc1.bias", 256, err); if (!b1) return DBAZ_EINVAL;
        std::vector<double> s0, t0, s1, t1;
        if (!bn_affine(nn, "bn_fc0", 512, s0, t0, err)) return DBAZ_EINVAL;
        if (!bn_affine(nn, "bn_fc1", 256, s1, t1, err)) return DBAZ_EINVAL;
        nn->sn_w0 = nn_upload(nn, pack_dense(*w0, 512, 1024));
        nn->sn_b0 = nn_upload(nn, *b0);
        nn->sn_ps0 = nn_upload(nn, f32v(s0));
        nn->sn_pt0 = nn_upload(nn, f32v(t0));
        nn->sn_w1 = nn_upload(nn, pack_dense(*w1, 256, 512));
        nn->sn_b1 = nn_upload(nn, *b1);
        nn->sn_ps1 = nn_upload(nn, f32v(s1));
        nn->sn_pt1 = nn_upload(nn, f32v(t1));
    }
    // heads: policy_fc [32][256], value_fc [1][256] in k_head_fc's job layout
    {
        auto wp = sd_get(nn, "policy_fc.weight", (size_t)32 * 256, err); if (!wp) return DBAZ_EINVAL;
        auto bp = sd_get(nn, "policy_fc.bias", 32, err); if (!bp) return DBAZ_EINVAL;
        auto wv = sd_get(nn, "value_fc.weight", 256, err); if (!wv) return DBAZ_EINVAL;
        auto bv = sd_get(nn, "value_fc.bias", 1, err); if (!bv) return DBAZ_EINVAL;
        const int K = 256, KC = K / 16, ntp = 2, ntv = 1, NJ = 3;
        std::vector<float> pk((size_t)NJ * KC * 64 * 4, 0.0f), bias((size_t)NJ * 16, 0.0f);
        for (int job = 0; job < NJ; job++)
            for (int kc = 0; kc < KC; kc++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 4; e++) {
                        int o = (job < ntp ? job : 0) * 16 + (lane & 15), k = kc * 16 + 4 * (lane >> 4) + e;
                        float v = 0.0f;
                        if (job < ntp) v = (*wp)[(size_t)o * K + k];
                        else if ((lane & 15) == 0) v = (*wv)[k];
                        pk[(((size_t)job * KC + kc) * 64 + lane) * 4 + e] = v;
                    }
        for (int i = 0; i < 32; i++) bias[i] = (*bp)[i];
        bias[32] = (*bv)[0];
        nn->wfc = nn_upload(nn, pk);
        nn->bfc = nn_upload(nn, bias);
        std::vector<float> one(1, 0.0f);
        nn->wv1 = nn_upload(nn, one);
        nn->bv1 = nn_upload(nn, one);
        nn->KP = K; nn->ntp = ntp; nn->ntv = ntv; nn->RS4 = dense_rs4(K);
        nn->fc_lds = ((size_t)4 * 16 * HEAD_MT * NJ * 16 + (size_t)16 * HEAD_MT * (NJ * 16 + 1)) * 4;
    }
    nn->sn_flat = nn_alloc<float>(nn, (size_t)nn->max_batch * 1024);
    nn->sn_h1 = nn_alloc<float>(nn, (size_t)nn->max_batch * 512);
    nn->hact = nn_alloc<float>(nn, (size_t)nn->max_batch * 2 * 256 + 16);
    if (nn->hact) (void)hipMemset(nn->hact, 0, ((size_t)nn->max_batch * 2 * 256 + 16) * sizeof(float));
    if (!nn->sn_flat || !nn->sn_h1 || !nn->hact || !nn->wfc) { err = "hipMalloc failed (SimpleNN buffers)"; return DBAZ_EDEVICE; }
    // LDS: two images of S*16 rows x 264 dwords (+ zero regions); S = 4 -> 141 KB
    const size_t s4 = (C + 8) / 4;
    nn->S = 4;
    nn->sn_lds = 2 * ((((size_t)nn->S * g.HW * s4 + 15) & ~(size_t)15) + 3 * s4) * 16;
    hipError_t he = hipFuncSetAttribute(nn->precision == 1 ? (const void *)k_simple_trunk<1> : (const void *)k_simple_trunk<0>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)nn->sn_lds);
    if (he == hipSuccess) he = hipFuncSetAttribute((const void *)k_dense, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * dense_rs4(1024) * 16);
    if (he == hipSuccess) he = hipFuncSetAttribute((const void *)k_head_fc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nn->fc_lds);
    if (he != hipSuccess) { err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(he); return DBAZ_EDEVICE; }
    nn->ready = true;
    return DBAZ_OK;
}

int nn_commit(NNState *nn, hipStream_t s, std::string &err)
{
    (void)s;
    if (nn->kind == DBAZ_EVAL_SIMPLENN) return commit_simplenn(nn, err);
    if (nn->kind != DBAZ_EVAL_RESNET) { err = "dbaz_nn_configure not called"; return DBAZ_ESTATE; }
    const Geo &g = nn->g;
    const int C = nn->C, Cr = nn->Craw, hc = nn->hc, vf = nn->vf, HW = g.HW, A = g.A, K = hc * HW;
    for (void *p : nn->allocs) (void)hipFree(p);
    nn->allocs.clear();
    nn->ready = false;
    // bn_input
    {
        std::vector<double> si, ti;
        if (!bn_affine(nn, "bn_input", 3, si, ti, err)) return DBAZ_EINVAL;
        std::vector<float> fs(si.begin(), si.end()), ft(ti.begin(), ti.end());
        nn->in_s = nn_upload(nn, fs);
        nn->in_t = nn_upload(nn, ft);
    }
    // conv0 + bn0 -> [9][3][C]
    {
        auto w = sd_get(nn, "resnet.conv0.weight", (size_t)Cr * 27, err); if (!w) return DBAZ_EINVAL;
        auto b = sd_get(nn, "resnet.conv0.bias", Cr, err); if (!b) return DBAZ_EINVAL;
        std::vector<double> sc, tc;
        if (!bn_affine(nn, "resnet.bn0", Cr, sc, tc, err)) return DBAZ_EINVAL;
        std::vector<float> pk((size_t)27 * C, 0.0f), bias(C, 0.0f);
        for (int co = 0; co < Cr; co++) {
            for (int ci = 0; ci < 3; ci++)
                for (int tap = 0; tap < 9; tap++)
                    pk[(size_t)(tap * 3 + ci) * C + co] = (float)((double)(*w)[((size_t)co * 3 + ci) * 9 + tap] * sc[co]);
            bias[co] = (float)((double)(*b)[co] * sc[co] + tc[co]);
        }
        nn->w0 = nn_upload(nn, pk);
        nn->b0 = nn_upload(nn, bias);
        nn->w0p = nullptr;
        if (nn->precision == 1 && C >= 32) {
            double mx = 0;
            for (float v : pk) mx = std::max(mx, fabs((double)v));
            int sw = 0;
            if (mx > 0) { int e; frexp(mx, &e); sw = 14 - e; }
            sw = std::max(-24, std::min(24, sw));
            const double wscale = ldexp(1.0, sw);
            const int n_ct = C / 16;
            std::vector<_Float16> hp((size_t)n_ct * 2 * 64 * 8, (_Float16)0.0f);
            for (int ct = 0; ct < n_ct; ct++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 8; e++) {
                        const int co = ct * 16 + (lane & 15), k = 8 * (lane >> 4) + e;
                        if (k >= 27) continue;
                        const float v = (float)((double)pk[(size_t)k * C + co] * wscale);
                        const _Float16 h = (_Float16)v;
                        const size_t base = (size_t)ct * 2 * 64 * 8;
                        hp[base + (size_t)lane * 8 + e] = h;
                        hp[base + 64 * 8 + (size_t)lane * 8 + e] = (_Float16)(v - (float)h);
                    }
            std::vector<float> asf(hp.size() / 2);
            memcpy(asf.data(), hp.data(), hp.size() * 2);
            nn->w0p = nn_upload(nn, asf);
            nn->osc0 = (float)ldexp(1.0, -sw);
        }
    }
    {
        std::vector<float> pk_all, bias_all, pk32_all, bias32_all;
        const bool both = nn->precision == 1 && C >= 32; // f16x3 handles keep the exact-f32 operands too (safety net)
        nn->osc_host.clear();
        for (int i = 0; i < nn->blocks; i++) {
            std::string p = "resnet.resblocks." + std::to_string(i);
            if (!pack_conv(nn, p + ".conv1", p + ".bn1", C, pk_all, bias_all, err, true, both ? &pk32_all : nullptr, both ? &bias32_all : nullptr)) return DBAZ_EINVAL;
            if (!pack_conv(nn, p + ".conv2", p + ".bn2", C, pk_all, bias_all, err, true, both ? &pk32_all : nullptr, both ? &bias32_all : nullptr)) return DBAZ_EINVAL;
        }
        nn->tw = nn_upload(nn, pk_all);
        nn->tb = nn_upload(nn, bias_all);
        nn->tw32 = nn->tb32 = nullptr;
        nn->ovf_flags = nullptr;
        if (both) {
            nn->tw32 = nn_upload(nn, pk32_all);
            nn->tb32 = nn_upload(nn, bias32_all);
            nn->ovf_flags = nn_alloc<int>(nn, (size_t)nn->max_batch + 16);
            if (!nn->tw32 || !nn->tb32 || !nn->ovf_flags) { err = "hipMalloc failed (f32 fallback weights)"; return DBAZ_EDEVICE; }
            (void)hipMemset(nn->ovf_flags, 0, ((size_t)nn->max_batch + 16) * sizeof(int));
        }
        if (nn->osc_host.empty()) nn->osc_host.push_back(1.0f);
        nn->tosc = nn_upload(nn, nn->osc_host);
        nn->overflow = nn_alloc<int>(nn, 4);
        if (!nn->tw || !nn->tb || !nn->tosc || !nn->overflow) { err = "hipMalloc failed (tower weights)"; return DBAZ_EDEVICE; }
        (void)hipMemset(nn->overflow, 0, 16);
#ifdef DBAZ_STAMP
        nn->stamp_out = nn_alloc<unsigned long long>(nn, (size_t)nn->max_batch * 8 * 10);
        (void)hipMemset(nn->stamp_out, 0, (size_t)nn->max_batch * 8 * 10 * 8);
#endif
    }
    // heads: conv1x1 + BN folded, rows [policy hc | value hc]
    {
        std::vector<float> hw((size_t)2 * hc * C, 0.0f), hb(2 * hc);
        const char *heads[2] = {"policy_head", "value_head"};
        for (int h = 0; h < 2; h++) {
            std::string p = heads[h];
            auto w = sd_get(nn, p + ".conv0.weight", (size_t)hc * Cr, err); if (!w) return DBAZ_EINVAL;
            auto b = sd_get(nn, p + ".conv0.bias", hc, err); if (!b) return DBAZ_EINVAL;
            std::vector<double> sc, tc;
            if (!bn_affine(nn, p + ".bn0", hc, sc, tc, err)) return DBAZ_EINVAL;
            for (int o = 0; o < hc; o++) {
                for (int c = 0; c < Cr; c++) hw[(size_t)(h * hc + o) * C + c] = (float)((double)(*w)[(size_t)o * Cr + c] * sc[o]);
                hb[h * hc + o] = (float)((double)(*b)[o] * sc[o] + tc[o]);
            }
        }
        nn->hw = nn_upload(nn, hw);
        nn->hb = nn_upload(nn, hb);
        nn->hwp = nullptr;
        const int OCP = (2 * hc + 15) & ~15, n_ct = OCP / 16;
        if (nn->precision == 1 && C >= 32 && (n_ct == 1 || n_ct == 2 || n_ct == 4 || n_ct == 8)) {
            // same operand format as the tower layers (pack_conv): weights * 2^sw_h split into halves
            double mx = 0;
            for (float v : hw) mx = std::max(mx, fabs((double)v));
            int sw = 0;
            if (mx > 0) { int e; frexp(mx, &e); sw = 14 - e; }
            sw = std::max(-24, std::min(24, sw));
            const double wscale = ldexp(1.0, sw);
            const int KS = C / 32;
            std::vector<_Float16> hp((size_t)n_ct * KS * 2 * 64 * 8, (_Float16)0.0f);
            for (int ct = 0; ct < n_ct; ct++)
                for (int ks = 0; ks < KS; ks++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int e = 0; e < 8; e++) {
                            const int oc = ct * 16 + (lane & 15), ci = ks * 32 + 8 * (lane >> 4) + e;
                            if (oc >= 2 * hc) continue;
                            const float v = (float)((double)hw[(size_t)oc * C + ci] * wscale);
                            const _Float16 h = (_Float16)v;
                            const size_t base = (((size_t)ct * KS + ks) * 2) * 64 * 8;
                            hp[base + (size_t)lane * 8 + e] = h;
                            hp[base + 64 * 8 + (size_t)lane * 8 + e] = (_Float16)(v - (float)h);
                        }
            std::vector<float> asf(hp.size() / 2);
            memcpy(asf.data(), hp.data(), hp.size() * 2);
            nn->hwp = nn_upload(nn, asf);
            nn->hosc = (float)ldexp(1.0, -(sw + ACT_SHIFT));
        }
        auto wp = sd_get(nn, "policy_head.fc.weight", (size_t)A * K, err); if (!wp) return DBAZ_EINVAL;
        auto bp = sd_get(nn, "policy_head.fc.bias", A, err); if (!bp) return DBAZ_EINVAL;
        auto w0 = sd_get(nn, "value_head.fc0.weight", (size_t)vf * K, err); if (!w0) return DBAZ_EINVAL;
        auto b0 = sd_get(nn, "value_head.fc0.bias", vf, err); if (!b0) return DBAZ_EINVAL;
        auto w1 = sd_get(nn, "value_head.fc1.weight", vf, err); if (!w1) return DBAZ_EINVAL;
        auto b1 = sd_get(nn, "value_head.fc1.bias", 1, err); if (!b1) return DBAZ_EINVAL;
        const int KP = (K + 15) & ~15, KC = KP / 16;
        const int ntp = (A + 15) / 16, ntv = (vf + 15) / 16, NJ = ntp + ntv;
        std::vector<float> pk((size_t)NJ * KC * 64 * 4, 0.0f), bias((size_t)NJ * 16, 0.0f);
        for (int job = 0; job < NJ; job++) {
            const bool pol = job < ntp;
            const int tile = pol ? job : job - ntp, nout = pol ? A : vf;
            const std::vector<float> &wm = pol ? *wp : *w0;
            for (int kc = 0; kc < KC; kc++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 4; e++) {
                        int o = tile * 16 + (lane & 15), k = kc * 16 + 4 * (lane >> 4) + e;
                        if (o < nout && k < K) pk[(((size_t)job * KC + kc) * 64 + lane) * 4 + e] = wm[(size_t)o * K + k];
                    }
            for (int i = 0; i < 16; i++) {
                int o = tile * 16 + i;
                if (o < nout) bias[job * 16 + i] = pol ? (*bp)[o] : (*b0)[o];
            }
        }
        nn->wfc = nn_upload(nn, pk);
        nn->bfc = nn_upload(nn, bias);
        nn->wv1 = nn_upload(nn, *w1);
        nn->bv1 = nn_upload(nn, *b1);
        nn->KP = KP; nn->ntp = ntp; nn->ntv = ntv;
        int rs4 = KP / 4;
        rs4 = ((rs4 + 15) / 16) * 16 + 2; // = 2 mod 16 float4 units
        nn->RS4 = rs4;
        nn->fc_lds = ((size_t)4 * 16 * HEAD_MT * NJ * 16 + (size_t)16 * HEAD_MT * (NJ * 16 + 1)) * 4;
        if (nn->fc_lds > 158 * 1024) { err = "head FC tile does not fit LDS"; return DBAZ_EINVAL; }
        if (hipFuncSetAttribute((const void *)k_head_fc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nn->fc_lds) != hipSuccess) {
            err = "hipFuncSetAttribute(k_head_fc) failed"; return DBAZ_EDEVICE;
        }
    }
    // k_head_fc reads whole 16-float chunks: up to 15 floats past a row's K when K % 16 != 0 (their weights are
    // zero, so the values only have to be finite): slack at the end, everything zero-initialised
    nn->hact = nn_alloc<float>(nn, (size_t)nn->max_batch * 2 * K + 16);
    if (nn->hact) (void)hipMemset(nn->hact, 0, ((size_t)nn->max_batch * 2 * K + 16) * sizeof(float));
    if (!nn->hact || !nn->wv1 || !nn->w0) { err = "hipMalloc failed (network buffers)"; return DBAZ_EDEVICE; }
    // conv workgroup geometry: S whole samples, NT position tiles of 16 rows (<= MAXT)
    const size_t lds_budget = 158 * 1024; // of 160 KiB: two ping-pong activation images
    // the idle image doubles as staging for conv0 (padded planes + 27*C weights) and the head convs
    auto lds_bytes = [&](int S_) {
        const size_t s4 = (C + 8) / 4;
        const size_t img = ((((size_t)S_ * HW * s4 + 15) & ~(size_t)15) + 3 * s4) * 4; // floats, incl. zero region
        const size_t need0 = (size_t)S_ * 3 * (g.H + 2) * (g.W + 2) + (size_t)27 * C;
        // head phase: conv1x1 weights (VALU path) + staged head activations + 16 floats of slack + the FC logits of the samples
        const size_t nj = (size_t)(A + 15) / 16 + (size_t)(vf + 15) / 16;
        const size_t need1 = (size_t)2 * hc * (C + 4) + (size_t)S_ * 2 * hc * HW + 16 + (size_t)S_ * (nj * 16 + 1);
        return (img + std::max(img, std::max(need0, need1))) * 4;
    };
    int S = (16 * MAXT) / HW;
    if (S < 1) S = 1;
    while (S > 1 && lds_bytes(S) > lds_budget) S--;
    if (lds_bytes(S) > lds_budget) { err = "board / channels / head_channels too large for the LDS-resident tower"; return DBAZ_EINVAL; }
    nn->S = S;
    nn->NT = (S * HW + 15) / 16;
    nn->conv_lds = lds_bytes(S);
    nn->NTT = nn->NT > 8 ? 7 : (nn->NT > 4 ? 4 : 2); // tiles per wave; two waves cover 2*NTT >= NT tiles
    // tail variants: <2,2> holds 64 rows, <4,4> 128 rows, <5,5> 160 rows
    nn->S_small = nn->S_mid = nn->S_big = 0;
    if (nn->NTT == 7) {
        nn->S_big = std::min(160 / HW, nn->S - 1);
        nn->S_mid = std::min(128 / HW, nn->S_big - 1);
        nn->S_small = std::min(64 / HW, nn->S_mid - 1);
        if (nn->S_big < 0) nn->S_big = 0;
    } else if (nn->NTT == 4) {
        nn->S_small = std::min(64 / HW, nn->S - 1);
    }
    if (nn->S_mid < 0) nn->S_mid = 0;
    if (nn->S_small < 0) nn->S_small = 0;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            nn->cus = cus;
    }
    // 32x32x16 tiling (debug build, nn_precision = 2): 8 position tiles of 32 rows per workgroup, rows of C + 4 dwords
    if (nn->mf32 && (C == 64 || C == 128)) {
        auto lds_mf = [&](int S_) {
            const size_t s4 = (C + 4) / 4;
            const size_t img = ((((size_t)S_ * HW * s4 + 15) & ~(size_t)15) + 3 * s4) * 4;
            const size_t need0 = (size_t)S_ * 3 * (g.H + 2) * (g.W + 2) + (size_t)27 * C;
            const size_t nj = (size_t)(A + 15) / 16 + (size_t)(vf + 15) / 16;
            const size_t need1 = (size_t)2 * hc * (C + 4) + (size_t)S_ * 2 * hc * HW + 16 + (size_t)S_ * (nj * 16 + 1);
            return (img + std::max(img, std::max(need0, need1))) * 4;
        };
        int Sm = MAXROWS / HW;
        if (Sm > 16) Sm = 16;
        while (Sm > 1 && lds_mf(Sm) > lds_budget) Sm--;
        if (Sm >= 1 && lds_mf(Sm) <= lds_budget) {
            nn->S_mf = Sm;
            nn->NT2 = (Sm * HW + 31) / 32 > 4 ? 2 : 1;
            nn->conv_lds_mf = lds_mf(Sm);
            nn->S_mf_tail = nn->NT2 == 2 ? std::min(128 / HW, Sm - 1) : 0; // <1>: 4 tiles of 32 rows
            if (nn->S_mf_tail < 0) nn->S_mf_tail = 0;
        } else {
            nn->mf32 = 0;
        }
    } else {
        nn->mf32 = 0;
    }
    if (nn->c2 && C == 64) {
        int Sc = MAXROWS / HW;
        if (Sc > 16) Sc = 16;
        // (the shipped variant keeps a two-slot weight ring of 16 KB behind the images; 1.1 KB of static LDS besides)
        const size_t ring = (nn->variant & (4 | 64)) ? (size_t)2 * 512 * 16 : 0, budget_c2 = (size_t)160 * 1024 - 1536;
        while (Sc > 1 && lds_bytes(Sc) + ring > budget_c2) Sc--;
        nn->S_c2 = Sc;
        nn->NT_c2 = ((Sc * HW + 15) / 16 + 3) / 4;           // tiles per wave (4 groups)
        nn->conv_lds_c2 = lds_bytes(Sc) + ring;
        // worth it only where its 4 x NT_c2 tiles are filled about as well as the one-cout-tile kernel's NT (9x9: 200 of 256
        // rows against 200 of 208)
        const double fill_c2 = (double)Sc * HW / (64.0 * nn->NT_c2), fill_1 = (double)nn->S * HW / (16.0 * nn->NT);
        // (0.9: 3x3 boards, 15 samples in 16 tiles against 13 in 13 -- their steps never fill a round of the main launch, but its
        // remainder bodies keep the residual stream in registers, which is worth 1.3 % there)
        if (fill_c2 < 0.9 * fill_1) nn->c2 = 0;
        if (nn->NT_c2 < 1 || nn->NT_c2 > 4 || lds_bytes(Sc) + ring > budget_c2) nn->c2 = 0;
    } else {
        nn->c2 = 0;
    }
    hipError_t he = tower_dispatch(nn, nullptr, TowerArgs(), nn->NTT, 0, true);
    nn->use_rem = (nn->precision == 1 && nn->NTT == 7 && C >= 32 && !nn->mf32) ? 1 : 0;
    // the two-cout-tile main launch keeps the residual stream in registers; its remainder bodies must round the same way, and
    // those live in k_tower_rem<64, 1>: no remainder launch, no two-cout-tile main launch
    if (!nn->use_rem) nn->c2 = 0;
    if (he == hipSuccess && nn->use_rem) he = tower_dispatch_rem(nn, nullptr, TowerArgs(), 0, true);
    if (he == hipSuccess && nn->c2) he = tower_dispatch_c2(nn, nullptr, TowerArgs(), nn->NT_c2, 0, true);

    if (he == hipSuccess && nn->mf32) he = tower_dispatch_mf(nn, nullptr, TowerArgs(), 2, 0, true);
    if (he == hipSuccess && nn->mf32) he = tower_dispatch_mf(nn, nullptr, TowerArgs(), 1, 0, true);
    if (he == hipSuccess && nn->tw32) he = tower_dispatch(nn, nullptr, TowerArgs(), nn->NTT, 0, true, 0);
    if (he == hipSuccess && nn->S_mid > 0) he = tower_dispatch(nn, nullptr, TowerArgs(), 4, 0, true);
    if (he == hipSuccess && nn->S_small > 0) he = tower_dispatch(nn, nullptr, TowerArgs(), 2, 0, true);
    if (he == hipSuccess && nn->S_big > 0) he = tower_dispatch(nn, nullptr, TowerArgs(), 5, 0, true);
    if (he != hipSuccess) { err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(he); return DBAZ_EDEVICE; }
    nn->ready = true;
    return DBAZ_OK;
}

void nn_forward(NNState *nn, hipStream_t s, const float *feat, const int32_t *list_dev, const int32_t *n_dev, int max_n,
                float *P, float *V, int AS, hipEvent_t ev_begin, hipEvent_t ev_end, int cut_round, int cut_defer, int32_t *n_used)
{
    const Geo &g = nn->g;
    const int hc = nn->hc, HW = g.HW;
    if (max_n > nn->max_batch) max_n = nn->max_batch;
    if (nn->kind == DBAZ_EVAL_SIMPLENN) {
        SimpleArgs sa;
        sa.feat = feat; sa.list = list_dev; sa.n_dev = n_dev; sa.w0 = nn->w0; sa.b0 = nn->b0; sa.s0 = nn->sn_s0; sa.t0 = nn->sn_t0;
        sa.tw = nn->tw; sa.tb = nn->tb; sa.ts = nn->sn_ts; sa.tt = nn->sn_tt; sa.tosc = nn->tosc; sa.flat = nn->sn_flat;
        sa.overflow = nn->overflow; sa.S = nn->S;
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        const int grid = (max_n + nn->S - 1) / nn->S;
        if (nn->precision == 1) hipLaunchKernelGGL(k_simple_trunk<1>, dim3(grid), dim3(CONV_THREADS), nn->sn_lds, s, g, sa);
        else hipLaunchKernelGGL(k_simple_trunk<0>, dim3(grid), dim3(CONV_THREADS), nn->sn_lds, s, g, sa);
        if (ev_end) (void)hipEventRecord(ev_end, s);
        DenseArgs d0;
        d0.n_dev = n_dev; d0.in = nn->sn_flat; d0.w = nn->sn_w0; d0.b = nn->sn_b0; d0.ps = nn->sn_ps0; d0.pt = nn->sn_pt0;
        d0.out = nn->sn_h1; d0.K = 1024; d0.O = 512; d0.RS4 = dense_rs4(1024); d0.dup = 0;
        hipLaunchKernelGGL(k_dense, dim3((max_n + 15) / 16), dim3(256), (size_t)16 * d0.RS4 * 16, s, d0);
        DenseArgs d1;
        d1.n_dev = n_dev; d1.in = nn->sn_h1; d1.w = nn->sn_w1; d1.b = nn->sn_b1; d1.ps = nn->sn_ps1; d1.pt = nn->sn_pt1;
        d1.out = nn->hact; d1.K = 512; d1.O = 256; d1.RS4 = dense_rs4(512); d1.dup = 1;
        hipLaunchKernelGGL(k_dense, dim3((max_n + 15) / 16), dim3(256), (size_t)16 * d1.RS4 * 16, s, d1);
        HeadArgs ha;
        ha.list = list_dev; ha.n_dev = n_dev; ha.hact = nn->hact; ha.wfc = nn->wfc; ha.bfc = nn->bfc; ha.wv1 = nn->wv1; ha.bv1 = nn->bv1;
        ha.P = P; ha.V = V; ha.K = 256; ha.KP = 256; ha.RS4 = nn->RS4; ha.ntp = nn->ntp; ha.ntv = nn->ntv; ha.vf = 0; ha.AS = AS;
        ha.value_direct = 1; ha.cut_round = 0; ha.cut_defer = 0; ha.n_used = nullptr;
        hipLaunchKernelGGL(k_head_fc, dim3((max_n + 16 * HEAD_MT - 1) / (16 * HEAD_MT)), dim3(256), nn->fc_lds, s, g, ha);
        return;
    }
    TowerArgs ta;
    ta.feat = feat; ta.list = list_dev; ta.n_dev = n_dev; ta.in_s = nn->in_s; ta.in_t = nn->in_t; ta.w0 = nn->w0; ta.b0 = nn->b0;
    ta.tw = nn->tw; ta.tb = nn->tb; ta.tosc = nn->tosc; ta.hw = nn->hw; ta.hb = nn->hb; ta.hwp = nn->hwp; ta.hosc = nn->hosc; ta.w0p = nn->w0p; ta.osc0 = nn->osc0; ta.hact = nn->hact;
    ta.overflow = nn->overflow; ta.ovf_flags = nn->ovf_flags; ta.fallback = 0; ta.S = nn->S; ta.nblocks = nn->blocks; ta.hc = hc;
    ta.wfc = nn->wfc; ta.bfc = nn->bfc; ta.wv1 = nn->wv1; ta.bv1 = nn->bv1; ta.P = P; ta.V = V; ta.KP = nn->KP; ta.ntp = nn->ntp;
    ta.ntv = nn->ntv; ta.vf = nn->vf; ta.AS = AS; ta.n_used = n_used;
    ta.stamp_out = nn->stamp_out;
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    ta.S_main = nn->S; ta.S_small = nn->S_small; ta.S_mid = nn->S_mid; ta.S_big = nn->S_big; ta.S_huge = 0; ta.cus = nn->cus;
    ta.cut_round = cut_round; ta.cut_defer = cut_defer;
    ta.role = 0; ta.S = nn->S;
    if (nn->c2) {
        // two cout tiles per wave (default for 64 channels) for the FULL rounds; what is left behind the last full round goes
        // to one round of the one-cout-tile kernels, whose workgroups come in finer sizes (1, 2, 3 samples, or all S of them
        // as role 4) -- or stays with this launch if it is more than such a round holds
        ta.S_main = ta.S = nn->S_c2; ta.S_huge = nn->S;
        (void)tower_dispatch_c2(nn, s, ta, nn->NT_c2, (max_n + nn->S_c2 - 1) / nn->S_c2, false);
    }
    if (nn->mf32 && !nn->c2) {
        // 32x32x16 tiling: main launch + one tail launch of half-size workgroups (same split rule, derived from n on the device)
        ta.S_main = ta.S = nn->S_mf; ta.S_small = nn->S_mf_tail; ta.S_mid = ta.S_big = ta.S_huge = 0;
        (void)tower_dispatch_mf(nn, s, ta, nn->NT2, (max_n + nn->S_mf - 1) / nn->S_mf, false);
        if (nn->S_mf_tail > 0) {
            ta.role = 1; ta.S = nn->S_mf_tail;
            (void)tower_dispatch_mf(nn, s, ta, 1, nn->cus, false);
        }
    } else {
    if (!nn->c2) (void)tower_dispatch(nn, s, ta, nn->NTT, (max_n + nn->S - 1) / nn->S, false);
    if (nn->use_rem) {     // the remainder sizes in one launch (the workgroups pick theirs)
        ta.role = -1;
        (void)tower_dispatch_rem(nn, s, ta, nn->cus, false);
    } else {
    if (nn->S_small > 0) { // tail <= cus * S_small samples: one round of <2,2> workgroups
        ta.role = 1; ta.S = nn->S_small;
        (void)tower_dispatch(nn, s, ta, 2, nn->cus, false);
    }
    if (nn->S_mid > 0) {   // tail <= cus * S_mid samples: one round of <4,4> workgroups
        ta.role = 2; ta.S = nn->S_mid;
        (void)tower_dispatch(nn, s, ta, 4, nn->cus, false);
    }
    if (nn->S_big > 0) {   // tail <= cus * S_big samples: one round of <5,5> workgroups
        ta.role = 3; ta.S = nn->S_big;
        (void)tower_dispatch(nn, s, ta, 5, nn->cus, false);
    }
    }
    }
    if (nn->precision == 1 && nn->tw32 && !nn->no_fallback) {
        // safety net of the f16x3 mode: samples whose workgroup saw an activation leave f16's range are redone by the
        // exact-f32 tower (its workgroups check the per-sample flags on the device and leave at once otherwise)
        ta.role = 0; ta.S = ta.S_main = nn->S; ta.S_small = ta.S_mid = ta.S_big = ta.S_huge = 0; ta.fallback = 1;
        ta.tw = nn->tw32; ta.tb = nn->tb32; ta.w0p = nullptr; ta.hwp = nullptr;
        (void)tower_dispatch(nn, s, ta, nn->NTT, std::min((max_n + nn->S - 1) / nn->S, nn->cus), false, 0);
    }
    if (ev_end) (void)hipEventRecord(ev_end, s);
    // (the head FCs, softmax and tanh ran inside the tower workgroups: head_fc_fused)
    (void)HW;
}

void nn_round_info(const NNState *nn, int *round, int *rem_max)
{
    *round = 0; *rem_max = 0;
    if (!nn || nn->kind != DBAZ_EVAL_RESNET || nn->mf32) return;
    *round = nn->cus * (nn->c2 ? nn->S_c2 : nn->S);
    const int s_rem = nn->c2 ? nn->S : std::max(nn->S_big, std::max(nn->S_mid, nn->S_small));
    *rem_max = nn->cus * s_rem;
}

double nn_flops_per_sample(const NNState *nn)
{
    // 2*MAC of conv + FC layers: conv0, 2*blocks tower convs, head convs, FCs
    const Geo &g = nn->g;
    if (nn->kind == DBAZ_EVAL_SIMPLENN) // conv0 + 3 padded convs + the unpadded conv4 + FCs (62.89 MFLOP)
        return 2.0 * (16 * 27 * 256 + 3.0 * 16 * 2304 * 256 + 4 * 2304 * 256 + 1024 * 512 + 512 * 256 + 256 * 33);
    const double HW = g.HW, C = nn->Craw, hc = nn->hc, K = hc * HW;
    double f = 2.0 * HW * 27 * C + 2.0 * nn->blocks * 2.0 * HW * 9 * C * C;
    f += 2.0 * 2.0 * HW * C * hc + 2.0 * K * g.A + 2.0 * K * nn->vf + 2.0 * nn->vf;
    return f;
}

const char *nn_tower_kernel_name(const NNState *nn) { (void)nn; return "k_tower"; }

// diagnostic build: copies the stamp sums of the last launch to the host
int nn_read_stamps(NNState *nn, unsigned long long *out, int n_wg)
{
    if (!nn || !nn->stamp_out) return -1;
    return (int)hipMemcpy(out, nn->stamp_out, (size_t)n_wg * 8 * 10 * 8, hipMemcpyDeviceToHost);
}

// f16x3 mode, networks WITHOUT the exact-f32 safety net (SimpleNN; ResNetZero narrower than 32 channels never runs
// f16x3): non-zero once an activation exceeded f16's range (results invalid: use precision 0)
int nn_overflowed(NNState *nn)
{
    if (!nn || !nn->overflow || nn->tw32) return 0;
    int v = 0;
    (void)hipMemcpy(&v, nn->overflow, 4, hipMemcpyDeviceToHost);
    return v;
}

// f16x3 mode with the safety net: samples re-evaluated by the exact-f32 tower so far
long long nn_fallback_evals(NNState *nn)
{
    if (!nn || !nn->overflow || !nn->tw32) return 0;
    int v[2] = {0, 0};
    (void)hipMemcpy(v, nn->overflow, 8, hipMemcpyDeviceToHost);
    return v[1];
}
