// replay.hip -- training DATA path on gfx950: from packed replay rows in HBM to training batches
// in HBM, no host round trip (the consumer side of the self-play path).
//
// Reference semantics restated here
//   * pi column            self_play.py:114-115   float64 visits / visits.sum()
//   * dataset arrays       utils/utils.py:66-80   (rows already selected/ordered by the host:
//                          training flag, where-clause, df.sample); pos_average =
//                          df.groupby(x columns).mean(): pandas' group_mean is a Kahan-compensated
//                          float64 sum in row order divided by the row count; groups come out in
//                          ascending lexicographic order of the x columns; .astype(float32)
//   * SymmetriesGenerator  dots_boxes/dots_boxes_nn.py:11-58  the 8 (flip, rotate) transforms of the
//                          two edge planes (sentinel column/row stay in place), same map on pi
//
// Kernels (all HBM-bound byte/integer work, one 64-lane wave per row or per group)
//   k_ds_stage   packed rows (RowMeta | x i16[3HW] | visits i32[A]) -> SoA staging + sort key
//   k_ds_gather_key / hipcub radix sort (LSD over the key words, stable) -> row permutation
//   k_ds_flags / hipcub inclusive scan / k_ds_gstart -> group boundaries
//   k_ds_mean    per group: Kahan means of pi (float64) and z, float32 results
//   k_make_batch gather by index + symmetry LUT -> float32 boards [n,3,H,W], pi [n,A], z [n,1]
//   k_symmetry   the same LUT applied to caller tensors (SymmetriesGenerator drop-in)
// The sort key is the x vector itself in compressed form: planes 0/1 hold 0/1 per cell and plane 2
// is constant (dots_boxes_game.py:96-100), so [bits of x_0..x_{A-1}, MSB first | x_A + 32768 as 16
// bits] orders rows exactly as the lexicographic comparison of the 3HW columns does; k_ds_stage
// verifies that the rows have this form.
// Compiled with -ffp-contract=off (Kahan sums must not be contracted).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "replay.h"

#define ROW_X_OFF 28 // sizeof(RowMeta)
#define ROW_Z_OFF 25

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes, bool keep, hipStream_t s)
    {
        if (bytes <= cap) return hipSuccess;
        size_t ncap = bytes + bytes / 2 + 256;
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, ncap);
        if (e != hipSuccess) return e;
        if (keep && p && cap) {
            e = hipMemcpyAsync(q, p, cap, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { (void)hipFree(q); return e; }
        }
        if (p) (void)hipFree(p);
        p = q;
        cap = ncap;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct ReplayDS {
    Geo g;
    int KW = 1;           // 64-bit words of the sort key
    int FP = 4;           // shorts per SoA x row: 3HW rounded up to 4 (8-byte aligned rows)
    int64_t n_stage = 0;  // rows staged
    int64_t n_out = 0;    // dataset rows (after pos_average)
    bool finished = false;
    DevBuf st_x, st_vis, st_z, st_key;       // staging
    DevBuf perm, perm2, keyw, keyw2, flag, gid, gstart, tmp, sel, err;
    DevBuf ds_x, ds_pi, ds_z;                // dataset
    DevBuf lut;                              // [8][A] symmetry source indices
    DevBuf idx;                              // batch indices
    bool lut_ready = false;
    // batches queued on the caller's stream: a ring of pinned index buffers (host -> device copies from pinned memory are
    // asynchronous; a slot is reused only after the copy that read it has completed) and of device index buffers
    static constexpr int IDX_RING = 4;
    int32_t *idx_pin[IDX_RING] = {nullptr, nullptr, nullptr, nullptr};
    int32_t *idx_dev[IDX_RING] = {nullptr, nullptr, nullptr, nullptr};
    size_t idx_cap[IDX_RING] = {0, 0, 0, 0};
    hipEvent_t idx_ev[IDX_RING] = {nullptr, nullptr, nullptr, nullptr};
    int idx_slot = 0;
};

#define RCHECK(call)                                                                    \
    do {                                                                                \
        hipError_t _e = (call);                                                         \
        if (_e != hipSuccess) {                                                         \
            err = std::string(#call) + " failed: " + hipGetErrorString(_e);             \
            return DBAZ_EDEVICE;                                                        \
        }                                                                               \
    } while (0)

// ------------------------------------------------------------------------------------
// symmetry tables (host)
// ------------------------------------------------------------------------------------
int rds_symmetry_lut(const Geo &g, int sym, int32_t *lut, std::string &err)
{
    if (sym < 0 || sym > 7) { err = "symmetry id must be 0..7"; return DBAZ_EINVAL; }
    const int H = g.H, W = g.W, HW = g.HW;
    const bool fy = sym & 1, fx = sym & 2, rot = sym & 4; // IDXS order: None,(1,),(2,),(1,2) then the same + rotate
    if (rot && H != W) { err = "the rotating symmetries (4..7) need a square board"; return DBAZ_EINVAL; }
    // S: flipped planes.  h plane without its last column, v plane without its last row.
    auto s_h = [&](int y, int x) { return x < W - 1 ? (fy ? H - 1 - y : y) * W + (fx ? W - 2 - x : x) : y * W + (W - 1); };
    auto s_v = [&](int y, int x) { return y < H - 1 ? HW + (fy ? H - 2 - y : y) * W + (fx ? W - 1 - x : x) : HW + (H - 1) * W + x; };
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            if (!rot) {
                lut[y * W + x] = s_h(y, x);
                lut[HW + y * W + x] = s_v(y, x);
            } else { // R(S): h' = transpose of S's v plane, v' = transpose of S's h plane
                lut[y * W + x] = s_v(x, y);
                lut[HW + y * W + x] = s_h(x, y);
            }
        }
    return DBAZ_OK;
}

static int ensure_lut(ReplayDS *d, hipStream_t s, std::string &err)
{
    if (d->lut_ready) return DBAZ_OK;
    const int A = d->g.A;
    std::vector<int32_t> h((size_t)8 * A, -1);
    const int nsym = d->g.H == d->g.W ? 8 : 4;
    for (int sym = 0; sym < nsym; sym++) {
        int rc = rds_symmetry_lut(d->g, sym, h.data() + (size_t)sym * A, err);
        if (rc) return rc;
    }
    RCHECK(d->lut.ensure(h.size() * 4, false, s));
    RCHECK(hipMemcpyAsync(d->lut.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, s));
    RCHECK(hipStreamSynchronize(s));
    d->lut_ready = true;
    return DBAZ_OK;
}

// ------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wave per selected row, rows taken in a block-uniform loop.  The 8-byte-aligned packed row is
// read with 8-byte lane loads into a per-wave LDS image; fields are then picked from LDS at their
// natural (2-byte) alignment and leave as 8-byte stores (x rows are padded to FP = F rounded up to 4
// shorts so that every SoA row starts 8-byte aligned).
__global__ void __launch_bounds__(256) k_ds_stage(const unsigned char *__restrict__ rows, int row_bytes, int64_t n_rows,
                                                  const int32_t *__restrict__ sel, int64_t n_sel, int64_t base, int F, int FP, int A,
                                                  int KW, int16_t *st_x, int32_t *st_vis, int8_t *st_z,
                                                  unsigned long long *st_key, int32_t *errflag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int rbp = (row_bytes + 15) & ~15;
    unsigned char *img = lds_raw + (size_t)wave * rbp;
    const int units = row_bytes >> 3;
    const int64_t stride = (int64_t)gridDim.x * nw;
    const int64_t iters = (n_sel + stride - 1) / stride;
    for (int64_t it = 0; it < iters; it++) {
        const int64_t i = it * stride + (int64_t)blockIdx.x * nw + wave;
        int64_t r = -1;
        if (i < n_sel) {
            r = sel ? (int64_t)sel[i] : i;
            if (r < 0 || r >= n_rows) {
                if (lane == 0) atomicOr(errflag, 2);
                r = -1;
            }
        }
        if (r >= 0) {
            const uint2 *src = reinterpret_cast<const uint2 *>(rows + (size_t)r * row_bytes);
            for (int u = lane; u < units; u += 64) reinterpret_cast<uint2 *>(img)[u] = src[u];
        }
        __syncthreads();
        if (r >= 0) {
            const int64_t o = base + i;
            const short *sx = reinterpret_cast<const short *>(img + ROW_X_OFF);
            const unsigned short *sv = reinterpret_cast<const unsigned short *>(img + ROW_X_OFF + 2 * (size_t)F);
            const short x2 = sx[A];
            bool bad = false;
            for (int u = lane; u < FP / 4; u += 64) {
                short v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = 4 * u + j;
                    v[j] = k < F ? sx[k] : (short)0;
                    if (k < A) bad |= (v[j] != 0 && v[j] != 1);
                    else if (k < F) bad |= (v[j] != x2);
                }
                uint2 pk;
                pk.x = (unsigned)(unsigned short)v[0] | ((unsigned)(unsigned short)v[1] << 16);
                pk.y = (unsigned)(unsigned short)v[2] | ((unsigned)(unsigned short)v[3] << 16);
                reinterpret_cast<uint2 *>(st_x + o * FP)[u] = pk;
            }
            for (int u = lane; u < A / 2; u += 64) {
                uint2 pk;
                pk.x = (unsigned)sv[4 * u] | ((unsigned)sv[4 * u + 1] << 16);
                pk.y = (unsigned)sv[4 * u + 2] | ((unsigned)sv[4 * u + 3] << 16);
                reinterpret_cast<uint2 *>(st_vis + o * A)[u] = pk;
            }
            if (lane == 0) st_z[o] = (int8_t)img[ROW_Z_OFF];
            // key: bit b of [x_0 .. x_{A-1} | 16 bits of x_A + 32768, MSB first]; word w = bits 64w..64w+63, MSB first
            const unsigned v16 = (unsigned)((int)x2 + 32768) & 0xFFFFu;
            for (int w = 0; w < KW; w++) {
                const int bidx = w * 64 + lane;
                int bit = 0;
                if (bidx < A) bit = sx[bidx] & 1;
                else if (bidx < A + 16) bit = (v16 >> (15 - (bidx - A))) & 1;
                const unsigned long long m = __ballot(bit);
                if (lane == 0) st_key[o * KW + w] = __brevll(m);
            }
            if (__ballot(bad) && lane == 0) atomicOr(errflag, 1);
        }
        __syncthreads();
    }
}

__global__ void k_iota(int32_t *p, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (int32_t)i;
}

__global__ void k_ds_gather_key(const unsigned long long *__restrict__ st_key, int KW, int w, const int32_t *__restrict__ perm,
                                unsigned long long *out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = st_key[(size_t)perm[i] * KW + w];
}

__global__ void k_ds_flags(const unsigned long long *__restrict__ st_key, int KW, const int32_t *__restrict__ perm, int32_t *flag,
                           int64_t n, int every)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int f = (i == 0) || every;
    if (!f) {
        const unsigned long long *a = st_key + (size_t)perm[i] * KW, *b = st_key + (size_t)perm[i - 1] * KW;
        for (int w = 0; w < KW; w++) f |= a[w] != b[w];
    }
    flag[i] = f;
}

__global__ void k_ds_gstart(const int32_t *__restrict__ flag, const int32_t *__restrict__ gid, int32_t *gstart, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) gstart[gid[i] - 1] = (int32_t)i;
    if (i == n - 1) gstart[gid[i]] = (int32_t)n;
}

// one wave per group; VPL values of the A pi columns per lane; rows of the group in staging order
template <int VPL>
__global__ void __launch_bounds__(256) k_ds_mean(const int16_t *__restrict__ st_x, const int32_t *__restrict__ st_vis,
                                                 const int8_t *__restrict__ st_z, const int32_t *__restrict__ perm,
                                                 const int32_t *__restrict__ gstart, int64_t n_groups, int FP, int A,
                                                 int16_t *ds_x, float *ds_pi, float *ds_z)
{
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const int i0 = gstart[g], i1 = gstart[g + 1];
    double sum[VPL], comp[VPL], zsum = 0.0, zcomp = 0.0;
#pragma unroll
    for (int j = 0; j < VPL; j++) sum[j] = comp[j] = 0.0;
    for (int i = i0; i < i1; i++) {
        const int64_t r = perm[i];
        int vis[VPL], part = 0;
#pragma unroll
        for (int j = 0; j < VPL; j++) {
            const int a = lane + 64 * j;
            vis[j] = a < A ? st_vis[r * A + a] : 0;
            part += vis[j];
        }
        const double tot = (double)wave_sum_i32(part);
#pragma unroll
        for (int j = 0; j < VPL; j++) { // pandas group_mean (Kahan)
            const double val = (double)vis[j] / tot;
            const double y = val - comp[j];
            const double t = sum[j] + y;
            comp[j] = t - sum[j] - y;
            sum[j] = t;
        }
        {
            const double val = (double)st_z[r];
            const double y = val - zcomp;
            const double t = zsum + y;
            zcomp = t - zsum - y;
            zsum = t;
        }
    }
    const double cnt = (double)(i1 - i0);
#pragma unroll
    for (int j = 0; j < VPL; j++) {
        const int a = lane + 64 * j;
        if (a < A) ds_pi[g * A + a] = (float)(sum[j] / cnt);
    }
    if (lane == 0) ds_z[g] = (float)(zsum / cnt);
    const int64_t r0 = perm[i0];
    for (int u = lane; u < FP / 4; u += 64)
        reinterpret_cast<uint2 *>(ds_x + g * FP)[u] = reinterpret_cast<const uint2 *>(st_x + r0 * FP)[u];
}

// One wave per output row (block-uniform loop): the dataset row (x padded to FP shorts, pi) enters
// a per-wave LDS image with 8-byte loads, the symmetry LUT sits in LDS once per block, and the
// float32 outputs leave coalesced.
__global__ void __launch_bounds__(256) k_make_batch(const int16_t *__restrict__ ds_x, const float *__restrict__ ds_pi,
                                                    const float *__restrict__ ds_z, const int32_t *__restrict__ idx, int n,
                                                    int64_t n_ds, const int32_t *__restrict__ lut /* null: identity */, int F, int FP,
                                                    int A, float *boards, float *pi, float *z, int32_t *errflag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int *lut_s = reinterpret_cast<int *>(lds_raw);
    const int per_wave = FP * 2 + A * 4; // both multiples of 8
    short *xrow = reinterpret_cast<short *>(lds_raw + (size_t)A * 4 + (size_t)wave * per_wave);
    float *prow = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(xrow) + FP * 2);
    for (int k = threadIdx.x; k < A; k += blockDim.x) lut_s[k] = lut ? lut[k] : k;
    const int stride = gridDim.x * nw;
    const int iters = (n + stride - 1) / stride;
    for (int it = 0; it < iters; it++) {
        const int i = it * stride + blockIdx.x * nw + wave;
        int64_t r = -1;
        if (i < n) {
            r = idx[i];
            if (r < 0 || r >= n_ds) {
                if (lane == 0) atomicOr(errflag, 2);
                r = -1;
            }
        }
        if (r >= 0) {
            for (int u = lane; u < FP / 4; u += 64) reinterpret_cast<uint2 *>(xrow)[u] = reinterpret_cast<const uint2 *>(ds_x + r * FP)[u];
            for (int u = lane; u < A / 2; u += 64) reinterpret_cast<uint2 *>(prow)[u] = reinterpret_cast<const uint2 *>(ds_pi + r * A)[u];
        }
        __syncthreads();
        if (r >= 0) {
            for (int k = lane; k < F; k += 64) boards[(size_t)i * F + k] = (float)xrow[k < A ? lut_s[k] : k];
            for (int u = lane; u < A / 2; u += 64) {
                float2 o2;
                o2.x = prow[lut_s[2 * u]];
                o2.y = prow[lut_s[2 * u + 1]];
                reinterpret_cast<float2 *>(pi + (size_t)i * A)[u] = o2;
            }
            if (lane == 0) z[i] = ds_z[r];
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_symmetry(const float *__restrict__ bin, const float *__restrict__ pin, int64_t n,
                                                  const int32_t *__restrict__ lut, int F, int A, float *bout, float *pout)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    if (bin)
        for (int k = lane; k < F; k += 64) bout[i * F + k] = bin[i * F + ((k < A && lut) ? lut[k] : k)];
    if (pin)
        for (int k = lane; k < A; k += 64) pout[i * A + k] = pin[i * A + (lut ? lut[k] : k)];
}

// ------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------
ReplayDS *rds_create(const Geo &g)
{
    ReplayDS *d = new ReplayDS();
    d->g = g;
    d->KW = (g.A + 16 + 63) / 64;
    d->FP = (3 * g.HW + 3) & ~3;
    return d;
}

void rds_destroy(ReplayDS *d)
{
    if (!d) return;
    DevBuf *all[] = {&d->st_x, &d->st_vis, &d->st_z, &d->st_key, &d->perm, &d->perm2, &d->keyw, &d->keyw2, &d->flag, &d->gid,
                     &d->gstart, &d->tmp, &d->sel, &d->err, &d->ds_x, &d->ds_pi, &d->ds_z, &d->lut, &d->idx};
    for (DevBuf *b : all) b->release();
    for (int i = 0; i < ReplayDS::IDX_RING; i++) {
        if (d->idx_ev[i]) { (void)hipEventSynchronize(d->idx_ev[i]); (void)hipEventDestroy(d->idx_ev[i]); }
        if (d->idx_pin[i]) (void)hipHostFree(d->idx_pin[i]);
        if (d->idx_dev[i]) (void)hipFree(d->idx_dev[i]);
    }
    delete d;
}

int64_t rds_size(const ReplayDS *d) { return d && d->finished ? d->n_out : 0; }

static int read_errflag(ReplayDS *d, hipStream_t s, int &flag, std::string &err)
{
    RCHECK(hipMemcpyAsync(&flag, d->err.p, 4, hipMemcpyDeviceToHost, s));
    RCHECK(hipStreamSynchronize(s));
    return DBAZ_OK;
}

int rds_begin(ReplayDS *d, std::string &err)
{
    (void)err;
    d->n_stage = 0;
    d->n_out = 0;
    d->finished = false;
    return DBAZ_OK;
}

int rds_add_rows(ReplayDS *d, hipStream_t s, const void *rows_dev, int64_t n_rows, int row_bytes, const int32_t *sel_host,
                 int64_t n_sel, std::string &err)
{
    const Geo &g = d->g;
    const int F = 3 * g.HW, A = g.A;
    const int need_rb = (int)((ROW_X_OFF + (size_t)F * 2 + (size_t)A * 4 + 7) & ~(size_t)7);
    if (d->finished) { err = "dataset already finished: call dbaz_dataset_begin first"; return DBAZ_ESTATE; }
    if (row_bytes != need_rb) { err = "row_bytes does not match this board's replay row"; return DBAZ_EINVAL; }
    if (n_rows < 0 || n_sel < 0 || (!rows_dev && n_rows > 0)) { err = "bad row arguments"; return DBAZ_EINVAL; }
    if (((uintptr_t)rows_dev & 7) != 0) { err = "replay rows must be 8-byte aligned"; return DBAZ_EINVAL; }
    if (!sel_host) n_sel = n_rows;
    if (n_sel == 0) return DBAZ_OK;
    if (d->n_stage + n_sel > 0x7FFFFFF0LL) { err = "dataset too large (int32 row indices)"; return DBAZ_EINVAL; }
    const int64_t tot = d->n_stage + n_sel;
    RCHECK(d->st_x.ensure((size_t)tot * d->FP * 2, true, s));
    RCHECK(d->st_vis.ensure((size_t)tot * A * 4, true, s));
    RCHECK(d->st_z.ensure((size_t)tot, true, s));
    RCHECK(d->st_key.ensure((size_t)tot * d->KW * 8, true, s));
    RCHECK(d->err.ensure(16, false, s));
    RCHECK(hipMemsetAsync(d->err.p, 0, 16, s));
    const int32_t *sel_dev = nullptr;
    if (sel_host) {
        RCHECK(d->sel.ensure((size_t)n_sel * 4, false, s));
        RCHECK(hipMemcpyAsync(d->sel.p, sel_host, (size_t)n_sel * 4, hipMemcpyHostToDevice, s));
        sel_dev = (const int32_t *)d->sel.p;
    }
    const unsigned blocks = (unsigned)std::min<int64_t>((n_sel + 3) / 4, 256 * 32);
    const size_t lds_stage = 4 * (size_t)((row_bytes + 15) & ~15);
    hipLaunchKernelGGL(k_ds_stage, dim3(blocks), dim3(256), lds_stage, s, (const unsigned char *)rows_dev, row_bytes, n_rows, sel_dev, n_sel,
                       d->n_stage, F, d->FP, A, d->KW, (int16_t *)d->st_x.p, (int32_t *)d->st_vis.p, (int8_t *)d->st_z.p,
                       (unsigned long long *)d->st_key.p, (int32_t *)d->err.p);
    RCHECK(hipGetLastError());
    int flag = 0;
    int rc = read_errflag(d, s, flag, err);
    if (rc) return rc;
    if (flag & 2) { err = "row selection index out of range"; return DBAZ_EINVAL; }
    if (flag & 1) { err = "rows are not BoxesState feature rows (edge planes must be 0/1, plane 2 constant)"; return DBAZ_EINVAL; }
    d->n_stage = tot;
    return DBAZ_OK;
}

int rds_finish(ReplayDS *d, hipStream_t s, int pos_average, const int32_t *order_host, int64_t *n_out, std::string &err)
{
    const Geo &g = d->g;
    const int A = g.A, KW = d->KW;
    const int64_t n = d->n_stage;
    if (d->finished) { err = "dataset already finished"; return DBAZ_ESTATE; }
    d->n_out = 0;
    if (n == 0) {
        d->finished = true;
        if (n_out) *n_out = 0;
        return DBAZ_OK;
    }
    const unsigned tb = 256, nb = (unsigned)((n + tb - 1) / tb);
    RCHECK(d->perm.ensure((size_t)n * 4, false, s));
    RCHECK(d->flag.ensure((size_t)n * 4, false, s));
    RCHECK(d->gid.ensure((size_t)n * 4, false, s));
    RCHECK(d->gstart.ensure((size_t)(n + 1) * 4, false, s));
    if (order_host) { // dataset order given as a permutation of the staged rows
        std::vector<uint8_t> seen((size_t)n, 0);
        for (int64_t i = 0; i < n; i++) {
            const int32_t o = order_host[i];
            if (o < 0 || o >= n || seen[o]) { err = "order is not a permutation of the staged rows"; return DBAZ_EINVAL; }
            seen[o] = 1;
        }
        RCHECK(hipMemcpyAsync(d->perm.p, order_host, (size_t)n * 4, hipMemcpyHostToDevice, s));
        RCHECK(hipStreamSynchronize(s));
    } else {
        hipLaunchKernelGGL(k_iota, dim3(nb), dim3(tb), 0, s, (int32_t *)d->perm.p, n);
    }
    if (pos_average) {
        RCHECK(d->perm2.ensure((size_t)n * 4, false, s));
        RCHECK(d->keyw.ensure((size_t)n * 8, false, s));
        RCHECK(d->keyw2.ensure((size_t)n * 8, false, s));
        size_t tb_sort = 0;
        RCHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb_sort, (unsigned long long *)d->keyw.p, (unsigned long long *)d->keyw2.p,
                                                  (int32_t *)d->perm.p, (int32_t *)d->perm2.p, (int)n, 0, 64, s));
        RCHECK(d->tmp.ensure(tb_sort, false, s));
        // LSD: least significant key word first; every pass is stable, so equal keys keep staging order
        for (int w = KW - 1; w >= 0; w--) {
            hipLaunchKernelGGL(k_ds_gather_key, dim3(nb), dim3(tb), 0, s, (const unsigned long long *)d->st_key.p, KW, w,
                               (const int32_t *)d->perm.p, (unsigned long long *)d->keyw.p, n);
            size_t bytes = d->tmp.cap;
            RCHECK(hipcub::DeviceRadixSort::SortPairs(d->tmp.p, bytes, (unsigned long long *)d->keyw.p, (unsigned long long *)d->keyw2.p,
                                                      (int32_t *)d->perm.p, (int32_t *)d->perm2.p, (int)n, 0, 64, s));
            std::swap(d->perm, d->perm2);
        }
    }
    hipLaunchKernelGGL(k_ds_flags, dim3(nb), dim3(tb), 0, s, (const unsigned long long *)d->st_key.p, KW, (const int32_t *)d->perm.p,
                       (int32_t *)d->flag.p, n, pos_average ? 0 : 1);
    size_t tb_scan = 0;
    RCHECK(hipcub::DeviceScan::InclusiveSum(nullptr, tb_scan, (int32_t *)d->flag.p, (int32_t *)d->gid.p, (int)n, s));
    RCHECK(d->tmp.ensure(tb_scan, false, s));
    size_t bytes = d->tmp.cap;
    RCHECK(hipcub::DeviceScan::InclusiveSum(d->tmp.p, bytes, (int32_t *)d->flag.p, (int32_t *)d->gid.p, (int)n, s));
    hipLaunchKernelGGL(k_ds_gstart, dim3(nb), dim3(tb), 0, s, (const int32_t *)d->flag.p, (const int32_t *)d->gid.p,
                       (int32_t *)d->gstart.p, n);
    int32_t m = 0;
    RCHECK(hipMemcpyAsync(&m, (const int32_t *)d->gid.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
    RCHECK(hipStreamSynchronize(s));
    RCHECK(d->ds_x.ensure((size_t)m * d->FP * 2, false, s));
    RCHECK(d->ds_pi.ensure((size_t)m * A * 4, false, s));
    RCHECK(d->ds_z.ensure((size_t)m * 4, false, s));
    const unsigned gb = (unsigned)((m + 3) / 4);
    const int vpl = (A + 63) / 64;
#define MEAN(V)                                                                                                              \
    hipLaunchKernelGGL(k_ds_mean<V>, dim3(gb), dim3(256), 0, s, (const int16_t *)d->st_x.p, (const int32_t *)d->st_vis.p,     \
                       (const int8_t *)d->st_z.p, (const int32_t *)d->perm.p, (const int32_t *)d->gstart.p, (int64_t)m, d->FP, A, \
                       (int16_t *)d->ds_x.p, (float *)d->ds_pi.p, (float *)d->ds_z.p)
    switch (vpl) {
    case 1: MEAN(1); break;
    case 2: MEAN(2); break;
    case 3: MEAN(3); break;
    case 4: MEAN(4); break;
    default: err = "board too large for the dataset kernels (A > 256)"; return DBAZ_EINVAL;
    }
#undef MEAN
    RCHECK(hipGetLastError());
    RCHECK(hipStreamSynchronize(s));
    d->n_out = m;
    d->finished = true;
    if (n_out) *n_out = m;
    return DBAZ_OK;
}

int rds_fetch(ReplayDS *d, hipStream_t s, int16_t *x, float *pi, float *z, std::string &err)
{
    if (!d->finished) { err = "no dataset: call dbaz_dataset_finish first"; return DBAZ_ESTATE; }
    const int F = 3 * d->g.HW, A = d->g.A;
    const int64_t m = d->n_out;
    if (m == 0) return DBAZ_OK;
    if (x) RCHECK(hipMemcpy2DAsync(x, (size_t)F * 2, d->ds_x.p, (size_t)d->FP * 2, (size_t)F * 2, (size_t)m, hipMemcpyDeviceToHost, s));
    if (pi) RCHECK(hipMemcpyAsync(pi, d->ds_pi.p, (size_t)m * A * 4, hipMemcpyDeviceToHost, s));
    if (z) RCHECK(hipMemcpyAsync(z, d->ds_z.p, (size_t)m * 4, hipMemcpyDeviceToHost, s));
    RCHECK(hipStreamSynchronize(s));
    return DBAZ_OK;
}

// on_caller_stream: the batch is QUEUED on `s` (the caller's own stream, e.g. torch's current one) and the call returns at once:
// the indices are checked on the host, nothing is read back.  Writing the caller's freshly allocated tensors from another stream
// would race with work the caller has queued on the memory's previous owner (a caching allocator recycles blocks in stream order).
int rds_batch(ReplayDS *d, hipStream_t s, const int32_t *idx_host, int n, int sym, float *boards_dev, float *pi_dev, float *z_dev,
              std::string &err, bool on_caller_stream)
{
    if (!d->finished) { err = "no dataset: call dbaz_dataset_finish first"; return DBAZ_ESTATE; }
    if (sym < 0 || sym > 7) { err = "symmetry id must be 0..7"; return DBAZ_EINVAL; }
    if (sym >= 4 && d->g.H != d->g.W) { err = "the rotating symmetries (4..7) need a square board"; return DBAZ_EINVAL; }
    if (n == 0) return DBAZ_OK;
    if (n < 0 || !idx_host || !boards_dev || !pi_dev || !z_dev) { err = "null argument"; return DBAZ_EINVAL; }
    const int F = 3 * d->g.HW, A = d->g.A;
    if (on_caller_stream)
        for (int i = 0; i < n; i++)
            if (idx_host[i] < 0 || (long long)idx_host[i] >= (long long)d->n_out) { err = "batch index out of range"; return DBAZ_EINVAL; }
    int rc = ensure_lut(d, s, err);
    if (rc) return rc;
    RCHECK(d->err.ensure(16, false, s));
    const int32_t *idx_dev = nullptr;
    if (on_caller_stream) {
        const int k = d->idx_slot;
        d->idx_slot = (k + 1) % ReplayDS::IDX_RING;
        if (d->idx_ev[k]) RCHECK(hipEventSynchronize(d->idx_ev[k])); // (only if the device is IDX_RING batches behind)
        else RCHECK(hipEventCreateWithFlags(&d->idx_ev[k], hipEventDisableTiming));
        if (d->idx_cap[k] < (size_t)n) {
            if (d->idx_pin[k]) RCHECK(hipHostFree(d->idx_pin[k]));
            if (d->idx_dev[k]) RCHECK(hipFree(d->idx_dev[k]));
            d->idx_pin[k] = nullptr; d->idx_dev[k] = nullptr; d->idx_cap[k] = 0;
            RCHECK(hipHostMalloc((void **)&d->idx_pin[k], (size_t)n * 4, hipHostMallocDefault));
            RCHECK(hipMalloc((void **)&d->idx_dev[k], (size_t)n * 4));
            d->idx_cap[k] = (size_t)n;
        }
        memcpy(d->idx_pin[k], idx_host, (size_t)n * 4);
        RCHECK(hipMemcpyAsync(d->idx_dev[k], d->idx_pin[k], (size_t)n * 4, hipMemcpyHostToDevice, s));
        idx_dev = d->idx_dev[k];
    } else {
        RCHECK(d->idx.ensure((size_t)n * 4, false, s));
        RCHECK(hipMemsetAsync(d->err.p, 0, 16, s));
        RCHECK(hipMemcpyAsync(d->idx.p, idx_host, (size_t)n * 4, hipMemcpyHostToDevice, s));
        idx_dev = (const int32_t *)d->idx.p;
    }
    const int32_t *lut = sym ? (const int32_t *)d->lut.p + (size_t)sym * A : nullptr;
    const size_t lds_batch = (size_t)A * 4 + 4 * ((size_t)d->FP * 2 + (size_t)A * 4);
    hipLaunchKernelGGL(k_make_batch, dim3(std::min((n + 3) / 4, 256 * 32)), dim3(256), lds_batch, s, (const int16_t *)d->ds_x.p,
                       (const float *)d->ds_pi.p, (const float *)d->ds_z.p, idx_dev, n, d->n_out, lut, F, d->FP, A,
                       boards_dev, pi_dev, z_dev,
                       (int32_t *)d->err.p);
    RCHECK(hipGetLastError());
    if (on_caller_stream) { // (indices were validated below the argument checks; ordering is the stream's)
        RCHECK(hipEventRecord(d->idx_ev[d->idx_slot == 0 ? ReplayDS::IDX_RING - 1 : d->idx_slot - 1], s));
        return DBAZ_OK;
    }
    int flag = 0;
    rc = read_errflag(d, s, flag, err); // also orders the batch before the caller's own stream
    if (rc) return rc;
    if (flag) { err = "batch index out of range"; return DBAZ_EINVAL; }
    return DBAZ_OK;
}

int rds_symmetry_apply(ReplayDS *d, hipStream_t s, int sym, const float *boards_in, const float *pol_in, int64_t n, float *boards_out,
                       float *pol_out, std::string &err)
{
    if (sym < 0 || sym > 7) { err = "symmetry id must be 0..7"; return DBAZ_EINVAL; }
    if (sym >= 4 && d->g.H != d->g.W) { err = "the rotating symmetries (4..7) need a square board"; return DBAZ_EINVAL; }
    if (n < 0 || (boards_in && !boards_out) || (pol_in && !pol_out)) { err = "bad arguments"; return DBAZ_EINVAL; }
    if ((boards_in && boards_in == boards_out) || (pol_in && pol_in == pol_out)) { err = "in-place symmetry is not supported"; return DBAZ_EINVAL; }
    if (n == 0) return DBAZ_OK;
    int rc = ensure_lut(d, s, err);
    if (rc) return rc;
    const int F = 3 * d->g.HW, A = d->g.A;
    const int32_t *lut = sym ? (const int32_t *)d->lut.p + (size_t)sym * A : nullptr;
    hipLaunchKernelGGL(k_symmetry, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, boards_in, pol_in, n, lut, F, A, boards_out, pol_out);
    RCHECK(hipGetLastError());
    RCHECK(hipStreamSynchronize(s));
    return DBAZ_OK;
}
