// tree.hip -- per-simulation MCTS kernels, one 64-lane wavefront per game.
//
// Restates the reference's sequential PUCT search (mcts.py, max_pending_evals = 1):
//   k_select          select_leaf      mcts.py:105-114  (+ children_ucb_score/best_child :91-103,
//                                                           lazy child creation utils/utils.py:51-58)
//   k_expand_backup   _search tail     mcts.py:186-199  (prior masking, expand :116-119, backup :121-132)
//   root_prep         UCT_search       mcts.py:211-226  (root renormalisation + Dirichlet mix)
//   k_search_begin    UCT_search       mcts.py:205-208
//   k_advance         init_mcts_tree   mcts.py:163-180  + SelfPlay.play_game / get_next_move /
//                                      get_datasets rows (self_play.py:27-35, 51-74, 95-156)
//
// Numerics follow numpy (NEP 50) exactly: UCB in float64, statistics in float32/int32,
// float32 prior sums with numpy's pairwise summation.  This TU is compiled with
// -ffp-contract=off: a fused multiply-add would change the roundings.
#include "rules.h"
#include "tree.h"

// ------------------------------------------------------------------------------------
// node access
// ------------------------------------------------------------------------------------
struct NodeMeta {
    GState st;
    int parent, move, flags, result, deepness;
};

__device__ __forceinline__ uint32_t *node_ptr(uint32_t *pool, const Geo &g, int idx)
{
    return pool + (size_t)idx * g.node_dw;
}

__device__ __forceinline__ NodeMeta load_meta(uint32_t *pool, const Geo &g, int idx)
{
    const uint32_t *nd = node_ptr(pool, g, idx);
    const uint4 a = *reinterpret_cast<const uint4 *>(nd);
    const uint4 b = *reinterpret_cast<const uint4 *>(nd + 4);
    const uint4 c = *reinterpret_cast<const uint4 *>(nd + 8);
    NodeMeta m;
    m.st.e0 = (uint64_t)a.x | ((uint64_t)a.y << 32);
    m.st.e1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
    m.st.e2 = (uint64_t)b.x | ((uint64_t)b.y << 32);
    m.st.e3 = (uint64_t)b.z | ((uint64_t)b.w << 32);
    m.parent = (int)c.x;
    m.move = (int)(int16_t)(c.y & 0xFFFFu);
    m.st.to_play = (int)((c.y >> 16) & 0xFFu);
    m.st.just_played = (int)((c.y >> 24) & 0xFFu) - 1;
    m.st.b2c0 = (int)(int16_t)(c.z & 0xFFFFu);
    m.st.b2c1 = (int)(int16_t)(c.z >> 16);
    m.flags = (int)(c.w & 0xFFu);
    m.result = (int)((c.w >> 8) & 0xFFu) - 1;
    m.deepness = (int)(c.w >> 16);
    return m;
}

__device__ __forceinline__ uint32_t pack_dw9(int move, int to_play, int just_played)
{
    return ((uint32_t)(uint16_t)(int16_t)move) | ((uint32_t)(to_play & 0xFF) << 16) |
           ((uint32_t)((just_played + 1) & 0xFF) << 24);
}
__device__ __forceinline__ uint32_t pack_dw10(int b0, int b1)
{
    return ((uint32_t)(uint16_t)(int16_t)b0) | ((uint32_t)(uint16_t)(int16_t)b1 << 16);
}
__device__ __forceinline__ uint32_t pack_dw11(int flags, int result, int deepness)
{
    return (uint32_t)(flags & 0xFF) | ((uint32_t)((result + 1) & 0xFF) << 8) | ((uint32_t)deepness << 16);
}

// UCTNode.__init__ (mcts.py:47-65): zeroed W / visits, no children; P is written at expand.
__device__ void init_node(uint32_t *pool, const Geo &g, int idx, const GState &st, int parent, int move,
                          int deepness, int lane)
{
    uint32_t *nd = node_ptr(pool, g, idx);
    int res = gs_result(st);
    int flags = (res != DBAZ_RESULT_NONE) ? NF_TERMINAL : 0;
    if (lane < META_DW) {
        uint32_t v = 0;
        switch (lane) {
        case 0: v = (uint32_t)st.e0; break;
        case 1: v = (uint32_t)(st.e0 >> 32); break;
        case 2: v = (uint32_t)st.e1; break;
        case 3: v = (uint32_t)(st.e1 >> 32); break;
        case 4: v = (uint32_t)st.e2; break;
        case 5: v = (uint32_t)(st.e2 >> 32); break;
        case 6: v = (uint32_t)st.e3; break;
        case 7: v = (uint32_t)(st.e3 >> 32); break;
        case 8: v = (uint32_t)parent; break;
        case 9: v = pack_dw9(move, st.to_play, st.just_played); break;
        case 10: v = pack_dw10(st.b2c0, st.b2c1); break;
        case 11: v = pack_dw11(flags, res, deepness); break;
        default: v = 0; break;
        }
        nd[lane] = v;
    }
    uint32_t *rows = nd + META_DW;
    for (int i = lane; i < g.AS; i += WAVE) {
        rows[g.AS + i] = 0u;             // W = 0.0f
        rows[2 * g.AS + i] = 0u;         // visits = 0
        rows[3 * g.AS + i] = 0xFFFFFFFFu; // child = -1
    }
}

// ------------------------------------------------------------------------------------
// Node pool of one game.  Re-rooting (init_mcts_tree, mcts.py:163-180) moves nothing: the chosen child
// simply becomes the root, and the old root -- its link to the kept child cut -- is handed to a
// collector that enumerates the dropped part of the tree a few nodes per simulation (ring `pend`:
// dropped nodes whose child row is still needed; stack `freel`: indices ready for reuse).  In steady state
// one node is created per simulation and GC_PER_STEP are recycled, so the pool never grows past
// the tree plus one move's worth of garbage.  (Round 1 compacted the kept subtree in place: 2.5-4 MB
// moved by one wave for 0.4-2 ms per move, on a CU the network's workgroups then could not use.)
// ------------------------------------------------------------------------------------
#ifndef GC_PER_STEP
#define GC_PER_STEP 2
#endif
struct PoolState { int n_nodes, n_free, head, tail; };

__device__ __forceinline__ PoolState pool_load(const Slot *S)
{
    PoolState q;
    q.n_nodes = S->n_nodes; q.n_free = S->n_free; q.head = S->pend_head; q.tail = S->pend_tail;
    return q;
}
__device__ __forceinline__ void pool_store(const PoolState &q, Slot *S)
{
    S->n_nodes = q.n_nodes; S->n_free = q.n_free; S->pend_head = q.head; S->pend_tail = q.tail;
}
__device__ __forceinline__ int ring_at(int i, int cap) { return i >= cap ? i - cap : i; }

// Takes up to NB (<= 4) nodes off the pending ring: their children join the ring, the nodes themselves the free
// stack.  Wave-cooperative, wave-uniform state.  Two halves so that the row loads of the batch (all in flight
// together) can overlap other work of the wave: gc_issue reserves the entries and requests their child rows,
// gc_finish consumes them.  Only entries that were on the ring at gc_issue are read.
template <int NB>
struct GcBatch {
    int k;
    int node[NB];
    int32_t c[NB][4];
};

template <int NB>
__device__ __forceinline__ void gc_issue(GcBatch<NB> &b, const Geo &g, const TreeBufs &B, int slot, uint32_t *pool, PoolState &q, int lane)
{
    const int avail = q.tail - q.head;
    b.k = avail < NB ? (avail > 0 ? avail : 0) : NB;
    if (b.k == 0) return;
    const int32_t *pend = B.pend + (size_t)slot * g.cap;
    const int h0 = q.head; // head stays in [0, cap), tail in [head, head + cap]
#pragma unroll
    for (int t = 0; t < NB; t++) b.node[t] = t < b.k ? pend[ring_at(h0 + t, g.cap)] : 0;
#pragma unroll
    for (int t = 0; t < NB; t++) {
        const int32_t *Crow = reinterpret_cast<const int32_t *>(node_ptr(pool, g, b.node[t]) + META_DW + 3 * g.AS);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = lane + WAVE * j;
            b.c[t][j] = (t < b.k && i < g.A) ? Crow[i] : -1;
        }
    }
    q.head += b.k;
}

template <int NB>
__device__ __forceinline__ void gc_finish(const GcBatch<NB> &b, const Geo &g, const TreeBufs &B, int slot, PoolState &q, int lane)
{
    if (b.k == 0) return;
    int32_t *pend = B.pend + (size_t)slot * g.cap, *fl = B.freel + (size_t)slot * g.cap;
    int tl = ring_at(q.tail, g.cap);
#pragma unroll
    for (int t = 0; t < NB; t++) {
        if (t < b.k) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (WAVE * j < g.A) {
                    const bool has = b.c[t][j] >= 0;
                    const unsigned long long m = __ballot(has);
                    if (has) pend[ring_at(tl + (int)__popcll(m & ((1ull << lane) - 1ull)), g.cap)] = b.c[t][j];
                    const int n = (int)__popcll(m);
                    tl = ring_at(tl + n, g.cap);
                    q.tail += n;
                }
            }
            if (lane == 0) fl[q.n_free] = b.node[t];
            q.n_free++;
        }
    }
    if (q.head >= g.cap) { q.head -= g.cap; q.tail -= g.cap; } // keep the counters small (tail - head <= cap always)
}

template <int NB>
__device__ __forceinline__ void pool_collect(const Geo &g, const TreeBufs &B, int slot, uint32_t *pool, PoolState &q, int lane)
{
    GcBatch<NB> b;
    gc_issue<NB>(b, g, B, slot, pool, q, lane);
    gc_finish<NB>(b, g, B, slot, q, lane);
}

// UCTNode creation needs an index: recycled first, then fresh, then whatever the collector can still find
__device__ __forceinline__ int pool_alloc(const Geo &g, const TreeBufs &B, int slot, uint32_t *pool, PoolState &q, int lane)
{
    if (q.n_free == 0 && q.n_nodes >= g.cap) {
        while (q.n_free == 0 && q.tail > q.head) {
            pool_collect<4>(g, B, slot, pool, q, lane);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the next batch reads ring entries this one pushed
        }
    }
    if (q.n_free > 0) {
        q.n_free--;
        return (B.freel + (size_t)slot * g.cap)[q.n_free]; // wave-uniform load
    }
    if (q.n_nodes < g.cap) return q.n_nodes++;
    return -1;
}

// ------------------------------------------------------------------------------------
// numpy pairwise summation on LDS data (<= 256 elements), result broadcast to the wave.
// Mirrors @TYPE@_pairwise_sum (numpy loops_utils.h.src): 8 strided accumulators per
// block of <= 128, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), tail added in order.
// ------------------------------------------------------------------------------------
template <typename T>
__device__ T np_pairwise_sum(const T *a, int n, int lane)
{
    int grp = lane >> 3, j = lane & 7;
    int s = 0, L = n;
    int n2 = 0;
    if (n > 128) {
        n2 = n / 2;
        n2 -= n2 % 8;
        if (grp == 0) { s = 0; L = n2; } else { s = n2; L = n - n2; }
    }
    T res;
    if (L < 8) {
        res = (T)(-0.0);
        for (int i = 0; i < L; i++)
            res += a[s + i];
    } else {
        T r = a[s + j];
        int lim = L - (L % 8);
        for (int i = 8; i < lim; i += 8)
            r += a[s + i + j];
        T t = r + __shfl_down(r, 1);
        T u = t + __shfl_down(t, 2);
        res = u + __shfl_down(u, 4);
        for (int i = lim; i < L; i++)
            res += a[s + i];
    }
    T b0 = __shfl(res, 0);
    if (n > 128) {
        T b1 = __shfl(res, 8);
        return b0 + b1;
    }
    return b0;
}

// ------------------------------------------------------------------------------------
// formula evaluators (restated from oracle ob_eval_formula; a pure function of the hash)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t fmix64(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

__device__ __forceinline__ uint64_t formula_hash(const GState &st)
{
    uint64_t h = 0x9E3779B97F4A7C15ULL;
    h = fmix64(h ^ st.e0);
    h = fmix64(h ^ st.e1);
    h = fmix64(h ^ st.e2);
    h = fmix64(h ^ st.e3);
    int b = st.to_play == 0 ? st.b2c0 : st.b2c1;
    h = fmix64(h ^ (uint64_t)(int64_t)(b + 512));
    return h;
}
__device__ __forceinline__ float formula_p(uint64_t h, int i, int kind)
{
    if (kind == DBAZ_EVAL_FORMULA_UNIFORM)
        return 1.0f;
    uint64_t t = fmix64(h + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ULL);
    uint32_t u = (uint32_t)(t >> 20) & 0xFFFFu;
    return (float)((u & 0xFFu) + 1u) * (float)(((u >> 8) & 0xFFu) + 1u);
}
__device__ __forceinline__ float formula_v(uint64_t h, int kind)
{
    if (kind == DBAZ_EVAL_FORMULA_UNIFORM)
        return 0.0f;
    uint64_t t = fmix64(h ^ 0xD6E8FEB86659FD93ULL);
    uint32_t u = (uint32_t)(t >> 17) & 0xFFFFu;
    return ((float)u - 32768.0f) / 32768.0f;
}

// ---- per-game transposition table (the reference's (p, v) cache by position hash, utils/proxies.py:35-43) ----
// Entry = tag24 | epoch8 | node index.  A hit is only taken after the candidate node of the LIVE tree has been
// checked: same network input (edges and the mover's boxes_to_close, exactly the reference's hash) and already
// expanded -- so stale entries (old epochs, renumbered nodes) can never return a wrong result, and (p, v) need not
// be stored: the twin's prior row IS masked(p) normalised for the same valid moves, its value sits in meta dword 12.
#define TT_PROBES 4
__device__ __forceinline__ int tt_mover_b2c(const GState &st) { return st.to_play == 0 ? st.b2c0 : st.b2c1; }
__device__ __forceinline__ unsigned tt_epoch_byte(int epoch) { return (unsigned)(epoch % 255) + 1u; }
__device__ __forceinline__ unsigned long long tt_entry(uint64_t h, unsigned ep, int idx)
{
    return ((h >> 40) << 40) | ((unsigned long long)ep << 32) | (unsigned long long)(unsigned)idx;
}

// ------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (production move sampling / Dirichlet noise; the reference
// uses numpy's global MT19937, which parity tests replace by injected draws)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox(uint4 c, uint2 k)
{
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ double u01(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}
// stream: (game, ply, element, purpose)
__device__ double rng_gamma(uint64_t seed, uint64_t game, uint32_t ply, uint32_t elem, double a)
{
    uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    double boost = 1.0;
    uint32_t ctr = 0;
    if (a < 1.0) {
        uint4 r = philox(make_uint4((uint32_t)game, (uint32_t)(game >> 32) ^ (ply << 8), elem, 0x40000000u), key);
        double u = u01(r.x, r.y);
        if (u < 1e-300) u = 1e-300;
        boost = pow(u, 1.0 / a);
        a += 1.0;
    }
    double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;; ctr++) {
        uint4 r = philox(make_uint4((uint32_t)game, (uint32_t)(game >> 32) ^ (ply << 8), elem, 0x80000000u + ctr), key);
        double u1 = u01(r.x, r.y), u2 = u01(r.z, r.w);
        if (u1 < 1e-300) u1 = 1e-300;
        double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        uint4 r2 = philox(make_uint4((uint32_t)game, (uint32_t)(game >> 32) ^ (ply << 8), elem, 0xC0000000u + ctr), key);
        double u = u01(r2.x, r2.y);
        if (u < 1.0 - 0.0331 * x * x * x * x || (u > 0 && log(u) < 0.5 * x * x + d * (1.0 - v + log(v))))
            return d * v * boost;
        if (ctr > 64) return d * boost; // unreachable in practice; bounded loop
    }
}

// ------------------------------------------------------------------------------------
// root prior renormalisation + Dirichlet mix, mcts.py:211-226 (wave-cooperative)
// ------------------------------------------------------------------------------------
__device__ void root_prep(const Geo &g, const SearchCfg &cfg, const TreeBufs &B, int slot, Slot *S,
                          uint32_t *pool, float *ldsf, double *ldsd, int lane)
{
    const int root = S->root;
    NodeMeta rm = load_meta(pool, g, root);
    const float *Prow = reinterpret_cast<const float *>(node_ptr(pool, g, root) + META_DW);
    double *rp = B.root_prior + (size_t)slot * g.AS;
    const int A = g.A;
    int prepped = S->root_prepped, isf64 = S->root_prior_f64;
    bool probs_f64;
    __syncthreads();
    if (!prepped || !isf64) {
        for (int i = lane; i < A; i += WAVE)
            ldsf[i] = prepped ? (float)rp[i] : Prow[i];
        __syncthreads();
        float cpsum = np_pairwise_sum<float>(ldsf, A, lane);
        probs_f64 = !(cpsum != 0.0f);
        __syncthreads();
        for (int i = lane; i < A; i += WAVE) {
            if (!probs_f64) ldsf[i] = ldsf[i] / cpsum; else ldsd[i] = 0.0;
        }
    } else {
        for (int i = lane; i < A; i += WAVE)
            ldsd[i] = rp[i];
        __syncthreads();
        double cpsum = np_pairwise_sum<double>(ldsd, A, lane);
        probs_f64 = true;
        __syncthreads();
        for (int i = lane; i < A; i += WAVE)
            ldsd[i] = (cpsum != 0.0) ? ldsd[i] / cpsum : 0.0;
    }
    __syncthreads();
    const double onemc = 1.0 - cfg.coeff;
    if (cfg.alpha > 0) {
        const bool ext = B.noise_valid[slot] != 0;
        const double *nin = B.noise_in + (size_t)slot * g.AS;
        double gsum = 0.0;
        double gam[4] = {0, 0, 0, 0};
        if (!ext) {
            // np.random.dirichlet over ALL A slots (mcts.py:220-222), own Philox stream
            int k = 0;
            for (int i = lane; i < A; i += WAVE, k++) {
                gam[k] = rng_gamma(cfg.seed, (uint64_t)S->game_idx, (uint32_t)S->move_idx, (uint32_t)i, cfg.alpha);
                gsum += gam[k];
            }
            for (int o = 32; o > 0; o >>= 1)
                gsum += __shfl_xor(gsum, o);
        }
        int k = 0;
        for (int i = lane; i < A; i += WAVE, k++) {
            double nz = ext ? nin[i] : gam[k] / gsum;
            nz = nz * (gs_valid(g, rm.st, i) ? 1.0 : 0.0);
            double a = probs_f64 ? onemc * ldsd[i] : (double)((float)onemc * ldsf[i]);
            rp[i] = a + cfg.coeff * nz;
        }
        S->root_prior_f64 = 1;
        if (lane == 0) B.noise_valid[slot] = 0;
    } else {
        const double cn = cfg.coeff * 0.0;
        for (int i = lane; i < A; i += WAVE)
            rp[i] = probs_f64 ? onemc * ldsd[i] + cn : (double)((float)onemc * ldsf[i] + (float)cn);
        S->root_prior_f64 = probs_f64 ? 1 : 0;
    }
    S->root_prepped = 1;
    __syncthreads();
}

// phase and a step stamp leave in ONE 8-byte store.  The driver pass (move choice, re-root, next search) of step t
// runs on a second stream NEXT TO k_select of step t: a slot whose search it starts carries stamp t, and k_select(t)
// -- which may see the slot's old or new phase word -- leaves slots stamped t alone; from step t+1 on (kernel boundary)
// the slot is a searching slot like any other.
__device__ __forceinline__ void set_phase_stamped(Slot *S, int phase, int step)
{
    *reinterpret_cast<volatile unsigned long long *>(&S->phase) = (unsigned long long)(unsigned)phase | ((unsigned long long)(unsigned)step << 32);
}

__device__ __forceinline__ int rule_num_reads(const Geo &g, const SearchCfg &cfg, const GState &st)
{
    // n_searches = min(4*factorial(nb_valid_moves), mcts_num_read), self_play.py:64-65
    int nb = gs_count_valid(g, st);
    double f = 4.0;
    for (int i = 2; i <= nb; i++) {
        f *= (double)i;
        if (f > (double)cfg.mcts_num_read) break;
    }
    return f < (double)cfg.mcts_num_read ? (int)f : cfg.mcts_num_read;
}

// UCT_search prologue (mcts.py:205-226) for one slot; num_reads < 0 = driver rule
__device__ void begin_search(const Geo &g, const SearchCfg &cfg, const TreeBufs &B, int slot, Slot *S,
                             uint32_t *pool, int num_reads, float *ldsf, double *ldsd, int lane)
{
    NodeMeta rm = load_meta(pool, g, S->root);
    if (rm.flags & NF_TERMINAL) { // play_game never searches a terminal root
        S->sims_left = 0;
        set_phase_stamped(S, PH_IDLE, cfg.step);
        return;
    }
    if (num_reads < 0) {
        num_reads = rule_num_reads(g, cfg, rm.st);
        // benchmark population (dbaz_selfplay_stagger): the slot's first search is cut short so that the slots'
        // move boundaries are spread over a whole search instead of all falling into the same step
        // (dbaz_selfplay_quickplay: the opening plies of the slot's first game are searched with a small budget -- a cheap way
        // to a population of positions that search-based play reaches; the staggered budget applies to the first full search)
        const int qu = S->quick_until;
        if (qu > 0 && S->move_idx < qu) {
            num_reads = min(num_reads, max(cfg.quick_reads, 1));
        } else {
            const int ffr = S->ff_reads;
            if (ffr > 0) {
                num_reads = min(num_reads, ffr);
                S->ff_reads = 0;
            }
        }
    }
    S->sims_left = num_reads;
    S->first_wave = 1;
    S->wave_sims = 0;
    if (rm.flags & NF_EXPANDED) {
        root_prep(g, cfg, B, slot, S, pool, ldsf, ldsd, lane);
        set_phase_stamped(S, num_reads > 0 ? PH_SIMS : PH_READY, cfg.step);
    } else {
        set_phase_stamped(S, PH_EXPAND_ROOT, cfg.step);
    }
}

__global__ void __launch_bounds__(WAVE) k_search_begin(Geo g, SearchCfg cfg, TreeBufs B, const int32_t *num_reads)
{
    __shared__ float ldsf[DBAZ_MAX_A];
    __shared__ double ldsd[DBAZ_MAX_A];
    int slot = blockIdx.x, lane = threadIdx.x;
    Slot *S = B.slots + slot;
    if (S->phase == PH_ERROR || S->game_idx < 0)
        return;
    uint32_t *pool = B.nodes + (size_t)slot * g.cap * g.node_dw;
    begin_search(g, cfg, B, slot, S, pool, num_reads ? num_reads[slot] : -1, ldsf, ldsd, lane);
}

// ------------------------------------------------------------------------------------
// select_leaf, mcts.py:105-114
// ------------------------------------------------------------------------------------
__device__ __forceinline__ bool cand_beats(double xa, int ia, bool ha, double xb, int ib, bool hb)
{
    // numpy argmax order: first maximum; the first NaN beats everything
    if (!ha) return false;
    if (!hb) return true;
    bool na = xa != xa, nb = xb != xb;
    if (na || nb) {
        if (na && nb) return ia < ib;
        return na;
    }
    return xa > xb || (xa == xb && ia < ib);
}

// The four rows of a node (this lane's children i = lane + 64 j) and its meta block are fetched
// together -- the descent then costs ONE dependent memory round trip per tree level (the next
// node's rows are requested the moment the child index is known, which itself comes out of the
// prefetched C row by shuffle) instead of four (rows in two dependent strides, C[best], meta).
#ifdef DBAZ_STAMP
// diagnostic build only (tools/stamp_select.sh): cycle stamps of a k_select wave, printed for a few slots
#define TSTAMP(var) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#else
#define TSTAMP(var) do { } while (0)
#endif

// pb_c's N-dependent factor  log((N + base + 1) / base) + cpuct  and  sqrt(N)  (mcts.py:92-94) from the
// host-libm tables.  N is wave-uniform: the loads are unconditional scalar loads (clamped index) so that
// they can be in flight together with the node's rows; beyond the table the device computes the terms
// (never reached with the default table sizing).
__device__ __forceinline__ void select_tab(const SearchCfg &cfg, const TreeBufs &B, int N, double &pbc, double &sq)
{
    const int i = min(N, cfg.table_n - 1);
    pbc = B.pbc_table[i];
    sq = B.sqrt_table[i];
}
__device__ __forceinline__ void select_tab_fix(const SearchCfg &cfg, int N, double &pbc, double &sq)
{
    if (N >= cfg.table_n) {
        pbc = log(((double)N + cfg.cpuct_base + 1.0) / cfg.cpuct_base) + cfg.cpuct;
        sq = sqrt((double)N);
    }
}

template <int NPL>
struct NodeRows {
    float P[NPL], W[NPL];
    uint32_t NS[NPL];
    int32_t C[NPL];
};

template <int NPL>
__device__ __forceinline__ void load_rows(NodeRows<NPL> &r, uint32_t *pool, const Geo &g, int idx, int lane)
{
    const uint32_t *nd = node_ptr(pool, g, idx) + META_DW;
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const int i = lane + WAVE * j;
        const bool ok = i < g.A;
        r.P[j] = ok ? reinterpret_cast<const float *>(nd)[i] : 0.0f;
        r.W[j] = ok ? reinterpret_cast<const float *>(nd + g.AS)[i] : 0.0f;
        r.NS[j] = ok ? nd[2 * g.AS + i] : 0u;
        r.C[j] = ok ? reinterpret_cast<const int32_t *>(nd + 3 * g.AS)[i] : -1;
    }
}

#ifndef SELECT_WAVES
#define SELECT_WAVES 16 // games per workgroup (one wave each)
#endif
// returns the model (0 / 1) whose network must evaluate this game's leaf, or -1
template <int NPL>
__device__ __forceinline__ int select_one(const Geo &g, const SearchCfg &cfg, const TreeBufs &B, int slot, int lane)
{
    Slot *S = B.slots + slot;
    const unsigned long long pw = *reinterpret_cast<volatile const unsigned long long *>(&S->phase); // phase | stamp << 32
    const int phase = (int)(unsigned)pw;
    if (phase != PH_EXPAND_ROOT && phase != PH_SIMS)
        return -1;
    if ((int)(unsigned)(pw >> 32) == cfg.step && cfg.driver_concurrent) // search started by this step's driver pass (see set_phase_stamped)
        return -1;
    if (cfg.eval_round > 0 && S->pending) {
        // the leaf of an earlier step whose evaluation was put off (full rounds only): path, leaf and features are still in place.
        // It asks for its place in the list at once, i.e. usually ahead of the leaves that are still being selected (a workgroup's
        // batched append below happens after the slowest of its 16 descents).  That is an ordering tendency, not a guarantee: the
        // append is an atomicAdd racing with other workgroups', so a leaf CAN land behind the cut again and wait one more step.
        // Nothing depends on it -- a put-off leaf only completes its simulation later; the game plays the same moves.
        if (lane == 0) {
            S->sel_step = cfg.step;
            const int pos = atomicAdd(B.n_eval, 1);
            B.eval_list[pos] = slot;
            S->eval_pos = pos;
        }
        return -1;
    }
    uint32_t *pool = B.nodes + (size_t)slot * g.cap * g.node_dw;
    PathEnt *path = B.path + (size_t)slot * g.dmax;
    const double *rprior = B.root_prior + (size_t)slot * g.AS;
    const int A = g.A;

    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts_mem = 0, ts_tab = 0, ts_ucb = 0, ts_arg = 0, tq0 = 0, tq1 = 0;
    (void)ts0; (void)ts1; (void)ts2; (void)ts3; (void)ts_mem; (void)ts_tab; (void)ts_ucb; (void)ts_arg; (void)tq0; (void)tq1;
    TSTAMP(ts0);
    const int root = S->root;
    int cur = root, depth = 0, in_move = -1;
    PoolState q = pool_load(S);
    GcBatch<GC_PER_STEP> gcb; // the collector's share of this simulation: rows requested now, consumed after the descent
    gcb.k = 0;
    if (cfg.gc_lazy <= 0 || q.n_free + (g.cap - q.n_nodes) < cfg.gc_lazy) gc_issue<GC_PER_STEP>(gcb, g, B, slot, pool, q, lane);
    int Nself = S->root_N;
    NodeMeta m = load_meta(pool, g, root);
    NodeRows<NPL> R;
    load_rows<NPL>(R, pool, g, root, lane);
    double pbc_cur, sq_cur;
    select_tab(cfg, B, Nself, pbc_cur, sq_cur);
    select_tab_fix(cfg, Nself, pbc_cur, sq_cur);
    double rp[NPL]; // float64 root priors (root_prep), used for the root level only
#pragma unroll
    for (int j = 0; j < NPL; j++) rp[j] = (lane + WAVE * j < A) ? rprior[lane + WAVE * j] : 0.0;
    in_move = m.move;
    // match play (self_play.py:59,237-239): the player to move at the ROOT selects the model for the
    // whole search of this move; odd games swap the seats (the reference shuffles them by worker pid)
    const int model = cfg.match_play ? ((m.st.to_play ^ (int)(S->game_idx & 1)) & 1) : 0;
    if ((m.flags & NF_EXPANDED) && !(m.flags & NF_TERMINAL)) {
        if (lane == 0)
            S->root_W = S->root_W - 1.0f; // current.total_value -= VIRTUAL_LOSS (root slot)
    }
    int err = 0, need_eval = -1;
    TSTAMP(ts1);
    while ((m.flags & NF_EXPANDED) && !(m.flags & NF_TERMINAL)) {
        uint32_t *nd = node_ptr(pool, g, cur);
        float *Wrow = reinterpret_cast<float *>(nd + META_DW + g.AS);
        uint32_t *NSrow = nd + META_DW + 2 * g.AS;
        int32_t *Crow = reinterpret_cast<int32_t *>(nd + META_DW + 3 * g.AS);
        // children_ucb_score, mcts.py:91-99 (float64 throughout); the N-dependent terms were requested
        // together with this node's rows (select_tab)
        TSTAMP(tq0);
        const double pbc0 = pbc_cur, sq = sq_cur;
#ifdef DBAZ_STAMP
        asm volatile("" ::"v"(pbc0), "v"(sq));
        TSTAMP(tq1); ts_tab += tq1 - tq0;
#endif
        // Branch-free scan of this lane's children (word j of the played / sentinel masks holds child lane + 64 j)
        const uint64_t ew[4] = {m.st.e0, m.st.e1, m.st.e2, m.st.e3};
        double bx = 0.0;
        int bj = 0;
        float bw = 0.0f;
        uint32_t bns = 0;
        int bc = -1;
        bool have = false;
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            const int i = lane + WAVE * j;
            const bool in = i < A;
            const double P = (cur == root) ? rp[j] : (double)R.P[j];
            const float w = R.W[j];
            const uint32_t ns = R.NS[j];
            const int n = (int)(ns & NS_MASK);
            const double sgn = (ns & NS_SAME) ? 1.0 : -1.0;
            const double t = sq / (double)(n + 1);
            const double pb_c = pbc0 * t;
            const double prior_score = pb_c * P;
            double value_score = (double)w / (double)(1 + n);
            value_score = value_score * sgn;
            const double score = prior_score + value_score;
            const bool valid = in && !(((ew[j] | g.sentinel[j]) >> lane) & 1ull);
            const double inval = valid ? 0.0 : 1.0;
            const double x = -1e12 * inval + score; // best_child, mcts.py:101-103
            // numpy argmax order inside the lane: first maximum, a NaN already held wins
            const bool take = in && (!have || (bx == bx && !(x <= bx)));
            bx = take ? x : bx;
            bj = take ? j : bj;
            bw = take ? w : bw;
            bns = take ? ns : bns;
            bc = take ? R.C[j] : bc;
            have = have || take;
        }
#ifdef DBAZ_STAMP
        asm volatile("" ::"v"(bx), "v"(bj));
        TSTAMP(tq0); ts_ucb += tq0 - tq1;
#endif
        int bi;
        if (__ballot(have && bx != bx) == 0ull) {
            // no NaN anywhere (always, in practice): wave maximum, then the lowest child index that attains it
            double mx = have ? bx : -INFINITY;
            for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
            bi = -1;
#pragma unroll
            for (int j = 0; j < NPL; j++) {
                const unsigned long long mk = __ballot(have && bx == mx && bj == j);
                if (bi < 0 && mk != 0ull) bi = WAVE * j + (__ffsll((long long)mk) - 1);
            }
        } else {
            // numpy's argmax with NaNs (the first NaN wins): generic butterfly
            bi = have ? lane + WAVE * bj : 0x7fffffff;
            double rx = bx;
            bool rh = have;
            for (int o = 32; o > 0; o >>= 1) {
                double ox = __shfl_xor(rx, o);
                int oi = __shfl_xor(bi, o);
                int oh = __shfl_xor((int)rh, o);
                if (cand_beats(ox, oi, oh != 0, rx, bi, rh)) { rx = ox; bi = oi; rh = true; }
            }
        }
#ifdef DBAZ_STAMP
        asm volatile("" ::"v"(bi));
        TSTAMP(tq1); ts_arg += tq1 - tq0;
#endif
        // the lane that owns child bi holds it as ITS running best (lanes scan their children in index order and
        // the wave-wide winner is some lane's own best), so its bw / bns / bc are the winner's W, N|sign and C
        // bi and everything read from its owner lane are wave-uniform: say so (scalar branches and loads below)
        bi = __builtin_amdgcn_readfirstlane(bi);
        const int owner = bi & (WAVE - 1);
        bw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bw), owner));
        bns = (uint32_t)__builtin_amdgcn_readlane((int)bns, owner);
        int child = __builtin_amdgcn_readlane(bc, owner);
        if (lane == 0) {
            PathEnt pe;
            pe.node = cur; pe.move_in = (int16_t)in_move; pe.to_play = (int16_t)m.st.to_play;
            path[depth] = pe;
        }
        depth++;
        if (child < 0) {
            // DictWithDefault.__missing__ -> UCTNode(game_state.play(move)), mcts.py:53-54
            child = pool_alloc(g, B, slot, pool, q, lane);
            if (child < 0) { err = DBAZ_EPOOL; depth--; break; }
            GState st = m.st;
            int r = gs_play(g, st, bi, nullptr);
            if (r < 0) { err = DBAZ_EILLEGAL; depth--; break; }
            bool same = (st.to_play == st.just_played);
            if (lane == 0) {
                Crow[bi] = child;
                NSrow[bi] = same ? NS_SAME : 0u; // child_player_changed slot (mcts.py:119), visits = 0
            }
            init_node(pool, g, child, st, cur, bi, m.deepness + 1, lane);
            NodeMeta cm;
            cm.st = st; cm.parent = cur; cm.move = bi;
            cm.result = gs_result(st);
            cm.flags = (cm.result != DBAZ_RESULT_NONE) ? NF_TERMINAL : 0;
            cm.deepness = m.deepness + 1;
            cur = child; in_move = bi; m = cm;
            break;
        }
#ifdef DBAZ_STAMP
        unsigned long long tl0, tl1;
        TSTAMP(tl0); // includes the drain of this level's stores (path entry)
#endif
        // everything the next level needs is requested at once: the child's pb_c / sqrt terms (its visit count is
        // already known from this node's N row), its meta block and its four rows
        const int nchild = (int)(bns & NS_MASK);
        double pbc_nx, sq_nx;
        select_tab(cfg, B, nchild, pbc_nx, sq_nx);
        NodeMeta cm = load_meta(pool, g, child);
        load_rows<NPL>(R, pool, g, child, lane); // rows of an unexpanded node are ignored
        select_tab_fix(cfg, nchild, pbc_nx, sq_nx);
#ifdef DBAZ_STAMP
        TSTAMP(tl1);
        ts_mem += tl1 - tl0;
#endif
        if ((cm.flags & NF_EXPANDED) && !(cm.flags & NF_TERMINAL)) {
            if (lane == owner)
                Wrow[bi] = bw - 1.0f; // VIRTUAL_LOSS on the node being left next iteration
            Nself = nchild;
            pbc_cur = pbc_nx;
            sq_cur = sq_nx;
        }
        cur = child; in_move = bi; m = cm;
    }
    TSTAMP(ts2);
    if (err) {
        if (lane == 0) { S->error = err; S->phase = PH_ERROR; }
        return -1;
    }
    gc_finish<GC_PER_STEP>(gcb, g, B, slot, q, lane);
    if (lane == 0) {
        PathEnt pe;
        pe.node = cur; pe.move_in = (int16_t)in_move; pe.to_play = (int16_t)m.st.to_play;
        path[depth] = pe;
        S->leaf = cur;
        S->sel_step = cfg.step;
        S->path_len = depth + 1;
        S->leaf_terminal = (m.flags & NF_TERMINAL) ? 1 : 0;
        S->leaf_result = m.result;
        S->leaf_to_play = m.st.to_play;
        pool_store(q, S);
        S->model = model;
        if (q.n_nodes > S->pool_high) S->pool_high = q.n_nodes;
    }
    if (!(m.flags & NF_TERMINAL)) {
        // get_features of the leaf as float32 planes (nn_batch_builder + nn.py:157)
        float *f = B.feat + (size_t)slot * 3 * g.HW;
        for (int i = lane; i < 3 * g.HW; i += WAVE)
            f[i] = (float)gs_feature(g, m.st, i);
        const int ev = model ? cfg.evaluator2 : cfg.evaluator;
        if (ev == DBAZ_EVAL_RESNET || ev == DBAZ_EVAL_SIMPLENN) need_eval = model;
        int hit = -1;
        // (the table exists for network evaluators, and for the formula evaluators when forced on -- transposition_cache
        // = 2 -- so that the hit path can be compared with the oracle bit for bit)
        if (B.tt && ev != DBAZ_EVAL_EXTERNAL) {
            // lanes 0..TT_PROBES-1 read the probe window; a candidate is verified against the live tree
            const uint64_t h = formula_hash(m.st);
            const unsigned long long *tt = B.tt + (size_t)slot * ((size_t)B.tt_mask + 1);
            unsigned long long *ttw = B.tt + (size_t)slot * ((size_t)B.tt_mask + 1);
            unsigned long long ent = 0;
            if (lane < TT_PROBES) ent = tt[((unsigned)h + (unsigned)lane) & (unsigned)B.tt_mask];
            unsigned long long cand = __ballot(lane < TT_PROBES && ent != 0ull && (ent >> 40) == (h >> 40));
            const int mb = tt_mover_b2c(m.st);
            while (cand && hit < 0) {
                const int pl = __ffsll((long long)cand) - 1;
                cand &= cand - 1;
                const int idx = (int)(unsigned)__shfl(ent, pl);
                if (idx >= 0 && idx < q.n_nodes && idx != cur) {
                    const NodeMeta tm = load_meta(pool, g, idx);
                    if ((tm.flags & NF_EXPANDED) && !(tm.flags & NF_TERMINAL) && tm.st.e0 == m.st.e0 && tm.st.e1 == m.st.e1 &&
                        tm.st.e2 == m.st.e2 && tm.st.e3 == m.st.e3 && tt_mover_b2c(tm.st) == mb) {
                        hit = idx;
                        // still in use: stamp the entry with the current epoch (old epochs are the preferred victims)
                        if (lane == 0)
                            ttw[((unsigned)h + (unsigned)pl) & (unsigned)B.tt_mask] = tt_entry(h, tt_epoch_byte(S->tt_epoch), idx);
                    }
                }
            }
            if (hit >= 0) need_eval = -1;
        }
        if (lane == 0) S->leaf_hit = hit;
    }
#ifdef DBAZ_STAMP
    TSTAMP(ts3);
    if (lane == 0 && (slot & 511) == 0)
        printf("SEL slot %d depth %d root %llu descent %llu leaf %llu mem %llu tab %llu ucb %llu arg %llu\n", slot, depth, ts1 - ts0, ts2 - ts1,
               ts3 - ts2, ts_mem, ts_tab, ts_ucb, ts_arg);
#endif
    return need_eval;
}

// One wave per game, SELECT_WAVES games per workgroup.  The leaves that need a network evaluation are
// appended to the per-model lists with ONE global atomic per workgroup and model (positions inside
// the workgroup come from an LDS counter): a per-wave atomicAdd on the single list counter serialises
// 8192 same-address atomics per step and was most of this kernel's duration.
template <int NPL>
__global__ void __launch_bounds__(WAVE * SELECT_WAVES) k_select(Geo g, SearchCfg cfg, TreeBufs B, int n_slots)
{
    __shared__ int s_cnt[2], s_base[2];
    const int slot = blockIdx.x * SELECT_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & (WAVE - 1);
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int model = -1;
    if (slot < n_slots) model = select_one<NPL>(g, cfg, B, slot, lane);
    int my = -1;
    if (model >= 0 && lane == 0) my = atomicAdd(&s_cnt[model], 1);
    __syncthreads();
    if (threadIdx.x < 2 && s_cnt[threadIdx.x] > 0) s_base[threadIdx.x] = atomicAdd(B.n_eval + threadIdx.x, s_cnt[threadIdx.x]);
    __syncthreads();
    if (my >= 0) {
        (model ? B.eval_list2 : B.eval_list)[s_base[model] + my] = slot;
        B.slots[slot].eval_pos = s_base[model] + my;
    }
}

// Full rounds only (SearchCfg.eval_round > 0).  A network launch costs whole rounds of workgroups (nn.hip): 5 306 leaves are four
// rounds of 1 280 and a remainder round that costs 40 % of a round for 3.5 % of the leaves.  When at most eval_defer_max leaves
// would be left behind the last full round, the network kernels take only the full rounds (nn.hip cut_n: every kernel of the step
// applies the same rule to the same count; k_head_fc leaves the count in n_eval[2]); the slots behind the cut keep their selected
// leaf -- k_expand_backup sets Slot::pending -- and ask again in the next step, at the head of the list (select_one).  Every game
// plays the same moves: only WHEN a slot's simulation completes changes.

// ------------------------------------------------------------------------------------
// UCT_search with K > 1 pending evaluations on ONE tree (mcts.py:228-239; players.AZPlayer).
// A wave of the search = up to K simulations selected ONE AFTER THE OTHER by the tree's wavefront, each leaving a
// virtual loss on its leaf path, then ONE batched evaluation of their leaves, then their expand + backup in the same
// order.  Virtual loss: exactly the reference's `total_value -= VIRTUAL_LOSS` on every node a simulation leaves
// (mcts.py:108-109; restored by backup's `+ VIRTUAL_LOSS`), plus what the reference lacks -- the visit is counted on
// every edge of the path, the leaf's included, at SELECT time instead of at backup time, so that the simulations of a
// wave spread over the tree instead of piling onto one leaf (with K = 64 the reference's in-flight leaves are
// 98 % duplicates).  A leaf that is selected again while its evaluation is pending is not evaluated twice (NF_INFLIGHT).
// With K = 1 every number equals the sequential kernels' (and the oracle's) bit for bit: a simulation never reads a
// count it has itself incremented.
// ------------------------------------------------------------------------------------
template <int NPL>
__device__ __forceinline__ void select_multi_one(const Geo &g, const SearchCfg &cfg, const TreeBufs &B, int slot, int lane)
{
    Slot *S = B.slots + slot;
    const int phase = S->phase;
    if (phase != PH_EXPAND_ROOT && phase != PH_SIMS) return;
    uint32_t *pool = B.nodes + (size_t)slot * g.cap * g.node_dw;
    const double *rprior = B.root_prior + (size_t)slot * g.AS;
    const int A = g.A, K = B.kmax;
    int width = min(max(cfg.pending, 1), K);
    if (S->first_wave) width = min(width, A);                 // max_pend = min(max_pending_evals, len(valid_moves))
    int n_sims = phase == PH_EXPAND_ROOT ? 1 : min(width, S->sims_left);
    PoolState q = pool_load(S);
    const int root = S->root;
    int err = 0, done = 0;
    for (int k = 0; k < n_sims && !err; k++) {
        PathEnt *path = B.path_m + ((size_t)slot * K + k) * g.dmax;
        pool_collect<GC_PER_STEP>(g, B, slot, pool, q, lane);
        int cur = root, depth = 0, in_move;
        int Nself = S->root_N;
        NodeMeta m = load_meta(pool, g, root);
        NodeRows<NPL> R;
        load_rows<NPL>(R, pool, g, root, lane);
        double pbc_cur, sq_cur;
        select_tab(cfg, B, Nself, pbc_cur, sq_cur);
        select_tab_fix(cfg, Nself, pbc_cur, sq_cur);
        double rp[NPL];
#pragma unroll
        for (int j = 0; j < NPL; j++) rp[j] = (lane + WAVE * j < A) ? rprior[lane + WAVE * j] : 0.0;
        in_move = m.move;
        const unsigned vv = cfg.virtual_visits ? 1u : 0u;
        if (lane == 0) {
            S->root_N = Nself + (int)vv;                        // virtual_visits: the visit is counted now (backup adds the value only)
            if ((m.flags & NF_EXPANDED) && !(m.flags & NF_TERMINAL)) S->root_W = S->root_W - 1.0f;
        }
        while ((m.flags & NF_EXPANDED) && !(m.flags & NF_TERMINAL)) {
            uint32_t *nd = node_ptr(pool, g, cur);
            float *Wrow = reinterpret_cast<float *>(nd + META_DW + g.AS);
            uint32_t *NSrow = nd + META_DW + 2 * g.AS;
            int32_t *Crow = reinterpret_cast<int32_t *>(nd + META_DW + 3 * g.AS);
            const double pbc0 = pbc_cur, sq = sq_cur;
            const uint64_t ew[4] = {m.st.e0, m.st.e1, m.st.e2, m.st.e3};
            double bx = 0.0;
            int bj = 0;
            float bw = 0.0f;
            uint32_t bns = 0;
            int bc = -1;
            bool have = false;
#pragma unroll
            for (int j = 0; j < NPL; j++) {
                const int i = lane + WAVE * j;
                const bool in = i < A;
                const double P = (cur == root) ? rp[j] : (double)R.P[j];
                const float w = R.W[j];
                const uint32_t ns = R.NS[j];
                const int n = (int)(ns & NS_MASK);
                const double sgn = (ns & NS_SAME) ? 1.0 : -1.0;
                const double t = sq / (double)(n + 1);
                const double pb_c = pbc0 * t;
                const double prior_score = pb_c * P;
                double value_score = (double)w / (double)(1 + n);
                value_score = value_score * sgn;
                const double score = prior_score + value_score;
                const bool valid = in && !(((ew[j] | g.sentinel[j]) >> lane) & 1ull);
                const double inval = valid ? 0.0 : 1.0;
                const double x = -1e12 * inval + score;
                const bool take = in && (!have || (bx == bx && !(x <= bx)));
                bx = take ? x : bx;
                bj = take ? j : bj;
                bw = take ? w : bw;
                bns = take ? ns : bns;
                bc = take ? R.C[j] : bc;
                have = have || take;
            }
            int bi = have ? lane + WAVE * bj : 0x7fffffff;
            {   // numpy argmax order (first maximum; the first NaN wins): generic butterfly
                double rx = bx;
                bool rh = have;
                for (int o = 32; o > 0; o >>= 1) {
                    double ox = __shfl_xor(rx, o);
                    int oi = __shfl_xor(bi, o);
                    int oh = __shfl_xor((int)rh, o);
                    if (cand_beats(ox, oi, oh != 0, rx, bi, rh)) { rx = ox; bi = oi; rh = true; }
                }
            }
            bi = __builtin_amdgcn_readfirstlane(bi);
            const int owner = bi & (WAVE - 1);
            bw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bw), owner));
            bns = (uint32_t)__builtin_amdgcn_readlane((int)bns, owner);
            int child = __builtin_amdgcn_readlane(bc, owner);
            if (lane == 0) {
                PathEnt pe;
                pe.node = cur; pe.move_in = (int16_t)in_move; pe.to_play = (int16_t)m.st.to_play;
                path[depth] = pe;
            }
            depth++;
            if (child < 0) {
                child = pool_alloc(g, B, slot, pool, q, lane);
                if (child < 0) { err = DBAZ_EPOOL; depth--; break; }
                GState st = m.st;
                int r = gs_play(g, st, bi, nullptr);
                if (r < 0) { err = DBAZ_EILLEGAL; depth--; break; }
                bool same = (st.to_play == st.just_played);
                if (lane == 0) {
                    Crow[bi] = child;
                    NSrow[bi] = (same ? NS_SAME : 0u) | vv;   // child_player_changed slot; the pending visit
                }
                init_node(pool, g, child, st, cur, bi, m.deepness + 1, lane);
                NodeMeta cm;
                cm.st = st; cm.parent = cur; cm.move = bi;
                cm.result = gs_result(st);
                cm.flags = (cm.result != DBAZ_RESULT_NONE) ? NF_TERMINAL : 0;
                cm.deepness = m.deepness + 1;
                cur = child; in_move = bi; m = cm;
                break;
            }
            const int nchild = (int)(bns & NS_MASK);
            double pbc_nx, sq_nx;
            select_tab(cfg, B, nchild, pbc_nx, sq_nx);
            NodeMeta cm = load_meta(pool, g, child);
            load_rows<NPL>(R, pool, g, child, lane);
            select_tab_fix(cfg, nchild, pbc_nx, sq_nx);
            if (lane == owner) {
                if (vv) NSrow[bi] = bns + 1u;                   // the visit of the edge into `child`, counted now
                if ((cm.flags & NF_EXPANDED) && !(cm.flags & NF_TERMINAL)) Wrow[bi] = bw - 1.0f; // VIRTUAL_LOSS on the node left next
            }
            if ((cm.flags & NF_EXPANDED) && !(cm.flags & NF_TERMINAL)) {
                Nself = nchild;
                pbc_cur = pbc_nx;
                sq_cur = sq_nx;
            }
            cur = child; in_move = bi; m = cm;
        }
        if (err) break;
        // leaf bookkeeping of simulation k
        const bool term = (m.flags & NF_TERMINAL) != 0;
        const bool dup = !term && (m.flags & NF_INFLIGHT);
        if (lane == 0) {
            PathEnt pe;
            pe.node = cur; pe.move_in = (int16_t)in_move; pe.to_play = (int16_t)m.st.to_play;
            path[depth] = pe;
            SimRec sr;
            sr.leaf = cur; sr.path_len = depth + 1; sr.terminal = term ? 1 : 0; sr.result = m.result; sr.to_play = m.st.to_play;
            sr.dup = dup ? 1 : 0;
            B.simrec[(size_t)slot * K + k] = sr;
            if (!term && !dup)
                node_ptr(pool, g, cur)[11] = pack_dw11(m.flags | NF_INFLIGHT, m.result, m.deepness);
        }
        if (!term && !dup) {
            float *f = B.feat_m + ((size_t)slot * K + k) * 3 * g.HW;
            for (int i = lane; i < 3 * g.HW; i += WAVE) f[i] = (float)gs_feature(g, m.st, i);
        }
        done = k + 1;
        // the next simulation of this wave reads what this one wrote (same wavefront: program order + a drain)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if (lane == 0) {
        if (err) { S->error = err; S->phase = PH_ERROR; }
        S->wave_sims = done;
        S->sel_step = cfg.step;
        if (phase == PH_SIMS) S->first_wave = 0; // (the root expansion is not one of the waves)
        pool_store(q, S);
        if (q.n_nodes > S->pool_high) S->pool_high = q.n_nodes;
    }
}

template <int NPL>
__global__ void __launch_bounds__(WAVE) k_select_multi(Geo g, SearchCfg cfg, TreeBufs B, int n_slots)
{
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= n_slots) return;
    select_multi_one<NPL>(g, cfg, B, slot, lane);
    __syncthreads();
    // evaluation list: the wave's simulations that need the network, in simulation order
    const Slot *S = B.slots + slot;
    if (S->phase == PH_ERROR) return;
    const int K = B.kmax, n = S->wave_sims;
    const bool nn_ev = cfg.evaluator == DBAZ_EVAL_RESNET || cfg.evaluator == DBAZ_EVAL_SIMPLENN;
    if (!nn_ev || S->sel_step != cfg.step) return;
    for (int k0 = 0; k0 < n; k0 += WAVE) {
        const int k = k0 + lane;
        bool need = false;
        if (k < n) {
            const SimRec sr = B.simrec[(size_t)slot * K + k];
            need = !sr.terminal && !sr.dup;
        }
        const unsigned long long mk = __ballot(need);
        int base = 0;
        if (lane == 0 && mk) base = atomicAdd(B.n_eval, (int)__popcll(mk));
        base = __shfl(base, 0);
        if (need) B.list_m[base + (int)__popcll(mk & ((1ull << lane) - 1ull))] = slot * K + k;
    }
}

__global__ void __launch_bounds__(WAVE) k_expand_backup_multi(Geo g, SearchCfg cfg, TreeBufs B)
{
    __shared__ float ldsf[DBAZ_MAX_A];
    __shared__ double ldsd[DBAZ_MAX_A];
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (slot == 0 && lane == 0) { B.n_eval[0] = 0; B.n_eval[1] = 0; B.drv_count[0] = 0; } // (drv_count: self-play in waves)
    Slot *S = B.slots + slot;
    const int phase = S->phase;
    if (phase != PH_EXPAND_ROOT && phase != PH_SIMS) return;
    if (S->sel_step != cfg.step) return;
    uint32_t *pool = B.nodes + (size_t)slot * g.cap * g.node_dw;
    const int A = g.A, K = B.kmax, n = S->wave_sims;
    const int ev = cfg.evaluator;
    const bool formula = ev == DBAZ_EVAL_FORMULA_HASH || ev == DBAZ_EVAL_FORMULA_UNIFORM;
    int sims_left = S->sims_left;
    for (int k = 0; k < n; k++) {
        const SimRec sr = B.simrec[(size_t)slot * K + k];
        const PathEnt *path = B.path_m + ((size_t)slot * K + k) * g.dmax;
        NodeMeta lm = load_meta(pool, g, sr.leaf);
        uint32_t *nd = node_ptr(pool, g, sr.leaf);
        float v;
        if (sr.terminal) {
            v = (float)lm.result;
        } else if (lm.flags & NF_EXPANDED) {
            v = __uint_as_float(nd[12]); // expanded by an earlier simulation of this wave (sr.dup): same (p, v)
        } else {
            float *Prow = reinterpret_cast<float *>(nd + META_DW);
            uint64_t h = 0;
            if (formula) h = formula_hash(lm.st);
            const float *ep = B.evalP_m + ((size_t)slot * K + k) * g.AS;
            __syncthreads();
            for (int i = lane; i < A; i += WAVE) {
                float p = formula ? formula_p(h, i, ev) : ep[i];
                ldsf[i] = p * (gs_valid(g, lm.st, i) ? 1.0f : 0.0f);
            }
            __syncthreads();
            float s = np_pairwise_sum<float>(ldsf, A, lane);
            const bool renorm = (s > 0.0f) && (s != 1.0f);
            for (int i = lane; i < A; i += WAVE) Prow[i] = renorm ? ldsf[i] / s : ldsf[i];
            v = formula ? formula_v(h, ev) : B.evalV_m[(size_t)slot * K + k];
            if (lane == 0) nd[12] = __float_as_uint(v);
        }
        if (lane == 0) nd[11] = pack_dw11((lm.flags | NF_EXPANDED) & ~NF_INFLIGHT, lm.result, lm.deepness);
        // backup: W += v_n + VIRTUAL_LOSS on every path node; N += 1 here (the reference) or already at selection (virtual_visits)
        const int tp = lm.st.to_play;
        for (int d = lane; d < sr.path_len; d += WAVE) {
            PathEnt pe = path[d];
            float vn = (pe.to_play == tp) ? v : -v;
            float add = vn + 1.0f;
            if (d == 0) {
                S->root_W = S->root_W + add;
                if (!cfg.virtual_visits) S->root_N = S->root_N + 1;
            } else {
                uint32_t *pn = node_ptr(pool, g, path[d - 1].node);
                float *Wr = reinterpret_cast<float *>(pn + META_DW + g.AS);
                Wr[pe.move_in] = Wr[pe.move_in] + add;
                if (!cfg.virtual_visits) pn[META_DW + 2 * g.AS + pe.move_in] += 1u;
            }
        }
        if (lane == 0) {
            const int term = sr.terminal;
            S->terminal_count += term;
            if (lm.deepness > S->max_deepness) S->max_deepness = lm.deepness;
            S->n_search += 1;
            S->sum_path += sr.path_len;
            S->n_term += term;
            S->n_eval += (term || sr.dup) ? 0 : 1;
            S->n_hit += sr.dup ? 1 : 0;
        }
        if (phase != PH_EXPAND_ROOT) sims_left--;
        // the next simulation's path shares nodes with this one (same wavefront: program order + a drain)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (phase == PH_EXPAND_ROOT) {
        root_prep(g, cfg, B, slot, S, pool, ldsf, ldsd, lane);
        if (lane == 0) set_phase_stamped(S, S->sims_left > 0 ? PH_SIMS : PH_READY, cfg.step);
    } else if (lane == 0) {
        S->sims_left = sims_left;
        if (sims_left <= 0) set_phase_stamped(S, PH_READY, cfg.step);
    }
    if (lane == 0) S->wave_sims = 0;
}

// ------------------------------------------------------------------------------------
// _search tail: prior masking (mcts.py:189-196), expand (:116-119), backup (:121-132)
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE) k_expand_backup(Geo g, SearchCfg cfg, TreeBufs B)
{
    __shared__ float ldsf[DBAZ_MAX_A];
    __shared__ double ldsd[DBAZ_MAX_A];
    const int slot = blockIdx.x, lane = threadIdx.x;
    // the step's last kernel resets the per-step counters for the next one (the evaluation lists were consumed by the network
    // launches before it, the driver list by the pass joined before them): no memset launches between the kernels
    if (slot == 0 && lane == 0) { B.n_eval[0] = 0; B.n_eval[1] = 0; B.drv_count[0] = 0; }
    Slot *S = B.slots + slot;
    const int phase = S->phase;
    if (phase != PH_EXPAND_ROOT && phase != PH_SIMS)
        return;
    // only slots for which THIS step's k_select left a leaf: the driver kernel (second stream) may have started a new
    // search in a slot while this step's network was running
    if (S->sel_step != cfg.step)
        return;
    uint32_t *pool = B.nodes + (size_t)slot * g.cap * g.node_dw;
    const PathEnt *path = B.path + (size_t)slot * g.dmax;
    const int A = g.A;
    const int leaf = S->leaf, plen = S->path_len;
    NodeMeta lm = load_meta(pool, g, leaf);
    uint32_t *nd = node_ptr(pool, g, leaf);
    float v;
    int hit = -1;
    if (!(lm.flags & NF_TERMINAL)) {
        float *Prow = reinterpret_cast<float *>(nd + META_DW);
        const int ev = (cfg.match_play && S->model) ? cfg.evaluator2 : cfg.evaluator;
        const bool formula = ev == DBAZ_EVAL_FORMULA_HASH || ev == DBAZ_EVAL_FORMULA_UNIFORM;
        if (B.tt) hit = S->leaf_hit;
        if (cfg.eval_round > 0 && hit < 0 && !formula) {
            // behind this step's cut: the network has not seen the leaf; keep it and ask again next step
            const bool late = S->eval_pos >= B.n_eval[2];
            if (lane == 0) S->pending = late ? 1 : 0;
            if (late) return;
        }
        if (hit >= 0) {
            // transposition: the twin's row is masked(p) / sum for the same valid moves, its v the same network output
            const uint32_t *tw = node_ptr(pool, g, hit);
            const float *Psrc = reinterpret_cast<const float *>(tw + META_DW);
            for (int i = lane; i < A; i += WAVE) Prow[i] = Psrc[i];
            v = __uint_as_float(tw[12]);
        } else {
            uint64_t h = 0;
            if (formula) h = formula_hash(lm.st);
            const float *ep = B.evalP + (size_t)slot * g.AS;
            for (int i = lane; i < A; i += WAVE) {
                float p = formula ? formula_p(h, i, ev) : ep[i];
                ldsf[i] = p * (gs_valid(g, lm.st, i) ? 1.0f : 0.0f); // child_priors * valid
            }
            __syncthreads();
            float s = np_pairwise_sum<float>(ldsf, A, lane);
            const bool renorm = (s > 0.0f) && (s != 1.0f);
            for (int i = lane; i < A; i += WAVE)
                Prow[i] = renorm ? ldsf[i] / s : ldsf[i];
            v = formula ? formula_v(h, ev) : B.evalV[slot];
        }
        if (B.tt && lane == 0) {
            nd[12] = __float_as_uint(v); // a later twin reads the value here
            if (hit < 0) {
                // insert: first empty / stale-epoch / same-tag entry of the probe window, else a hash-chosen victim
                const uint64_t h = formula_hash(lm.st);
                unsigned long long *tt = B.tt + (size_t)slot * ((size_t)B.tt_mask + 1);
                const unsigned ep = tt_epoch_byte(S->tt_epoch);
                int victim = (int)((h >> 20) & (TT_PROBES - 1));
                for (int q = TT_PROBES - 1; q >= 0; q--) {
                    const unsigned long long e0 = tt[((unsigned)h + (unsigned)q) & (unsigned)B.tt_mask];
                    if (e0 == 0ull || ((unsigned)(e0 >> 32) & 0xFFu) != ep || (e0 >> 40) == (h >> 40)) victim = q;
                }
                tt[((unsigned)h + (unsigned)victim) & (unsigned)B.tt_mask] = tt_entry(h, ep, leaf);
            }
        }
    } else {
        v = (float)lm.result; // get_result(), python int
    }
    if (lane == 0)
        nd[11] = pack_dw11(lm.flags | NF_EXPANDED, lm.result, lm.deepness);
    // backup: every path node (root ... leaf) gets W += v_n + VIRTUAL_LOSS, N += 1
    const int tp = lm.st.to_play;
    for (int d = lane; d < plen; d += WAVE) {
        PathEnt pe = path[d];
        float vn = (pe.to_play == tp) ? v : -v;
        float add = vn + 1.0f;
        if (d == 0) {
            S->root_W = S->root_W + add;
            S->root_N = S->root_N + 1;
        } else {
            uint32_t *pn = node_ptr(pool, g, path[d - 1].node);
            float *Wr = reinterpret_cast<float *>(pn + META_DW + g.AS);
            uint32_t *NSr = pn + META_DW + 2 * g.AS;
            Wr[pe.move_in] = Wr[pe.move_in] + add;
            NSr[pe.move_in] = NSr[pe.move_in] + 1u;
        }
    }
    __syncthreads();
    if (lane == 0) {
        const int term = (lm.flags & NF_TERMINAL) ? 1 : 0;
        S->terminal_count += term;
        if (lm.deepness > S->max_deepness) S->max_deepness = lm.deepness;
        S->n_search += 1;
        S->sum_path += plen;
        S->n_term += term;
        S->n_eval += (term || hit >= 0) ? 0 : 1;
        S->n_hit += hit >= 0 ? 1 : 0;
    }
    if (phase == PH_EXPAND_ROOT) {
        root_prep(g, cfg, B, slot, S, pool, ldsf, ldsd, lane);
        if (lane == 0) set_phase_stamped(S, S->sims_left > 0 ? PH_SIMS : PH_READY, cfg.step);
    } else if (lane == 0) {
        int left = S->sims_left - 1;
        S->sims_left = left;
        if (left <= 0) set_phase_stamped(S, PH_READY, cfg.step);
    }
}

// ------------------------------------------------------------------------------------
// init_mcts_tree (mcts.py:163-180): O(1) re-root.  reuse_tree: the chosen child becomes the root where it
// lies (its subtree keeps its node indices, so transposition entries stay valid), the old root -- link to
// the kept child cut -- goes onto the collector's ring.  Otherwise the whole pool is dropped.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void pool_reset(Slot *S)
{
    S->root = 0;
    S->n_nodes = 1;
    S->n_free = 0;
    S->pend_head = 0;
    S->pend_tail = 0;
}

// re-root slot on `move`; returns 0 or an error code.  wave-uniform.
// need_free > 0 (the self-play driver): the kept subtree holds at most min(`carried`, nodes of the whole tree - 1) nodes (every
// node below the chosen child was created by a distinct simulation through it); if that plus need_free exceeds the pool, the next
// search could run out of nodes half-way.  The reference's trees are unbounded Python objects; here such a move starts from a fresh root instead (as with
// reuse_tree = 0, counted in Slot::pool_resets) -- a trained network that puts nearly all visits on one move for many plies in a
// row (a chain) keeps nearly everything, move after move.  need_free = 0 (dbaz_advance): no check, exhaustion stays an error.
__device__ int reroot(const Geo &g, const TreeBufs &B, int slot, Slot *S, uint32_t *pool, int move,
                      int reuse, int lane, int need_free = 0)
{
    const int root = S->root;
    NodeMeta rm = load_meta(pool, g, root);
    uint32_t *nd0 = node_ptr(pool, g, root);
    if (move < 0 || move >= g.A)
        return DBAZ_EILLEGAL;
    int child = (int)nd0[META_DW + 3 * g.AS + move];
    int carried = (int)(nd0[META_DW + 2 * g.AS + move] & NS_MASK);
    GState st;
    int child_deep;
    if (child >= 0) {
        NodeMeta cm = load_meta(pool, g, child);
        st = cm.st;
        child_deep = cm.deepness;
    } else {
        st = rm.st;
        if (gs_play(g, st, move, nullptr) < 0)
            return DBAZ_EILLEGAL;
        child_deep = rm.deepness + 1;
        carried = 0;
    }
    __syncthreads();
    if (reuse && child >= 0 && need_free > 0 && carried + need_free > g.cap) {
        // visits over-count nodes (a revisit of a terminal leaf creates none: end games carry thousands of visits on a few
        // hundred nodes), so take the exact count before giving the subtree up: with the collector drained, the nodes in use
        // are those of the current tree (root, kept child and siblings)
        PoolState q = pool_load(S);
        while (q.tail > q.head) {
            pool_collect<4>(g, B, slot, pool, q, lane);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the next batch reads ring entries this one pushed
        }
        if (lane == 0) pool_store(q, S);
        const int live = q.n_nodes - q.n_free - 1;
        if ((carried < live ? carried : live) + need_free > g.cap) {
            reuse = 0;
            if (lane == 0) S->pool_resets++;
        }
        __syncthreads();
    }
    if (reuse && child >= 0) {
        const int tail = S->pend_tail;
        if (lane == 0) {
            nd0[META_DW + 3 * g.AS + move] = 0xFFFFFFFFu;           // the kept subtree is no child of the dropped root
            node_ptr(pool, g, child)[8] = 0xFFFFFFFFu;               // new root: parent = None
            (B.pend + (size_t)slot * g.cap)[ring_at(tail, g.cap)] = root;
        }
        S->pend_tail = tail + 1;
        S->root = child;
        S->deepness_correction = child_deep;
        S->tree_size = carried;
    } else if (reuse) {
        // children[move] did not exist: a fresh, unexpanded node becomes the root; nothing else survives
        pool_reset(S);
        init_node(pool, g, 0, st, -1, move, child_deep, lane);
        S->deepness_correction = child_deep;
        S->tree_size = carried;
    } else {
        // create_root_uct_node(child.game_state); next_node.move = move
        pool_reset(S);
        init_node(pool, g, 0, st, -1, move, 1, lane);
        S->deepness_correction = 0;
        S->tree_size = 0;
    }
    // new TreeRoot: fresh defaultdict slots and statistics (mcts.py:22-31)
    S->root_N = 0;
    S->root_W = 0.0f;
    S->root_prepped = 0;
    S->root_prior_f64 = 0;
    S->max_deepness = 0;
    S->terminal_count = 0;
    S->tt_epoch = S->tt_epoch + 1; // entries of older moves become the preferred victims (they stay valid while their node lives)
    __syncthreads();
    return 0;
}

// reset slot to a fresh game from the empty board (create_root_uct_node(BoxesState()))
__device__ void fresh_game(const Geo &g, const SearchCfg &cfg, const TreeBufs &B, int slot, Slot *S,
                           uint32_t *pool, long long game_idx, int lane)
{
    GState st;
    gs_init(g, st);
    int ff = S->ff_plies;
    int plies = 0;
    if (ff > 0) {
        // synthetic mid-game population (benchmark only): uniformly random legal plies
        uint2 key = make_uint2((uint32_t)cfg.seed ^ 0xA5A5A5A5u, (uint32_t)(cfg.seed >> 32));
        for (int t = 0; t < ff; t++) {
            int nb = gs_count_valid(g, st);
            if (nb <= 1) break;
            uint4 r = philox(make_uint4((uint32_t)game_idx, (uint32_t)(game_idx >> 32), (uint32_t)t, 0x11111111u), key);
            int pick = (int)(r.x % (uint32_t)nb);
            int mv = -1;
            for (int i = 0; i < g.A; i++) {
                if (gs_valid(g, st, i)) {
                    if (pick == 0) { mv = i; break; }
                    pick--;
                }
            }
            GState trial = st;
            gs_play(g, trial, mv, nullptr);
            if (gs_result(trial) != DBAZ_RESULT_NONE) break;
            st = trial;
            plies++;
        }
        S->ff_plies = 0;
    }
    init_node(pool, g, 0, st, -1, -1, 1, lane);
    pool_reset(S);
    S->root_N = 0;
    S->root_W = 0.0f;
    S->root_prepped = 0;
    S->root_prior_f64 = 0;
    S->deepness_correction = 0;
    S->max_deepness = 0;
    S->terminal_count = 0;
    S->tree_size = 0;
    S->move_idx = 0;
    S->n_rows = 0;
    S->game_idx = game_idx;
    S->temperature = 1.0;
    S->phase = PH_IDLE;
    S->sel_step = 0; // no leaf pending (step numbers start at 1)
    S->tt_epoch = S->tt_epoch + 1;
    (void)plies;
    __syncthreads();
}

__global__ void __launch_bounds__(WAVE) k_set_positions(Geo g, SearchCfg cfg, TreeBufs B, const int16_t *moves,
                                                        const int32_t *offsets)
{
    const int slot = blockIdx.x, lane = threadIdx.x;
    Slot *S = B.slots + slot;
    uint32_t *pool = B.nodes + (size_t)slot * g.cap * g.node_dw;
    GState st;
    gs_init(g, st);
    int err = 0;
    if (moves && offsets) {
        for (int i = offsets[slot]; i < offsets[slot + 1]; i++)
            if (gs_play(g, st, moves[i], nullptr) < 0) { err = DBAZ_EILLEGAL; break; }
    }
    S->ff_plies = 0;
    S->ff_reads = 0;
    S->quick_until = 0;
    fresh_game(g, cfg, B, slot, S, pool, slot, lane);
    init_node(pool, g, 0, st, -1, -1, 1, lane);
    if (lane == 0) {
        S->error = err;
        S->phase = err ? PH_ERROR : PH_IDLE;
        S->n_search = S->sum_path = S->n_eval = S->n_term = S->n_hit = 0;
        S->pool_high = 1;
        S->pool_resets = 0;
        S->pending = 0;
    }
}

// manual init_mcts_tree on every slot with moves[slot] >= 0
__global__ void __launch_bounds__(WAVE) k_advance_manual(Geo g, SearchCfg cfg, TreeBufs B, const int32_t *moves,
                                                         int reuse)
{
    const int slot = blockIdx.x, lane = threadIdx.x;
    Slot *S = B.slots + slot;
    if (S->phase == PH_ERROR || S->game_idx < 0)
        return;
    int mv = moves[slot];
    if (mv < 0)
        return;
    uint32_t *pool = B.nodes + (size_t)slot * g.cap * g.node_dw;
    int err = reroot(g, B, slot, S, pool, mv, reuse, lane);
    if (lane == 0) {
        if (err) { S->error = err; S->phase = PH_ERROR; }
        else { S->phase = PH_IDLE; S->move_idx += 1; }
    }
}

// ------------------------------------------------------------------------------------
// self-play driver: SelfPlay.play_game / get_next_move / get_datasets (self_play.py)
// ------------------------------------------------------------------------------------
__device__ bool try_emit(const Geo &g, const TreeBufs &B, int slot, Slot *S, uint32_t *pool, int lane)
{
    // rows of the finished game -> output buffer, z per row (self_play.py:105-112)
    NodeMeta rm = load_meta(pool, g, S->root); // terminal root
    const int n = S->n_rows;
    int base = 0;
    if (lane == 0) {
        base = atomicAdd(B.out_count, n);
        if (base + n > B.max_out) {
            atomicSub(B.out_count, n);
            base = -1;
        }
    }
    base = __shfl(base, 0);
    if (base < 0)
        return false;
    const int F = 3 * g.HW, A = g.A, rcap = g.E + 1;
    const int winner = rm.st.just_played;
    const int zt = rm.result;
    for (int r = 0; r < n; r++) {
        const int16_t *sx = B.row_x + ((size_t)slot * rcap + r) * F;
        int16_t *dx = B.out_x + (size_t)(base + r) * F;
        for (int i = lane; i < F; i += WAVE) dx[i] = sx[i];
        const int32_t *sv = B.row_vis + ((size_t)slot * rcap + r) * A;
        int32_t *dv = B.out_vis + (size_t)(base + r) * A;
        for (int i = lane; i < A; i += WAVE) dv[i] = sv[i];
        if (lane == 0) {
            RowMeta mm = B.row_meta[(size_t)slot * rcap + r];
            mm.z = (int8_t)((mm.player == winner) ? zt : -zt);
            B.out_meta[base + r] = mm;
        }
    }
    if (lane == 0) {
        atomicAdd((unsigned long long *)B.games_finished, 1ull);
    }
    return true;
}

__device__ void next_game_or_idle(const Geo &g, const SearchCfg &cfg, const TreeBufs &B, int slot, Slot *S,
                                  uint32_t *pool, float *ldsf, double *ldsd, int lane);

__device__ void start_move_search(const Geo &g, const SearchCfg &cfg, const TreeBufs &B, int slot, Slot *S,
                                  uint32_t *pool, float *ldsf, double *ldsd, int lane)
{
    // play_game loop head (self_play.py:57-66): temperature schedule, sims budget, search
    const int i = S->move_idx;
    for (int k = 0; k < cfg.n_temp; k++)
        if (cfg.temp_idx[k] == i) S->temperature = cfg.temp_val[k];
    // teacher-forced Dirichlet vector for this (game, ply), if scripted
    long long gi = S->game_idx - B.first_game;
    if (cfg.alpha > 0 && B.script_noise && gi >= 0 && gi < B.n_script && B.script_has_noise[gi] && i <= g.E) {
        const double *src = B.script_noise + ((size_t)gi * (g.E + 1) + i) * g.A;
        double *dst = B.noise_in + (size_t)slot * g.AS;
        for (int k = lane; k < g.A; k += WAVE) dst[k] = src[k];
        if (lane == 0) B.noise_valid[slot] = 1;
        __syncthreads();
        __threadfence_block();
    }
    begin_search(g, cfg, B, slot, S, pool, -1, ldsf, ldsd, lane);
}

__device__ void next_game_or_idle(const Geo &g, const SearchCfg &cfg, const TreeBufs &B, int slot, Slot *S,
                                  uint32_t *pool, float *ldsf, double *ldsd, int lane)
{
    long long gidx = 0;
    if (lane == 0)
        gidx = (long long)atomicAdd((unsigned long long *)B.next_game, 1ull);
    gidx = __shfl(gidx, 0);
    if (gidx >= B.last_game) {
        S->game_idx = -1;
        S->phase = PH_IDLE;
        return;
    }
    S->quick_until = 0; // only a slot's first game has quick plies
    fresh_game(g, cfg, B, slot, S, pool, gidx, lane);
    start_move_search(g, cfg, B, slot, S, pool, ldsf, ldsd, lane);
}

__global__ void __launch_bounds__(WAVE) k_selfplay_start(Geo g, SearchCfg cfg, TreeBufs B)
{
    __shared__ float ldsf[DBAZ_MAX_A];
    __shared__ double ldsd[DBAZ_MAX_A];
    const int slot = blockIdx.x, lane = threadIdx.x;
    Slot *S = B.slots + slot;
    uint32_t *pool = B.nodes + (size_t)slot * g.cap * g.node_dw;
    if (lane == 0) {
        S->error = 0;
        S->n_search = S->sum_path = S->n_eval = S->n_term = S->n_hit = 0;
        S->pool_high = 1;
        S->pool_resets = 0;
        S->pending = 0;
    }
    // deterministic initial assignment: slot i takes game first+i (the dispenser starts behind them)
    long long gidx = B.first_game + slot;
    if (gidx >= B.last_game) {
        S->game_idx = -1;
        S->phase = PH_IDLE;
        return;
    }
    fresh_game(g, cfg, B, slot, S, pool, gidx, lane);
    start_move_search(g, cfg, B, slot, S, pool, ldsf, ldsd, lane);
}

// One pass of the driver for every slot whose reads are done (PH_READY) or whose finished
// game still waits for output space (PH_EMIT).
// (the slots come from k_driver_scan's list: a step in which no slot needs the driver costs two tiny launches instead
// of one workgroup per game squeezing in between the network's workgroups)
__device__ void advance_one(const Geo &g, const SearchCfg &cfg, const TreeBufs &B, int slot, float *ldsf, double *ldsd,
                            int lane)
{
    Slot *S = B.slots + slot;
    const int phase = S->phase;
    if (phase != PH_READY && phase != PH_EMIT)
        return;
    uint32_t *pool = B.nodes + (size_t)slot * g.cap * g.node_dw;
    const int A = g.A, F = 3 * g.HW, rcap = g.E + 1;
    if (phase == PH_EMIT) {
        if (try_emit(g, B, slot, S, pool, lane))
            next_game_or_idle(g, cfg, B, slot, S, pool, ldsf, ldsd, lane);
        return;
    }
    const int root = S->root;
    NodeMeta rm = load_meta(pool, g, root);
    const uint32_t *NS0 = node_ptr(pool, g, root) + META_DW + 2 * g.AS;
    // ---- get_next_move (self_play.py:27-35) ----
    int vmax = 0;
    long long vsum = 0;
    for (int i = lane; i < A; i += WAVE) {
        int vc = (int)(NS0[i] & NS_MASK);
        vmax = vc > vmax ? vc : vmax;
        vsum += vc;
    }
    for (int o = 32; o > 0; o >>= 1) {
        int ov = __shfl_xor(vmax, o);
        vmax = ov > vmax ? ov : vmax;
        vsum += __shfl_xor(vsum, o);
    }
    int mv = -1;
    const int ply = S->move_idx;
    long long gi = S->game_idx - B.first_game;
    if (B.script_moves && gi >= 0 && gi < B.n_script && ply <= g.E)
        mv = B.script_moves[(size_t)gi * (g.E + 1) + ply];
    if (mv < 0) {
        // probs = (vc/vc.max())**(1/T); probs /= probs.sum(); np.random.choice(A, p=probs)
        const double invT = 1.0 / S->temperature;
        for (int i = lane; i < A; i += WAVE) {
            int vc = (int)(NS0[i] & NS_MASK);
            ldsd[i] = pow((double)vc / (double)vmax, invT);
        }
        __syncthreads();
        double ps = np_pairwise_sum<double>(ldsd, A, lane);
        __syncthreads();
        if (lane == 0) {
            double c = 0.0;
            for (int i = 0; i < A; i++) { c += ldsd[i] / ps; ldsd[i] = c; } // cumsum
        }
        __syncthreads();
        const double last = ldsd[A - 1];
        uint2 key = make_uint2((uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32));
        uint4 r = philox(make_uint4((uint32_t)S->game_idx, (uint32_t)(S->game_idx >> 32), (uint32_t)ply, 0x22222222u), key);
        const double u = u01(r.x, r.y);
        int cnt = 0;
        for (int i = lane; i < A; i += WAVE)
            cnt += (ldsd[i] / last <= u) ? 1 : 0; // searchsorted(side='right')
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        mv = cnt;
        if (mv >= A) { // unreachable (cdf[-1]/last == 1 > u); fall back to the most visited move
            int bv = -1, bidx = 0;
            for (int i = 0; i < A; i++) { int vc = (int)(NS0[i] & NS_MASK); if (vc > bv) { bv = vc; bidx = i; } }
            mv = bidx;
        }
        __syncthreads();
    }
    // ---- row of get_datasets for this root (self_play.py:107-117) ----
    const int r = S->n_rows;
    if (r < rcap) {
        int16_t *rx = B.row_x + ((size_t)slot * rcap + r) * F;
        for (int i = lane; i < F; i += WAVE) rx[i] = (int16_t)gs_feature(g, rm.st, i);
        int32_t *rv = B.row_vis + ((size_t)slot * rcap + r) * A;
        for (int i = lane; i < A; i += WAVE) rv[i] = (int32_t)(NS0[i] & NS_MASK);
        if (lane == 0) {
            RowMeta mm;
            mm.game_idx = (int32_t)S->game_idx;
            mm.move_idx = (int16_t)ply;
            mm.move = (int16_t)rm.move;
            mm.played = (int16_t)mv;
            mm.max_deepness = (int16_t)(S->max_deepness - S->deepness_correction);
            mm.tree_size = S->tree_size;
            mm.terminal_count = S->terminal_count;
            mm.q_value = S->root_W / (float)(1 + S->root_N); // TreeRoot.get_tree_stats, mcts.py:33-36
            mm.player = (int8_t)rm.st.to_play;
            mm.z = 0;
            mm.pad = 0;
            B.row_meta[(size_t)slot * rcap + r] = mm;
        }
        S->n_rows = r + 1;
    }
    __syncthreads();
    // ---- init_mcts_tree ----
    int err = reroot(g, B, slot, S, pool, mv, cfg.reuse_tree, lane, cfg.mcts_num_read + 2);
    if (err) {
        if (lane == 0) { S->error = err; S->phase = PH_ERROR; }
        return;
    }
    S->move_idx = ply + 1;
    if (lane == 0) atomicAdd((unsigned long long *)B.moves_played, 1ull);
    NodeMeta nm = load_meta(pool, g, S->root);
    if (nm.flags & NF_TERMINAL) {
        if (try_emit(g, B, slot, S, pool, lane))
            next_game_or_idle(g, cfg, B, slot, S, pool, ldsf, ldsd, lane);
        else
            S->phase = PH_EMIT;
        return;
    }
    start_move_search(g, cfg, B, slot, S, pool, ldsf, ldsd, lane);
}

// which slots need the driver: finished reads or a blocked emit (the pass runs after the previous step's expand/backup)
__global__ void __launch_bounds__(256) k_driver_scan(SearchCfg cfg, TreeBufs B, int n_slots)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool need = false;
    if (i < n_slots) {
        const int phase = B.slots[i].phase;
        need = phase == PH_EMIT || phase == PH_READY;
    }
    const unsigned long long m = __ballot(need);
    const int lane = threadIdx.x & (WAVE - 1);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(B.drv_count, (int)__popcll(m));
    base = __shfl(base, 0);
    if (need) B.drv_list[base + (int)__popcll(m & ((1ull << lane) - 1ull))] = i;
}

__global__ void __launch_bounds__(WAVE) k_advance_auto(Geo g, SearchCfg cfg, TreeBufs B)
{
    __shared__ float ldsf[DBAZ_MAX_A];
    __shared__ double ldsd[DBAZ_MAX_A];
    const int lane = threadIdx.x;
    const int n = *B.drv_count;
    for (int li = blockIdx.x; li < n; li += gridDim.x) {
        advance_one(g, cfg, B, B.drv_list[li], ldsf, ldsd, lane);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// readback helpers + batched rules kernels (one thread per state)
// ------------------------------------------------------------------------------------
__global__ void k_get_roots(Geo g, TreeBufs B, int n_slots, double *priors, float *tv, int32_t *nv, int32_t *changed,
                            int32_t *stats, float *q, float *root_tv, int32_t *root_nv, uint64_t *edges,
                            int16_t *b2c2, int8_t *to_play, int8_t *just_played, int8_t *result, int8_t *expanded)
{
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= n_slots) return;
    Slot *S = B.slots + slot;
    uint32_t *pool = B.nodes + (size_t)slot * g.cap * g.node_dw;
    NodeMeta rm = load_meta(pool, g, S->root);
    const uint32_t *nd = node_ptr(pool, g, S->root);
    const int A = g.A;
    for (int i = lane; i < A; i += WAVE) {
        uint32_t ns = nd[META_DW + 2 * g.AS + i];
        int c = (int)nd[META_DW + 3 * g.AS + i];
        if (priors) {
            double p;
            if (S->root_prepped) p = B.root_prior[(size_t)slot * g.AS + i];
            else p = (rm.flags & NF_EXPANDED) && !(rm.flags & NF_TERMINAL)
                         ? (double)reinterpret_cast<const float *>(nd + META_DW)[i] : 0.0;
            priors[(size_t)slot * A + i] = p;
        }
        if (tv) tv[(size_t)slot * A + i] = reinterpret_cast<const float *>(nd + META_DW + g.AS)[i];
        if (nv) nv[(size_t)slot * A + i] = (int32_t)(ns & NS_MASK);
        if (changed) {
            // child_player_changed: 1 until the child is expanded, then +/-1 (mcts.py:61-62,119)
            int ch = 1;
            if (c >= 0) {
                NodeMeta cm = load_meta(pool, g, c);
                if (cm.flags & NF_EXPANDED) ch = (ns & NS_SAME) ? 1 : -1;
            }
            changed[(size_t)slot * A + i] = ch;
        }
    }
    if (lane == 0) {
        if (stats) {
            stats[slot * 3 + 0] = S->max_deepness - S->deepness_correction;
            stats[slot * 3 + 1] = S->tree_size;
            stats[slot * 3 + 2] = S->terminal_count;
        }
        if (q) q[slot] = S->root_W / (float)(1 + S->root_N);
        if (root_tv) root_tv[slot] = S->root_W;
        if (root_nv) root_nv[slot] = S->root_N;
        if (edges) {
            edges[slot * 4 + 0] = rm.st.e0; edges[slot * 4 + 1] = rm.st.e1;
            edges[slot * 4 + 2] = rm.st.e2; edges[slot * 4 + 3] = rm.st.e3;
        }
        if (b2c2) { b2c2[slot * 2] = (int16_t)rm.st.b2c0; b2c2[slot * 2 + 1] = (int16_t)rm.st.b2c1; }
        if (to_play) to_play[slot] = (int8_t)rm.st.to_play;
        if (just_played) just_played[slot] = (int8_t)rm.st.just_played;
        if (result) result[slot] = (int8_t)gs_result(rm.st);
        if (expanded) expanded[slot] = (rm.flags & NF_EXPANDED) ? 1 : 0;
    }
}

__global__ void k_get_leaves(Geo g, TreeBufs B, int n_slots, int16_t *leaf_x, uint8_t *need_eval, int32_t *n_active)
{
    const int slot = blockIdx.x, lane = threadIdx.x;
    Slot *S = B.slots + slot;
    const bool active = S->phase == PH_EXPAND_ROOT || S->phase == PH_SIMS;
    const bool need = active && !S->leaf_terminal;
    const int F = 3 * g.HW;
    const float *f = B.feat + (size_t)slot * F;
    for (int i = lane; i < F; i += WAVE)
        leaf_x[(size_t)slot * F + i] = need ? (int16_t)f[i] : (int16_t)0;
    if (lane == 0) {
        need_eval[slot] = need ? 1 : 0;
        if (active) atomicAdd(n_active, 1);
    }
}

// Slot array -> SlotSummary (out must be zeroed except first_error_slot = 0x7fffffff)
__global__ void __launch_bounds__(256) k_slot_summary(TreeBufs B, int n_slots, SlotSummary *out)
{
    unsigned long long ns = 0, ne = 0, nh = 0, nt = 0, sp = 0, nr = 0;
    int active = 0, error = 0, blocked = 0, high = 0, ferr = 0x7fffffff;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += gridDim.x * blockDim.x) {
        const Slot &S = B.slots[i];
        ns += (unsigned long long)S.n_search; ne += (unsigned long long)S.n_eval; nh += (unsigned long long)S.n_hit;
        nt += (unsigned long long)S.n_term; sp += (unsigned long long)S.sum_path; nr += (unsigned long long)S.pool_resets;
        const int ph = S.phase;
        if (ph == PH_ERROR) { error++; ferr = min(ferr, i); }
        else if (S.game_idx >= 0 && ph != PH_IDLE) active++;
        if (ph == PH_EMIT) blocked++;
        high = max(high, S.pool_high);
    }
    for (int o = 32; o > 0; o >>= 1) {
        ns += __shfl_xor(ns, o); ne += __shfl_xor(ne, o); nh += __shfl_xor(nh, o); nt += __shfl_xor(nt, o); sp += __shfl_xor(sp, o);
        nr += __shfl_xor(nr, o);
        active += __shfl_xor(active, o); error += __shfl_xor(error, o); blocked += __shfl_xor(blocked, o);
        high = max(high, __shfl_xor(high, o)); ferr = min(ferr, __shfl_xor(ferr, o));
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        atomicAdd(&out->n_search, ns); atomicAdd(&out->n_eval, ne); atomicAdd(&out->n_hit, nh); atomicAdd(&out->n_term, nt);
        atomicAdd(&out->sum_path, sp); atomicAdd(&out->n_reset, nr);
        atomicAdd(&out->active, active); atomicAdd(&out->error, error); atomicAdd(&out->blocked, blocked);
        atomicMax(&out->pool_high, high);
        if (ferr != 0x7fffffff) atomicMin(&out->first_error_slot, ferr);
    }
}
// second pass (only when an error was seen): the code of the lowest failing slot
__global__ void k_slot_error_code(TreeBufs B, SlotSummary *out)
{
    if (out->first_error_slot != 0x7fffffff) out->first_error_code = B.slots[out->first_error_slot].error;
}

// wall-clock cut-off of UCT_search (mcts.py:232-233): the reads that were not started are dropped
__global__ void k_stop_search(TreeBufs B, int n_slots)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    Slot *S = B.slots + i;
    if (S->phase == PH_SIMS) { S->sims_left = 0; S->phase = PH_READY; }
}

__global__ void k_count_active(TreeBufs B, int n_slots, int32_t *out /*[3]: searching, ready, error*/)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int ph = i < n_slots ? B.slots[i].phase : PH_IDLE;
    // one atomic per wave and class (same-address atomics serialise)
    const unsigned long long m0 = __ballot(ph == PH_EXPAND_ROOT || ph == PH_SIMS);
    const unsigned long long m1 = __ballot(ph == PH_READY || ph == PH_EMIT);
    const unsigned long long m2 = __ballot(ph == PH_ERROR);
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        if (m0) atomicAdd(out + 0, (int)__popcll(m0));
        if (m1) atomicAdd(out + 1, (int)__popcll(m1));
        if (m2) atomicAdd(out + 2, (int)__popcll(m2));
    }
}

__global__ void k_rules(Geo g, int op, int n, uint64_t *edges, int16_t *b2c2, int8_t *to_play, int8_t *just_played,
                        const int32_t *moves, int8_t *n_closed, int8_t *closed_lc, uint8_t *valid, int8_t *result,
                        int16_t *x)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    GState s;
    gs_init(g, s);
    if (op != 0) {
        if (edges) { s.e0 = edges[4 * i]; s.e1 = edges[4 * i + 1]; s.e2 = edges[4 * i + 2]; s.e3 = edges[4 * i + 3]; }
        if (b2c2) { s.b2c0 = b2c2[2 * i]; s.b2c1 = b2c2[2 * i + 1]; }
        if (to_play) s.to_play = to_play[i];
        if (just_played) s.just_played = just_played[i];
    }
    bool wb = false;
    switch (op) {
    case 0: wb = true; break;                                  // __init__
    case 1:                                                    // get_valid_moves
        for (int k = 0; k < g.A; k++) valid[(size_t)i * g.A + k] = gs_valid(g, s, k) ? 1 : 0;
        break;
    case 2: {                                                  // play_
        int cl[4] = {-1, -1, -1, -1};
        int r = gs_play(g, s, moves[i], cl);
        n_closed[i] = (int8_t)r;
        if (closed_lc) for (int k = 0; k < 4; k++) closed_lc[4 * i + k] = (int8_t)((r > 0 && k < 2 * r) ? cl[k] : -1);
        wb = r >= 0;
        break;
    }
    case 3: result[i] = (int8_t)gs_result(s); break;           // get_result
    case 4:                                                    // get_features
        for (int k = 0; k < 3 * g.HW; k++) x[(size_t)i * 3 * g.HW + k] = (int16_t)gs_feature(g, s, k);
        break;
    }
    if (wb) {
        edges[4 * i] = s.e0; edges[4 * i + 1] = s.e1; edges[4 * i + 2] = s.e2; edges[4 * i + 3] = s.e3;
        b2c2[2 * i] = (int16_t)s.b2c0; b2c2[2 * i + 1] = (int16_t)s.b2c1;
        to_play[i] = (int8_t)s.to_play;
        just_played[i] = (int8_t)s.just_played;
    }
}

// ------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------
void tree_launch_search_begin(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots,
                              const int32_t *num_reads_dev)
{
    hipLaunchKernelGGL(k_search_begin, dim3(n_slots), dim3(WAVE), 0, s, g, c, B, num_reads_dev);
}
void tree_launch_select(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots)
{
    switch ((g.A + WAVE - 1) / WAVE) {
    case 1: hipLaunchKernelGGL(k_select<1>, dim3((n_slots + SELECT_WAVES - 1) / SELECT_WAVES), dim3(WAVE * SELECT_WAVES), 0, s, g, c, B, n_slots); break;
    case 2: hipLaunchKernelGGL(k_select<2>, dim3((n_slots + SELECT_WAVES - 1) / SELECT_WAVES), dim3(WAVE * SELECT_WAVES), 0, s, g, c, B, n_slots); break;
    case 3: hipLaunchKernelGGL(k_select<3>, dim3((n_slots + SELECT_WAVES - 1) / SELECT_WAVES), dim3(WAVE * SELECT_WAVES), 0, s, g, c, B, n_slots); break;
    default: hipLaunchKernelGGL(k_select<4>, dim3((n_slots + SELECT_WAVES - 1) / SELECT_WAVES), dim3(WAVE * SELECT_WAVES), 0, s, g, c, B, n_slots); break;
    }
}
void tree_launch_select_multi(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots)
{
    switch ((g.A + WAVE - 1) / WAVE) {
    case 1: hipLaunchKernelGGL(k_select_multi<1>, dim3(n_slots), dim3(WAVE), 0, s, g, c, B, n_slots); break;
    case 2: hipLaunchKernelGGL(k_select_multi<2>, dim3(n_slots), dim3(WAVE), 0, s, g, c, B, n_slots); break;
    case 3: hipLaunchKernelGGL(k_select_multi<3>, dim3(n_slots), dim3(WAVE), 0, s, g, c, B, n_slots); break;
    default: hipLaunchKernelGGL(k_select_multi<4>, dim3(n_slots), dim3(WAVE), 0, s, g, c, B, n_slots); break;
    }
}
void tree_launch_expand_backup_multi(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots)
{
    hipLaunchKernelGGL(k_expand_backup_multi, dim3(n_slots), dim3(WAVE), 0, s, g, c, B);
}
void tree_launch_expand_backup(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots)
{
    hipLaunchKernelGGL(k_expand_backup, dim3(n_slots), dim3(WAVE), 0, s, g, c, B);
}
void tree_launch_set_positions(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots,
                               const int16_t *moves_dev, const int32_t *offsets_dev)
{
    hipLaunchKernelGGL(k_set_positions, dim3(n_slots), dim3(WAVE), 0, s, g, c, B, moves_dev, offsets_dev);
}
void tree_launch_advance_manual(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots,
                                const int32_t *moves_dev, int reuse)
{
    hipLaunchKernelGGL(k_advance_manual, dim3(n_slots), dim3(WAVE), 0, s, g, c, B, moves_dev, reuse);
}
void tree_launch_selfplay_start(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots)
{
    hipLaunchKernelGGL(k_selfplay_start, dim3(n_slots), dim3(WAVE), 0, s, g, c, B);
}
void tree_launch_advance_auto(hipStream_t s, const Geo &g, const SearchCfg &c, const TreeBufs &B, int n_slots)
{
    hipLaunchKernelGGL(k_driver_scan, dim3((n_slots + 255) / 256), dim3(256), 0, s, c, B, n_slots);
    hipLaunchKernelGGL(k_advance_auto, dim3(n_slots < 1024 ? n_slots : 1024), dim3(WAVE), 0, s, g, c, B);
}
void tree_launch_get_roots(hipStream_t s, const Geo &g, const TreeBufs &B, int n_slots, double *priors, float *tv,
                           int32_t *nv, int32_t *changed, int32_t *stats, float *q, float *root_tv, int32_t *root_nv,
                           uint64_t *edges, int16_t *b2c2, int8_t *to_play, int8_t *just_played, int8_t *result,
                           int8_t *expanded)
{
    hipLaunchKernelGGL(k_get_roots, dim3(n_slots), dim3(WAVE), 0, s, g, B, n_slots, priors, tv, nv, changed, stats, q,
                       root_tv, root_nv, edges, b2c2, to_play, just_played, result, expanded);
}
void tree_launch_get_leaves(hipStream_t s, const Geo &g, const TreeBufs &B, int n_slots, int16_t *leaf_x,
                            uint8_t *need_eval, int32_t *n_active)
{
    hipLaunchKernelGGL(k_get_leaves, dim3(n_slots), dim3(WAVE), 0, s, g, B, n_slots, leaf_x, need_eval, n_active);
}
void tree_launch_stop_search(hipStream_t s, const TreeBufs &B, int n_slots)
{
    hipLaunchKernelGGL(k_stop_search, dim3((n_slots + 255) / 256), dim3(256), 0, s, B, n_slots);
}
void tree_launch_count_active(hipStream_t s, const TreeBufs &B, int n_slots, int32_t *out3)
{
    hipLaunchKernelGGL(k_count_active, dim3((n_slots + 255) / 256), dim3(256), 0, s, B, n_slots, out3);
}
void tree_launch_slot_summary(hipStream_t s, const TreeBufs &B, int n_slots, SlotSummary *out_dev)
{
    int blocks = (n_slots + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(k_slot_summary, dim3(blocks), dim3(256), 0, s, B, n_slots, out_dev);
    hipLaunchKernelGGL(k_slot_error_code, dim3(1), dim3(1), 0, s, B, out_dev);
}
void tree_launch_rules(hipStream_t s, const Geo &g, int op, int n, uint64_t *edges, int16_t *b2c2, int8_t *to_play,
                       int8_t *just_played, const int32_t *moves, int8_t *n_closed, int8_t *closed_lc, uint8_t *valid,
                       int8_t *result, int16_t *x)
{
    hipLaunchKernelGGL(k_rules, dim3((n + 127) / 128), dim3(128), 0, s, g, op, n, edges, b2c2, to_play, just_played,
                       moves, n_closed, closed_lc, valid, result, x);
}
