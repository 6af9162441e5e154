// engine.hip -- C ABI (include/dbaz.h) of the MI355X self-play rollout engine.
// Host-side orchestration only: device memory, the per-step launch sequence
//   select -> [policy/value network] -> expand+backup -> driver advance
// on one HIP stream, HIP-event timing, and host<->device staging for the boundary.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <numeric>
#include <string>
#include <vector>

#include "common.h"
#include "nn.h"
#include "replay.h"
#include "tree.h"

static thread_local std::string g_create_error; // message of a failed dbaz_create (no handle to keep it in); per thread

struct dbaz_engine {
    dbaz_config cfg;
    Geo g;
    SearchCfg sc;
    TreeBufs B;
    int n_slots = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // re-rooting / game turnover (k_advance_auto) runs here, next to the network
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string err;
    std::vector<void *> allocs;
    NNState *nn = nullptr;   // the model that dbaz_nn_* calls address (nns[cur_model])
    NNState *nns[2] = {nullptr, nullptr};
    int cur_model = 0;
    // staging (device) for the boundary
    void *stage = nullptr;
    size_t stage_bytes = 0;
    int32_t *d_small = nullptr; // 16 ints of scratch counters
    // counters / timing
    int64_t steps = 0;
    bool timing = false;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double ms_total = 0, ms_nn_tower = 0;
    int64_t nn_launches = 0;
    int64_t steps_at_t0 = 0;
    // scripts (host copies kept until selfplay_start uploads them)
    std::vector<int16_t> script_moves;
    std::vector<double> script_noise;
    std::vector<uint8_t> script_has_noise;
    int64_t script_first = -1;
    int n_script = 0;
    std::vector<int32_t> ff_plies, ff_reads, quick_plies;
    SlotSummary *d_sum = nullptr;
    bool selfplay = false;
    bool late_join = false; // the driver pass is joined behind the network launch instead of in front of it (sim_step)
    int eval_round = 0, eval_defer_max = 0; // full rounds only (tree.hip, above k_select): set when the network is committed
    bool search_open = false;
    int search_iters_left = 0;
    // packed replay rows
    void *replay_dev = nullptr;
    size_t replay_bytes = 0;
    ReplayDS *rds = nullptr;    // training data path (replay.hip): the selected dataset, created on first use
    ReplayDS *rds_all[DBAZ_MAX_DATASETS] = {nullptr, nullptr, nullptr, nullptr};
    int cur_ds = 0;
};

static int set_error(dbaz_engine *e, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_create_error = buf;
    return code;
}

// one handle = one GPU: every entry point re-selects the handle's device, so that a host that
// switched the current HIP device in between (torch with several GPUs) cannot mis-route a launch
#define USE_DEVICE(e)                                                                   \
    do {                                                                                \
        if (e) HIP_CHECK_RET(e, hipSetDevice((e)->cfg.device));                         \
    } while (0)

extern "C" const char *dbaz_last_error(const dbaz_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }
extern "C" int dbaz_version(void) { return DBAZ_ABI_VERSION; }
extern "C" int dbaz_nodes_per_slot(const dbaz_engine *e) { return e ? e->g.cap : 0; }
#ifdef DBAZ_STAMP
// diagnostic builds (tools/stamp_build_run.sh) only: the shipped library does not export it, include/dbaz.h does not declare it
int nn_read_stamps(NNState *nn, unsigned long long *out, int n_wg);
extern "C" int dbaz_debug_read_stamps(dbaz_engine *e, unsigned long long *out, int n_wg);
#endif

template <typename T>
static int dmalloc(dbaz_engine *e, T **p, size_t count, bool zero = true)
{
    void *q = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    HIP_CHECK_RET(e, hipMalloc(&q, bytes));
    e->allocs.push_back(q);
    if (zero) HIP_CHECK_RET(e, hipMemsetAsync(q, 0, bytes, e->stream));
    *p = (T *)q;
    return 0;
}

static int ensure_stage(dbaz_engine *e, size_t bytes)
{
    if (bytes <= e->stage_bytes) return 0;
    if (e->stage) { HIP_CHECK_RET(e, hipStreamSynchronize(e->stream)); HIP_CHECK_RET(e, hipFree(e->stage)); e->stage = nullptr; }
    bytes = (bytes + 4095) & ~(size_t)4095;
    HIP_CHECK_RET(e, hipMalloc(&e->stage, bytes));
    e->stage_bytes = bytes;
    return 0;
}

// carve aligned sub-buffers out of the staging area
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *b) : base((char *)b) {}
    template <typename T> T *take(size_t n)
    {
        off = (off + 255) & ~(size_t)255;
        T *p = (T *)(base + off);
        off += n * sizeof(T);
        return p;
    }
    static size_t need(std::initializer_list<size_t> sizes)
    {
        size_t t = 0;
        for (size_t s : sizes) t = ((t + 255) & ~(size_t)255) + s;
        return t + 256;
    }
};

static int upload_tables(dbaz_engine *e)
{
    const SearchCfg &sc = e->sc;
    const long long tn = sc.table_n;
    std::vector<double> pbc(tn), sq(tn);
    for (long long n = 0; n < tn; n++) {
        pbc[n] = log(((double)n + sc.cpuct_base + 1.0) / sc.cpuct_base) + sc.cpuct;
        sq[n] = sqrt((double)n);
    }
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    HIP_CHECK_RET(e, hipMemcpy((void *)e->B.pbc_table, pbc.data(), tn * sizeof(double), hipMemcpyHostToDevice));
    HIP_CHECK_RET(e, hipMemcpy((void *)e->B.sqrt_table, sq.data(), tn * sizeof(double), hipMemcpyHostToDevice));
    return DBAZ_OK;
}

extern "C" int dbaz_set_search_params(dbaz_engine *e, double cpuct, double cpuct_base, double noise_alpha, double noise_coeff)
{
    if (!e) return DBAZ_EINVAL;
    USE_DEVICE(e);
    if (!(cpuct_base > 0)) return set_error(e, DBAZ_EINVAL, "cpuct_base must be > 0");
    const bool tables = cpuct != e->sc.cpuct || cpuct_base != e->sc.cpuct_base;
    e->sc.cpuct = cpuct; e->sc.cpuct_base = cpuct_base; e->sc.alpha = noise_alpha; e->sc.coeff = noise_coeff;
    return tables ? upload_tables(e) : DBAZ_OK;
}

extern "C" int dbaz_create(const dbaz_config *cfg, dbaz_engine **out)
{
    if (!cfg || !out) return set_error(nullptr, DBAZ_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->rows < 1 || cfg->cols < 1 || 2 * (cfg->rows + 1) * (cfg->cols + 1) > DBAZ_MAX_A)
        return set_error(nullptr, DBAZ_EINVAL, "board %dx%d unsupported (A must be <= %d)", cfg->rows, cfg->cols, DBAZ_MAX_A);
    if (cfg->n_slots < 1) return set_error(nullptr, DBAZ_EINVAL, "n_slots must be >= 1");
    if (cfg->mcts_num_read < 0 || cfg->cpuct_base <= 0) return set_error(nullptr, DBAZ_EINVAL, "bad search parameters");
    if (cfg->evaluator < 0 || cfg->evaluator > DBAZ_EVAL_EXTERNAL) return set_error(nullptr, DBAZ_EINVAL, "bad evaluator");
    if (cfg->match_play && (cfg->evaluator2 < 0 || cfg->evaluator2 >= DBAZ_EVAL_EXTERNAL || cfg->evaluator == DBAZ_EVAL_EXTERNAL))
        return set_error(nullptr, DBAZ_EINVAL, "match play needs two device evaluators");
    if (cfg->n_temp < 0 || cfg->n_temp > 8) return set_error(nullptr, DBAZ_EINVAL, "n_temp must be in 0..8");
    if (cfg->transposition_cache < 0 || cfg->transposition_cache > 2) return set_error(nullptr, DBAZ_EINVAL, "transposition_cache must be 0, 1 or 2");
    if (cfg->max_pending_evals < 0 || cfg->max_pending_evals > 1024) return set_error(nullptr, DBAZ_EINVAL, "max_pending_evals must be in 0..1024");
    if (cfg->max_pending_evals > 1 && (cfg->evaluator == DBAZ_EVAL_EXTERNAL || cfg->match_play))
        return set_error(nullptr, DBAZ_EINVAL, "max_pending_evals > 1 needs a device evaluator and no match play");
#ifdef DBAZ_DEBUG
    if (cfg->nn_precision < 0 || cfg->nn_precision > 14) return set_error(nullptr, DBAZ_EINVAL, "nn_precision must be in 0..14 (debug build)");
#else
    if (cfg->nn_precision < 0 || cfg->nn_precision > 1) return set_error(nullptr, DBAZ_EINVAL, "nn_precision must be 0 (exact f32) or 1 (f16x3)");
#endif
    if (cfg->eval_round < -1 || cfg->eval_defer_max < 0) return set_error(nullptr, DBAZ_EINVAL, "bad eval_round / eval_defer_max");
    if (cfg->selfplay_pending && cfg->max_pending_evals <= 1) return set_error(nullptr, DBAZ_EINVAL, "selfplay_pending needs max_pending_evals > 1");
    int ndev = 0;
    hipError_t herr = hipGetDeviceCount(&ndev);
    if (herr != hipSuccess || ndev <= 0)
        return set_error(nullptr, DBAZ_EDEVICE, "no HIP device available (%s): the engine has no CPU fallback",
                         herr != hipSuccess ? hipGetErrorString(herr) : "device count 0");
    if (cfg->device < 0 || cfg->device >= ndev) return set_error(nullptr, DBAZ_EINVAL, "device %d out of range (%d devices)", cfg->device, ndev);
    dbaz_engine *e = new dbaz_engine();
    e->cfg = *cfg;
    e->n_slots = cfg->n_slots;
#define CREATE_CHECK(call)                                                                         \
    do {                                                                                           \
        int _r = (call);                                                                           \
        if (_r) { g_create_error = e->err; dbaz_destroy(e); return _r; }                           \
    } while (0)
#define CREATE_HIP(call)                                                                           \
    do {                                                                                           \
        hipError_t _h = (call);                                                                    \
        if (_h != hipSuccess) {                                                                    \
            set_error(nullptr, DBAZ_EDEVICE, "%s: %s", #call, hipGetErrorString(_h));              \
            dbaz_destroy(e);                                                                       \
            return DBAZ_EDEVICE;                                                                   \
        }                                                                                          \
    } while (0)
    CREATE_HIP(hipSetDevice(cfg->device));
    CREATE_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    CREATE_HIP(hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking));
    CREATE_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    CREATE_HIP(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    CREATE_HIP(hipEventCreate(&e->ev_t0));
    CREATE_HIP(hipEventCreate(&e->ev_t1));

    Geo &g = e->g;
    memset(&g, 0, sizeof(g));
    g.rows = cfg->rows; g.cols = cfg->cols; g.H = g.rows + 1; g.W = g.cols + 1; g.HW = g.H * g.W;
    g.A = 2 * g.HW; g.AS = (g.A + 3) & ~3; g.B = g.rows * g.cols; g.E = 2 * g.rows * g.cols + g.rows + g.cols;
    g.node_dw = META_DW + 4 * g.AS;
    if (cfg->nodes_per_slot > 0) {
        g.cap = cfg->nodes_per_slot;
    } else {
        // default: 10 searches' worth of nodes; up to 40 where all pools together stay within a third of the device's memory
        // (a trained network keeps most of its tree from move to move: dbaz_counters.pool_resets)
        const long long base = 10LL * (cfg->mcts_num_read + 2), wide = 4 * base;
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess || mem_total == 0) mem_total = (size_t)288 << 30;
        const long long fit = (long long)(mem_total / 3) / ((long long)cfg->n_slots * g.node_dw * 4);
        g.cap = (int)std::min(wide, std::max(base, fit));
    }
    if (g.cap < 8) g.cap = 8;
    g.dmax = g.E + 4;
    for (int c = 0; c < g.W; c++) { int i = (1 * g.H + g.rows) * g.W + c; g.sentinel[i >> 6] |= 1ull << (i & 63); }
    for (int l = 0; l < g.H; l++) { int i = (0 * g.H + l) * g.W + g.cols; g.sentinel[i >> 6] |= 1ull << (i & 63); }
    for (int i = 0; i < g.A; i++) g.amask[i >> 6] |= 1ull << (i & 63);

    SearchCfg &sc = e->sc;
    memset(&sc, 0, sizeof(sc));
    sc.cpuct = cfg->cpuct; sc.cpuct_base = cfg->cpuct_base; sc.alpha = cfg->noise_alpha; sc.coeff = cfg->noise_coeff;
    sc.mcts_num_read = cfg->mcts_num_read; sc.reuse_tree = cfg->reuse_tree; sc.n_temp = cfg->n_temp;
    for (int i = 0; i < cfg->n_temp; i++) { sc.temp_idx[i] = cfg->temp_idx[i]; sc.temp_val[i] = cfg->temp_val[i]; }
    sc.evaluator = cfg->evaluator; sc.seed = cfg->seed;
    sc.match_play = cfg->match_play ? 1 : 0; sc.evaluator2 = cfg->evaluator2;
    sc.gc_lazy = (cfg->debug_flags & DBAZ_DBG_LAZY_GC) ? 64 : 0;
    sc.pending = cfg->max_pending_evals > 1 ? cfg->max_pending_evals : 1;
    // self-play in waves follows the reference's bookkeeping (visits added at backup, mcts.py:121-126); the single-tree search
    // of players.AZPlayer defaults to counted virtual visits (dbaz_set_pending changes either)
    sc.virtual_visits = cfg->selfplay_pending ? 0 : 1;

    TreeBufs &B = e->B;
    memset(&B, 0, sizeof(B));
    const size_t ns = e->n_slots;
    const int F = 3 * g.HW, rcap = g.E + 1;
    B.max_out = cfg->max_out_rows > 0 ? cfg->max_out_rows : (int32_t)std::min<size_t>(std::max<size_t>(4096, 2 * ns * rcap), (size_t)1 << 26);
    if (B.max_out < rcap) { set_error(nullptr, DBAZ_EINVAL, "max_out_rows must hold at least one game (%d rows)", rcap); dbaz_destroy(e); return DBAZ_EINVAL; }
    CREATE_CHECK(dmalloc(e, &B.nodes, ns * g.cap * g.node_dw, false));
    CREATE_CHECK(dmalloc(e, &B.slots, ns));
    CREATE_CHECK(dmalloc(e, &B.path, ns * g.dmax));
    CREATE_CHECK(dmalloc(e, &B.root_prior, ns * g.AS));
    CREATE_CHECK(dmalloc(e, &B.noise_in, ns * g.AS));
    CREATE_CHECK(dmalloc(e, &B.noise_valid, ns));
    CREATE_CHECK(dmalloc(e, &B.feat, ns * F));
    CREATE_CHECK(dmalloc(e, &B.evalP, ns * g.AS));
    CREATE_CHECK(dmalloc(e, &B.evalV, ns));
    CREATE_CHECK(dmalloc(e, &B.eval_list, ns));
    CREATE_CHECK(dmalloc(e, &B.eval_list2, ns));
    CREATE_CHECK(dmalloc(e, &B.n_eval, 4));
    CREATE_CHECK(dmalloc(e, &B.pend, ns * g.cap, false));
    CREATE_CHECK(dmalloc(e, &B.freel, ns * g.cap, false));
    B.kmax = cfg->max_pending_evals > 1 ? cfg->max_pending_evals : 1;
    if (B.kmax > 1) {
        const size_t nk = ns * (size_t)B.kmax;
        CREATE_CHECK(dmalloc(e, &B.simrec, nk));
        CREATE_CHECK(dmalloc(e, &B.path_m, nk * g.dmax));
        CREATE_CHECK(dmalloc(e, &B.feat_m, nk * F));
        CREATE_CHECK(dmalloc(e, &B.evalP_m, nk * g.AS));
        CREATE_CHECK(dmalloc(e, &B.evalV_m, nk));
        CREATE_CHECK(dmalloc(e, &B.list_m, nk));
    }
    CREATE_CHECK(dmalloc(e, &B.drv_list, ns));
    CREATE_CHECK(dmalloc(e, &B.drv_count, 4));
    B.tt = nullptr;
    B.tt_mask = 0;
    {
        auto nn_ev = [](int ev) { return ev == DBAZ_EVAL_RESNET || ev == DBAZ_EVAL_SIMPLENN; };
        // off in two-model match play, like the reference (self_play.py:230): a kept twin may belong to the other model
        // transposition_cache = 2 forces the table on for the formula evaluators too (their (p, v) is a pure function
        // of the same key): the hit path -- twin's prior row, v from its meta block, re-insertion at re-root -- then runs
        // under the oracle-pinned searches of the test-suite
        const bool formula_ev = cfg->evaluator == DBAZ_EVAL_FORMULA_HASH || cfg->evaluator == DBAZ_EVAL_FORMULA_UNIFORM;
        if (!cfg->match_play && ((cfg->transposition_cache == 0 && nn_ev(cfg->evaluator)) ||
                                 (cfg->transposition_cache == 2 && (nn_ev(cfg->evaluator) || formula_ev)))) {
            size_t tcap = 64;
            // (measured round 2: 1x / 2x / 16x the pool size give 38.5 / 38.6 / 38.6 % hits on the headline workload)
            while (tcap < 2 * (size_t)g.cap) tcap <<= 1; // <= 50 % load even when every node of the pool is a distinct position
            CREATE_CHECK(dmalloc(e, &B.tt, ns * tcap));
            B.tt_mask = (int32_t)(tcap - 1);
        }
    }
    CREATE_CHECK(dmalloc(e, &B.row_x, ns * rcap * F, false));
    CREATE_CHECK(dmalloc(e, &B.row_vis, ns * rcap * g.A, false));
    CREATE_CHECK(dmalloc(e, &B.row_meta, ns * rcap, false));
    CREATE_CHECK(dmalloc(e, &B.out_x, (size_t)B.max_out * F, false));
    CREATE_CHECK(dmalloc(e, &B.out_vis, (size_t)B.max_out * g.A, false));
    CREATE_CHECK(dmalloc(e, &B.out_meta, (size_t)B.max_out, false));
    CREATE_CHECK(dmalloc(e, &B.out_count, 4));
    CREATE_CHECK(dmalloc(e, &B.next_game, 2));
    CREATE_CHECK(dmalloc(e, &B.games_finished, 2));
    CREATE_CHECK(dmalloc(e, &B.moves_played, 2));
    CREATE_CHECK(dmalloc(e, &e->d_small, 16));
    CREATE_CHECK(dmalloc(e, &e->d_sum, 1));
    // log / sqrt tables evaluated with the HOST libm (what python's math.log/math.sqrt call),
    // mcts.py:92-94
    {
        long long tn = (long long)g.E * (cfg->mcts_num_read + 2) + 4;
        if (tn < 4096) tn = 4096;
        if (tn > (1 << 22)) tn = 1 << 22;
        sc.table_n = (int)tn;
        double *d_pbc, *d_sq;
        CREATE_CHECK(dmalloc(e, &d_pbc, tn, false));
        CREATE_CHECK(dmalloc(e, &d_sq, tn, false));
        B.pbc_table = d_pbc;
        B.sqrt_table = d_sq;
        CREATE_CHECK(upload_tables(e));
    }
    B.first_game = 0;
    B.last_game = 0;
    e->nns[0] = nn_create(g, e->n_slots * B.kmax, cfg->nn_precision, (cfg->debug_flags & DBAZ_DBG_NO_FALLBACK) != 0);
    e->nns[1] = nn_create(g, e->n_slots, cfg->nn_precision, (cfg->debug_flags & DBAZ_DBG_NO_FALLBACK) != 0);
    e->nn = e->nns[0];
    CREATE_HIP(hipStreamSynchronize(e->stream));
    // every slot starts as an idle empty board
    tree_launch_set_positions(e->stream, g, sc, B, e->n_slots, nullptr, nullptr);
    CREATE_HIP(hipStreamSynchronize(e->stream));
    e->late_join = (cfg->debug_flags & DBAZ_DBG_EARLY_JOIN) == 0;
    *out = e;
    return DBAZ_OK;
}

extern "C" void dbaz_destroy(dbaz_engine *e)
{
    if (!e) return;
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->stream2) (void)hipStreamSynchronize(e->stream2);
    for (int i = 0; i < 2; i++) if (e->nns[i]) nn_destroy(e->nns[i]);
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->stage) (void)hipFree(e->stage);
    if (e->replay_dev) (void)hipFree(e->replay_dev);
    for (int i = 0; i < DBAZ_MAX_DATASETS; i++) if (e->rds_all[i]) rds_destroy(e->rds_all[i]);
    for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->ev_t0) (void)hipEventDestroy(e->ev_t0);
    if (e->ev_t1) (void)hipEventDestroy(e->ev_t1);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->stream2) (void)hipStreamDestroy(e->stream2);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int dbaz_sync(dbaz_engine *e)
{
    if (!e) return DBAZ_EINVAL;
    USE_DEVICE(e);
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    return DBAZ_OK;
}

// ---------------------------------------------------------------- rules (G1-G6)
static int rules_call(dbaz_engine *e, int op, int32_t n, uint64_t *edges, int16_t *b2c2, int8_t *to_play,
                      int8_t *just_played, const int32_t *moves, int8_t *n_closed, int8_t *closed_lc, uint8_t *valid,
                      int8_t *result, int16_t *x)
{
    if (!e) return DBAZ_EINVAL;
    USE_DEVICE(e);
    if (n < 0) return set_error(e, DBAZ_EINVAL, "n < 0");
    if (n == 0) return DBAZ_OK;
    const Geo &g = e->g;
    const size_t N = n;
    const int F = 3 * g.HW;
    size_t need = Carver::need({N * 32, N * 4, N, N, N * 4, N, N * 4, N * g.A, N, N * F * 2});
    int r = ensure_stage(e, need);
    if (r) return r;
    Carver cv(e->stage);
    uint64_t *d_e = cv.take<uint64_t>(N * 4);
    int16_t *d_b = cv.take<int16_t>(N * 2);
    int8_t *d_tp = cv.take<int8_t>(N);
    int8_t *d_jp = cv.take<int8_t>(N);
    int32_t *d_mv = cv.take<int32_t>(N);
    int8_t *d_nc = cv.take<int8_t>(N);
    int8_t *d_cl = cv.take<int8_t>(N * 4);
    uint8_t *d_v = cv.take<uint8_t>(N * g.A);
    int8_t *d_r = cv.take<int8_t>(N);
    int16_t *d_x = cv.take<int16_t>(N * F);
    hipStream_t s = e->stream;
    if (op != 0) {
        if (edges) HIP_CHECK_RET(e, hipMemcpyAsync(d_e, edges, N * 32, hipMemcpyHostToDevice, s));
        if (b2c2) HIP_CHECK_RET(e, hipMemcpyAsync(d_b, b2c2, N * 4, hipMemcpyHostToDevice, s));
        if (to_play) HIP_CHECK_RET(e, hipMemcpyAsync(d_tp, to_play, N, hipMemcpyHostToDevice, s));
        if (just_played) HIP_CHECK_RET(e, hipMemcpyAsync(d_jp, just_played, N, hipMemcpyHostToDevice, s));
    }
    if (moves) HIP_CHECK_RET(e, hipMemcpyAsync(d_mv, moves, N * 4, hipMemcpyHostToDevice, s));
    tree_launch_rules(s, g, op, n, edges ? d_e : nullptr, b2c2 ? d_b : nullptr, to_play ? d_tp : nullptr,
                      just_played ? d_jp : nullptr, d_mv, d_nc, closed_lc ? d_cl : nullptr, d_v, d_r, d_x);
    HIP_CHECK_RET(e, hipGetLastError());
    if (op == 0 || op == 2) {
        HIP_CHECK_RET(e, hipMemcpyAsync(edges, d_e, N * 32, hipMemcpyDeviceToHost, s));
        HIP_CHECK_RET(e, hipMemcpyAsync(b2c2, d_b, N * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK_RET(e, hipMemcpyAsync(to_play, d_tp, N, hipMemcpyDeviceToHost, s));
        HIP_CHECK_RET(e, hipMemcpyAsync(just_played, d_jp, N, hipMemcpyDeviceToHost, s));
    }
    if (op == 2) {
        HIP_CHECK_RET(e, hipMemcpyAsync(n_closed, d_nc, N, hipMemcpyDeviceToHost, s));
        if (closed_lc) HIP_CHECK_RET(e, hipMemcpyAsync(closed_lc, d_cl, N * 4, hipMemcpyDeviceToHost, s));
    }
    if (op == 1) HIP_CHECK_RET(e, hipMemcpyAsync(valid, d_v, N * g.A, hipMemcpyDeviceToHost, s));
    if (op == 3) HIP_CHECK_RET(e, hipMemcpyAsync(result, d_r, N, hipMemcpyDeviceToHost, s));
    if (op == 4) HIP_CHECK_RET(e, hipMemcpyAsync(x, d_x, N * F * 2, hipMemcpyDeviceToHost, s));
    HIP_CHECK_RET(e, hipStreamSynchronize(s));
    return DBAZ_OK;
}

extern "C" int dbaz_rules_init(dbaz_engine *e, int32_t n, uint64_t *edges, int16_t *b2c2, int8_t *to_play, int8_t *just_played)
{
    if (!e || !edges || !b2c2 || !to_play || !just_played) return e ? set_error(e, DBAZ_EINVAL, "null buffer") : DBAZ_EINVAL;
    return rules_call(e, 0, n, edges, b2c2, to_play, just_played, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}
extern "C" int dbaz_rules_valid_moves(dbaz_engine *e, int32_t n, const uint64_t *edges, uint8_t *valid)
{
    if (!e || !edges || !valid) return e ? set_error(e, DBAZ_EINVAL, "null buffer") : DBAZ_EINVAL;
    return rules_call(e, 1, n, (uint64_t *)edges, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, valid, nullptr, nullptr);
}
extern "C" int dbaz_rules_play(dbaz_engine *e, int32_t n, uint64_t *edges, int16_t *b2c2, int8_t *to_play,
                               int8_t *just_played, const int32_t *moves, int8_t *n_closed, int8_t *closed_lc)
{
    if (!e || !edges || !b2c2 || !to_play || !just_played || !moves || !n_closed)
        return e ? set_error(e, DBAZ_EINVAL, "null buffer") : DBAZ_EINVAL;
    int r = rules_call(e, 2, n, edges, b2c2, to_play, just_played, moves, n_closed, closed_lc, nullptr, nullptr, nullptr);
    if (r) return r;
    for (int i = 0; i < n; i++)
        if (n_closed[i] < 0)
            return set_error(e, DBAZ_EILLEGAL, "Illegal move: %d (state %d)", moves[i], i);
    return DBAZ_OK;
}
extern "C" int dbaz_rules_result(dbaz_engine *e, int32_t n, const int16_t *b2c2, const int8_t *to_play, int8_t *result)
{
    if (!e || !b2c2 || !to_play || !result) return e ? set_error(e, DBAZ_EINVAL, "null buffer") : DBAZ_EINVAL;
    return rules_call(e, 3, n, nullptr, (int16_t *)b2c2, (int8_t *)to_play, nullptr, nullptr, nullptr, nullptr, nullptr, result, nullptr);
}
extern "C" int dbaz_rules_features(dbaz_engine *e, int32_t n, const uint64_t *edges, const int16_t *b2c2,
                                   const int8_t *to_play, int16_t *x)
{
    if (!e || !edges || !b2c2 || !to_play || !x) return e ? set_error(e, DBAZ_EINVAL, "null buffer") : DBAZ_EINVAL;
    return rules_call(e, 4, n, (uint64_t *)edges, (int16_t *)b2c2, (int8_t *)to_play, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, x);
}

// ---------------------------------------------------------------- network (N1-N3)
extern "C" int dbaz_nn_configure(dbaz_engine *e, int32_t kind, int32_t channels, int32_t blocks, int32_t head_channels, int32_t value_fc)
{
    if (!e) return DBAZ_EINVAL;
    std::string err;
    int r = nn_configure(e->nn, kind, channels, blocks, head_channels, value_fc, err);
    if (r) return set_error(e, r, "%s", err.c_str());
    return DBAZ_OK;
}
extern "C" int dbaz_nn_select_model(dbaz_engine *e, int32_t model)
{
    if (!e) return DBAZ_EINVAL;
    if (model < 0 || model > 1) return set_error(e, DBAZ_EINVAL, "model must be 0 or 1");
    e->cur_model = model;
    e->nn = e->nns[model];
    return DBAZ_OK;
}

extern "C" int dbaz_nn_set_tensor(dbaz_engine *e, const char *key, const float *data, int64_t numel)
{
    if (!e || !key || !data) return e ? set_error(e, DBAZ_EINVAL, "null argument") : DBAZ_EINVAL;
    std::string err;
    int r = nn_set_tensor(e->nn, key, data, numel, err);
    if (r) return set_error(e, r, "%s", err.c_str());
    return DBAZ_OK;
}
extern "C" int dbaz_nn_commit(dbaz_engine *e)
{
    if (!e) return DBAZ_EINVAL;
    USE_DEVICE(e);
    std::string err;
    int r = nn_commit(e->nn, e->stream, err);
    if (r) return set_error(e, r, "%s", err.c_str());
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    if (e->nn == e->nns[0]) {
        nn_round_info(e->nn, &e->eval_round, &e->eval_defer_max);
        // dbaz_config.eval_round: -1 switches the cut off, r > 0 forces a round size (and eval_defer_max the largest left-over
        // that is put off) so that small test runs go through the same path
        if (e->cfg.eval_round < 0) e->eval_round = 0;
        else if (e->cfg.eval_round > 0) {
            e->eval_round = e->cfg.eval_round;
            e->eval_defer_max = e->cfg.eval_defer_max > 0 ? e->cfg.eval_defer_max : e->cfg.eval_round - 1;
        }
        if (e->eval_defer_max <= 0) e->eval_round = 0;
    }
    return DBAZ_OK;
}
extern "C" int dbaz_nn_predict(dbaz_engine *e, int32_t n, const float *X, float *p, float *v)
{
    if (!e || !X || !p || !v || n < 0) return e ? set_error(e, DBAZ_EINVAL, "bad argument") : DBAZ_EINVAL;
    USE_DEVICE(e);
    if (!nn_ready(e->nn)) return set_error(e, DBAZ_ESTATE, "network weights not committed (dbaz_nn_commit)");
    const Geo &g = e->g;
    const int F = 3 * g.HW;
    const int chunk = e->n_slots;
    size_t need = Carver::need({(size_t)chunk * F * 4, (size_t)chunk * g.AS * 4, (size_t)chunk * 4, 16});
    int r = ensure_stage(e, need);
    if (r) return r;
    Carver cv(e->stage);
    float *d_x = cv.take<float>((size_t)chunk * F);
    float *d_p = cv.take<float>((size_t)chunk * g.AS);
    float *d_v = cv.take<float>(chunk);
    int32_t *d_n = cv.take<int32_t>(4);
    std::vector<float> hp((size_t)chunk * g.AS);
    for (int off = 0; off < n; off += chunk) {
        int m = std::min(chunk, n - off);
        HIP_CHECK_RET(e, hipMemcpyAsync(d_x, X + (size_t)off * F, (size_t)m * F * 4, hipMemcpyHostToDevice, e->stream));
        HIP_CHECK_RET(e, hipMemcpyAsync(d_n, &m, 4, hipMemcpyHostToDevice, e->stream));
        nn_forward(e->nn, e->stream, d_x, nullptr, d_n, m, d_p, d_v, g.AS, nullptr, nullptr);
        HIP_CHECK_RET(e, hipGetLastError());
        HIP_CHECK_RET(e, hipMemcpyAsync(hp.data(), d_p, (size_t)m * g.AS * 4, hipMemcpyDeviceToHost, e->stream));
        HIP_CHECK_RET(e, hipMemcpyAsync(v + off, d_v, (size_t)m * 4, hipMemcpyDeviceToHost, e->stream));
        HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
        for (int i = 0; i < m; i++)
            memcpy(p + (size_t)(off + i) * g.A, hp.data() + (size_t)i * g.AS, (size_t)g.A * 4);
    }
    if (nn_overflowed(e->nn))
        return set_error(e, DBAZ_EDEVICE, "nn_precision=1: an activation left the f16 range; use nn_precision=0 for this network");
    return DBAZ_OK;
}

// ---------------------------------------------------------------- search (M1-M9)
// Slot array reduced on the device (64 bytes cross PCIe instead of n_slots * sizeof(Slot))
static int device_summary(dbaz_engine *e, SlotSummary *out)
{
    SlotSummary init;
    memset(&init, 0, sizeof(init));
    init.first_error_slot = 0x7fffffff;
    HIP_CHECK_RET(e, hipMemcpyAsync(e->d_sum, &init, sizeof(init), hipMemcpyHostToDevice, e->stream));
    tree_launch_slot_summary(e->stream, e->B, e->n_slots, e->d_sum);
    HIP_CHECK_RET(e, hipGetLastError());
    HIP_CHECK_RET(e, hipMemcpyAsync(out, e->d_sum, sizeof(*out), hipMemcpyDeviceToHost, e->stream));
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    return DBAZ_OK;
}

static int report_slot_error(dbaz_engine *e, const SlotSummary &s)
{
    if (s.error == 0) return DBAZ_OK;
    const int i = s.first_error_slot;
    if (s.first_error_code == DBAZ_EPOOL)
        return set_error(e, DBAZ_EPOOL, "slot %d: node pool exhausted (%d nodes); raise nodes_per_slot", i, e->g.cap);
    if (s.first_error_code == DBAZ_EILLEGAL) return set_error(e, DBAZ_EILLEGAL, "Illegal move (slot %d)", i);
    return set_error(e, DBAZ_ESTATE, "slot %d stopped with error %d", i, s.first_error_code);
}

static int check_slot_errors(dbaz_engine *e)
{
    SlotSummary s;
    int r = device_summary(e, &s);
    if (r) return r;
    return report_slot_error(e, s);
}

extern "C" int dbaz_set_positions(dbaz_engine *e, const int16_t *moves, const int32_t *offsets)
{
    if (!e) return DBAZ_EINVAL;
    USE_DEVICE(e);
    if ((moves == nullptr) != (offsets == nullptr)) return set_error(e, DBAZ_EINVAL, "moves and offsets must both be given");
    e->selfplay = false;
    e->search_open = false;
    const int16_t *d_m = nullptr;
    const int32_t *d_o = nullptr;
    if (offsets) {
        size_t nm = offsets[e->n_slots];
        int r = ensure_stage(e, Carver::need({nm * 2 + 16, (size_t)(e->n_slots + 1) * 4}));
        if (r) return r;
        Carver cv(e->stage);
        int16_t *pm = cv.take<int16_t>(nm + 8);
        int32_t *po = cv.take<int32_t>(e->n_slots + 1);
        if (nm) HIP_CHECK_RET(e, hipMemcpyAsync(pm, moves, nm * 2, hipMemcpyHostToDevice, e->stream));
        HIP_CHECK_RET(e, hipMemcpyAsync(po, offsets, (size_t)(e->n_slots + 1) * 4, hipMemcpyHostToDevice, e->stream));
        d_m = pm; d_o = po;
    }
    tree_launch_set_positions(e->stream, e->g, e->sc, e->B, e->n_slots, d_m, d_o);
    HIP_CHECK_RET(e, hipGetLastError());
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    return check_slot_errors(e);
}

static hipEvent_t next_event(dbaz_engine *e)
{
    if (e->ev_used == e->ev_pool.size()) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        e->ev_pool.push_back(ev);
    }
    return e->ev_pool[e->ev_used++];
}

// one WAVE of up to K simulations for every searching tree (max_pending_evals = K > 1): K sequential selections with
// virtual loss per tree, one batched evaluation of all their leaves, expand + backup in selection order
// with_driver (self-play with dbaz_config.selfplay_pending: every search of a game runs in these waves, the reference's
// self_play.py:27-30 with max_async_searches = K): the driver pass -- move choice, row emission, re-root, next search's root
// preparation -- runs first, on the same stream; a slot it starts takes its first wave in this very step
static int sim_wave(dbaz_engine *e, bool with_driver)
{
    hipStream_t s = e->stream;
    const bool use_nn = e->sc.evaluator == DBAZ_EVAL_RESNET || e->sc.evaluator == DBAZ_EVAL_SIMPLENN;
    e->sc.step = (int)(e->steps & 0x3FFFFFFF) + 1;
    e->sc.driver_concurrent = 0;
    e->sc.eval_round = 0;
    if (with_driver) tree_launch_advance_auto(s, e->g, e->sc, e->B, e->n_slots);
    tree_launch_select_multi(s, e->g, e->sc, e->B, e->n_slots);
    if (use_nn) {
        nn_forward(e->nns[0], s, e->B.feat_m, e->B.list_m, e->B.n_eval, e->n_slots * e->B.kmax, e->B.evalP_m, e->B.evalV_m, e->g.AS, nullptr, nullptr);
        e->nn_launches++;
    }
    tree_launch_expand_backup_multi(s, e->g, e->sc, e->B, e->n_slots);
    e->steps++;
    HIP_CHECK_RET(e, hipGetLastError());
    return DBAZ_OK;
}

// one simulation step for every searching slot
static int sim_step(dbaz_engine *e, bool with_driver)
{
    if (e->B.kmax > 1 && (!with_driver || e->cfg.selfplay_pending)) return sim_wave(e, with_driver);
    hipStream_t s = e->stream;
    auto is_nn = [](int ev) { return ev == DBAZ_EVAL_RESNET || ev == DBAZ_EVAL_SIMPLENN; };
    const bool use_nn = is_nn(e->sc.evaluator);
    const bool use_nn2 = e->sc.match_play && is_nn(e->sc.evaluator2);
    e->sc.step = (int)(e->steps & 0x3FFFFFFF) + 1; // never 0 (a fresh Slot's stamp)
    e->sc.driver_concurrent = with_driver ? 1 : 0;
    if (with_driver) {
        // The driver pass (move choice, O(1) re-root, row emission, game turnover, next search's root preparation) only
        // touches slots whose reads are done (PH_READY / PH_EMIT); k_select only touches slots that are searching and
        // skips the ones this very pass starts (stamp).  So the pass runs on a second stream NEXT TO k_select -- both are
        // short, latency-bound kernels that leave most of the chip idle.  It is joined behind the network launch (see below):
        // almost always it has finished by the time k_select has, and when it has not (small boards: a move ends somewhere
        // every step and k_select is short) its few single-wave workgroups cost the network less than waiting for them.
        HIP_CHECK_RET(e, hipEventRecord(e->ev_fork, s));
        HIP_CHECK_RET(e, hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
        tree_launch_advance_auto(e->stream2, e->g, e->sc, e->B, e->n_slots);
        HIP_CHECK_RET(e, hipEventRecord(e->ev_join, e->stream2));
    }
    // full rounds only (tree.hip, above k_select): self-play stepping, one network, one leaf per slot
    e->sc.eval_round = 0;
    if (with_driver && use_nn && !use_nn2 && !e->sc.match_play && e->B.kmax <= 1 && e->eval_round > 0 && e->n_slots > e->eval_round) {
        e->sc.eval_round = e->eval_round;
        e->sc.eval_defer_max = e->eval_defer_max;
    }
    tree_launch_select(s, e->g, e->sc, e->B, e->n_slots); // (the lists' counters were zeroed by the previous k_expand_backup)
    const bool late_join = with_driver && e->late_join;
    if (with_driver && !late_join) HIP_CHECK_RET(e, hipStreamWaitEvent(s, e->ev_join, 0));
    if (use_nn) {
        hipEvent_t a = nullptr, b = nullptr;
        if (e->timing) { a = next_event(e); b = next_event(e); }
        nn_forward(e->nns[0], s, e->B.feat, e->B.eval_list, e->B.n_eval, e->n_slots, e->B.evalP, e->B.evalV, e->g.AS, a, b,
                   e->sc.eval_round, e->sc.eval_defer_max, e->B.n_eval + 2);
        e->nn_launches++;
    }
    if (use_nn2) {
        nn_forward(e->nns[1], s, e->B.feat, e->B.eval_list2, e->B.n_eval + 1, e->n_slots, e->B.evalP, e->B.evalV, e->g.AS, nullptr, nullptr);
        e->nn_launches++;
    }
    // the driver pass may run on under the network: nothing there reads what it writes (leaf lists and features come from
    // k_select, which skipped the slots the pass starts); k_expand_backup does (it clears the pass's slot list).
    // 3x3: 11.9 -> 13.2 M expansions/s, 6x6: +1.5 %, 9x9: +2.8 % against joining in front of the network
    if (late_join) HIP_CHECK_RET(e, hipStreamWaitEvent(s, e->ev_join, 0));
    tree_launch_expand_backup(s, e->g, e->sc, e->B, e->n_slots);
    e->steps++;
    HIP_CHECK_RET(e, hipGetLastError());
    return DBAZ_OK;
}

static int upload_search_inputs(dbaz_engine *e, const int32_t *num_reads, const double *noise, const int32_t **d_reads)
{
    const Geo &g = e->g;
    *d_reads = nullptr;
    int r = ensure_stage(e, Carver::need({(size_t)e->n_slots * 4, (size_t)e->n_slots * g.AS * 8}));
    if (r) return r;
    Carver cv(e->stage);
    int32_t *pr = cv.take<int32_t>(e->n_slots);
    if (num_reads) {
        HIP_CHECK_RET(e, hipMemcpyAsync(pr, num_reads, (size_t)e->n_slots * 4, hipMemcpyHostToDevice, e->stream));
        *d_reads = pr;
    }
    if (noise && e->sc.alpha > 0) {
        std::vector<double> padded((size_t)e->n_slots * g.AS, 0.0);
        for (int i = 0; i < e->n_slots; i++)
            memcpy(padded.data() + (size_t)i * g.AS, noise + (size_t)i * g.A, (size_t)g.A * 8);
        HIP_CHECK_RET(e, hipMemcpyAsync(e->B.noise_in, padded.data(), padded.size() * 8, hipMemcpyHostToDevice, e->stream));
        std::vector<int32_t> ones(e->n_slots, 1);
        HIP_CHECK_RET(e, hipMemcpyAsync(e->B.noise_valid, ones.data(), (size_t)e->n_slots * 4, hipMemcpyHostToDevice, e->stream));
        HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    } else {
        HIP_CHECK_RET(e, hipMemsetAsync(e->B.noise_valid, 0, (size_t)e->n_slots * 4, e->stream));
    }
    return DBAZ_OK;
}

static int count_phases(dbaz_engine *e, int32_t out[4])
{
    HIP_CHECK_RET(e, hipMemsetAsync(e->d_small, 0, 16, e->stream));
    tree_launch_count_active(e->stream, e->B, e->n_slots, e->d_small);
    HIP_CHECK_RET(e, hipMemcpyAsync(out, e->d_small, 16, hipMemcpyDeviceToHost, e->stream));
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    return DBAZ_OK;
}

extern "C" int dbaz_search_begin(dbaz_engine *e, const int32_t *num_reads, const double *noise)
{
    if (!e) return DBAZ_EINVAL;
    USE_DEVICE(e);
    if (e->selfplay) return set_error(e, DBAZ_ESTATE, "self-play in progress; call dbaz_set_positions first");
    const int32_t *d_reads;
    int r = upload_search_inputs(e, num_reads, noise, &d_reads);
    if (r) return r;
    e->sc.step = 0;
    tree_launch_search_begin(e->stream, e->g, e->sc, e->B, e->n_slots, d_reads);
    HIP_CHECK_RET(e, hipGetLastError());
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    int mx = e->sc.mcts_num_read;
    if (num_reads) { mx = 0; for (int i = 0; i < e->n_slots; i++) mx = std::max(mx, num_reads[i]); }
    e->search_iters_left = e->B.kmax > 1 ? (mx + e->sc.pending - 1) / e->sc.pending + 2 : mx + 1;
    e->search_open = true;
    return DBAZ_OK;
}

extern "C" int dbaz_set_pending(dbaz_engine *e, int32_t k, int32_t virtual_visits)
{
    if (!e) return DBAZ_EINVAL;
    if (e->B.kmax <= 1) return set_error(e, DBAZ_ESTATE, "the handle was created with max_pending_evals <= 1");
    if (k < 1 || k > e->B.kmax) return set_error(e, DBAZ_EINVAL, "pending evaluations must be in 1..%d", e->B.kmax);
    e->sc.pending = k;
    e->sc.virtual_visits = virtual_visits ? 1 : 0;
    return DBAZ_OK;
}

extern "C" int dbaz_search_timed(dbaz_engine *e, const int32_t *num_reads, const double *noise, double time_limit_s)
{
    if (!e) return DBAZ_EINVAL;
    if (e->sc.evaluator == DBAZ_EVAL_EXTERNAL)
        return set_error(e, DBAZ_ESTATE, "external evaluator: use dbaz_search_begin/dbaz_select/dbaz_expand_backup");
    if ((e->sc.evaluator == DBAZ_EVAL_RESNET || e->sc.evaluator == DBAZ_EVAL_SIMPLENN) && !nn_ready(e->nns[0]))
        return set_error(e, DBAZ_ESTATE, "network weights not committed (dbaz_nn_commit)");
    // end_time = time.time() + (time_limit or 120), mcts.py:201-203
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(time_limit_s > 0 ? time_limit_s : 120.0);
    int r = dbaz_search_begin(e, num_reads, noise);
    if (r) return r;
    // the root expansion is not subject to the clock (mcts.py:207-208); afterwards the clock is looked at between waves
    // (one wave = K reads per tree; with K = 1 every 8 steps)
    const int per_check = e->B.kmax > 1 ? 1 : 8;
    bool first = true;
    for (;;) {
        for (int i = 0; i < (first ? 1 : per_check); i++) {
            r = sim_step(e, false);
            if (r) return r;
        }
        first = false;
        int32_t c[4];
        r = count_phases(e, c);
        if (r) return r;
        if (c[2] > 0) { e->search_open = false; return check_slot_errors(e); }
        if (c[0] == 0) break;
        if (std::chrono::steady_clock::now() > t_end) { // mcts.py:232-233: the remaining reads are not started
            tree_launch_stop_search(e->stream, e->B, e->n_slots);
            HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
            break;
        }
    }
    e->search_open = false;
    return DBAZ_OK;
}

extern "C" int dbaz_search(dbaz_engine *e, const int32_t *num_reads, const double *noise)
{
    if (!e) return DBAZ_EINVAL;
    if (e->sc.evaluator == DBAZ_EVAL_EXTERNAL)
        return set_error(e, DBAZ_ESTATE, "external evaluator: use dbaz_search_begin/dbaz_select/dbaz_expand_backup");
    if ((e->sc.evaluator == DBAZ_EVAL_RESNET || e->sc.evaluator == DBAZ_EVAL_SIMPLENN) && !nn_ready(e->nns[0]))
        return set_error(e, DBAZ_ESTATE, "network weights not committed (dbaz_nn_commit)");
    int r = dbaz_search_begin(e, num_reads, noise);
    if (r) return r;
    for (;;) {
        for (int i = 0; i < e->search_iters_left; i++) {
            r = sim_step(e, false);
            if (r) return r;
        }
        int32_t c[4];
        r = count_phases(e, c);
        if (r) return r;
        if (c[2] > 0) { e->search_open = false; return check_slot_errors(e); }
        if (c[0] == 0) break;
        e->search_iters_left = 8;
    }
    e->search_open = false;
    return DBAZ_OK;
}

extern "C" int dbaz_select(dbaz_engine *e, int32_t *n_active, int16_t *leaf_x, uint8_t *need_eval)
{
    if (!e || !n_active || !leaf_x || !need_eval) return e ? set_error(e, DBAZ_EINVAL, "null argument") : DBAZ_EINVAL;
    USE_DEVICE(e);
    if (!e->search_open) return set_error(e, DBAZ_ESTATE, "dbaz_search_begin not called");
    const Geo &g = e->g;
    const size_t F = 3 * g.HW, ns = e->n_slots;
    int r = ensure_stage(e, Carver::need({ns * F * 2, ns, 16}));
    if (r) return r;
    Carver cv(e->stage);
    int16_t *d_x = cv.take<int16_t>(ns * F);
    uint8_t *d_ne = cv.take<uint8_t>(ns);
    int32_t *d_na = cv.take<int32_t>(4);
    e->sc.step = (e->sc.step & 0x3FFFFFFF) + 1;
    e->sc.driver_concurrent = 0;
    tree_launch_select(e->stream, g, e->sc, e->B, e->n_slots);
    HIP_CHECK_RET(e, hipMemsetAsync(d_na, 0, 16, e->stream));
    tree_launch_get_leaves(e->stream, g, e->B, e->n_slots, d_x, d_ne, d_na);
    HIP_CHECK_RET(e, hipGetLastError());
    HIP_CHECK_RET(e, hipMemcpyAsync(leaf_x, d_x, ns * F * 2, hipMemcpyDeviceToHost, e->stream));
    HIP_CHECK_RET(e, hipMemcpyAsync(need_eval, d_ne, ns, hipMemcpyDeviceToHost, e->stream));
    HIP_CHECK_RET(e, hipMemcpyAsync(n_active, d_na, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    int32_t c[4];
    r = count_phases(e, c);
    if (r) return r;
    if (c[2] > 0) return check_slot_errors(e);
    return DBAZ_OK;
}

extern "C" int dbaz_expand_backup(dbaz_engine *e, const float *p, const float *v)
{
    if (!e || !p || !v) return e ? set_error(e, DBAZ_EINVAL, "null argument") : DBAZ_EINVAL;
    USE_DEVICE(e);
    if (!e->search_open) return set_error(e, DBAZ_ESTATE, "dbaz_search_begin not called");
    const Geo &g = e->g;
    std::vector<float> padded((size_t)e->n_slots * g.AS, 0.0f);
    for (int i = 0; i < e->n_slots; i++)
        memcpy(padded.data() + (size_t)i * g.AS, p + (size_t)i * g.A, (size_t)g.A * 4);
    HIP_CHECK_RET(e, hipMemcpyAsync(e->B.evalP, padded.data(), padded.size() * 4, hipMemcpyHostToDevice, e->stream));
    HIP_CHECK_RET(e, hipMemcpyAsync(e->B.evalV, v, (size_t)e->n_slots * 4, hipMemcpyHostToDevice, e->stream));
    tree_launch_expand_backup(e->stream, g, e->sc, e->B, e->n_slots);
    e->steps++;
    HIP_CHECK_RET(e, hipGetLastError());
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    return DBAZ_OK;
}

extern "C" int dbaz_get_roots(dbaz_engine *e, double *priors, float *total_value, int32_t *visits, int32_t *changed,
                              int32_t *stats, float *q_value, float *root_tv, int32_t *root_nv)
{
    if (!e) return DBAZ_EINVAL;
    USE_DEVICE(e);
    const Geo &g = e->g;
    const size_t ns = e->n_slots, A = g.A;
    int r = ensure_stage(e, Carver::need({ns * A * 8, ns * A * 4, ns * A * 4, ns * A * 4, ns * 12, ns * 4, ns * 4, ns * 4}));
    if (r) return r;
    Carver cv(e->stage);
    double *d_p = cv.take<double>(ns * A);
    float *d_tv = cv.take<float>(ns * A);
    int32_t *d_nv = cv.take<int32_t>(ns * A);
    int32_t *d_ch = cv.take<int32_t>(ns * A);
    int32_t *d_st = cv.take<int32_t>(ns * 3);
    float *d_q = cv.take<float>(ns);
    float *d_rtv = cv.take<float>(ns);
    int32_t *d_rnv = cv.take<int32_t>(ns);
    tree_launch_get_roots(e->stream, g, e->B, e->n_slots, d_p, d_tv, d_nv, d_ch, d_st, d_q, d_rtv, d_rnv, nullptr,
                          nullptr, nullptr, nullptr, nullptr, nullptr);
    HIP_CHECK_RET(e, hipGetLastError());
    hipStream_t s = e->stream;
    if (priors) HIP_CHECK_RET(e, hipMemcpyAsync(priors, d_p, ns * A * 8, hipMemcpyDeviceToHost, s));
    if (total_value) HIP_CHECK_RET(e, hipMemcpyAsync(total_value, d_tv, ns * A * 4, hipMemcpyDeviceToHost, s));
    if (visits) HIP_CHECK_RET(e, hipMemcpyAsync(visits, d_nv, ns * A * 4, hipMemcpyDeviceToHost, s));
    if (changed) HIP_CHECK_RET(e, hipMemcpyAsync(changed, d_ch, ns * A * 4, hipMemcpyDeviceToHost, s));
    if (stats) HIP_CHECK_RET(e, hipMemcpyAsync(stats, d_st, ns * 12, hipMemcpyDeviceToHost, s));
    if (q_value) HIP_CHECK_RET(e, hipMemcpyAsync(q_value, d_q, ns * 4, hipMemcpyDeviceToHost, s));
    if (root_tv) HIP_CHECK_RET(e, hipMemcpyAsync(root_tv, d_rtv, ns * 4, hipMemcpyDeviceToHost, s));
    if (root_nv) HIP_CHECK_RET(e, hipMemcpyAsync(root_nv, d_rnv, ns * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK_RET(e, hipStreamSynchronize(s));
    return DBAZ_OK;
}

extern "C" int dbaz_get_root_states(dbaz_engine *e, uint64_t *edges, int16_t *b2c2, int8_t *to_play,
                                    int8_t *just_played, int8_t *result, int8_t *expanded)
{
    if (!e) return DBAZ_EINVAL;
    USE_DEVICE(e);
    const size_t ns = e->n_slots;
    int r = ensure_stage(e, Carver::need({ns * 32, ns * 4, ns, ns, ns, ns}));
    if (r) return r;
    Carver cv(e->stage);
    uint64_t *d_e = cv.take<uint64_t>(ns * 4);
    int16_t *d_b = cv.take<int16_t>(ns * 2);
    int8_t *d_tp = cv.take<int8_t>(ns), *d_jp = cv.take<int8_t>(ns), *d_r = cv.take<int8_t>(ns), *d_x = cv.take<int8_t>(ns);
    tree_launch_get_roots(e->stream, e->g, e->B, e->n_slots, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                          nullptr, nullptr, d_e, d_b, d_tp, d_jp, d_r, d_x);
    HIP_CHECK_RET(e, hipGetLastError());
    hipStream_t s = e->stream;
    if (edges) HIP_CHECK_RET(e, hipMemcpyAsync(edges, d_e, ns * 32, hipMemcpyDeviceToHost, s));
    if (b2c2) HIP_CHECK_RET(e, hipMemcpyAsync(b2c2, d_b, ns * 4, hipMemcpyDeviceToHost, s));
    if (to_play) HIP_CHECK_RET(e, hipMemcpyAsync(to_play, d_tp, ns, hipMemcpyDeviceToHost, s));
    if (just_played) HIP_CHECK_RET(e, hipMemcpyAsync(just_played, d_jp, ns, hipMemcpyDeviceToHost, s));
    if (result) HIP_CHECK_RET(e, hipMemcpyAsync(result, d_r, ns, hipMemcpyDeviceToHost, s));
    if (expanded) HIP_CHECK_RET(e, hipMemcpyAsync(expanded, d_x, ns, hipMemcpyDeviceToHost, s));
    HIP_CHECK_RET(e, hipStreamSynchronize(s));
    return DBAZ_OK;
}

extern "C" int dbaz_advance(dbaz_engine *e, const int32_t *moves, int32_t reuse_tree)
{
    if (!e || !moves) return e ? set_error(e, DBAZ_EINVAL, "null argument") : DBAZ_EINVAL;
    USE_DEVICE(e);
    if (e->selfplay) return set_error(e, DBAZ_ESTATE, "self-play in progress");
    int r = ensure_stage(e, Carver::need({(size_t)e->n_slots * 4}));
    if (r) return r;
    Carver cv(e->stage);
    int32_t *d_m = cv.take<int32_t>(e->n_slots);
    HIP_CHECK_RET(e, hipMemcpyAsync(d_m, moves, (size_t)e->n_slots * 4, hipMemcpyHostToDevice, e->stream));
    tree_launch_advance_manual(e->stream, e->g, e->sc, e->B, e->n_slots, d_m, reuse_tree);
    HIP_CHECK_RET(e, hipGetLastError());
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    int rr = check_slot_errors(e);
    if (rr == DBAZ_EILLEGAL || rr == DBAZ_EPOOL) {
        // leave the other slots usable: the failing slot stays in PH_ERROR until dbaz_set_positions
    }
    return rr;
}

// ---------------------------------------------------------------- self-play driver (D1-D3)
extern "C" int dbaz_selfplay_script(dbaz_engine *e, int64_t game_idx, const int16_t *moves, int32_t n_moves, const double *noise)
{
    if (!e || !moves || n_moves < 0) return e ? set_error(e, DBAZ_EINVAL, "bad argument") : DBAZ_EINVAL;
    const Geo &g = e->g;
    const int rc = g.E + 1;
    if (n_moves > rc) return set_error(e, DBAZ_EINVAL, "script longer than a game (%d > %d)", n_moves, rc);
    if (e->script_first < 0) e->script_first = game_idx;
    int64_t rel = game_idx - e->script_first;
    if (rel < 0 || rel > (1 << 20)) return set_error(e, DBAZ_EINVAL, "scripts must be given in increasing game order");
    if (rel >= e->n_script) {
        e->n_script = (int)rel + 1;
        e->script_moves.resize((size_t)e->n_script * rc, -1);
        e->script_noise.resize((size_t)e->n_script * rc * g.A, 0.0);
        e->script_has_noise.resize(e->n_script, 0);
    }
    for (int i = 0; i < rc; i++) e->script_moves[(size_t)rel * rc + i] = i < n_moves ? moves[i] : (int16_t)-1;
    if (noise) {
        memcpy(e->script_noise.data() + (size_t)rel * rc * g.A, noise, (size_t)n_moves * g.A * 8);
        e->script_has_noise[rel] = 1;
    }
    return DBAZ_OK;
}

extern "C" int dbaz_selfplay_fastforward(dbaz_engine *e, const int32_t *plies)
{
    if (!e || !plies) return e ? set_error(e, DBAZ_EINVAL, "null argument") : DBAZ_EINVAL;
    e->ff_plies.assign(plies, plies + e->n_slots);
    return DBAZ_OK;
}

extern "C" int dbaz_selfplay_stagger(dbaz_engine *e, const int32_t *first_reads)
{
    if (!e || !first_reads) return e ? set_error(e, DBAZ_EINVAL, "null argument") : DBAZ_EINVAL;
    for (int i = 0; i < e->n_slots; i++)
        if (first_reads[i] < 0) return set_error(e, DBAZ_EINVAL, "first_reads[%d] < 0", i);
    e->ff_reads.assign(first_reads, first_reads + e->n_slots);
    return DBAZ_OK;
}

extern "C" int dbaz_selfplay_quickplay(dbaz_engine *e, const int32_t *plies, int32_t reads)
{
    if (!e || !plies) return e ? set_error(e, DBAZ_EINVAL, "null argument") : DBAZ_EINVAL;
    if (reads < 1) return set_error(e, DBAZ_EINVAL, "reads must be >= 1");
    for (int i = 0; i < e->n_slots; i++)
        if (plies[i] < 0) return set_error(e, DBAZ_EINVAL, "plies[%d] < 0", i);
    e->quick_plies.assign(plies, plies + e->n_slots);
    e->sc.quick_reads = reads;
    return DBAZ_OK;
}

extern "C" int dbaz_selfplay_start(dbaz_engine *e, int64_t n_games, int64_t first_game_idx)
{
    if (!e || n_games < 0) return e ? set_error(e, DBAZ_EINVAL, "bad argument") : DBAZ_EINVAL;
    USE_DEVICE(e);
    if (e->sc.evaluator == DBAZ_EVAL_EXTERNAL) return set_error(e, DBAZ_ESTATE, "self-play needs a device evaluator");
    if ((e->sc.evaluator == DBAZ_EVAL_RESNET || e->sc.evaluator == DBAZ_EVAL_SIMPLENN) && !nn_ready(e->nns[0]))
        return set_error(e, DBAZ_ESTATE, "network weights not committed (dbaz_nn_commit)");
    if (e->sc.match_play && (e->sc.evaluator2 == DBAZ_EVAL_RESNET || e->sc.evaluator2 == DBAZ_EVAL_SIMPLENN) && !nn_ready(e->nns[1]))
        return set_error(e, DBAZ_ESTATE, "network weights of model 1 not committed (dbaz_nn_select_model(1), dbaz_nn_commit)");
    TreeBufs &B = e->B;
    hipStream_t s = e->stream;
    HIP_CHECK_RET(e, hipStreamSynchronize(s));
    B.first_game = first_game_idx;
    B.last_game = first_game_idx + n_games;
    long long next = first_game_idx + std::min<int64_t>(n_games, e->n_slots);
    HIP_CHECK_RET(e, hipMemcpy(B.next_game, &next, 8, hipMemcpyHostToDevice));
    HIP_CHECK_RET(e, hipMemset(B.games_finished, 0, 8));
    HIP_CHECK_RET(e, hipMemset(B.moves_played, 0, 8));
    HIP_CHECK_RET(e, hipMemset(B.out_count, 0, 4));
    HIP_CHECK_RET(e, hipMemset(B.noise_valid, 0, (size_t)e->n_slots * 4));
    // scripts
    B.script_moves = nullptr; B.script_noise = nullptr; B.script_has_noise = nullptr; B.n_script = 0;
    if (e->n_script > 0) {
        if (e->script_first != first_game_idx) return set_error(e, DBAZ_EINVAL, "scripts start at game %lld, self-play at %lld", (long long)e->script_first, (long long)first_game_idx);
        int16_t *dm; double *dn; uint8_t *dh;
        int r;
        if ((r = dmalloc(e, &dm, e->script_moves.size(), false))) return r;
        if ((r = dmalloc(e, &dn, e->script_noise.size(), false))) return r;
        if ((r = dmalloc(e, &dh, e->script_has_noise.size(), false))) return r;
        HIP_CHECK_RET(e, hipMemcpy(dm, e->script_moves.data(), e->script_moves.size() * 2, hipMemcpyHostToDevice));
        HIP_CHECK_RET(e, hipMemcpy(dn, e->script_noise.data(), e->script_noise.size() * 8, hipMemcpyHostToDevice));
        HIP_CHECK_RET(e, hipMemcpy(dh, e->script_has_noise.data(), e->script_has_noise.size(), hipMemcpyHostToDevice));
        B.script_moves = dm; B.script_noise = dn; B.script_has_noise = dh; B.n_script = e->n_script;
        e->n_script = 0; e->script_first = -1;
        e->script_moves.clear(); e->script_noise.clear(); e->script_has_noise.clear();
    }
    // fast-forward plies -> Slot.ff_plies
    {
        std::vector<Slot> hs(e->n_slots);
        HIP_CHECK_RET(e, hipMemcpy(hs.data(), B.slots, sizeof(Slot) * e->n_slots, hipMemcpyDeviceToHost));
        for (int i = 0; i < e->n_slots; i++) {
            hs[i].ff_plies = e->ff_plies.empty() ? 0 : e->ff_plies[i];
            hs[i].ff_reads = e->ff_reads.empty() ? 0 : e->ff_reads[i];
            hs[i].quick_until = e->quick_plies.empty() ? 0 : e->quick_plies[i];
        }
        e->ff_reads.clear();
        e->quick_plies.clear();
        HIP_CHECK_RET(e, hipMemcpy(B.slots, hs.data(), sizeof(Slot) * e->n_slots, hipMemcpyHostToDevice));
        e->ff_plies.clear();
    }
    e->sc.step = 0; // searches started here carry stamp 0, which no step ever has
    tree_launch_selfplay_start(s, e->g, e->sc, B, e->n_slots);
    HIP_CHECK_RET(e, hipGetLastError());
    HIP_CHECK_RET(e, hipStreamSynchronize(s));
    e->selfplay = true;
    e->search_open = false;
    e->steps = 0;
    return DBAZ_OK;
}

extern "C" int dbaz_step(dbaz_engine *e, int32_t k)
{
    if (!e || k < 0) return e ? set_error(e, DBAZ_EINVAL, "bad argument") : DBAZ_EINVAL;
    USE_DEVICE(e);
    if (!e->selfplay) return set_error(e, DBAZ_ESTATE, "dbaz_selfplay_start not called");
    for (int i = 0; i < k; i++) {
        int r = sim_step(e, true);
        if (r) return r;
    }
    return DBAZ_OK;
}

static int slot_summary(dbaz_engine *e, dbaz_counters *c, SlotSummary *raw = nullptr)
{
    SlotSummary s;
    int r = device_summary(e, &s);
    if (r) return r;
    c->expansions = (int64_t)s.n_search;
    c->nn_evals = (int64_t)s.n_eval;
    c->cache_hits = (int64_t)s.n_hit;
    c->pool_resets = (int64_t)s.n_reset;
    c->terminal_leaves = (int64_t)s.n_term;
    c->sum_path = (int64_t)s.sum_path;
    c->active_slots = s.active;
    c->error_slots = s.error;
    c->blocked_slots = s.blocked;
    c->pool_high_water = s.pool_high;
    if (raw) *raw = s;
    return DBAZ_OK;
}

extern "C" int dbaz_get_counters(dbaz_engine *e, dbaz_counters *out)
{
    if (!e || !out) return DBAZ_EINVAL;
    USE_DEVICE(e);
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    memset(out, 0, sizeof(*out));
    int r = slot_summary(e, out);
    if (r) return r;
    long long gf = 0, mp = 0;
    int32_t oc = 0;
    HIP_CHECK_RET(e, hipMemcpy(&gf, e->B.games_finished, 8, hipMemcpyDeviceToHost));
    HIP_CHECK_RET(e, hipMemcpy(&mp, e->B.moves_played, 8, hipMemcpyDeviceToHost));
    HIP_CHECK_RET(e, hipMemcpy(&oc, e->B.out_count, 4, hipMemcpyDeviceToHost));
    out->steps = e->steps;
    out->games_finished = gf;
    out->moves_played = mp;
    out->rows_ready = oc;
    out->ms_total = e->ms_total;
    out->ms_nn = e->ms_nn_tower;
    out->ms_nn_tower = e->ms_nn_tower;
    out->ms_tree = e->ms_total - e->ms_nn_tower;
    out->nn_launches = e->nn_launches;
    out->f32_fallback_evals = (int32_t)std::min<long long>(nn_fallback_evals(e->nns[0]) + nn_fallback_evals(e->nns[1]), 0x7fffffff);
    if (nn_overflowed(e->nns[0]) || nn_overflowed(e->nns[1]))
        return set_error(e, DBAZ_EDEVICE, "nn_precision=1: an activation left the f16 range; use nn_precision=0 for this network");
    return DBAZ_OK;
}

extern "C" int dbaz_run(dbaz_engine *e, int64_t max_steps)
{
    if (!e) return DBAZ_EINVAL;
    if (!e->selfplay) return set_error(e, DBAZ_ESTATE, "dbaz_selfplay_start not called");
    int64_t done = 0, last_exp = -1, last_moves = -1;
    int stalls = 0;
    for (;;) {
        dbaz_counters c;
        int r = dbaz_get_counters(e, &c);
        if (r) return r;
        if (c.error_slots > 0) return check_slot_errors(e);
        if (c.active_slots == 0) return DBAZ_OK;
        // output buffer full: the caller must fetch.  Only after stepping in THIS call, so that a
        // call made right after a drain lets the blocked slots emit.
        if (c.blocked_slots == c.active_slots && done > 0) return DBAZ_OK;
        if (c.expansions == last_exp && c.moves_played == last_moves) {
            if (++stalls >= 4) return set_error(e, DBAZ_ESTATE, "self-play made no progress for %d polls", stalls);
        } else stalls = 0;
        last_exp = c.expansions; last_moves = c.moves_played;
        if (max_steps > 0 && done >= max_steps) return DBAZ_OK;
        int chunk = 64;
        if (max_steps > 0) chunk = (int)std::min<int64_t>(chunk, max_steps - done);
        r = dbaz_step(e, chunk);
        if (r) return r;
        done += chunk;
    }
}

extern "C" int dbaz_timing_begin(dbaz_engine *e)
{
    if (!e) return DBAZ_EINVAL;
    USE_DEVICE(e);
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    e->timing = true;
    e->ev_used = 0;
    e->nn_launches = 0;
    e->ms_total = e->ms_nn_tower = 0;
    e->steps_at_t0 = e->steps;
    HIP_CHECK_RET(e, hipEventRecord(e->ev_t0, e->stream));
    return DBAZ_OK;
}

extern "C" int dbaz_timing_end(dbaz_engine *e)
{
    if (!e) return DBAZ_EINVAL;
    USE_DEVICE(e);
    if (!e->timing) return set_error(e, DBAZ_ESTATE, "dbaz_timing_begin not called");
    HIP_CHECK_RET(e, hipEventRecord(e->ev_t1, e->stream));
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    float ms = 0;
    HIP_CHECK_RET(e, hipEventElapsedTime(&ms, e->ev_t0, e->ev_t1));
    e->ms_total = ms;
    double tower = 0;
    for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
        float t = 0;
        HIP_CHECK_RET(e, hipEventElapsedTime(&t, e->ev_pool[i], e->ev_pool[i + 1]));
        tower += t;
    }
    e->ms_nn_tower = tower;
    e->timing = false;
    return DBAZ_OK;
}

extern "C" int dbaz_fetch_samples(dbaz_engine *e, int32_t max_rows, int32_t *n_rows, int32_t *game_idx, int16_t *move_idx,
                                  int16_t *move, int8_t *player, int16_t *x, int32_t *visits, double *pi, int8_t *z,
                                  int16_t *max_deepness, int32_t *tree_size, int32_t *terminal_count, float *q_value,
                                  int16_t *played)
{
    if (!e || !n_rows) return e ? set_error(e, DBAZ_EINVAL, "null argument") : DBAZ_EINVAL;
    USE_DEVICE(e);
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    const Geo &g = e->g;
    const int F = 3 * g.HW, A = g.A;
    int32_t n = 0;
    HIP_CHECK_RET(e, hipMemcpy(&n, e->B.out_count, 4, hipMemcpyDeviceToHost));
    *n_rows = n;
    if (max_rows < n) {
        if (max_rows == 0) return DBAZ_OK; // size query
        return set_error(e, DBAZ_EINVAL, "%d rows ready but max_rows = %d", n, max_rows);
    }
    if (n == 0) return DBAZ_OK;
    std::vector<RowMeta> meta(n);
    std::vector<int16_t> hx((size_t)n * F);
    std::vector<int32_t> hv((size_t)n * A);
    // one trip: the three row arrays and the counter reset are queued on the stream, then ONE synchronisation
    HIP_CHECK_RET(e, hipMemcpyAsync(meta.data(), e->B.out_meta, sizeof(RowMeta) * n, hipMemcpyDeviceToHost, e->stream));
    HIP_CHECK_RET(e, hipMemcpyAsync(hx.data(), e->B.out_x, (size_t)n * F * 2, hipMemcpyDeviceToHost, e->stream));
    HIP_CHECK_RET(e, hipMemcpyAsync(hv.data(), e->B.out_vis, (size_t)n * A * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_CHECK_RET(e, hipMemsetAsync(e->B.out_count, 0, 4, e->stream));
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        if (meta[a].game_idx != meta[b].game_idx) return meta[a].game_idx < meta[b].game_idx;
        return meta[a].move_idx < meta[b].move_idx;
    });
    for (int r = 0; r < n; r++) {
        const RowMeta &m = meta[order[r]];
        const int src = order[r];
        if (game_idx) game_idx[r] = m.game_idx;
        if (move_idx) move_idx[r] = m.move_idx;
        if (move) move[r] = m.move;
        if (player) player[r] = m.player;
        if (x) memcpy(x + (size_t)r * F, hx.data() + (size_t)src * F, (size_t)F * 2);
        if (visits) memcpy(visits + (size_t)r * A, hv.data() + (size_t)src * A, (size_t)A * 4);
        if (pi) {
            // policies.append(child_number_visits / (vs or 1.0)), self_play.py:114-115
            long long vs = 0;
            for (int a = 0; a < A; a++) vs += hv[(size_t)src * A + a];
            const double den = vs ? (double)vs : 1.0;
            for (int a = 0; a < A; a++) pi[(size_t)r * A + a] = (double)hv[(size_t)src * A + a] / den;
        }
        if (z) z[r] = m.z;
        if (max_deepness) max_deepness[r] = m.max_deepness;
        if (tree_size) tree_size[r] = m.tree_size;
        if (terminal_count) terminal_count[r] = m.terminal_count;
        if (q_value) q_value[r] = m.q_value;
        if (played) played[r] = m.played;
    }
    return DBAZ_OK;
}

#ifdef DBAZ_STAMP
extern "C" int dbaz_debug_read_stamps(dbaz_engine *e, unsigned long long *out, int n_wg)
{
    if (!e) return DBAZ_EINVAL;
    (void)hipStreamSynchronize(e->stream);
    return nn_read_stamps(e->nn, out, n_wg) == 0 ? DBAZ_OK : DBAZ_ESTATE;
}
#endif

extern "C" int dbaz_replay_rows_dev(dbaz_engine *e, void **rows_dev, int32_t *n_rows, int32_t *row_bytes)
{
    USE_DEVICE(e);
    // replay row (fixed stride): RowMeta (24 B) | x int16[3HW] | visits int32[A], padded to 8 B
    if (!e || !rows_dev || !n_rows || !row_bytes) return e ? set_error(e, DBAZ_EINVAL, "null argument") : DBAZ_EINVAL;
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    const Geo &g = e->g;
    const int F = 3 * g.HW, A = g.A;
    int32_t n = 0;
    HIP_CHECK_RET(e, hipMemcpy(&n, e->B.out_count, 4, hipMemcpyDeviceToHost));
    const int rb = (int)((sizeof(RowMeta) + (size_t)F * 2 + (size_t)A * 4 + 7) & ~(size_t)7);
    size_t need = std::max<size_t>((size_t)n * rb, 16);
    if (need > e->replay_bytes) {
        if (e->replay_dev) HIP_CHECK_RET(e, hipFree(e->replay_dev));
        e->replay_dev = nullptr;
        HIP_CHECK_RET(e, hipMalloc(&e->replay_dev, need));
        e->replay_bytes = need;
    }
    if (n > 0) {
        char *dst = (char *)e->replay_dev;
        HIP_CHECK_RET(e, hipMemcpy2DAsync(dst, rb, e->B.out_meta, sizeof(RowMeta), sizeof(RowMeta), n, hipMemcpyDeviceToDevice, e->stream));
        HIP_CHECK_RET(e, hipMemcpy2DAsync(dst + sizeof(RowMeta), rb, e->B.out_x, (size_t)F * 2, (size_t)F * 2, n, hipMemcpyDeviceToDevice, e->stream));
        HIP_CHECK_RET(e, hipMemcpy2DAsync(dst + sizeof(RowMeta) + (size_t)F * 2, rb, e->B.out_vis, (size_t)A * 4, (size_t)A * 4, n, hipMemcpyDeviceToDevice, e->stream));
        HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    }
    *rows_dev = e->replay_dev;
    *n_rows = n;
    *row_bytes = rb;
    return DBAZ_OK;
}

// ---- training data path (replay.hip) ---------------------------------------------------------
static Geo make_geo(int rows, int cols)
{
    Geo g;
    memset(&g, 0, sizeof(g));
    g.rows = rows; g.cols = cols; g.H = rows + 1; g.W = cols + 1; g.HW = g.H * g.W; g.A = 2 * g.HW;
    return g;
}

#define RDS(e)                                                                          \
    do {                                                                                \
        if (!(e)) return DBAZ_EINVAL;                                                   \
        USE_DEVICE(e);                                                                  \
        if (!(e)->rds_all[(e)->cur_ds]) (e)->rds_all[(e)->cur_ds] = rds_create((e)->g); \
        (e)->rds = (e)->rds_all[(e)->cur_ds];                                           \
    } while (0)
#define RDS_RET(e, call)                                                                \
    do {                                                                                \
        std::string _err;                                                               \
        int _rc = (call);                                                               \
        if (_rc) return set_error(e, _rc, "%s", _err.c_str());                          \
        return DBAZ_OK;                                                                 \
    } while (0)

extern "C" int dbaz_dataset_select(dbaz_engine *e, int32_t which)
{
    if (!e) return DBAZ_EINVAL;
    if (which < 0 || which >= DBAZ_MAX_DATASETS) return set_error(e, DBAZ_EINVAL, "dataset index must be 0..%d", DBAZ_MAX_DATASETS - 1);
    e->cur_ds = which;
    return DBAZ_OK;
}
extern "C" int dbaz_replay_rows_clear(dbaz_engine *e)
{
    if (!e) return DBAZ_EINVAL;
    USE_DEVICE(e);
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    HIP_CHECK_RET(e, hipMemset(e->B.out_count, 0, 4));
    return DBAZ_OK;
}
extern "C" int dbaz_dataset_begin(dbaz_engine *e)
{
    RDS(e);
    RDS_RET(e, rds_begin(e->rds, _err));
}
extern "C" int dbaz_dataset_add_rows(dbaz_engine *e, const void *rows_dev, int64_t n_rows, int32_t row_bytes, const int32_t *sel,
                                     int64_t n_sel)
{
    RDS(e);
    HIP_CHECK_RET(e, hipStreamSynchronize(e->stream));
    RDS_RET(e, rds_add_rows(e->rds, e->stream, rows_dev, n_rows, row_bytes, sel, n_sel, _err));
}
extern "C" int dbaz_dataset_finish(dbaz_engine *e, int32_t pos_average, const int32_t *order, int64_t *n_out)
{
    RDS(e);
    RDS_RET(e, rds_finish(e->rds, e->stream, pos_average, order, n_out, _err));
}
extern "C" int dbaz_dataset_fetch(dbaz_engine *e, int16_t *x, float *pi, float *z)
{
    RDS(e);
    RDS_RET(e, rds_fetch(e->rds, e->stream, x, pi, z, _err));
}
extern "C" int dbaz_dataset_batch(dbaz_engine *e, const int32_t *idx, int32_t n, int32_t sym, float *boards_dev, float *pi_dev,
                                  float *z_dev)
{
    RDS(e);
    RDS_RET(e, rds_batch(e->rds, e->stream, idx, n, sym, boards_dev, pi_dev, z_dev, _err));
}
extern "C" int dbaz_dataset_batch_on(dbaz_engine *e, const int32_t *idx, int32_t n, int32_t sym, float *boards_dev, float *pi_dev,
                                     float *z_dev, void *stream)
{
    RDS(e);
    RDS_RET(e, rds_batch(e->rds, (hipStream_t)stream, idx, n, sym, boards_dev, pi_dev, z_dev, _err, true));
}
extern "C" int dbaz_symmetry_apply(dbaz_engine *e, int32_t sym, const float *boards_in_dev, const float *pol_in_dev, int64_t n,
                                   float *boards_out_dev, float *pol_out_dev)
{
    RDS(e);
    RDS_RET(e, rds_symmetry_apply(e->rds, e->stream, sym, boards_in_dev, pol_in_dev, n, boards_out_dev, pol_out_dev, _err));
}
extern "C" int dbaz_symmetry_table(int32_t rows, int32_t cols, int32_t sym, int32_t *lut_out)
{
    if (rows < 1 || cols < 1 || !lut_out) return set_error(nullptr, DBAZ_EINVAL, "bad arguments");
    std::string err;
    int rc = rds_symmetry_lut(make_geo(rows, cols), sym, lut_out, err);
    if (rc) return set_error(nullptr, rc, "%s", err.c_str());
    return DBAZ_OK;
}
