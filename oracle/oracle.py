"""ctypes binding of the CPU ORACLE (oracle/dbaz_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under dotsboxesaz_amd/ imports it.

Parity status: pinned against golden vectors produced by importing the
reference (tests/golden/gen_golden.py) -- see tests/test_oracle_*.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
OB_MAX_A = 256
RESULT_NONE = 2


def build(force=False):
    """Compile liboracle.so with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "dbaz_oracle.c")
    hdr = os.path.join(_HERE, "dbaz_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Dims(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("A", C.c_int), ("B", C.c_int)]


class State(C.Structure):
    _fields_ = [("board", C.c_uint8 * OB_MAX_A), ("just_played", C.c_int32), ("to_play", C.c_int32),
                ("b2c2", C.c_int32 * 2), ("hash_bits", C.c_uint64 * 4), ("hash_b2c2", C.c_int32)]

    def copy(self):
        s = State()
        C.memmove(C.byref(s), C.byref(self), C.sizeof(State))
        return s

    def hash_int(self):
        return sum(int(self.hash_bits[w]) << (64 * w) for w in range(4))


class SearchParams(C.Structure):
    _fields_ = [("cpuct", C.c_double), ("cpuct_base", C.c_double), ("alpha", C.c_double), ("coeff", C.c_double)]


class SelfPlayParams(C.Structure):
    _fields_ = [("mcts_num_read", C.c_int), ("sp", SearchParams), ("reuse_tree", C.c_int),
                ("n_temp", C.c_int), ("temp_idx", C.c_int * 8), ("temp_val", C.c_double * 8)]


class Game(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("result", C.c_int32), ("winner", C.c_int32),
                ("n_search", C.c_int64), ("sum_path", C.c_int64),
                ("move", C.POINTER(C.c_int16)), ("player", C.POINTER(C.c_int8)),
                ("x", C.POINTER(C.c_int16)), ("visits", C.POINTER(C.c_int32)),
                ("pi", C.POINTER(C.c_double)), ("z", C.POINTER(C.c_int64)),
                ("max_deepness", C.POINTER(C.c_int32)), ("tree_size", C.POINTER(C.c_int32)),
                ("terminal_count", C.POINTER(C.c_int32)), ("q_value", C.POINTER(C.c_float)),
                ("played", C.POINTER(C.c_int16))]


EVAL_FN = C.CFUNCTYPE(None, C.POINTER(Dims), C.POINTER(State), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p)
CHOICE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)
NOISE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.c_double, C.c_void_p)
MOVE_FN = C.CFUNCTYPE(None, C.c_int, C.c_void_p)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.ob_dims_init.argtypes = [C.POINTER(Dims), C.c_int, C.c_int]
        L.ob_state_init.argtypes = [C.POINTER(Dims), C.POINTER(State)]
        L.ob_valid_moves.argtypes = [C.POINTER(Dims), C.POINTER(State), C.c_void_p]
        L.ob_get_result.argtypes = [C.POINTER(State)]
        L.ob_get_result.restype = C.c_int
        L.ob_play_.argtypes = [C.POINTER(Dims), C.POINTER(State), C.c_int, C.c_void_p]
        L.ob_play_.restype = C.c_int
        L.ob_features.argtypes = [C.POINTER(Dims), C.POINTER(State), C.c_void_p]
        L.ob_np_sum_f32.argtypes = [C.c_void_p, C.c_int]
        L.ob_np_sum_f32.restype = C.c_float
        L.ob_np_sum_f64.argtypes = [C.c_void_p, C.c_int]
        L.ob_np_sum_f64.restype = C.c_double
        L.ob_eval_formula.argtypes = [C.POINTER(Dims), C.POINTER(State), C.c_void_p, C.c_void_p, C.c_void_p]
        L.ob_tree_create.argtypes = [C.POINTER(Dims), C.POINTER(State)]
        L.ob_tree_create.restype = C.c_void_p
        L.ob_tree_free.argtypes = [C.c_void_p]
        L.ob_uct_search.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(SearchParams),
                                    C.c_void_p, C.c_void_p]
        L.ob_uct_search.restype = C.c_int
        L.ob_uct_search_pending.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(SearchParams),
                                            C.c_void_p, C.c_void_p]
        L.ob_uct_search_pending.restype = C.c_int
        L.ob_tree_advance.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ob_tree_advance.restype = C.c_void_p
        L.ob_tree_state.argtypes = [C.c_void_p]
        L.ob_tree_state.restype = C.POINTER(State)
        L.ob_tree_is_terminal.argtypes = [C.c_void_p]
        L.ob_tree_is_expanded.argtypes = [C.c_void_p]
        L.ob_tree_root_arrays.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.ob_tree_stats.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.ob_tree_root_slot.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ob_tree_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ob_play_game.argtypes = [C.POINTER(Dims), C.POINTER(State), C.POINTER(SelfPlayParams),
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ob_play_game.restype = C.POINTER(Game)
        L.ob_game_free.argtypes = [C.POINTER(Game)]
        L.ob_choice_xorshift.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ob_choice_xorshift.restype = C.c_int
        _lib = L
    return _lib


def dims(rows, cols):
    d = Dims()
    lib().ob_dims_init(C.byref(d), rows, cols)
    return d


def new_state(d):
    s = State()
    lib().ob_state_init(C.byref(d), C.byref(s))
    return s


def valid_moves(d, s):
    out = np.zeros(d.A, dtype=np.uint8)
    lib().ob_valid_moves(C.byref(d), C.byref(s), out.ctypes.data)
    return out.astype(bool)


def get_result(s):
    r = lib().ob_get_result(C.byref(s))
    return None if r == RESULT_NONE else r


def play_(d, s, move):
    """Mutates s; returns the list of closed (l, c) boxes; raises ValueError if illegal."""
    closed = np.zeros(4, dtype=np.int32)
    n = lib().ob_play_(C.byref(d), C.byref(s), int(move), closed.ctypes.data)
    if n < 0:
        raise ValueError("Illegal move: %d" % move)
    return [(int(closed[2 * i]), int(closed[2 * i + 1])) for i in range(n)]


def features(d, s):
    out = np.zeros(3 * d.H * d.W, dtype=np.int16)
    lib().ob_features(C.byref(d), C.byref(s), out.ctypes.data)
    return out.reshape(3, d.H, d.W)


def board_array(d, s):
    return np.frombuffer(bytes(s.board), dtype=np.uint8)[: d.A].reshape(2, d.H, d.W).copy()


def state_from_moves(d, moves):
    s = new_state(d)
    for m in moves:
        play_(d, s, m)
    return s


def np_sum_f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return np.float32(lib().ob_np_sum_f32(a.ctypes.data, a.size))


def np_sum_f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return np.float64(lib().ob_np_sum_f64(a.ctypes.data, a.size))


def eval_formula(d, s, kind=0):
    p = np.zeros(d.A, dtype=np.float32)
    v = C.c_float()
    k = C.c_int(kind)
    lib().ob_eval_formula(C.byref(d), C.byref(s), p.ctypes.data, C.byref(v), C.byref(k))
    return p, np.float32(v.value)


class Evaluator:
    """Wraps either a formula kind (int) or a python callable
    fn(dims, state) -> (p float32[A], v float) as an ob_eval_fn."""

    def __init__(self, fn_or_kind):
        self.n_calls = 0
        if isinstance(fn_or_kind, int):
            self._kind = C.c_int(fn_or_kind)
            self.fn_ptr = C.cast(lib().ob_eval_formula, C.c_void_p)
            self.user = C.cast(C.pointer(self._kind), C.c_void_p)
        else:
            pyfn = fn_or_kind

            def _cb(dp, sp, pp, vp, _user):
                self.n_calls += 1
                d = dp.contents
                p, v = pyfn(d, sp.contents)
                p = np.ascontiguousarray(p, dtype=np.float32).ravel()
                C.memmove(pp, p.ctypes.data, 4 * d.A)
                vp[0] = float(np.float32(np.asarray(v).ravel()[0]))

            self._cb = EVAL_FN(_cb)
            self.fn_ptr = C.cast(self._cb, C.c_void_p)
            self.user = None


def search_params(cpuct=(1.25, 19652), dirichlet=(0.0, 0.0)):
    return SearchParams(float(cpuct[0]), float(cpuct[1]), float(dirichlet[0]), float(dirichlet[1]))


class Tree:
    """create_root_uct_node / UCT_search / init_mcts_tree of mcts.py."""

    def __init__(self, d, state):
        self.d = d
        self.ptr = lib().ob_tree_create(C.byref(d), C.byref(state))

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().ob_tree_free(self.ptr)
            self.ptr = None

    def search(self, num_reads, evaluator, cpuct=(1.25, 19652), dirichlet=(0.0, 0.0), noise=None, max_pending=1):
        """UCT_search; max_pending > 1: mcts.py:228-239 under an evaluator that suspends once per call (waves)."""
        sp = search_params(cpuct, dirichlet)
        vis = np.zeros(self.d.A, dtype=np.int32)
        nz = None
        if dirichlet[0] > 0:
            nz = np.ascontiguousarray(noise, dtype=np.float64)
            assert nz.size == self.d.A
        lib().ob_uct_search_pending(self.ptr, int(num_reads), int(max_pending), evaluator.fn_ptr, evaluator.user, C.byref(sp),
                                    nz.ctypes.data if nz is not None else None, vis.ctypes.data)
        return vis

    def advance(self, move, reuse_tree=True):
        p = lib().ob_tree_advance(self.ptr, int(move), int(bool(reuse_tree)))
        if not p:
            raise ValueError("Illegal move: %d" % move)
        self.ptr = p

    @property
    def state(self):
        return lib().ob_tree_state(self.ptr).contents.copy()

    @property
    def is_terminal(self):
        return bool(lib().ob_tree_is_terminal(self.ptr))

    @property
    def is_expanded(self):
        return bool(lib().ob_tree_is_expanded(self.ptr))

    def root_arrays(self):
        A = self.d.A
        pri = np.zeros(A, dtype=np.float64)
        tv = np.zeros(A, dtype=np.float32)
        nv = np.zeros(A, dtype=np.int32)
        pc = np.zeros(A, dtype=np.int32)
        lib().ob_tree_root_arrays(self.ptr, pri.ctypes.data, tv.ctypes.data, nv.ctypes.data, pc.ctypes.data)
        return pri, tv, nv, pc

    def stats(self):
        md, ts, tc = C.c_int32(), C.c_int32(), C.c_int32()
        q = C.c_float()
        lib().ob_tree_stats(self.ptr, C.byref(md), C.byref(ts), C.byref(tc), C.byref(q))
        return md.value, ts.value, tc.value, np.float32(q.value)

    def root_slot(self):
        tv = C.c_float()
        nv = C.c_int32()
        lib().ob_tree_root_slot(self.ptr, C.byref(tv), C.byref(nv))
        return np.float32(tv.value), nv.value

    def live_nodes(self):
        """nodes of the current tree, root included"""
        f = lib().ob_tree_live_nodes
        f.restype, f.argtypes = C.c_int64, [C.c_void_p]
        return int(f(self.ptr))

    def counters(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        lib().ob_tree_counters(self.ptr, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value


def selfplay_params(mcts_num_read=800, cpuct=(1.25, 19652), noise=(0.8, 0.25), reuse_tree=True,
                    temperature=None):
    temperature = {0: 1.0, 12: 0.02} if temperature is None else temperature
    pp = SelfPlayParams()
    pp.mcts_num_read = int(mcts_num_read)
    pp.sp = search_params(cpuct, noise)
    pp.reuse_tree = int(bool(reuse_tree))
    items = sorted(temperature.items())
    assert len(items) <= 8
    pp.n_temp = len(items)
    for i, (k, v) in enumerate(items):
        pp.temp_idx[i] = int(k)
        pp.temp_val[i] = float(v)
    return pp


def play_game(d, pp, evaluator, start=None, choice=None, noise=None, forced_moves=None, rng_state=None, on_move=None):
    """SelfPlay.play_game + its rows of get_datasets.

    choice(probs)->move and noise(A, alpha)->f64[A] are python callables (pass
    np.random.choice / np.random.dirichlet closures to replay the reference's RNG
    stream); when None the built-in xorshift hooks are used (cpu_baseline leg).
    """
    L = lib()
    start = new_state(d) if start is None else start
    keep = []
    state = C.c_uint64(rng_state if rng_state is not None else 0x9E3779B97F4A7C15)
    if choice is None:
        ch_ptr, ch_user = C.cast(L.ob_choice_xorshift, C.c_void_p), C.cast(C.pointer(state), C.c_void_p)
    else:
        def _ch(pp_, n, _u):
            return int(choice(np.ctypeslib.as_array(pp_, shape=(n,)).copy()))
        cb = CHOICE_FN(_ch)
        keep.append(cb)
        ch_ptr, ch_user = C.cast(cb, C.c_void_p), None
    if noise is None:
        nz_ptr, nz_user = C.cast(L.ob_noise_xorshift, C.c_void_p), C.cast(C.pointer(state), C.c_void_p)
    else:
        def _nz(out, n, alpha, _u):
            v = np.ascontiguousarray(noise(n, alpha), dtype=np.float64)
            C.memmove(out, v.ctypes.data, 8 * n)
        cb2 = NOISE_FN(_nz)
        keep.append(cb2)
        nz_ptr, nz_user = C.cast(cb2, C.c_void_p), None
    fm = None
    nf = 0
    if forced_moves is not None:
        fm = np.ascontiguousarray(forced_moves, dtype=np.int16)
        nf = fm.size
    mv_ptr = None
    if on_move is not None:
        cb3 = MOVE_FN(lambda tp, _u: on_move(int(tp)))
        keep.append(cb3)
        mv_ptr = C.cast(cb3, C.c_void_p)
    gp = L.ob_play_game(C.byref(d), C.byref(start), C.byref(pp), evaluator.fn_ptr, evaluator.user,
                        ch_ptr, ch_user, nz_ptr, nz_user, fm.ctypes.data if fm is not None else None, nf, mv_ptr, None)
    g = gp.contents
    n, A, F = g.n_rows, d.A, 3 * d.H * d.W

    def arr(ptr, shape, dt):
        return np.ctypeslib.as_array(ptr, shape=shape).astype(dt, copy=True)

    out = dict(
        n_rows=n, result=g.result, winner=g.winner, n_search=g.n_search, sum_path=g.sum_path,
        move=arr(g.move, (n,), np.int16), player=arr(g.player, (n,), np.int8),
        x=arr(g.x, (n, F), np.int16), visits=arr(g.visits, (n, A), np.int32),
        pi=arr(g.pi, (n, A), np.float64), z=arr(g.z, (n,), np.int64),
        max_deepness=arr(g.max_deepness, (n,), np.int32), tree_size=arr(g.tree_size, (n,), np.int32),
        terminal_count=arr(g.terminal_count, (n,), np.int32), q_value=arr(g.q_value, (n,), np.float32),
        played=arr(g.played, (n,), np.int16),
    )
    L.ob_game_free(gp)
    return out
