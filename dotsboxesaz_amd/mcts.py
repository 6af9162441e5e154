"""Host-side mirror of the reference's search contract (mcts.py:156-244).

    root = create_root_uct_node(game_state)
    visits = await UCT_search(root, num_reads, async_nn, cpuct, max_pending_evals, dirichlet)
    root = init_mcts_tree(root, move, reuse_tree)

The tree lives on the GPU (one engine slot, external-evaluator mode): select / expand /
backup / re-root are the HIP kernels; `async_nn(game_state) -> (p[A], v[1])` is any python
coroutine, as in the reference.  Semantics are the reference's sequential ones
with a python `async_nn` (max_pending_evals is then accepted and ignored); a root created with a network
(`create_root_uct_node(state, nn=...)`) is searched on the device with up to max_pending_evals simulations
of the one tree in flight (see players.py).  `time_limit` works as in the reference
(mcts.py:201-203,232-233, default 120 s): players.AZPlayer's `UCT_search(root, int(1e12), ..., time_limit=t)`.
Dirichlet noise is drawn from numpy's global RNG exactly where the reference draws it.
"""
import time
from collections import namedtuple

import numpy as np

from .engine import Engine

TreeStats = namedtuple("TreeStats", ["max_deepness", "tree_size", "terminal_count", "q_value"])


class _LeafState:
    """What async_nn may ask of a leaf: features, hash, action count, valid moves."""

    def __init__(self, x):
        self._x = x

    def get_features(self):
        return self._x

    def get_actions_size(self):
        return 2 * self._x.shape[1] * self._x.shape[2]

    def get_valid_moves(self, as_indices=False):
        H, W = self._x.shape[1:]
        m = self._x[:2].reshape(-1) == 0
        m = m.copy()
        m.reshape(2, H, W)[1, H - 1, :] = False
        m.reshape(2, H, W)[0, :, W - 1] = False
        return np.argwhere(m).ravel().tolist() if as_indices else m

    def get_hash(self):
        bits = 0
        for i in np.nonzero(self._x[:2].reshape(-1))[0]:
            bits |= 1 << int(i)
        return (bits, float(self._x[2, 0, 0]) / 2.0)

    def __hash__(self):
        return hash(self.get_hash())

    def __eq__(self, other):
        return self.get_hash() == other.get_hash()


class UCTNode:
    """Handle on the root of a device-resident tree."""

    def __init__(self, engine, game_state, move):
        self._e = engine
        self.game_state = game_state
        self.move = move
        self._roots = None

    def _refresh(self):
        self._roots = self._e.roots()
        return self._roots

    @property
    def child_number_visits(self):
        return (self._roots or self._refresh())["visits"][0]

    @property
    def child_total_value(self):
        return (self._roots or self._refresh())["total_value"][0]

    @property
    def child_priors(self):
        return (self._roots or self._refresh())["priors"][0]

    @property
    def is_expanded(self):
        return bool(self._e.root_states()["expanded"][0])

    @property
    def is_terminal(self):
        return self.game_state.get_result() is not None

    def get_tree_stats(self):
        r = self._roots or self._refresh()
        return TreeStats(int(r["stats"][0, 0]), int(r["stats"][0, 1]), int(r["stats"][0, 2]), r["q"][0])


def create_root_uct_node(game_state, nodes_per_slot=0, mcts_num_read=800, nn=None, max_pending_evals=64, nn_precision=1,
                         device=0):
    """mcts.py:156-160.  With `nn` (a NeuralNetWrapper, or a weight container such as nn.ResNetZero) the
    tree's evaluator is the HIP network of this handle and UCT_search keeps up to `max_pending_evals`
    simulations of the tree in flight (players.AZPlayer's use); without it, `async_nn` of UCT_search is
    awaited for every leaf (any python coroutine, sequential search)."""
    r, c = game_state._dim
    model = getattr(nn, "model", nn)
    if model is None:
        e = Engine(r, c, 1, mcts_num_read=mcts_num_read, evaluator="external", nodes_per_slot=nodes_per_slot, device=device)
    else:
        e = Engine(r, c, 1, mcts_num_read=mcts_num_read, evaluator=model.kind, nodes_per_slot=nodes_per_slot or (1 << 19),
                   nn_precision=nn_precision if model.kind == "resnet" else 0, device=device,
                   max_pending_evals=max(1, int(max_pending_evals)))
        e.load_state_dict(model.state_dict(), model.kind, **model.shape)
    # to_play and the scores depend on the ORDER of the moves, so the position is installed on the
    # device by replaying the state's recorded move history
    e.set_positions([list(game_state._moves)])
    node = UCTNode(e, game_state, None)
    node._device_nn = model is not None
    return node


async def UCT_search(root_node, num_reads, async_nn, cpuct=(1.25, 19652), max_pending_evals=64, dirichlet=(0.0, 0.0),
                     time_limit=None, virtual_visits=True):
    """mcts.py:183-244 -> np.int32[A] root child visit counts.  virtual_visits (device evaluator only): False = the
    reference's bookkeeping of pending simulations, True = their visits also count at selection time."""
    e = root_node._e
    alpha, coeff = dirichlet
    e.set_search_params(cpuct, dirichlet)
    noise = None
    if alpha > 0:
        # valid_actions*alpha is alpha for EVERY slot in the reference (mcts.py:220-222)
        noise = np.random.dirichlet(np.full(e.A, alpha), 1).reshape(1, e.A)
    reads = int(min(int(num_reads), 2 ** 31 - 1))

    if getattr(root_node, "_device_nn", False):
        # evaluator on the device: waves of up to max_pending_evals simulations of this tree, one batched network
        # launch per wave, the clock read between waves (dbaz_search_timed); `async_nn` is not called
        kmax = int(e.cfg.max_pending_evals)
        if kmax > 1:
            e.set_pending(max(1, min(int(max_pending_evals), kmax)), virtual_visits)
        e.search_timed(time_limit, reads, noise)
        root_node._refresh()
        return root_node.child_number_visits

    # drive select / evaluate / expand+backup by hand so that async_nn can be awaited
    import ctypes as C
    end_time = time.time() + (time_limit if time_limit else 120)  # mcts.py:201-203
    nr = np.full(1, reads, np.int32)
    first = not root_node.is_expanded  # the root expansion is not subject to the clock (mcts.py:207-208)
    e._ck(e._L.dbaz_search_begin(e.h, nr.ctypes.data, noise.ctypes.data if noise is not None else None))
    x = np.zeros((1, 3, e.H, e.W), np.int16)
    need = np.zeros(1, np.uint8)
    na = C.c_int32()
    P = np.zeros((1, e.A), np.float32)
    V = np.zeros(1, np.float32)
    while True:
        if not first and time.time() > end_time:  # mcts.py:232-233: silent early exit
            break
        first = False
        e._ck(e._L.dbaz_select(e.h, C.byref(na), x.ctypes.data, need.ctypes.data))
        if na.value == 0:
            break
        if need[0]:
            p, v = await async_nn(_LeafState(x[0].copy()))
            P[0] = np.asarray(p, np.float32)
            V[0] = np.float32(np.asarray(v).ravel()[0])
        e._ck(e._L.dbaz_expand_backup(e.h, P.ctypes.data, V.ctypes.data))
    root_node._refresh()
    return root_node.child_number_visits


def init_mcts_tree(previous_node, move, reuse_tree=True):
    """mcts.py:163-180"""
    e = previous_node._e
    e.advance([int(move)], reuse_tree)
    gs = previous_node.game_state.play(int(move))
    node = UCTNode(e, gs, int(move))
    node._device_nn = getattr(previous_node, "_device_nn", False)
    return node
