"""Host-side mirror of the reference's game API (game.py:1-74, dots_boxes_game.py:10-155).

`BoxesState` keeps the reference's method names, argument meaning and error behaviour
(ValueError on an illegal move), but every rule evaluation is a call into the HIP rules
kernels through the C ABI (a batch of one).  It exists so that reference-side code and tests
run unchanged; throughput work uses the batched Engine API instead.
"""
import copy

import numpy as np

from .engine import Engine

_ENGINES = {}


def _rules_engine(rows, cols):
    key = (rows, cols)
    if key not in _ENGINES:
        _ENGINES[key] = Engine(rows, cols, 1, mcts_num_read=1, nodes_per_slot=8)
    return _ENGINES[key]


class GameState:
    __slots__ = []

    def __hash__(self):
        return self.get_hash().__hash__()

    def __eq__(self, other):
        return self.get_hash() == other.get_hash()


class BoxesState(GameState):
    __slots__ = ("hash", "_st", "_dim", "_moves")
    BOARD_DIM = (3, 3)
    FEATURES_SHAPE = (3, 4, 4)
    NB_ACTIONS = 32
    NB_BOXES = 9

    @staticmethod
    def init_static_fields(dims):
        # the reference passes ((rows, cols),) -- dots_boxes_game.py:21-28, configuration.py:79
        BoxesState.BOARD_DIM = tuple(dims[0])
        r, c = BoxesState.BOARD_DIM
        BoxesState.FEATURES_SHAPE = (3, r + 1, c + 1)
        BoxesState.NB_ACTIONS = 2 * (r + 1) * (c + 1)
        BoxesState.NB_BOXES = r * c

    def __init__(self):
        self._dim = tuple(BoxesState.BOARD_DIM)
        self._st = _rules_engine(*self._dim).rules_init(1)
        self.hash = (0, 0)
        self._moves = []

    # ---- reference attributes --------------------------------------------------------
    @property
    def to_play(self):
        return int(self._st["to_play"][0])

    @property
    def just_played(self):
        jp = int(self._st["just_played"][0])
        return None if jp < 0 else jp

    @property
    def boxes_to_close(self):
        return [self._st["b2c2"][0, 0] / 2.0, self._st["b2c2"][0, 1] / 2.0]

    @property
    def board(self):
        r, c = self._dim
        H, W = r + 1, c + 1
        b = np.zeros(2 * H * W, dtype=np.uint8)
        bits = self._bits()
        for i in range(2 * H * W):
            if (bits >> i) & 1:
                b[i] = 255
        b = b.reshape(2, H, W)
        b[1, r, :] = 1
        b[0, :, c] = 1
        return b

    def _bits(self):
        return sum(int(self._st["edges"][0, w]) << (64 * w) for w in range(4))

    def _eng(self):
        return _rules_engine(*self._dim)

    # ---- GameState API ---------------------------------------------------------------
    def get_actions_size(self):
        return 2 * (self._dim[0] + 1) * (self._dim[1] + 1)

    def get_valid_moves(self, as_indices=False):
        m = self._eng().rules_valid_moves(self._st)[0]
        return np.argwhere(m).ravel().tolist() if as_indices else m

    def get_result(self):
        r = int(self._eng().rules_result(self._st)[0])
        return None if r == 2 else r

    def play_(self, move):
        try:
            nc, lc = self._eng().rules_play(self._st, [int(move)])
        except ValueError:
            raise ValueError("Illegal move: " + str(move))
        self._moves.append(int(move))
        # _update_hash (dots_boxes_game.py:106-109) with a full-width python int bitmask
        self.hash = (self._bits(), self.boxes_to_close[self.to_play])
        return [(int(lc[0, 2 * i]), int(lc[0, 2 * i + 1])) for i in range(int(nc[0]))]

    def play(self, move):
        new_state = copy.deepcopy(self)
        new_state.play_(move)
        return new_state

    def __deepcopy__(self, memo):
        s = BoxesState.__new__(BoxesState)
        s._dim = self._dim
        s._st = {k: v.copy() for k, v in self._st.items()}
        s.hash = self.hash
        s._moves = list(self._moves)
        return s

    def get_features(self):
        return self._eng().rules_features(self._st)[0]

    def get_hash(self):
        return self.hash

    def __repr__(self):
        return "BoxesState(%dx%d, bits=%#x, to_play=%d, boxes_to_close=%s, result=%s)" % (
            self._dim[0], self._dim[1], self._bits(), self.to_play, self.boxes_to_close, self.get_result())


def nn_batch_builder(*game_states):
    """dots_boxes_game.py:148-155"""
    return np.stack([gs[0].get_features() for gs in game_states], axis=0)
