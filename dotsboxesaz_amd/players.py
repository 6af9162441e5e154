"""Host-side mirror of players.AZPlayer's move request (players.py:55-69; web/dotsandboxesagent_az.py:89-122).

The reference runs a process that, per request (game_state, generation, time_limit), searches ONE tree for
`time_limit` seconds -- `UCT_search(node, int(1e12), nn, mcts_cpuct, max_async_searches, (0.0, 0.0), time_limit)`
-- and answers the most visited valid move.  Here the tree, the network and the clock-bounded search live on the
GPU (dotsboxesaz_amd.mcts with a device evaluator: waves of max_async_searches simulations with virtual loss);
this class only keeps the per-generation weight containers and shapes the request/response.  No process, no
queues: the reference's are transport, not part of the path.
"""
import asyncio

import numpy as np

from . import mcts


def _get(d, k, default=None):
    if isinstance(d, dict):
        return d.get(k, default)
    return getattr(d, k, default)


class AZPlayer:
    def __init__(self, params, time_limit, device=0, nn_precision=1):
        self.params, self.time_limit, self.device, self.nn_precision = params, time_limit, device, nn_precision
        self.models = {}

    def _load_model(self, generation):
        """players.py:25-32"""
        if generation in self.models:
            return self.models[generation]
        model = _get(_get(self.params, "nn"), "model_class")(self.params)
        if generation is not None:
            model.load_parameters(generation)
        self.models[generation] = model
        return model

    def get_move(self, game_state, generation=None, time_limit=None, model=None):
        """One request of players.py:52-69: returns (move or None, visit counts, reads per second)."""
        import time
        mc = _get(_get(self.params, "self_play"), "mcts")
        model = model if model is not None else self._load_model(generation)
        node = mcts.create_root_uct_node(game_state, nn=model, max_pending_evals=int(_get(mc, "max_async_searches", 64)),
                                         nn_precision=self.nn_precision, device=self.device)
        t0 = time.time()
        loop = asyncio.new_event_loop()
        try:
            policy = loop.run_until_complete(mcts.UCT_search(node, int(1e12), None, tuple(_get(mc, "mcts_cpuct", (1.25, 19652))),
                                                             int(_get(mc, "max_async_searches", 64)), (0.0, 0.0),
                                                             time_limit if time_limit is not None else self.time_limit))
        finally:
            loop.close()
        dt = time.time() - t0
        policy = np.array(policy, copy=True)
        node._e.close()
        if policy.sum() > 0:
            policy *= np.asarray(game_state.get_valid_moves(), dtype=policy.dtype)
            return int(policy.argmax()), policy, float(policy.sum()) / dt  # we greedily return the best action
        return None, policy, 0.0
