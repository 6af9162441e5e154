"""MI355X-native self-play rollout engine for Dots & Boxes AlphaZero (see README.md / DESIGN.md).

Modules (host-side mirrors of the reference's call contracts over the C ABI of libdbaz_hip.so):
    engine      Engine: one handle per GPU (rules, search, self-play driver, network, replay/dataset kernels)
    game        BoxesState                         (dots_boxes/dots_boxes_game.py)
    mcts        UCT_search, create_root_uct_node   (mcts.py)
    nn          ResNetZero / SimpleNN containers, NeuralNetWrapper (nn.py, dots_boxes/dots_boxes_nn.py)
    self_play   SelfPlay, generate_games, compute_elo, RCCL replay all-gather (self_play.py)
    train_data  ReplayStore, ReplayDataset, DeviceLoader, SymmetriesGenerator (utils/utils.py, dots_boxes_nn.py)
    train       AlphaZeroLoss, train(), checkpoints (nn.py) -- optimizer step on torch-ROCm
    coach       Coach.learn_to_play (coach.py) with the replay resident in HBM
    build       hipcc build of the shared library

Nothing is imported eagerly: the HIP library is loaded (or a clear error raised) on first use.
"""
__version__ = "0.1.0"
