"""ctypes binding of libdbaz_hip.so (C ABI: include/dbaz.h).

This is the stub a reference maintainer would add.  There
is no fallback: if the HIP library is missing or cannot be loaded, importing
the product path raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DBAZ_LIB") or os.path.join(_HERE, "libdbaz_hip.so")  # DBAZ_LIB: diagnostic builds (tools/)

OK, EINVAL, EILLEGAL, EDEVICE, EPOOL, ESTATE = range(6)
RESULT_NONE = 2
EVAL_FORMULA_HASH, EVAL_FORMULA_UNIFORM, EVAL_RESNET, EVAL_SIMPLENN, EVAL_EXTERNAL = range(5)
ABI_VERSION = 3  # DBAZ_ABI_VERSION of include/dbaz.h
DBG_EARLY_JOIN, DBG_NO_FALLBACK = 1, 2

# every symbol include/dbaz.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "dbaz_last_error", "dbaz_version", "dbaz_build_info", "dbaz_nodes_per_slot", "dbaz_create", "dbaz_destroy", "dbaz_sync",
    "dbaz_rules_init", "dbaz_rules_valid_moves", "dbaz_rules_play", "dbaz_rules_result", "dbaz_rules_features",
    "dbaz_nn_configure", "dbaz_nn_select_model", "dbaz_nn_set_tensor", "dbaz_nn_commit", "dbaz_nn_predict",
    "dbaz_set_search_params", "dbaz_set_positions", "dbaz_search", "dbaz_search_timed", "dbaz_set_pending", "dbaz_search_begin", "dbaz_select", "dbaz_expand_backup",
    "dbaz_get_roots", "dbaz_get_root_states", "dbaz_advance",
    "dbaz_selfplay_start", "dbaz_selfplay_script", "dbaz_selfplay_fastforward", "dbaz_selfplay_stagger", "dbaz_selfplay_quickplay", "dbaz_step", "dbaz_run",
    "dbaz_get_counters", "dbaz_timing_begin", "dbaz_timing_end", "dbaz_fetch_samples", "dbaz_replay_rows_dev",
    "dbaz_replay_rows_clear", "dbaz_dataset_select", "dbaz_dataset_begin", "dbaz_dataset_add_rows", "dbaz_dataset_finish", "dbaz_dataset_fetch", "dbaz_dataset_batch", "dbaz_dataset_batch_on",
    "dbaz_symmetry_apply", "dbaz_symmetry_table",
    "dbaz_trainer_last_error", "dbaz_trainer_create", "dbaz_trainer_destroy", "dbaz_trainer_forward", "dbaz_trainer_backward",
    "dbaz_trainer_net_forward", "dbaz_trainer_net_backward",
    "dbaz_bn2d_workspace_bytes", "dbaz_bn2d_forward", "dbaz_bn2d_backward",
    "dbaz_az_loss_workspace_bytes", "dbaz_az_loss", "dbaz_sgd_step",
]


class NetTensors(C.Structure):
    """dbaz_net_tensors (include/dbaz.h): device pointers to a ResNetZero's parameters -- or to where their gradients go."""
    _fields_ = [(k, C.c_void_p) for k in (
        "bn_input_w", "bn_input_b", "conv0_w", "conv0_b", "bn0_w", "bn0_b", "blk_conv_w", "blk_conv_b", "blk_bn_w", "blk_bn_b",
        "ph_conv_w", "ph_conv_b", "ph_bn_w", "ph_bn_b", "ph_fc_w", "ph_fc_b", "vh_conv_w", "vh_conv_b", "vh_bn_w", "vh_bn_b",
        "vh_fc0_w", "vh_fc0_b", "vh_fc1_w", "vh_fc1_b")]


class NetRunning(C.Structure):
    """dbaz_net_running: the BatchNorm layers' running statistics."""
    _fields_ = [(k, C.c_void_p) for k in ("bn_input_mean", "bn_input_var", "bn0_mean", "bn0_var", "blk_mean", "blk_var", "ph_mean", "ph_var",
                                          "vh_mean", "vh_var")]


class Config(C.Structure):
    _fields_ = [
        ("rows", C.c_int32), ("cols", C.c_int32), ("n_slots", C.c_int32), ("nodes_per_slot", C.c_int32),
        ("mcts_num_read", C.c_int32), ("cpuct", C.c_double), ("cpuct_base", C.c_double),
        ("noise_alpha", C.c_double), ("noise_coeff", C.c_double), ("reuse_tree", C.c_int32),
        ("n_temp", C.c_int32), ("temp_idx", C.c_int32 * 8), ("temp_val", C.c_double * 8),
        ("evaluator", C.c_int32), ("device", C.c_int32), ("seed", C.c_uint64), ("max_out_rows", C.c_int32),
        ("nn_precision", C.c_int32), ("match_play", C.c_int32), ("evaluator2", C.c_int32),
        ("transposition_cache", C.c_int32), ("max_pending_evals", C.c_int32), ("selfplay_pending", C.c_int32),
        ("eval_round", C.c_int32), ("eval_defer_max", C.c_int32), ("debug_flags", C.c_uint32), ("reserved", C.c_int32 * 3),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("steps", C.c_int64), ("expansions", C.c_int64), ("nn_evals", C.c_int64), ("terminal_leaves", C.c_int64),
        ("sum_path", C.c_int64), ("games_finished", C.c_int64), ("rows_ready", C.c_int64), ("moves_played", C.c_int64),
        ("pool_high_water", C.c_int64), ("active_slots", C.c_int32), ("error_slots", C.c_int32), ("blocked_slots", C.c_int32), ("f32_fallback_evals", C.c_int32),
        ("ms_total", C.c_double), ("ms_tree", C.c_double), ("ms_nn", C.c_double), ("nn_launches", C.c_int64),
        ("ms_nn_tower", C.c_double), ("cache_hits", C.c_int64), ("pool_resets", C.c_int64),
    ]


class DbazError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


_lib = None


def load():
    """Load the HIP library or raise -- never falls back to a CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -m dotsboxesaz_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # A process that also uses torch must load torch's bundled HIP runtime BEFORE this library pulls in
    # the system one: in the other order torch's later initialisation finds "No HIP GPUs" (two runtimes
    # competing for the device; measured on ROCm 7.2 + torch 2.10/rocm7.0).  The library itself has no
    # torch dependency -- a process without torch simply runs on the system runtime.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    L.dbaz_version.restype = C.c_int
    if L.dbaz_version() != ABI_VERSION:
        raise ImportError("%s has ABI version %d, this binding expects %d: rebuild with `python -m dotsboxesaz_amd.build`"
                          % (LIB_PATH, L.dbaz_version(), ABI_VERSION))
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.dbaz_last_error.argtypes = [vp]
    L.dbaz_last_error.restype = C.c_char_p
    L.dbaz_version.restype = C.c_int
    L.dbaz_nodes_per_slot.argtypes = [vp]
    L.dbaz_build_info.restype = C.c_char_p
    L.dbaz_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.dbaz_destroy.argtypes = [vp]
    L.dbaz_destroy.restype = None
    L.dbaz_sync.argtypes = [vp]
    L.dbaz_rules_init.argtypes = [vp, i32, vp, vp, vp, vp]
    L.dbaz_rules_valid_moves.argtypes = [vp, i32, vp, vp]
    L.dbaz_rules_play.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.dbaz_rules_result.argtypes = [vp, i32, vp, vp, vp]
    L.dbaz_rules_features.argtypes = [vp, i32, vp, vp, vp, vp]
    L.dbaz_nn_configure.argtypes = [vp, i32, i32, i32, i32, i32]
    L.dbaz_nn_select_model.argtypes = [vp, i32]
    L.dbaz_nn_set_tensor.argtypes = [vp, C.c_char_p, vp, i64]
    L.dbaz_nn_commit.argtypes = [vp]
    L.dbaz_nn_predict.argtypes = [vp, i32, vp, vp, vp]
    L.dbaz_set_search_params.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double]
    L.dbaz_set_positions.argtypes = [vp, vp, vp]
    L.dbaz_search.argtypes = [vp, vp, vp]
    L.dbaz_search_timed.argtypes = [vp, vp, vp, C.c_double]
    L.dbaz_set_pending.argtypes = [vp, i32, i32]
    L.dbaz_search_begin.argtypes = [vp, vp, vp]
    L.dbaz_select.argtypes = [vp, vp, vp, vp]
    L.dbaz_expand_backup.argtypes = [vp, vp, vp]
    L.dbaz_get_roots.argtypes = [vp] + [vp] * 8
    L.dbaz_get_root_states.argtypes = [vp] + [vp] * 6
    L.dbaz_advance.argtypes = [vp, vp, i32]
    L.dbaz_selfplay_start.argtypes = [vp, i64, i64]
    L.dbaz_selfplay_script.argtypes = [vp, i64, vp, i32, vp]
    L.dbaz_selfplay_fastforward.argtypes = [vp, vp]
    L.dbaz_selfplay_stagger.argtypes = [vp, vp]
    L.dbaz_selfplay_quickplay.argtypes = [vp, vp, i32]
    L.dbaz_step.argtypes = [vp, i32]
    L.dbaz_run.argtypes = [vp, i64]
    L.dbaz_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.dbaz_timing_begin.argtypes = [vp]
    L.dbaz_timing_end.argtypes = [vp]
    L.dbaz_fetch_samples.argtypes = [vp, i32, vp] + [vp] * 13
    L.dbaz_replay_rows_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]
    L.dbaz_trainer_last_error.argtypes = [vp]
    L.dbaz_trainer_last_error.restype = C.c_char_p
    L.dbaz_trainer_create.argtypes = [i32, i32, i32, i32, i32, i32, C.POINTER(vp)]
    L.dbaz_trainer_destroy.argtypes = [vp]
    L.dbaz_trainer_destroy.restype = None
    L.dbaz_trainer_forward.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.dbaz_trainer_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.dbaz_trainer_net_forward.argtypes = [vp, i32, vp, C.POINTER(NetTensors), C.POINTER(NetRunning), i32, i32, i32, vp, vp, vp]
    L.dbaz_trainer_net_backward.argtypes = [vp, vp, vp, vp, C.POINTER(NetTensors), C.POINTER(NetTensors), vp]
    L.dbaz_bn2d_workspace_bytes.argtypes = [i32]
    L.dbaz_bn2d_forward.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, C.c_float, C.c_float, i32, vp, vp, vp, vp, vp]
    L.dbaz_bn2d_backward.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.dbaz_az_loss_workspace_bytes.argtypes = []
    L.dbaz_az_loss.argtypes = [vp, vp, vp, vp, i32, i32, C.c_float, vp, vp, vp, vp, vp]
    L.dbaz_sgd_step.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, i32, C.c_float, C.c_float, C.c_float, vp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("dbaz_last_error", "dbaz_build_info", "dbaz_destroy", "dbaz_trainer_last_error", "dbaz_trainer_destroy", "dbaz_bn2d_workspace_bytes",
                        "dbaz_az_loss_workspace_bytes"):
            fn.restype = C.c_int
    L.dbaz_bn2d_workspace_bytes.restype = C.c_int64
    L.dbaz_az_loss_workspace_bytes.restype = C.c_int64
    _lib = L
    return L


def build_info():
    """{"src": hash of all kernel sources, "nn": hash of the network kernels} of the LOADED library (csrc/buildinfo.cpp)."""
    info = load().dbaz_build_info().decode()
    return dict(kv.split("=", 1) for kv in info.split() if "=" in kv)


def check(handle, rc):
    if rc == OK:
        return
    msg = load().dbaz_last_error(handle)
    msg = msg.decode() if msg else "error %d" % rc
    if rc == EILLEGAL:
        raise ValueError(msg)  # the reference raises ValueError for illegal moves
    raise DbazError(rc, msg)
