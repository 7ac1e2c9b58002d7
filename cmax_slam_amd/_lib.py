"""ctypes loader for libcmaxhip.so (the C ABI declared in include/cmax_hip.h).

There is no fallback of any kind: if the HIP extension is missing or fails to load, importing the
evaluators raises.  Nothing in this package touches oracle/.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("CMAX_HIP_SO") or os.path.join(_HERE, "libcmaxhip.so")  # (the override: A/B runs of two builds on one GPU box, tools/ab_builds.sh)

OK, ERR_INVALID_ARG, ERR_EVENT_RANGE, ERR_HIP, ERR_SPLINE_RANGE, ERR_STATE, ERR_TIME_ORDER = range(7)
VARIANCE, MEAN_SQUARE, GRADIENT_MAGNITUDE = 0, 1, 2
GRAD_PLANES, GRAD_ADJOINT = 0, 1
DT_U8, DT_F32, DT_F64 = 0, 1, 2
OP_SUM, OP_MAX = 0, 1
GROUP_AUTO, GROUP_RCCL, GROUP_DIRECT = 0, 1, 2
SCHED_BACKGROUND, SCHED_NORMAL, SCHED_URGENT = -1, 0, 1
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)
OPT_GRAD_MODE, OPT_SPLAT_MODE, OPT_REUSE_IMAGE, OPT_SPIN_WAIT, OPT_DETERMINISTIC, OPT_TAIL_FINALIZE = 1, 2, 3, 4, 5, 6
OPT_COMPOSITE_IMAGE, OPT_FOLD_BATCH, OPT_GATED_DF, OPT_CHAIN_SOLVE, OPT_FUSED_IMAGE = 8, 9, 10, 11, 12
DIAG_FORCE_CROSS_DEVICE = 1  # cmx_diag_set key (include/cmax_hip_diag.h)
PLANE_IL_OLD, PLANE_IL_NEW, PLANE_IWE, PLANE_DERIV0 = 0, 1, 2, 16
T_SPLAT, T_IMAGE, T_POSE, T_GATHER, T_ZERO, T_COMM, T_FINAL, T_BATCH, T_COUNT = 0, 1, 2, 3, 4, 5, 6, 7, 8
T_NAMES = ("splat", "image", "pose", "gather", "zero", "comm", "final", "batch")

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_u16p = C.POINTER(C.c_uint16)
c_i64p = C.POINTER(C.c_int64)
ctx_p = C.c_void_p

# every symbol include/cmax_hip.h declares: (restype, argtypes)
class AosLayout(C.Structure):  # cmx_aos_layout: record size + byte offsets of (uint16 x, uint16 y, uint32 sec, uint32 nsec)
    _fields_ = [("stride", C.c_size_t), ("off_x", C.c_size_t), ("off_y", C.c_size_t), ("off_sec", C.c_size_t), ("off_nsec", C.c_size_t)]


import numpy as _np  # noqa: E402
# dvs_msgs::Event as the reference's std::vector holds it: {uint16 x, y; ros::Time ts {uint32 sec, nsec}; bool polarity} = 16 bytes
DVS_EVENT_DTYPE = _np.dtype({"names": ["x", "y", "sec", "nsec", "polarity"], "formats": ["<u2", "<u2", "<u4", "<u4", "u1"],
                             "offsets": [0, 2, 4, 8, 12], "itemsize": 16})


def aos_layout_of(arr):
    """cmx_aos_layout of a numpy structured array with fields x, y, sec, nsec."""
    f = arr.dtype.fields
    return AosLayout(arr.dtype.itemsize, f["x"][1], f["y"][1], f["sec"][1], f["nsec"][1])


def dvs_events(x, y, t_ns, polarity=None):
    """SoA -> the reference's AoS records (test / bench helper: what a ROS host already has in msg->events)."""
    ev = _np.zeros(len(x), DVS_EVENT_DTYPE)
    ev["x"], ev["y"] = x, y
    t = _np.asarray(t_ns, _np.int64)
    ev["sec"], ev["nsec"] = t // 1000000000, t % 1000000000
    if polarity is not None:
        ev["polarity"] = polarity
    return ev


SYMBOLS = {
    "cmx_version": (C.c_char_p, []),
    "cmx_device_count": (C.c_int, []),
    "cmx_last_error": (C.c_char_p, [ctx_p]),
    "cmx_status_string": (C.c_char_p, [C.c_int]),
    "cmx_destroy": (None, [ctx_p]),
    "cmx_set_option": (C.c_int, [ctx_p, C.c_int, C.c_int]),
    "cmx_set_stream": (C.c_int, [ctx_p, C.c_void_p]),
    "cmx_set_stream_priority": (C.c_int, [ctx_p, C.c_int]),
    "cmx_set_cu_mask": (C.c_int, [ctx_p, C.POINTER(C.c_uint32), C.c_int]),
    "cmx_diag_set": (C.c_int, [C.c_int, C.c_int]),
    "cmx_group_transport_info": (C.c_int, [ctx_p, C.POINTER(C.c_int), C.POINTER(C.c_int), c_dp, c_dp]),
    "cmx_set_sched_class": (C.c_int, [ctx_p, C.c_int]),
    "cmx_backend_create_group": (C.c_int, [C.POINTER(ctx_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, c_dp, C.c_int, C.c_int,
                                           C.c_int]),
    "cmx_group_info": (C.c_int, [ctx_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), c_i64p, c_dp]),
    "cmx_backend_get_pose_table": (C.c_int, [ctx_p, C.c_int, c_dp, c_fp, C.POINTER(C.c_int), c_i64p, C.POINTER(C.c_int)]),
    "cmx_frontend_create": (C.c_int, [C.POINTER(ctx_p), C.c_int, C.c_int, C.c_int, c_dp]),
    "cmx_frontend_set_packet": (C.c_int, [ctx_p, C.c_int64, c_u16p, c_u16p, c_i64p, C.c_int64, C.c_double, C.c_double,
                                          C.c_double, C.c_double, C.c_int, C.c_double, C.c_int]),
    "cmx_frontend_eval": (C.c_int, [ctx_p, c_dp, c_dp, c_dp]),
    "cmx_frontend_prepare": (C.c_int, [ctx_p, c_dp]),
    "cmx_backend_prepare": (C.c_int, [ctx_p, c_dp]),
    "cmx_frontend_eval_many": (C.c_int, [ctx_p, C.c_int, c_dp, c_dp, c_dp]),
    "cmx_backend_eval_many": (C.c_int, [ctx_p, C.c_int, c_dp, c_dp, c_dp]),
    "cmx_hint_next_df": (C.c_int, [ctx_p, C.c_double, C.c_int]),
    "cmx_frontend_eval_each": (C.c_int, [ctx_p, C.c_int, c_dp, c_dp, c_dp]),
    "cmx_backend_eval_each": (C.c_int, [ctx_p, C.c_int, c_dp, c_dp, c_dp]),
    "cmx_frontend_get_iwe": (C.c_int, [ctx_p, c_dp, C.c_int, c_fp, c_fp]),
    "cmx_backend_create": (C.c_int, [C.POINTER(ctx_p), C.c_int, C.c_int, C.c_int, c_dp, C.c_int, C.c_int]),
    "cmx_backend_set_window": (C.c_int, [ctx_p, C.c_int64, c_u16p, c_u16p, c_i64p, C.c_int, C.c_int, c_dp, C.c_int64,
                                         C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int, c_fp]),
    "cmx_backend_eval": (C.c_int, [ctx_p, c_dp, c_dp, c_dp]),
    "cmx_backend_get_plane": (C.c_int, [ctx_p, C.c_int, c_fp]),
    "cmx_backend_get_alpha": (C.c_int, [ctx_p, c_dp]),
    "cmx_backend_update_map": (C.c_int, [ctx_p, C.c_int]),
    "cmx_backend_mark_visited": (C.c_int, [ctx_p, c_dp, C.c_int]),
    "cmx_backend_reset_map": (C.c_int, [ctx_p]),
    "cmx_backend_get_map": (C.c_int, [ctx_p, c_fp, C.POINTER(C.c_uint8)]),
    "cmx_backend_set_map": (C.c_int, [ctx_p, c_fp, C.POINTER(C.c_uint8)]),
    "cmx_events_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_size_t]),
    "cmx_events_create_group": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_size_t]),
    "cmx_events_devices": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "cmx_events_destroy": (None, [C.c_void_p]),
    "cmx_events_last_error": (C.c_char_p, [C.c_void_p]),
    "cmx_events_push": (C.c_int, [C.c_void_p, C.c_int64, c_u16p, c_u16p, c_i64p]),
    "cmx_events_push_aos": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(AosLayout)]),
    "cmx_frontend_set_packet_aos": (C.c_int, [ctx_p, C.c_int64, C.c_void_p, C.POINTER(AosLayout), C.c_int64, C.c_double, C.c_double,
                                            C.c_double, C.c_double, C.c_int, C.c_double, C.c_int]),
    "cmx_backend_set_window_aos": (C.c_int, [ctx_p, C.c_int64, C.c_void_p, C.POINTER(AosLayout), C.c_int, C.c_int, c_dp, C.c_int64,
                                           C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int, c_fp]),
    "cmx_events_drop_before": (C.c_int, [C.c_void_p, C.c_int64]),
    "cmx_events_begin": (C.c_int64, [C.c_void_p]),
    "cmx_events_end": (C.c_int64, [C.c_void_p]),
    "cmx_frontend_set_packet_from": (C.c_int, [ctx_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                               C.c_double, C.c_double, C.c_int, C.c_double, C.c_int]),
    "cmx_backend_set_window_from": (C.c_int, [ctx_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, c_dp, C.c_int64,
                                              C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int, c_fp]),
    "cmx_traj_temp_start_ns": (C.c_int64, [C.c_double, C.c_int, C.c_double]),
    "cmx_accum_capacity": (C.c_size_t, [ctx_p]),
    "cmx_set_accum_buffer": (C.c_int, [ctx_p, C.c_void_p, C.c_size_t]),
    "cmx_accum_ptr": (C.c_void_p, [ctx_p]),
    "cmx_accum_count": (C.c_size_t, [ctx_p]),
    "cmx_frontend_accumulate": (C.c_int, [ctx_p, c_dp, C.c_int]),
    "cmx_frontend_finish": (C.c_int, [ctx_p, c_dp, c_dp]),
    "cmx_backend_accumulate": (C.c_int, [ctx_p, c_dp, C.c_int]),
    "cmx_backend_finish": (C.c_int, [ctx_p, c_dp, c_dp]),
    "cmx_frontend_finish_begin": (C.c_int, [ctx_p, C.c_int]),
    "cmx_frontend_finish_end": (C.c_int, [ctx_p, c_dp, c_dp]),
    "cmx_backend_finish_begin": (C.c_int, [ctx_p, C.c_int]),
    "cmx_backend_finish_end": (C.c_int, [ctx_p, c_dp, c_dp]),
    "cmx_grad_ptr": (C.c_void_p, [ctx_p]),
    "cmx_grad_count": (C.c_size_t, [ctx_p]),
    "cmx_set_grad_buffer": (C.c_int, [ctx_p, C.c_void_p, C.c_size_t]),
    "cmx_comm_unique_id": (C.c_int, [C.c_char_p]),
    "cmx_comm_attach": (C.c_int, [ctx_p, C.c_char_p, C.c_int, C.c_int]),
    "cmx_comm_detach": (C.c_int, [ctx_p]),
    "cmx_comm_attach_custom": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "cmx_comm_info": (C.c_int, [ctx_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cmx_frontend_solve": (C.c_int, [ctx_p, c_dp, C.c_void_p]),
    "cmx_backend_solve": (C.c_int, [ctx_p, C.c_int, c_dp, C.c_void_p]),
    "cmx_frcg_minimize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_dp, C.c_double, C.c_double,
                                    C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "cmx_frcg_minimize_hinted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_dp, C.c_double,
                                           C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "cmx_integrate_ang_vel": (C.c_int, [C.c_int, c_i64p, c_dp, C.c_int64, c_dp, c_i64p, c_dp, C.c_int, c_i64p, c_dp,
                                        C.POINTER(C.c_int)]),
    "cmx_num_ctrl_poses": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_double]),
    "cmx_fit_ctrl_poses": (C.c_int, [C.c_int, C.c_int, c_i64p, c_dp, C.c_double, C.c_double, C.c_int, c_dp]),
    "cmx_traj_incremental_update": (C.c_int, [C.c_int, c_dp, C.c_int, C.c_int, c_dp]),
    "cmx_traj_evaluate": (C.c_int, [C.c_int, C.c_int, c_dp, C.c_int64, C.c_int64, C.c_int64, c_dp]),
    "cmx_bearing_lut": (C.c_int, [C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "cmx_get_stats": (C.c_int, [ctx_p, c_dp, C.c_int]),
    "cmx_abi_version": (C.c_int, []),
    "cmx_timing_enable": (C.c_int, [ctx_p, C.c_int]),
    "cmx_timing_get": (C.c_int, [ctx_p, c_dp, c_i64p]),
}



class SolveReport(C.Structure):
    _fields_ = [("iterations", C.c_int), ("status", C.c_int), ("n_f", C.c_int), ("n_df", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double)]


_LIB = None


class CmaxHipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("cmax-hip status %d: %s" % (status, msg))
        self.status = status


def build(force=False):
    """Compile libcmaxhip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(SO_PATH):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s"] + (["-B"] if force else []))
    return SO_PATH


def lib():
    """Load the HIP extension; raises if it is missing (no CPU fallback exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("libcmaxhip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback for the event-warping path)")
        # PyTorch-ROCm bundles its own HIP runtime (same SONAME, libamdhip64.so.7).  A process that uses both must
        # load torch's copy first, otherwise torch finds the device already owned by the system runtime and reports
        # "No HIP GPUs are available".  Loading torch first makes both share one runtime; hosts that never use torch
        # (the C++/ROS integration) are unaffected.  Opt out with CMAX_HIP_NO_TORCH=1.
        if not os.environ.get("CMAX_HIP_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(ctx, status):
    if status != OK:
        L = lib()
        msg = L.cmx_last_error(ctx).decode() if ctx else ""
        raise CmaxHipError(status, (L.cmx_status_string(status).decode() + (": " + msg if msg else "")))
