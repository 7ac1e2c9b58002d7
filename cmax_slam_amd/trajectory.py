"""Host-side mirror of the reference's trajectory bookkeeping around a back-end window.

Thin ctypes layer over the host C++ entry points of libcmaxhip.so (csrc/cmx_trajinit.cpp); names follow the
reference:

  integrateAngVel        PoseGraphOptimizer::integrateAngVel       src/backend/pose_graph_optimizer.cpp:191-222
  Trajectory             Linear/CubicTrajectory                    src/backend/trajectory.cpp, include/backend/trajectory.h
    .generateCtrlPoses   :205-214 / :480-489      .pushbackCtrlPoses   :81-84 / :324-327
    .incrementalUpdate   :221-238 / :491-499      .evaluate            :86-110 / :329-355
    .temp_window         what CopyAndIncrementalUpdate (:240-263 / :501-522) hands the event warper
  bearing_lut            CMaxSLAM::precomputeBearingVectors        src/cmax_slam.cpp:106-120

Nothing here touches the GPU; quaternions are (x, y, z, w) float64, stamps int64 ns.
"""
import ctypes as C

import numpy as np

from . import _lib
from .evaluator import CmaxHipError


def _chk(rc, what):
    if rc != _lib.OK:
        raise CmaxHipError(rc, "%s: %s" % (what, _lib.lib().cmx_status_string(rc).decode()))


def _d(a):
    return a.ctypes.data_as(_lib.c_dp)


def _i64(a):
    return a.ctypes.data_as(_lib.c_i64p)


def integrateAngVel(pose_latest, ang_vel_subset, ang_vel_prev, first_time_window):
    """pose_latest = (t_ns, quat); ang_vel_subset = (t_ns[n], omega[n,3]); ang_vel_prev = (t_ns, omega).
    Returns ((pose_t_ns[m], pose_quat[m,4]), ang_vel_prev')."""
    t = np.ascontiguousarray(ang_vel_subset[0], np.int64)
    w = np.ascontiguousarray(ang_vel_subset[1], np.float64).reshape(-1, 3)
    if len(t) != len(w):
        raise ValueError("ang_vel_subset: stamps and vectors differ in length")
    q0 = np.ascontiguousarray(pose_latest[1], np.float64)
    prev_t = np.array([int(ang_vel_prev[0])], np.int64)
    prev_w = np.array(ang_vel_prev[1], np.float64, copy=True)
    out_t = np.zeros(max(len(t), 1), np.int64)
    out_q = np.zeros((max(len(t), 1), 4))
    m = C.c_int(0)
    _chk(_lib.lib().cmx_integrate_ang_vel(len(t), _i64(t), _d(w), int(pose_latest[0]), _d(q0), _i64(prev_t), _d(prev_w),
                                          int(bool(first_time_window)), _i64(out_t), _d(out_q), C.byref(m)),
         "cmx_integrate_ang_vel")
    return (out_t[:m.value].copy(), out_q[:m.value].copy()), (int(prev_t[0]), prev_w)


def bearing_lut(W, H, K, D=None, R=None, P=None):
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    opt = [None if a is None else np.ascontiguousarray(a, np.float64).reshape(n) for a, n in ((D, 5), (R, 9), (P, 12))]
    lut = np.zeros((H, W, 3))
    _chk(_lib.lib().cmx_bearing_lut(int(W), int(H), _d(K), *[None if a is None else _d(a) for a in opt], _d(lut)),
         "cmx_bearing_lut")
    return lut


class Trajectory:
    """Uniform cumulative B-spline on SO(3): spline_degree 1 -> order 2 (linear), 3 -> order 4 (cubic)."""

    def __init__(self, spline_degree, t_beg_ns, dt_knots):
        if spline_degree not in (1, 3):
            raise ValueError("spline_degree must be 1 or 3")
        self.order = 2 if spline_degree == 1 else 4
        self.t_beg_ns = int(t_beg_ns)                       # config.t_beg.toNSec()
        self.t_beg = _sec(self.t_beg_ns)                    # t_beg_ = config.t_beg.toSec()
        self.dt_knots = float(dt_knots)
        self.dt_ns = int(1e9 * self.dt_knots)               # int64_t(1e9 * config.dt_knots)
        self.knots = np.zeros((0, 4))

    def size(self):
        return len(self.knots)

    def generateCtrlPoses(self, poses, t_beg_ns, t_end_ns):
        """poses = (t_ns[m], quat[m,4]) -> new control poses fitted over [t_beg, t_end]."""
        n = _lib.lib().cmx_num_ctrl_poses(self.order, int(t_beg_ns), int(t_end_ns), self.dt_knots)
        t = np.ascontiguousarray(poses[0], np.int64)
        q = np.ascontiguousarray(poses[1], np.float64).reshape(-1, 4)
        out = np.zeros((max(n, 0), 4))
        _chk(_lib.lib().cmx_fit_ctrl_poses(self.order, len(t), _i64(t), _d(q), _sec(int(t_beg_ns)), self.dt_knots, n,
                                           _d(out)), "cmx_fit_ctrl_poses")
        return out

    def pushbackCtrlPoses(self, cps):
        self.knots = np.ascontiguousarray(np.vstack([self.knots, np.asarray(cps, np.float64).reshape(-1, 4)]))

    def incrementalUpdate(self, drotv, idx_beg):
        d = np.ascontiguousarray(drotv, np.float64).reshape(-1)
        _chk(_lib.lib().cmx_traj_incremental_update(self.size(), _d(self.knots), int(idx_beg), len(d), _d(d)),
             "cmx_traj_incremental_update")

    def evaluate(self, t_ns):
        q = np.zeros(4)
        _chk(_lib.lib().cmx_traj_evaluate(self.order, self.size(), _d(self.knots), self.t_beg_ns, self.dt_ns, int(t_ns),
                                          _d(q)), "cmx_traj_evaluate")
        return q

    def temp_window(self, idx_traj_beg):
        """(knots[idx_traj_beg:], start_ns, dt_ns) of the temporary trajectory the cost functor warps with; start_ns
        keeps the reference's double -> ns truncation (trajectory.cpp:255-256)."""
        start_ns = _lib.lib().cmx_traj_temp_start_ns(self.t_beg, int(idx_traj_beg), self.dt_knots)
        return np.ascontiguousarray(self.knots[idx_traj_beg:]), int(start_ns), self.dt_ns


def _sec(t_ns):
    """ros::Time::toSec()"""
    return float(t_ns // 1000000000) + 1e-9 * float(t_ns % 1000000000)
