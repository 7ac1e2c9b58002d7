"""ctypes binding of the CPU oracle (oracle/liboracle.so) and of the compiled-reference Basalt shim.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under cmax_slam_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

VARIANCE, MEAN_SQUARE, GRADIENT_MAGNITUDE = 0, 1, 2

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_u16p = C.POINTER(C.c_uint16)
c_i64p = C.POINTER(C.c_int64)


class FeCfg(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("lut", c_dp), ("fx", C.c_double), ("fy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("batch", C.c_int), ("sigma", C.c_double),
                ("measure", C.c_int)]


class BeCfg(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("lut", c_dp), ("Wp", C.c_int), ("Hp", C.c_int),
                ("batch", C.c_int), ("sample_rate", C.c_int), ("sigma", C.c_double), ("measure", C.c_int),
                ("order", C.c_int), ("K", C.c_int), ("start_ns", C.c_int64), ("dt_ns", C.c_int64),
                ("num_fixed", C.c_int), ("t_next_win_beg_ns", C.c_int64)]


class BeState(C.Structure):
    _fields_ = [("IL_old", c_fp), ("IL_new", c_fp), ("IL", c_fp), ("IG", c_fp), ("IGp", c_fp),
                ("alpha", C.c_double), ("first_iter", C.c_int)]


def build(force=False):
    """Compile liboracle.so (and _ref/libbasalt_ref.so when the reference tree is present)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean"])
    # make is incremental: a stale library after a source edit is rebuilt, an up-to-date one costs milliseconds
    subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so", "liboracle_mt.so", "liboracle_f64.so"])
    if os.path.isdir("/root/reference/thirdparty/basalt-headers"):
        ref = os.path.join(_HERE, "_ref", "libbasalt_ref.so")
        if force or not os.path.exists(ref):
            subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_gauss_ksize.restype = C.c_int
        L.orc_gauss_ksize.argtypes = [C.c_double]
        L.orc_gauss_kernel.argtypes = [C.c_int, C.c_double, c_fp]
        L.orc_gaussian_blur.argtypes = [c_fp, C.c_int, C.c_int, C.c_int, C.c_double]
        L.orc_contrast.restype = C.c_double
        L.orc_contrast.argtypes = [c_fp, C.c_int, C.POINTER(c_fp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_dp]
        L.orc_time_batch_ns.restype = C.c_int64
        L.orc_time_batch_ns.argtypes = [C.c_int64, C.c_int64]
        L.orc_time_to_sec.restype = C.c_double
        L.orc_time_to_sec.argtypes = [C.c_int64]
        L.orc_fe_iwe.argtypes = [C.POINTER(FeCfg), C.c_int64, c_u16p, c_u16p, c_i64p, C.c_int64, c_dp, c_fp, c_fp, C.c_int]
        L.orc_fe_eval.argtypes = [C.POINTER(FeCfg), C.c_int64, c_u16p, c_u16p, c_i64p, C.c_int64, c_dp, c_dp, c_dp]
        L.orc_so3_spline_eval.argtypes = [C.c_int, C.c_int, c_dp, C.c_int64, C.c_int64, C.c_int64, c_dp, c_dp, c_dp,
                                          C.POINTER(C.c_int)]
        L.orc_so3_left_update.argtypes = [c_dp, c_dp]
        L.orc_so3_exp.argtypes = [c_dp, c_dp]
        L.orc_so3_log.argtypes = [c_dp, c_dp]
        L.orc_traj_temp_start_ns.restype = C.c_int64
        L.orc_traj_temp_start_ns.argtypes = [C.c_double, C.c_int, C.c_double]
        L.orc_be_iwe.argtypes = [C.POINTER(BeCfg), C.POINTER(BeState), C.c_int64, c_u16p, c_u16p, c_i64p, c_dp, c_fp, c_fp]
        L.orc_be_eval.argtypes = [C.POINTER(BeCfg), C.POINTER(BeState), C.c_int64, c_u16p, c_u16p, c_i64p, c_dp, c_dp,
                                  c_dp, c_dp, c_fp]
        L.orc_be_warp_batch.restype = C.c_int
        L.orc_be_warp_batch.argtypes = [C.POINTER(BeCfg), C.POINTER(BeState), c_u16p, c_u16p, c_i64p, C.c_int64, C.c_int64,
                                        c_dp, c_fp]
        L.orc_be_alpha.restype = C.c_double
        L.orc_be_alpha.argtypes = [c_fp, c_fp, C.c_int]
        L.orc_equirect_project.argtypes = [C.c_int, C.c_int, c_dp, c_dp, c_fp]
        c_u8p = C.POINTER(C.c_uint8)
        L.orc_be_update_ig.argtypes = [c_fp, c_fp, c_u8p, C.c_int, C.c_int]
        L.orc_be_mark_visited.argtypes = [C.c_int, C.c_int, c_dp, C.c_int, C.c_int, c_dp, C.c_int, c_u8p]
        L.orc_so3_mul.argtypes = [c_dp, c_dp, c_dp]
        L.orc_integrate_ang_vel.restype = C.c_int
        L.orc_integrate_ang_vel.argtypes = [C.c_int, c_i64p, c_dp, C.c_int64, c_dp, C.POINTER(C.c_int64), c_dp, C.c_int,
                                            c_i64p, c_dp]
        L.orc_num_ctrl_poses.restype = C.c_int
        L.orc_num_ctrl_poses.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_double]
        L.orc_fit_ctrl_poses.restype = C.c_int
        L.orc_fit_ctrl_poses.argtypes = [C.c_int, C.c_int, c_i64p, c_dp, C.c_double, C.c_double, C.c_int, c_dp]
        L.orc_fullpiv_qr_solve.restype = C.c_int
        L.orc_fullpiv_qr_solve.argtypes = [C.c_int, C.c_int, c_dp, c_dp, c_dp]
        L.orc_bearing_lut.argtypes = [C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp]
        _LIB = L
    return _LIB


_MT = None


def mt_lib():
    """The all-cores OpenMP variant (liboracle_mt.so): NOT the reference, which is single-threaded."""
    global _MT
    if _MT is None:
        so = os.path.join(_HERE, "liboracle_mt.so")
        if not os.path.exists(so):
            build()
        M = C.CDLL(so)
        M.orc_mt_max_threads.restype = C.c_int
        M.orc_fe_eval_mt.argtypes = [C.POINTER(FeCfg), C.c_int64, c_u16p, c_u16p, c_i64p, C.c_int64, c_dp, C.c_int, c_dp, c_dp]
        M.orc_be_eval_mt.argtypes = [C.POINTER(BeCfg), C.POINTER(BeState), C.c_int64, c_u16p, c_u16p, c_i64p, c_dp, c_dp,
                                     C.c_int, c_dp, c_dp]
        _MT = M
    return _MT


def ref_lib():
    """The reference's vendored Basalt compiled from /root/reference (None when not built)."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libbasalt_ref.so")
        if not os.path.exists(so):
            return None
        R = C.CDLL(so)
        R.ref_so3_spline_eval.argtypes = [C.c_int, C.c_int, c_dp, C.c_int64, C.c_int64, C.c_int64, c_dp, c_dp, c_dp,
                                          C.POINTER(C.c_int)]
        R.ref_so3_exp.argtypes = [c_dp, c_dp]
        R.ref_so3_log.argtypes = [c_dp, c_dp]
        R.ref_so3_left_update.argtypes = [c_dp, c_dp]
        R.ref_so3_mul.argtypes = [c_dp, c_dp, c_dp]
        R.ref_fullpiv_qr_solve.restype = C.c_int
        R.ref_fullpiv_qr_solve.argtypes = [C.c_int, C.c_int, c_dp, c_dp, c_dp]
        _REF = R
    return _REF


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _fp(a):
    return a.ctypes.data_as(c_fp) if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ----------------------------------------------------------------------------- OpenCV-semantics helpers
def gaussian_blur(img, sigma):
    a = np.array(img, dtype=np.float32, order="C", copy=True)
    H, W = a.shape[:2]
    cn = 1 if a.ndim == 2 else a.shape[2]
    lib().orc_gaussian_blur(_fp(a), W, H, cn, float(sigma))
    return a


def gauss_kernel(sigma):
    n = lib().orc_gauss_ksize(float(sigma))
    k = np.zeros(n, np.float32)
    lib().orc_gauss_kernel(n, float(sigma), _fp(k))
    return k


def contrast(img, planes, measure, want_grad=True):
    """img: HxW fp32; planes: PxHxW fp32 (contiguous planes)."""
    img = _c(img, np.float32)
    H, W = img.shape
    planes = _c(planes, np.float32) if planes is not None else None
    P = 0 if planes is None else planes.shape[0]
    arr = (c_fp * max(P, 1))()
    for k in range(P):
        arr[k] = planes[k].ctypes.data_as(c_fp)
    g = np.zeros(max(P, 1))
    val = lib().orc_contrast(_fp(img), H * W, arr, 1, P, int(measure), W, H, _dp(g) if (want_grad and P) else None)
    return val, (g[:P] if want_grad and P else None)


def time_batch_ns(t0, t1):
    return int(lib().orc_time_batch_ns(int(t0), int(t1)))


# ----------------------------------------------------------------------------- front end
class Frontend:
    """Holds one event packet + camera; mirrors the state AngVelEstimator hands to local_contrast_fdf."""

    def __init__(self, W, H, lut, fx, fy, cx, cy, batch=100, sigma=1.0, measure=VARIANCE):
        self.lut = _c(lut, np.float64).reshape(-1)
        assert self.lut.size == W * H * 3
        self.cfg = FeCfg(W, H, _dp(self.lut), fx, fy, cx, cy, batch, sigma, measure)
        self.W, self.H = W, H

    def set_packet(self, x, y, t_ns, t_ref_ns):
        self.x, self.y, self.t = _c(x, np.uint16), _c(y, np.uint16), _c(t_ns, np.int64)
        self.t_ref = int(t_ref_ns)

    def _ev(self):
        return (len(self.x), self.x.ctypes.data_as(c_u16p), self.y.ctypes.data_as(c_u16p),
                self.t.ctypes.data_as(c_i64p), self.t_ref)

    def iwe(self, omega, deriv=False, blur=True):
        om = _c(omega, np.float64)
        iwe = np.zeros((self.H, self.W), np.float32)
        d = np.zeros((self.H, self.W, 3), np.float32) if deriv else None
        rc = lib().orc_fe_iwe(C.byref(self.cfg), *self._ev(), _dp(om), _fp(iwe), _fp(d), int(blur))
        if rc:
            raise ValueError("oracle front-end: invalid event coordinates")
        return (iwe, d) if deriv else iwe

    def eval(self, omega, want_grad=True):
        """returns (contrast, grad[3] or None) -- the GSL glue returns the negatives."""
        om = _c(omega, np.float64)
        c = C.c_double()
        g = np.zeros(3)
        rc = lib().orc_fe_eval(C.byref(self.cfg), *self._ev(), _dp(om), C.byref(c), _dp(g) if want_grad else None)
        if rc:
            raise ValueError("oracle front-end: invalid event coordinates")
        return c.value, (g if want_grad else None)

    def eval_allcores(self, omega, want_grad=True, nthreads=0):
        """The same evaluation on `nthreads` host threads (0 = all): thread-private images, NOT the reference."""
        om = _c(omega, np.float64)
        c = C.c_double()
        g = np.zeros(3)
        rc = mt_lib().orc_fe_eval_mt(C.byref(self.cfg), *self._ev(), _dp(om), int(nthreads), C.byref(c),
                                     _dp(g) if want_grad else None)
        if rc:
            raise ValueError("oracle front-end (all cores) failed rc=%d" % rc)
        return c.value, (g if want_grad else None)



# ----------------------------------------------------------------------------- spline
def spline_eval(order, knots_xyzw, start_ns, dt_ns, t_ns, jac=True, use_ref=False):
    k = _c(knots_xyzw, np.float64).reshape(-1, 4)
    q = np.zeros(4)
    R = np.zeros(9)
    J = np.zeros(9 * order)
    idx = C.c_int(-1)
    fn = ref_lib().ref_so3_spline_eval if use_ref else lib().orc_so3_spline_eval
    rc = fn(order, k.shape[0], _dp(k), int(start_ns), int(dt_ns), int(t_ns), _dp(q), _dp(R), _dp(J) if jac else None,
            C.byref(idx))
    if rc:
        raise ValueError("spline evaluate: time outside the knot range (Basalt would assert)")
    return q, R.reshape(3, 3), (J.reshape(order, 3, 3) if jac else None), idx.value


def so3_exp(w, use_ref=False):
    w = _c(w, np.float64)
    q = np.zeros(4)
    (ref_lib().ref_so3_exp if use_ref else lib().orc_so3_exp)(_dp(w), _dp(q))
    return q


def so3_log(q, use_ref=False):
    q = _c(q, np.float64)
    w = np.zeros(3)
    (ref_lib().ref_so3_log if use_ref else lib().orc_so3_log)(_dp(q), _dp(w))
    return w


def left_update(knots_xyzw, drotv, num_fixed, use_ref=False):
    k = np.array(knots_xyzw, dtype=np.float64, order="C", copy=True).reshape(-1, 4)
    d = _c(drotv, np.float64).reshape(-1, 3)
    fn = ref_lib().ref_so3_left_update if use_ref else lib().orc_so3_left_update
    for i in range(num_fixed, k.shape[0]):
        fn(_dp(k[i]), _dp(d[i - num_fixed]))
    return k


def so3_mul(a, b, use_ref=False):
    a, b = _c(a, np.float64), _c(b, np.float64)
    out = np.zeros(4)
    (ref_lib().ref_so3_mul if use_ref else lib().orc_so3_mul)(_dp(a), _dp(b), _dp(out))
    return out


# ----------------------------------------------------------------------------- control-pose initialisation
def fullpiv_qr_solve(A, b, use_ref=False):
    """x = A.fullPivHouseholderQr().solve(b) and Eigen's rank (restated, or the vendored Eigen itself)."""
    A = _c(A, np.float64)
    b = _c(b, np.float64)
    x = np.zeros(A.shape[1])
    fn = ref_lib().ref_fullpiv_qr_solve if use_ref else lib().orc_fullpiv_qr_solve
    rank = fn(A.shape[0], A.shape[1], _dp(A), _dp(b), _dp(x))
    return x, rank


def integrate_ang_vel(t_ns, ang_vel, pose_t_ns, pose_quat, prev_t_ns, prev_ang_vel, first_time_window):
    """Returns (pose_t_ns[m], pose_quat[m,4], prev_t_ns', prev_ang_vel')."""
    t = _c(t_ns, np.int64)
    w = _c(ang_vel, np.float64).reshape(-1, 3)
    pq = _c(pose_quat, np.float64)
    pt = C.c_int64(int(prev_t_ns))
    pw = np.array(prev_ang_vel, dtype=np.float64, copy=True)
    ot = np.zeros(max(len(t), 1), np.int64)
    oq = np.zeros((max(len(t), 1), 4))
    m = lib().orc_integrate_ang_vel(len(t), t.ctypes.data_as(c_i64p), _dp(w), int(pose_t_ns), _dp(pq), C.byref(pt), _dp(pw),
                                    int(bool(first_time_window)), ot.ctypes.data_as(c_i64p), _dp(oq))
    return ot[:m].copy(), oq[:m].copy(), pt.value, pw


def num_ctrl_poses(order, t_beg_ns, t_end_ns, dt_knots):
    return int(lib().orc_num_ctrl_poses(int(order), int(t_beg_ns), int(t_end_ns), float(dt_knots)))


def fit_ctrl_poses(order, t_ns, quats, t_beg, dt_knots, num_cps):
    t = _c(t_ns, np.int64)
    q = _c(quats, np.float64).reshape(-1, 4)
    out = np.zeros((num_cps, 4))
    rc = lib().orc_fit_ctrl_poses(int(order), len(t), t.ctypes.data_as(c_i64p), _dp(q), float(t_beg), float(dt_knots),
                                  int(num_cps), _dp(out))
    if rc:
        raise ValueError("fitCtrlPoses: the reference's CHECK / Eigen index assertion would fire")
    return out


def bearing_lut(W, H, K, D, R=None, P=None):
    K = _c(K, np.float64).reshape(9)
    D = _c(D, np.float64).reshape(5)
    R = _c(np.eye(3) if R is None else R, np.float64).reshape(9)
    if P is None:
        P = np.concatenate([K.reshape(3, 3), np.zeros((3, 1))], axis=1)
    P = _c(P, np.float64).reshape(12)
    lut = np.zeros((H, W, 3))
    lib().orc_bearing_lut(int(W), int(H), _dp(K), _dp(D), _dp(R), _dp(P), _dp(lut))
    return lut


def traj_temp_start_ns(t_beg, idx_traj_beg, dt_knots):
    return int(lib().orc_traj_temp_start_ns(float(t_beg), int(idx_traj_beg), float(dt_knots)))


# ----------------------------------------------------------------------------- back end
class Backend:
    """Mirrors the EventWarper state PoseGraphOptimizer sets up before a window solve."""

    def __init__(self, W, H, lut, Wp, Hp, order, batch=100, sample_rate=1, sigma=1.0, measure=VARIANCE):
        self.lut = _c(lut, np.float64).reshape(-1)
        assert self.lut.size == W * H * 3
        self.W, self.H, self.Wp, self.Hp, self.order = W, H, Wp, Hp, order
        self.batch, self.sample_rate, self.sigma, self.measure = batch, sample_rate, sigma, measure
        z = lambda: np.zeros((Hp, Wp), np.float32)
        self.IL_old, self.IL_new, self.IL, self.IG, self.IGp = z(), z(), z(), z(), z()
        self.state = BeState(_fp(self.IL_old), _fp(self.IL_new), _fp(self.IL), _fp(self.IG), _fp(self.IGp), 0.0, 1)

    def set_window(self, x, y, t_ns, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns, IG=None):
        self.x, self.y, self.t = _c(x, np.uint16), _c(y, np.uint16), _c(t_ns, np.int64)
        self.knots = _c(knots_xyzw, np.float64).reshape(-1, 4).copy()
        self.K = self.knots.shape[0]
        self.cfg = BeCfg(self.W, self.H, _dp(self.lut), self.Wp, self.Hp, self.batch, self.sample_rate, self.sigma,
                         self.measure, self.order, self.K, int(start_ns), int(dt_ns), int(num_fixed),
                         int(t_next_win_beg_ns))
        self.num_fixed = num_fixed
        if IG is not None:
            self.IG[...] = IG
        else:
            self.IG[...] = 0
        self.state.first_iter = 1  # setFirstIter(true), pose_graph_optimizer.cpp:293
        self.state.alpha = 0.0

    @property
    def alpha(self):
        return self.state.alpha

    # --- global-map upkeep (event_pano_warper.cpp:81-126)
    def update_ig(self, max_update_times):
        if not hasattr(self, "update_times"):
            self.update_times = np.zeros((self.Hp, self.Wp), np.uint8)
        lib().orc_be_update_ig(_fp(self.IG), _fp(self.IL_old), self.update_times.ctypes.data_as(C.POINTER(C.c_uint8)),
                               self.IG.size, int(max_update_times))

    def mark_visited(self, quat_xyzw, radius=3):
        if not hasattr(self, "update_times"):
            self.update_times = np.zeros((self.Hp, self.Wp), np.uint8)
        q = _c(quat_xyzw, np.float64)
        lib().orc_be_mark_visited(self.W, self.H, _dp(self.lut), self.Wp, self.Hp, _dp(q), int(radius),
                                  self.update_times.ctypes.data_as(C.POINTER(C.c_uint8)))

    def _ev(self):
        return (len(self.x), self.x.ctypes.data_as(c_u16p), self.y.ctypes.data_as(c_u16p),
                self.t.ctypes.data_as(c_i64p))

    def iwe(self, drotv, planes=False):
        """IWE (blurred) and optionally the P derivative planes (blurred) at the updated trajectory."""
        k = left_update(self.knots, drotv, self.num_fixed)
        P = 3 * (self.K - self.num_fixed)
        iwe = np.zeros((self.Hp, self.Wp), np.float32)
        pl = np.zeros((P, self.Hp, self.Wp), np.float32) if planes else None
        rc = lib().orc_be_iwe(C.byref(self.cfg), C.byref(self.state), *self._ev(), _dp(k), _fp(iwe), _fp(pl))
        if rc:
            raise ValueError("oracle back-end failed rc=%d" % rc)
        return (iwe, pl) if planes else iwe

    def accumulate_raw(self, drotv, planes=False):
        """The vote loop alone (event_pano_warper.cpp:173-196): zero IL_old / IL_new (+ the P derivative planes), warp
        and accumulate THIS object's events batch by batch -- no cv::add, no alpha, no blur.  What one rank of a sharded
        window holds before the exchange (tests/test_dist_gloo.py).  Returns (IL_old, IL_new, planes or None)."""
        k = left_update(self.knots, drotv, self.num_fixed)
        P = 3 * (self.K - self.num_fixed)
        n = len(self.x)
        self.IL_old[...] = 0
        self.IL_new[...] = 0
        pl = np.zeros((P, self.Hp, self.Wp), np.float32) if planes else None
        beg = 0
        while beg < n - 1:   # :188  a trailing single-event batch is skipped
            end = beg + self.batch if n - beg > self.batch else n
            rc = lib().orc_be_warp_batch(C.byref(self.cfg), C.byref(self.state), self.x.ctypes.data_as(c_u16p),
                                         self.y.ctypes.data_as(c_u16p), self.t.ctypes.data_as(c_i64p), beg, end, _dp(k),
                                         _fp(pl))
            if rc:
                raise ValueError("oracle back-end failed rc=%d" % rc)
            beg += self.batch
        return self.IL_old, self.IL_new, pl

    def eval(self, drotv, want_grad=True):
        d = _c(drotv, np.float64).reshape(-1)
        P = 3 * (self.K - self.num_fixed)
        assert d.size == P
        c = C.c_double()
        g = np.zeros(max(P, 1))
        rc = lib().orc_be_eval(C.byref(self.cfg), C.byref(self.state), *self._ev(), _dp(self.knots), _dp(d),
                               C.byref(c), _dp(g) if want_grad else None, None)
        if rc:
            raise ValueError("oracle back-end failed rc=%d" % rc)
        return c.value, (g[:P] if want_grad else None)

    def eval_allcores(self, drotv, want_grad=True, nthreads=0):
        """The same evaluation on `nthreads` host threads (0 = all): thread-private planes, NOT the reference."""
        d = _c(drotv, np.float64).reshape(-1)
        P = 3 * (self.K - self.num_fixed)
        assert d.size == P
        c = C.c_double()
        g = np.zeros(max(P, 1))
        rc = mt_lib().orc_be_eval_mt(C.byref(self.cfg), C.byref(self.state), *self._ev(), _dp(self.knots), _dp(d),
                                     int(nthreads), C.byref(c), _dp(g) if want_grad else None)
        if rc:
            raise ValueError("oracle back-end (all cores) failed rc=%d" % rc)
        return c.value, (g[:P] if want_grad else None)


def equirect_project(Wp, Hp, P, jac=True):
    P = _c(P, np.float64)
    px = np.zeros(2)
    J = np.zeros(6, np.float32)
    lib().orc_equirect_project(Wp, Hp, _dp(P), _dp(px), _fp(J) if jac else None)
    return px, (J.reshape(2, 3) if jac else None)


# ----------------------------------------------------------------------------- exact (all-fp64) arbiter
_EXACT = None


def exact_lib():
    """liboracle_f64.so: the oracle's sources with float := double (oracle/exact_f64.c)."""
    global _EXACT
    if _EXACT is None:
        so = os.path.join(_HERE, "liboracle_f64.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle_f64.so"])
        _EXACT = C.CDLL(so)
    return _EXACT


class _BeState64(C.Structure):
    _fields_ = [("IL_old", c_dp), ("IL_new", c_dp), ("IL", c_dp), ("IG", c_dp), ("IGp", c_dp), ("alpha", C.c_double),
                ("first_iter", C.c_int)]


class BackendExact:
    """Backend's eval() in exact (fp64) arithmetic; same constructor / set_window arguments."""

    def __init__(self, W, H, lut, Wp, Hp, order, batch=100, sample_rate=1, sigma=1.0, measure=VARIANCE):
        self.lut = _c(lut, np.float64).reshape(-1)
        self.W, self.H, self.Wp, self.Hp, self.order = W, H, Wp, Hp, order
        self.batch, self.sample_rate, self.sigma, self.measure = batch, sample_rate, sigma, measure
        self.bufs = [np.zeros(Wp * Hp) for _ in range(5)]
        self.state = _BeState64(*[_dp(b) for b in self.bufs], 0.0, 1)

    def set_window(self, x, y, t_ns, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns, IG=None):
        self.x, self.y, self.t = _c(x, np.uint16), _c(y, np.uint16), _c(t_ns, np.int64)
        self.knots = _c(knots_xyzw, np.float64).reshape(-1, 4).copy()
        self.K, self.num_fixed = self.knots.shape[0], num_fixed
        self.cfg = BeCfg(self.W, self.H, _dp(self.lut), self.Wp, self.Hp, self.batch, self.sample_rate, self.sigma,
                         self.measure, self.order, self.K, int(start_ns), int(dt_ns), int(num_fixed), int(t_next_win_beg_ns))
        self.bufs[3][:] = 0 if IG is None else np.asarray(IG, np.float64).reshape(-1)
        self.state.first_iter = 1
        self.state.alpha = 0.0

    @property
    def alpha(self):
        return self.state.alpha

    def eval(self, drotv, want_grad=True):
        d = _c(drotv, np.float64).reshape(-1)
        P = 3 * (self.K - self.num_fixed)
        c = C.c_double()
        g = np.zeros(max(P, 1))
        rc = exact_lib().orc_be_eval(C.byref(self.cfg), C.byref(self.state), C.c_int64(len(self.x)),
                                     self.x.ctypes.data_as(c_u16p), self.y.ctypes.data_as(c_u16p),
                                     self.t.ctypes.data_as(c_i64p), _dp(self.knots), _dp(d), C.byref(c),
                                     _dp(g) if want_grad else None, None)
        if rc:
            raise ValueError("exact back-end failed rc=%d" % rc)
        return c.value, (g[:P] if want_grad else None)


class FrontendExact:
    """Frontend's eval() in exact (fp64) arithmetic."""

    def __init__(self, W, H, lut, fx, fy, cx, cy, batch=100, sigma=1.0, measure=VARIANCE):
        self.lut = _c(lut, np.float64).reshape(-1)
        self.cfg = FeCfg(W, H, _dp(self.lut), fx, fy, cx, cy, batch, sigma, measure)

    def set_packet(self, x, y, t_ns, t_ref_ns):
        self.x, self.y, self.t = _c(x, np.uint16), _c(y, np.uint16), _c(t_ns, np.int64)
        self.t_ref = int(t_ref_ns)

    def eval(self, omega, want_grad=True):
        om = _c(omega, np.float64)
        c = C.c_double()
        g = np.zeros(3)
        rc = exact_lib().orc_fe_eval(C.byref(self.cfg), C.c_int64(len(self.x)), self.x.ctypes.data_as(c_u16p),
                                     self.y.ctypes.data_as(c_u16p), self.t.ctypes.data_as(c_i64p), C.c_int64(self.t_ref),
                                     _dp(om), C.byref(c), _dp(g) if want_grad else None)
        if rc:
            raise ValueError("exact front-end failed rc=%d" % rc)
        return c.value, (g if want_grad else None)
