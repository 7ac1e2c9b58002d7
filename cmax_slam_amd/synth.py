"""Seeded synthetic event data for the benchmark configs.

Scene = short great-circle arcs ("edges") on the unit sphere, so motion-compensated IWEs are sharp and
atomic contention is realistic; 10 % uniformly random noise events; time-sorted int64-ns timestamps
starting at t0 = 1.0 s.  Polarity is omitted (the reference never reads it:
local_image_warped_events.cpp:148-151, event_pano_warper.cpp:300-310).
"""
from dataclasses import dataclass

import numpy as np
from scipy.spatial.transform import Rotation as Rot

SEED0 = 20240314
T0_NS = 1_000_000_000


def pinhole_lut(W, H, fx, fy, cx, cy):
    """Zero-distortion bearing LUT ((x-cx)/fx, (y-cy)/fy, 1), index y*W+x (cmax_slam.cpp:106-120)."""
    xs = (np.arange(W, dtype=np.float64) - cx) / fx
    ys = (np.arange(H, dtype=np.float64) - cy) / fy
    lut = np.empty((H, W, 3), np.float64)
    lut[..., 0] = xs[None, :]
    lut[..., 1] = ys[:, None]
    lut[..., 2] = 1.0
    return lut.reshape(-1, 3)


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


class _Scene:
    """n_arcs random great-circle arcs whose start lies within `cone` rad of one of axis_dirs."""

    def __init__(self, rng, n_arcs, axis_dirs, cone):
        k = rng.integers(0, len(axis_dirs), n_arcs)
        tilt = Rot.from_rotvec(_unit(rng.normal(size=(n_arcs, 3))) * (cone * np.sqrt(rng.random(n_arcs)))[:, None])
        self.a = tilt.apply(axis_dirs[k])
        self.tang = _unit(np.cross(self.a, rng.normal(size=(n_arcs, 3))))
        self.length = rng.uniform(0.05, 0.30, n_arcs)

    def sample(self, rng, n):
        which = rng.integers(0, len(self.length), n)
        s = rng.random(n) * self.length[which]
        return np.cos(s)[:, None] * self.a[which] + np.sin(s)[:, None] * self.tang[which]


def _project_pinhole(p, fx, fy, cx, cy, W, H):
    z = p[:, 2]
    ok = z > 1e-3
    u = np.where(ok, fx * p[:, 0] / np.where(ok, z, 1) + cx, -1)
    v = np.where(ok, fy * p[:, 1] / np.where(ok, z, 1) + cy, -1)
    xi, yi = np.rint(u).astype(np.int64), np.rint(v).astype(np.int64)
    ok &= (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
    return xi, yi, ok


@dataclass
class FrontendPacket:
    W: int
    H: int
    fx: float
    fy: float
    cx: float
    cy: float
    x: np.ndarray
    y: np.ndarray
    t_ns: np.ndarray
    t_ref_ns: int
    omega_true: np.ndarray
    batch: int = 100
    sigma: float = 1.0

    @property
    def lut(self):
        return pinhole_lut(self.W, self.H, self.fx, self.fy, self.cx, self.cy)


def frontend_packet(N, W, H, fx, fy, cx, cy, T=0.05, omega_true=(0.6, -0.9, 0.4), seed=SEED0, noise=0.10,
                    n_arcs=200):
    """One front-end event packet; the reference time is the packet's mid time (ang_vel_estimator.cpp:85-97)."""
    rng = np.random.default_rng(seed)
    omega = np.asarray(omega_true, np.float64)
    t_ref = T / 2
    cone = 1.15 * np.arctan(np.hypot(W / 2 / fx, H / 2 / fy))
    scene = _Scene(rng, n_arcs, np.array([[0, 0, 1.0]]), cone)
    n_sig_total = int(round(N * (1 - noise)))
    out_x, out_y, out_t = [], [], []
    need = n_sig_total
    while need > 0:
        m = int(need * 1.6) + 1024
        pts = scene.sample(rng, m)
        t = rng.random(m) * T
        # observed bearing at time t:  p_t = R(-omega (t - t_ref)) X_ref
        p_t = Rot.from_rotvec(-omega[None, :] * (t - t_ref)[:, None]).apply(pts)
        xi, yi, ok = _project_pinhole(p_t, fx, fy, cx, cy, W, H)
        out_x.append(xi[ok]); out_y.append(yi[ok]); out_t.append(t[ok])
        need -= int(ok.sum())
    x = np.concatenate(out_x)[:n_sig_total]
    y = np.concatenate(out_y)[:n_sig_total]
    t = np.concatenate(out_t)[:n_sig_total]
    n_noise = N - len(x)
    x = np.concatenate([x, rng.integers(0, W, n_noise)])
    y = np.concatenate([y, rng.integers(0, H, n_noise)])
    t = np.concatenate([t, rng.random(n_noise) * T])
    order = np.argsort(t, kind="stable")
    t_ns = T0_NS + np.floor(t[order] * 1e9).astype(np.int64)
    return FrontendPacket(W, H, fx, fy, cx, cy, x[order].astype(np.uint16), y[order].astype(np.uint16), t_ns,
                          T0_NS + int(round(t_ref * 1e9)), omega)


# ------------------------------------------------------------------ SO(3) cumulative B-spline (generation only)
def _blend_coeffs(order, u):
    u = np.asarray(u, np.float64)
    if order == 2:
        return np.stack([np.ones_like(u), u], -1)
    if order == 4:
        return np.stack([np.ones_like(u), (5 + 3 * u - 3 * u ** 2 + u ** 3) / 6, (1 + 3 * u + 3 * u ** 2 - 2 * u ** 3) / 6,
                         u ** 3 / 6], -1)
    raise ValueError("order must be 2 or 4")


def spline_rotations(order, knots_xyzw, start_ns, dt_ns, t_ns):
    """Vectorised So3Spline<order>::evaluate value (so3_spline.h:218-274), for data generation."""
    st = np.asarray(t_ns, np.int64) - start_ns
    s = st // dt_ns
    u = (st % dt_ns) / float(dt_ns)
    co = _blend_coeffs(order, u)
    K = Rot.from_quat(knots_xyzw)
    res = K[s]
    for i in range(order - 1):
        d = (K[s + i].inv() * K[s + i + 1]).as_rotvec()
        res = res * Rot.from_rotvec(d * co[:, i + 1][:, None])
    return res


@dataclass
class BackendWindow:
    W: int
    H: int
    fx: float
    fy: float
    cx: float
    cy: float
    Wp: int
    Hp: int
    order: int
    x: np.ndarray
    y: np.ndarray
    t_ns: np.ndarray
    knots_true: np.ndarray   # K x 4 (x,y,z,w)
    knots_init: np.ndarray   # perturbed start
    start_ns: int
    dt_ns: int
    num_fixed: int
    t_next_win_beg_ns: int
    batch: int = 100
    sample_rate: int = 1
    sigma: float = 1.0

    @property
    def lut(self):
        return pinhole_lut(self.W, self.H, self.fx, self.fy, self.cx, self.cy)

    @property
    def K(self):
        return self.knots_true.shape[0]

    @property
    def P(self):
        return 3 * (self.K - self.num_fixed)


def backend_window(N, W, H, fx, fy, cx, cy, Wp, Hp, order, K, num_fixed, T, dt_knots=0.05, seed=SEED0, noise=0.10,
                   n_arcs=200, knot_sigma=0.02, init_sigma=0.01, win_stride=None, slab=None):
    """One back-end BA window: events over [t0, t0+T), a smooth random SO(3) spline with K knots
    (cumulative exp(N(0, knot_sigma^2 I))) and a perturbed start.

    slab = (r, n): only the N events of time slab r of n, i.e. with t in [r T/n, (r+1) T/n) -- same trajectory, scene and
    window description as the other slabs, events drawn from a slab-specific stream.  Concatenating the n slabs gives
    one time-sorted window of n*N events: a rank of a sharded run generates its own slab only (config 4)."""
    rng = np.random.default_rng(seed)
    dt_ns = int(round(dt_knots * 1e9))
    start_ns = T0_NS
    assert T <= (K - order + 1) * dt_knots + 1e-12, "window longer than the spline support"
    q = Rot.identity()
    ks = []
    for _ in range(K):
        ks.append(q.as_quat())
        q = Rot.from_rotvec(rng.normal(0, knot_sigma, 3)) * q
    knots = np.array(ks)
    pert = rng.normal(0, init_sigma, (K, 3))
    pert[:num_fixed] = 0
    knots_init = (Rot.from_rotvec(pert) * Rot.from_quat(knots)).as_quat()

    # scene directions around the camera's optical axis along the path
    mids = spline_rotations(order, knots, start_ns, dt_ns, start_ns + (np.linspace(0, T, 9)[:-1] * 1e9).astype(np.int64))
    axes = mids.apply(np.array([0, 0, 1.0]))
    cone = 1.15 * np.arctan(np.hypot(W / 2 / fx, H / 2 / fy))
    scene = _Scene(rng, n_arcs, axes, cone)
    n_sig_total = int(round(N * (1 - noise)))
    t_lo, t_span = 0.0, T
    if slab is not None:
        r, n_slabs = slab
        assert 0 <= r < n_slabs
        t_lo, t_span = T * r / n_slabs, T / n_slabs
        rng = np.random.default_rng([seed, 7919 + r])  # events only: everything above came from the window's own stream
    # camera orientation on a 5 us grid (generation only; 5e-6 rad at 1 rad/s, far below a pixel)
    grid_ns = 5_000
    n_grid = int(T * 1e9) // grid_ns + 2
    grid_R = spline_rotations(order, knots, start_ns, dt_ns,
                              np.minimum(start_ns + np.arange(n_grid, dtype=np.int64) * grid_ns,
                                         start_ns + int(T * 1e9) - 1)).as_matrix()
    xs, ys, ts = [], [], []
    need = n_sig_total
    while need > 0:
        m = min(int(need * 1.8) + 1024, 4_000_000)
        pts = scene.sample(rng, m)
        t = t_lo + rng.random(m) * t_span
        tn = start_ns + np.floor(t * 1e9).astype(np.int64)
        Rt = grid_R[(tn - start_ns + grid_ns // 2) // grid_ns]
        p_cam = np.einsum("nji,nj->ni", Rt, pts)  # e_ray_w = R * e_ray_cam  (event_pano_warper.cpp:269)
        xi, yi, ok = _project_pinhole(p_cam, fx, fy, cx, cy, W, H)
        xs.append(xi[ok]); ys.append(yi[ok]); ts.append(tn[ok])
        need -= int(ok.sum())
    x = np.concatenate(xs)[:n_sig_total]
    y = np.concatenate(ys)[:n_sig_total]
    tn = np.concatenate(ts)[:n_sig_total]
    n_noise = N - len(x)
    x = np.concatenate([x, rng.integers(0, W, n_noise)])
    y = np.concatenate([y, rng.integers(0, H, n_noise)])
    tn = np.concatenate([tn, start_ns + np.floor((t_lo + rng.random(n_noise) * t_span) * 1e9).astype(np.int64)])
    o = np.argsort(tn, kind="stable")
    stride = T / 2 if win_stride is None else win_stride
    return BackendWindow(W, H, fx, fy, cx, cy, Wp, Hp, order, x[o].astype(np.uint16), y[o].astype(np.uint16), tn[o],
                         knots, knots_init, start_ns, dt_ns, num_fixed, start_ns + int(round(stride * 1e9)))


# ------------------------------------------------------------------ the BASELINE.json configs
HANDHELD_K = dict(fx=588.10, fy=593.99, cx=339.83, cy=242.43)  # launch/ecrot_handheld.launch:50


def config1(N=100_000, seed=SEED0 + 1):
    """ecrot_synth front end: 100k events, 240x180 (CPU-runnable reference case)."""
    return frontend_packet(N, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=seed)


def config2(N=1_000_000, seed=SEED0 + 2):
    """1M events, 640x480 front end -- the configuration the headline metric is quoted on."""
    return frontend_packet(N, 640, 480, seed=seed, **HANDHELD_K)


def config3(N=5_000_000, seed=SEED0 + 3, Wp=1024, Hp=1024):
    """Back-end BA: cubic, K=10 (3 fixed => P=21), 5M events, 1024x1024 panorama."""
    return backend_window(N, 640, 480, Wp=Wp, Hp=Hp, order=4, K=10, num_fixed=3, T=0.35, seed=seed, **HANDHELD_K)


def config4_slab(rank, world=8, per_gpu=5_000_000, seed=SEED0 + 4, Wp=1024, Hp=1024):
    """Back-end BA sliding window sharded over `world` GPUs (40M events over 8): the window of config 3 with
    world*per_gpu events, as `world` time slabs of per_gpu events each; rank r generates (and owns) slab r.  per_gpu is a
    multiple of the batch size, so slab boundaries are batch boundaries and the concatenation of the slabs has exactly the
    batches the single-GPU evaluation of the whole window has (dist.batch_range hands rank r precisely slab r)."""
    return backend_window(per_gpu, 640, 480, Wp=Wp, Hp=Hp, order=4, K=10, num_fixed=3, T=0.35, seed=seed,
                          slab=(rank, world), **HANDHELD_K)


def config5_slab(rank, world=8, per_gpu=2_500_000, seed=SEED0 + 5, Wp=4096, Hp=2048):
    """Config 5 (1280x720 sensor, linear K=5, 0 fixed => P=15, 4096x2048 map) as `world` time slabs of per_gpu events."""
    return backend_window(per_gpu, 1280, 720, 1000.0, 1000.0, 639.5, 359.5, Wp=Wp, Hp=Hp, order=2, K=5, num_fixed=0, T=0.2,
                          seed=seed, slab=(rank, world))


def concat_slabs(slabs):
    """The whole window of a list of time slabs (in rank order)."""
    import copy
    w = copy.copy(slabs[0])
    w.x = np.concatenate([s.x for s in slabs])
    w.y = np.concatenate([s.y for s in slabs])
    w.t_ns = np.concatenate([s.t_ns for s in slabs])
    assert np.all(np.diff(w.t_ns) >= 0)
    return w


def config5(N=20_000_000, seed=SEED0 + 5, Wp=4096, Hp=2048):
    """1280x720 sensor, linear K=5, 0 fixed (P=15), 4096x2048 map."""
    return backend_window(N, 1280, 720, 1000.0, 1000.0, 639.5, 359.5, Wp=Wp, Hp=Hp, order=2, K=5, num_fixed=0,
                          T=0.2, seed=seed)


# ------------------------------------------------------------------ a continuous stream (end-to-end example)
@dataclass
class EventStream:
    W: int
    H: int
    fx: float
    fy: float
    cx: float
    cy: float
    x: np.ndarray
    y: np.ndarray
    t_ns: np.ndarray
    grid_ns: int             # ground-truth sampling step
    grid_quat: np.ndarray    # world<-camera orientation (x,y,z,w) at T0_NS + k*grid_ns
    grid_omega: np.ndarray   # body-frame angular velocity at the same stamps [rad/s]

    @property
    def lut(self):
        return pinhole_lut(self.W, self.H, self.fx, self.fy, self.cx, self.cy)

    def quat_at(self, t_ns):
        k = np.clip((np.asarray(t_ns, np.int64) - T0_NS + self.grid_ns // 2) // self.grid_ns, 0, len(self.grid_quat) - 1)
        return self.grid_quat[k]

    def omega_at(self, t_ns):
        k = np.clip((np.asarray(t_ns, np.int64) - T0_NS + self.grid_ns // 2) // self.grid_ns, 0, len(self.grid_quat) - 1)
        return self.grid_omega[k]


def event_stream(rate, T, W, H, fx, fy, cx, cy, seed=SEED0, noise=0.10, n_arcs=400, omega_mean=(0.1, 0.9, 0.15),
                 omega_amp=(0.5, 0.4, 0.5), omega_hz=(1.3, 0.7, 1.9), yaw0_deg=0.0):
    """rate*T events over [t0, t0+T) seen by a camera whose body-frame angular velocity is
    omega_mean + omega_amp*sin(2 pi f t + phase): R(t+h) = R(t) exp(omega h) (the post-multiplied integration the
    back end itself uses, pose_graph_optimizer.cpp:205-210), world<-camera, R(t0) = rotation by yaw0 about Y
    (pose_graph_optimizer.cpp:88-93).  Scene = great-circle arcs spread over the band the camera sweeps."""
    rng = np.random.default_rng(seed)
    grid_ns = 10_000
    n_grid = int(T * 1e9) // grid_ns + 2
    tg = np.arange(n_grid) * grid_ns * 1e-9
    phase = rng.uniform(0, 2 * np.pi, 3)
    omega = np.asarray(omega_mean) + np.asarray(omega_amp) * np.sin(2 * np.pi * np.asarray(omega_hz) * tg[:, None] + phase)
    th = np.deg2rad(yaw0_deg)
    R = Rot.from_matrix(np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]]))
    # cumulative product of the per-step increments (midpoint rule)
    inc = Rot.from_rotvec(0.5 * (omega[:-1] + omega[1:]) * (grid_ns * 1e-9)).as_matrix()
    grid_R = np.empty((n_grid, 3, 3))
    grid_R[0] = R.as_matrix()
    for i in range(n_grid - 1):
        grid_R[i + 1] = grid_R[i] @ inc[i]
    quats = Rot.from_matrix(grid_R).as_quat()
    grid_R = Rot.from_quat(quats).as_matrix()  # re-orthonormalised
    axes = grid_R[:: max(n_grid // 24, 1), :, 2]
    cone = 1.25 * np.arctan(np.hypot(W / 2 / fx, H / 2 / fy))
    scene = _Scene(rng, n_arcs, axes, cone)
    N = int(round(rate * T))
    n_sig = int(round(N * (1 - noise)))
    xs, ys, ts = [], [], []
    need = n_sig
    while need > 0:
        m = min(int(need * 2.5) + 1024, 4_000_000)
        pts = scene.sample(rng, m)
        tn = T0_NS + np.floor(rng.random(m) * T * 1e9).astype(np.int64)
        Rt = grid_R[(tn - T0_NS + grid_ns // 2) // grid_ns]
        p_cam = np.einsum("nji,nj->ni", Rt, pts)
        xi, yi, ok = _project_pinhole(p_cam, fx, fy, cx, cy, W, H)
        xs.append(xi[ok]); ys.append(yi[ok]); ts.append(tn[ok])
        need -= int(ok.sum())
    x = np.concatenate(xs)[:n_sig]
    y = np.concatenate(ys)[:n_sig]
    tn = np.concatenate(ts)[:n_sig]
    n_noise = N - len(x)
    x = np.concatenate([x, rng.integers(0, W, n_noise)])
    y = np.concatenate([y, rng.integers(0, H, n_noise)])
    tn = np.concatenate([tn, T0_NS + np.floor(rng.random(n_noise) * T * 1e9).astype(np.int64)])
    o = np.argsort(tn, kind="stable")
    return EventStream(W, H, fx, fy, cx, cy, x[o].astype(np.uint16), y[o].astype(np.uint16), tn[o], grid_ns, quats, omega)
