"""oracle/iwe_numpy.py -- a SECOND, independently written restatement (numpy) of the per-event stage of both IWE builders.

TEST INFRASTRUCTURE ONLY (same rule as the rest of oracle/).  oracle/frontend.c and oracle/backend.c restate the reference
in C; this file restates the same reference lines again, from the sources, in vectorised numpy, so that the rows of
SURVEY.md section 8(a) that no golden vector can pin here (F2-F4, B3, B4, B6: OpenCV / ROS absent, the reference ships no
fixtures) rest on two restatements that agree instead of one.  tests/test_oracle_iwe_numpy.py compares the two.  PARITY
UNPINNED against the reference itself, like oracle/*.c.

What is restated, and from where
  front end   src/frontend/local_image_warped_events.cpp:10-38 (batches of event_batch_size), :58-169 (per batch: shared
              time = first + (last - first) * 0.5 in ros::Time / ros::Duration arithmetic; per event: first-order rotation,
              canonical projection, intrinsics, 2x3 Jacobian chain in double, bilinear votes and signed-weight derivative
              votes in float), src/utils/image_geom_util.cpp:7-58
  back end    src/backend/event_pano_warper.cpp:167-196 (batch cursor: `beg < end - 1`), :233-336 (per batch: pose at the shared time; per event with stride
              event_sample_rate restarting at each batch: R b, equirectangular projection, float Jacobian chain
              dpm_drb * (-[rb]x) * ddrot_ddrot_cp, IL_old / IL_new split, j = 3 (idx_cp_beg - num_cps_fixed) + i >= 0),
              include/backend/equirectangular_camera.h:18-45
  ros::Time   roscpp's Time / Duration (un-vendored): toSec() = sec + 1e-9 nsec; Duration * double = Duration(toSec() * s);
              Duration(double) = fromSec: sec = floor(t), nsec = round((t - sec) 1e9), then carry
  cv::Mat A*B gemm on CV_32F accumulates each entry in double and rounds once (OpenCV's GEMMSingleMul<float, double>); the
              fixed-size cv::Matx products accumulate in their own element type, k = 0, 1, 2 in order, starting from 0

The pose of a batch (B5) is NOT restated here: it is taken from the reference's own Basalt spline compiled from
/root/reference (oracle/_ref, pinned), or from oracle/so3_spline.c where that library is not present.

Accumulation order: votes are added event by event, corner by corner (np.add.at is unbuffered and walks its index array in
order), i.e. in the reference's order -- the float images can be compared bit for bit."""
import math

import numpy as np

F32 = np.float32


# ---- ros::Time / ros::Duration ----------------------------------------------------------------------------------
def _to_sec(ns):
    sec, nsec = divmod(int(ns), 1_000_000_000)
    return float(sec) + 1e-9 * float(nsec)


def _duration_from_sec(t):
    sec = math.floor(t)
    nsec = int(round((t - sec) * 1e9))  # boost::math::round: half away from zero; (t - sec) >= 0 here
    if (t - sec) * 1e9 - math.floor((t - sec) * 1e9) == 0.5:
        nsec = int(math.floor((t - sec) * 1e9)) + 1
    sec = int(sec) + nsec // 1_000_000_000
    nsec = nsec % 1_000_000_000
    return sec * 1_000_000_000 + nsec


def batch_time_ns(t_first_ns, t_last_ns):
    """time_first + (time_last - time_first) * 0.5"""
    d = int(t_last_ns) - int(t_first_ns)              # Time - Time: exact in (sec, nsec)
    half = _duration_from_sec(_to_sec(d) * 0.5)       # Duration * 0.5
    return int(t_first_ns) + half                     # Time + Duration: exact


def _add_votes(img, yy, xx, w):
    """img[yy + {0,0,1,1}, xx + {0,1,0,1}] += w[:, 0..3], event by event, corner by corner (the reference's order)."""
    n = len(xx)
    iy = np.empty(4 * n, np.int64)
    ix = np.empty(4 * n, np.int64)
    iy[0::4], iy[1::4], iy[2::4], iy[3::4] = yy, yy, yy + 1, yy + 1
    ix[0::4], ix[1::4], ix[2::4], ix[3::4] = xx, xx + 1, xx, xx + 1
    np.add.at(img, (iy, ix), w.reshape(-1).astype(F32))


def _weights(dx, dy):
    one = F32(1)
    return np.stack([(one - dx) * (one - dy), dx * (one - dy), (one - dx) * dy, dx * dy], axis=1).astype(F32)


def _dweights(r0, r1, dx, dy):
    """the four signed-weight expressions, float arithmetic: r0*(-(1-dy)) + r1*(-(1-dx)), r0*(1-dy) + r1*(-dx), ..."""
    one = F32(1)
    return np.stack([r0 * (-(one - dy)) + r1 * (-(one - dx)),
                     r0 * (one - dy) + r1 * (-dx),
                     r0 * (-dy) + r1 * (one - dx),
                     r0 * dy + r1 * dx], axis=1).astype(F32)


# ---- front end ---------------------------------------------------------------------------------------------------
def frontend_iwe(x, y, t_ns, t_ref_ns, lut, W, H, fx, fy, cx, cy, omega, batch, want_deriv=True):
    """(image_warped HxW float32, image_warped_deriv HxWx3 float32 or None), before any blur."""
    x = np.asarray(x, np.int64); y = np.asarray(y, np.int64); t = np.asarray(t_ns, np.int64)
    n = len(x)
    img = np.zeros((H, W), F32)
    der = np.zeros((H, W, 3), F32) if want_deriv else None
    if n == 0:
        return img, der
    lut = np.asarray(lut, np.float64).reshape(-1, 3)
    om = np.asarray(omega, np.float64)
    nb = (n + batch - 1) // batch
    tref = _to_sec(t_ref_ns)
    dt_b = np.empty(nb)
    for b in range(nb):
        b0, b1 = b * batch, min((b + 1) * batch, n)
        dt_b[b] = _to_sec(batch_time_ns(t[b0], t[b1 - 1])) - tref
    dt = dt_b[np.arange(n) // batch]
    p = lut[y * W + x]                                     # bearing vectors
    dr = om[None, :] * dt[:, None]                         # delta_rot = ang_vel * dt
    px, py, pz = p[:, 0], p[:, 1], p[:, 2]
    rx = px + (dr[:, 1] * pz - dr[:, 2] * py)              # p + delta_rot x p
    ry = py + (dr[:, 2] * px - dr[:, 0] * pz)
    rz = pz + (dr[:, 0] * py - dr[:, 1] * px)
    iz = 1.0 / rz
    xn, yn = rx * iz, ry * iz
    u = fx * xn + cx
    v = fy * yn + cy
    xx = np.trunc(u).astype(np.int64)                      # int xx = ev_warped_pt.x
    yy = np.trunc(v).astype(np.int64)
    ok = (1 <= xx) & (xx < W - 2) & (1 <= yy) & (yy < H - 2)
    dx = (u - xx).astype(F32)
    dy = (v - yy).astype(F32)
    _add_votes(img, yy[ok], xx[ok], _weights(dx[ok], dy[ok]))
    if want_deriv:
        # 2x3 = [canonicalProjection's 2x3] * [(-dt) p]x, Matx product: s = 0; s += a(i,k) b(k,j), k = 0..2
        vx, vy, vz = (-dt) * px, (-dt) * py, (-dt) * pz
        zero = np.zeros(n)
        S = [[zero, -vz, vy], [vz, zero, -vx], [-vy, vx, zero]]           # cross2Matrix(v)
        A = [[iz, zero, -xn * iz], [zero, iz, -yn * iz]]
        Jc = [[((0.0 + A[i][0] * S[0][j]) + A[i][1] * S[1][j]) + A[i][2] * S[2][j] for j in range(3)] for i in range(2)]
        K = [[fx, 0.0], [0.0, fy]]
        Jw = [[(0.0 + K[i][0] * Jc[0][j]) + K[i][1] * Jc[1][j] for j in range(3)] for i in range(2)]
        for k in range(3):
            r0 = Jw[0][k].astype(F32)[ok]                  # Point3f(r0m(0), ...): double -> float
            r1 = Jw[1][k].astype(F32)[ok]
            _add_votes(der[:, :, k], yy[ok], xx[ok], _dweights(r0, r1, dx[ok], dy[ok]))
    return img, der


# ---- back end ----------------------------------------------------------------------------------------------------
def _project_equirect(P, fx, fy, cxp, cyp):
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    phi = np.arctan2(x, z)
    rho = np.sqrt(x * x + y * y + z * z)
    theta = np.arcsin(y / rho)
    ydr = y / rho
    xdz = x / z
    tmp1 = fx / ((1 + xdz * xdz) * z)
    tmp2 = -fy / np.sqrt(1 - ydr * ydr)
    tmp3 = ydr / (rho * rho)
    J = np.zeros((len(x), 2, 3), F32)                      # cv::Matx23f: every entry cast from double
    J[:, 0, 0] = tmp1
    J[:, 0, 2] = -tmp1 * xdz
    J[:, 1, 0] = tmp2 * tmp3 * x
    J[:, 1, 1] = tmp2 * (tmp3 * y - 1 / rho)
    J[:, 1, 2] = tmp2 * tmp3 * z
    return cxp + phi * fx, cyp + theta * fy, J


def backend_iwe(x, y, t_ns, lut, W, Wp, Hp, order, num_fixed, t_next_win_beg_ns, batch, sample_rate, pose_of, want_deriv=True):
    """pose_of(t_ns) -> (R 3x3 float64, J (order,3,3) float64 = d_val_d_knot blocks, idx_cp_beg).
    want_deriv: False / 0 = images only, or the number of derivative planes P = 3 * (number of knots - num_fixed).
    Returns (IL_old, IL_new, planes [P, Hp, Wp] or None), before composition and blur."""
    x = np.asarray(x, np.int64); y = np.asarray(y, np.int64); t = np.asarray(t_ns, np.int64)
    n = len(x)
    il_old, il_new = np.zeros((Hp, Wp), F32), np.zeros((Hp, Wp), F32)
    nP = int(want_deriv) if not isinstance(want_deriv, bool) else 0
    planes = np.zeros((nP, Hp, Wp), F32) if nP else None
    lut = np.asarray(lut, np.float64).reshape(-1, 3)
    fx = (Wp / 360.0) * 180.0 / math.pi                    # focalFromFOV(size, 360, 180)
    fy = (Hp / 180.0) * 180.0 / math.pi
    cxp, cyp = Wp / 2.0, Hp / 2.0
    b0 = 0
    while b0 < n - 1:                                      # :188-190: `ev_batch_beg < end() - 1` -- a trailing batch of ONE event is never warped
        b1 = b0 + batch if n - b0 > batch else n
        tb = batch_time_ns(t[b0], t[b1 - 1])
        R, Jk, idx = pose_of(tb)
        b0_next = b0 + batch
        sel = np.arange(b0, b1, sample_rate)               # the stride restarts at every batch
        bv = lut[y[sel] * W + x[sel]]
        ray = (R[None, :, 0] * bv[:, 0:1] + R[None, :, 1] * bv[:, 1:2]) + R[None, :, 2] * bv[:, 2:3]   # R b, k = 0, 1, 2
        pxm, pym, dpm_drb = _project_equirect(ray, fx, fy, cxp, cyp)
        xx = np.trunc(pxm).astype(np.int64)
        yy = np.trunc(pym).astype(np.int64)
        dx = (pxm - xx).astype(F32)
        dy = (pym - yy).astype(F32)
        ok = (1 <= xx) & (xx < Wp - 2) & (1 <= yy) & (yy < Hp - 2)
        old = t[sel] < t_next_win_beg_ns
        w = _weights(dx, dy)
        m = ok & old
        _add_votes(il_old, yy[m], xx[m], w[m])
        m = ok & ~old
        _add_votes(il_new, yy[m], xx[m], w[m])
        if nP:
            rb = ray.astype(F32)                           # Matx33f(0, rb.z, -rb.y, ...): double -> float
            z0 = np.zeros(len(sel), F32)
            D = [[z0, rb[:, 2], -rb[:, 1]], [-rb[:, 2], z0, rb[:, 0]], [rb[:, 1], -rb[:, 0], z0]]
            # Matx23f * Matx33f: float accumulation from 0, k = 0..2
            dd = np.zeros((len(sel), 2, 3), F32)
            for i in range(2):
                for j in range(3):
                    s = np.zeros(len(sel), F32)
                    for k in range(3):
                        s = (s + dpm_drb[:, i, k] * D[k][j]).astype(F32)
                    dd[:, i, j] = s
            # cv::Mat (2x3 float) * cv::Mat (3 x 3n float): every entry accumulated in double, rounded once
            Jf = np.concatenate([Jk[k].astype(F32) for k in range(order)], axis=1)       # 3 x 3n, float
            jac = _gemm_f32(dd, Jf)
            for i in range(3 * order):
                j = 3 * (idx - num_fixed) + i
                if j < 0:
                    continue
                dw = _dweights(jac[:, 0, i], jac[:, 1, i], dx, dy)
                _add_votes(planes[j], yy[ok], xx[ok], dw[ok])
        b0 = b0_next
    return il_old, il_new, planes


def _gemm_f32(A, B):
    """per event: (2x3 float) * (3xm float) with double accumulation in k order, rounded to float once"""
    a = A.astype(np.float64)
    b = B.astype(np.float64)
    out = np.zeros((A.shape[0], 2, B.shape[1]))
    for k in range(3):
        out = out + a[:, :, k:k + 1] * b[None, k:k + 1, :]
    return out.astype(F32)


# ---- global-map upkeep and alpha (event_pano_warper.cpp:81-165) ----------------------------------------------------
def q_to_R(q_xyzw):
    """Eigen::Quaterniond::toRotationMatrix() (Sophus::SO3d::matrix()), same operation order."""
    x, y, z, w = (float(v) for v in q_xyzw)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1.0 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1.0 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1.0 - (txx + tyy)]])


def mark_visited(update_times, q_xyzw, lut, W, H, Wp, Hp, radius):
    """setUpdateTimesIG (:81-107): every sensor pixel warped by the pose marks a (2 radius + 1)^2 neighbourhood of a mask --
    with the reference's row test `0 <= y_mask + j` --, then cv::add(times, mask, times) on CV_8U (saturating)."""
    R = q_to_R(q_xyzw)
    lut = np.asarray(lut, np.float64).reshape(-1, 3)
    ray = (R[None, :, 0] * lut[:, 0:1] + R[None, :, 1] * lut[:, 1:2]) + R[None, :, 2] * lut[:, 2:3]
    fx = (Wp / 360.0) * 180.0 / math.pi
    fy = (Hp / 180.0) * 180.0 / math.pi
    pxm = Wp / 2.0 + np.arctan2(ray[:, 0], ray[:, 2]) * fx
    pym = Hp / 2.0 + np.arcsin(ray[:, 1] / np.sqrt(ray[:, 0] * ray[:, 0] + ray[:, 1] * ray[:, 1] + ray[:, 2] * ray[:, 2])) * fy
    ic = np.trunc(pxm).astype(np.int64)
    ir = np.trunc(pym).astype(np.int64)
    mask = np.zeros((Hp, Wp), np.uint8)
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            xm, ym = ic + i, ir + j
            ok = (0 <= ym + j) & (ym < Hp) & (0 <= xm) & (xm < Wp) & (ym >= 0)   # (ym >= 0: .at() with a negative row is UB in
            mask[ym[ok], xm[ok]] = 1                                             #  the reference; no pixel goes there in-bounds)
    update_times[:] = np.minimum(update_times.astype(np.int32) + mask, 255).astype(np.uint8)


def update_ig(IG, IL_old, update_times, max_update_times):
    """updateIG (:109-126)"""
    m = update_times <= max_update_times
    IG[m] = (IG[m] + IL_old[m]).astype(F32)


def alpha(IGp, IL):
    """updateAlpha (:134-165): ratio of the event densities num / area with area = sum(1 - exp(-I)) (float images, double sums)."""
    if np.count_nonzero(IGp) < 1:
        return 0.0

    def density(I):
        e = np.exp((F32(-1.0) * I).astype(F32)).astype(F32)     # cv::exp on CV_32F of the float MatExpr -(1/lambda0) * I
        area = float(np.sum((F32(1) - e).astype(F32), dtype=np.float64))
        return float(np.sum(I, dtype=np.float64)) / area
    return density(IL) / density(IGp)
