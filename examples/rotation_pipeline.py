#!/usr/bin/env python
"""ROS-free end-to-end run: events in -> angular velocities -> trajectory + panoramic map out.

An EXAMPLE host around the drop-in evaluator, not part of the product: it replays, single-threaded and in the order
the reference's two threads would interleave on a recorded bag, the control logic the reference keeps in

    AngVelEstimator::pushEvent / slideWindow          src/frontend/ang_vel_estimator.cpp:68-183
    PoseGraphOptimizer::pushAngVel / Run / getEventSubset / getAngVelSubset / processTimeWindow / slideWindow
                                                      src/backend/pose_graph_optimizer.cpp:60-189, :244-376

and calls, for everything that is on the accelerated path or next to it,

    FrontendEvaluator.setupProblemAndOptimize   (cmx_frontend_solve)       one packet  -> omega
    trajectory.integrateAngVel / Trajectory.generateCtrlPoses              omegas      -> new control poses
    BackendEvaluator.setupProblemAndOptimize    (cmx_backend_solve)        one window  -> refined control poses
    BackendEvaluator.updateIG / setUpdateTimesIG                           map upkeep on the device
    EventStore                                                             events uploaded once, sliced on device

Usage:  python examples/rotation_pipeline.py [--seconds 1.2] [--rate 2e6] [--degree 1|3]
Prints the angular-velocity RMSE of the front end and the orientation error of dead reckoning vs the refined
trajectory against the synthetic ground truth.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmax_slam_amd import evaluator, synth, trajectory  # noqa: E402

NS = 1_000_000_000


def _dur_ns(sec):
    """ros::Duration(double): fromSec -> floor seconds + rounded nanoseconds."""
    s = int(np.floor(sec))
    return s * NS + int(round((sec - s) * 1e9))


class Params:
    """Defaults of launch/ecrot_synth.launch:12-41 (sensor-size dependent ones scaled by the caller)."""
    num_events_per_packet = 60000   # ~30 ms of the synthetic 2 Mev/s stream (ecrot_synth.launch:22 uses 70000)
    dt_ang_vel = 0.01
    event_batch_size = 100
    frontend_event_sample_rate = 1
    backend_event_sample_rate = 1
    frontend_blur_sigma = 1.0
    backend_blur_sigma = 1.0
    time_window_size = 0.2
    sliding_window_stride = 0.1
    spline_degree = 1
    dt_knots = 0.05
    pano_height = 512
    backend_min_ev_rate = 10000
    max_update_times = 200
    Y_angle = 0.0
    deterministic = False           # CMX_OPT_DETERMINISTIC on both contexts: the same bits on every run


def run_pipeline(stream, prm=None, use_event_store=True, log=None):
    """stream: synth.EventStream (or anything with x, y, t_ns, W, H, fx, fy, cx, cy, lut).  Returns a dict."""
    prm = prm or Params()
    x, y, t = stream.x, stream.y, stream.t_ns
    n_total = len(t)
    fe = evaluator.FrontendEvaluator(stream.W, stream.H, stream.lut)
    be = evaluator.BackendEvaluator(stream.W, stream.H, stream.lut, 2 * prm.pano_height, prm.pano_height)
    fe.set_fast_path()
    be.set_fast_path()
    if prm.deterministic:
        fe.set_deterministic(True)
        be.set_deterministic(True)
    store = None
    if use_event_store:
        store = evaluator.EventStore(stream.W, stream.H, n_total)
        store.push(x, y, t)

    # ------------------------------------------------------------------ front end: packets (pushEvent)
    dt_av = _dur_ns(prm.dt_ang_vel)
    half = prm.num_events_per_packet // 2
    time_packet = int(t[0]) + _dur_ns(prm.dt_ang_vel * 0.5)
    time_get_subset = time_packet
    packets = []            # (idx_subset_beg, idx_subset_end, time_packet)
    subset_ts_map = []      # (event stamp, event index) -- ev_subset_ts_map_
    i = 0
    while True:
        # first event (not yet consumed) whose stamp exceeds the cursor
        i = max(i, int(np.searchsorted(t, time_get_subset, side="right")))
        if i >= n_total:
            break
        total = i + 1       # num_event_total_ after this event is appended
        packets.append((max(total - half, 0), total + half))
        subset_ts_map.append((int(t[i]), i))
        time_get_subset += dt_av
        i += 1
    ang_vel = np.zeros(3)
    ang_vels = []           # (time_packet, omega) pushed to the back end
    fe_ms = []
    for beg, end in packets:
        if n_total <= end:  # the reference waits for num_event_total_ > idx_subset_end: packet never completes
            break
        span = (int(t[end - 1]) - int(t[beg])) * 1e-9
        if span > 10 * prm.dt_ang_vel:
            ang_vel = np.zeros(3)
        else:
            t0 = time.perf_counter()
            args = (time_packet, stream.fx, stream.fy, stream.cx, stream.cy, prm.event_batch_size, prm.frontend_blur_sigma)
            if store is not None:
                fe.set_packet_from(store, beg, end - beg, *args)
            else:
                fe.set_packet(x[beg:end], y[beg:end], t[beg:end], *args)
            ang_vel, _rep = fe.setupProblemAndOptimize(ang_vel)
            fe_ms.append((time.perf_counter() - t0) * 1e3)
        ang_vels.append((time_packet, np.array(ang_vel, copy=True)))
        time_packet += dt_av
    if log:
        log("front end: %d packets, %.2f ms per packet (set + solve)" % (len(ang_vels), float(np.mean(fe_ms))))

    # ------------------------------------------------------------------ back end: sliding windows (Run)
    order = 2 if prm.spline_degree == 1 else 4
    win_size, win_stride = _dur_ns(prm.time_window_size), _dur_ns(prm.sliding_window_stride)
    cp_stride = int(round(prm.sliding_window_stride / prm.dt_knots))
    min_num_ev_per_win = (prm.time_window_size * prm.backend_min_ev_rate /
                          (prm.backend_event_sample_rate * prm.frontend_event_sample_rate))
    av_t = np.array([a[0] for a in ang_vels], np.int64)
    av_w = np.array([a[1] for a in ang_vels])
    map_t = np.array([m[0] for m in subset_ts_map], np.int64)
    map_i = np.array([m[1] for m in subset_ts_map], np.int64)
    # pushAngVel, first call
    t_win_beg = int(av_t[0]); t_win_end = t_win_beg + win_size
    t_av_beg, t_av_end = t_win_beg, t_win_end
    traj = trajectory.Trajectory(prm.spline_degree, t_win_beg, prm.dt_knots)
    ang_vel_prev = (int(av_t[0]), av_w[0].copy())
    th = prm.Y_angle * np.pi / 180
    pose_latest = (int(av_t[0]), _quat_from_matrix(np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0],
                                                             [-np.sin(th), 0, np.cos(th)]])))
    first_time_window, count_window, idx_cp_opt_beg = True, 0, 0
    av_used = 0             # frontend_ang_vel_.erase(begin, iter_end)
    dead_reckoning = [pose_latest]
    dr_pose = pose_latest
    dr_prev = ang_vel_prev
    reports, be_ms = [], []
    while av_t[-1] > t_win_end:  # isReadyFrontendPoses
        # getEventSubset(t_win_beg, t_win_end): packet-level search, then 100-event steps back from the end
        kb = int(np.searchsorted(map_t, t_win_beg, side="right"))
        ke = int(np.searchsorted(map_t, t_win_end, side="left"))
        if kb >= len(map_i) or ke >= len(map_i):
            break
        ev_beg, ev_end = int(map_i[kb]), int(map_i[ke])
        t_end_mod = t_win_end - _dur_ns(1e-6)
        while t[ev_end] > t_end_mod:
            ev_end -= 100
            if ev_end <= ev_beg:
                ev_end = ev_beg + 1
                break
        # getAngVelSubset(t_av_beg, t_av_end): stamps in (beg, end)
        a0 = max(int(np.searchsorted(av_t, t_av_beg, side="right")), av_used)
        a1 = int(np.searchsorted(av_t, t_av_end, side="left"))
        sub = (av_t[a0:a1], av_w[a0:a1])
        av_used = a1
        # processTimeWindow
        poses, ang_vel_prev = trajectory.integrateAngVel(pose_latest, sub, ang_vel_prev, first_time_window)
        dr_poses, dr_prev = trajectory.integrateAngVel(dr_pose, sub, dr_prev, first_time_window)
        if len(dr_poses[0]):
            dr_pose = (int(dr_poses[0][-1]), dr_poses[1][-1])
            dead_reckoning += list(zip(dr_poses[0].tolist(), dr_poses[1]))
        cps_new = traj.generateCtrlPoses(poses, t_av_beg, t_av_end)
        if first_time_window:
            idx_cp_opt_beg = 3 if prm.spline_degree == 3 else 1
            first_time_window = False
        else:
            cps_new = cps_new[(3 if prm.spline_degree == 3 else 1):]
        traj.pushbackCtrlPoses(cps_new)
        idx_cp_traj_beg = count_window * cp_stride
        idx_cp_opt_beg = max(idx_cp_traj_beg, idx_cp_opt_beg)
        num_cp_opt = traj.size() - idx_cp_opt_beg
        n_ev = ev_end - ev_beg
        if n_ev > min_num_ev_per_win:
            t0 = time.perf_counter()
            knots, start_ns, dt_ns = traj.temp_window(idx_cp_traj_beg)
            args = (order, knots, start_ns, dt_ns, idx_cp_opt_beg - idx_cp_traj_beg, t_win_beg + win_stride)
            kw = dict(event_batch_size=prm.event_batch_size, event_sample_rate=prm.backend_event_sample_rate,
                      blur_sigma=prm.backend_blur_sigma, IG="resident")  # the global map never leaves the device
            if store is not None:
                be.set_window_from(store, ev_beg, n_ev, *args, **kw)
            else:
                be.set_window(x[ev_beg:ev_end], y[ev_beg:ev_end], t[ev_beg:ev_end], *args, **kw)
            drotv, rep = be.setupProblemAndOptimize()
            traj.incrementalUpdate(drotv, idx_cp_opt_beg)
            be.updateIG(prm.max_update_times)
            # PoseGraphOptimizer::setUpdateTimesIG: FOV visit map every 0.05 s over the stride
            t_check = t_win_beg
            while t_check < t_win_beg + win_stride:
                be.setUpdateTimesIG(traj.evaluate(t_check), 3)
                t_check += _dur_ns(0.05)
            be_ms.append((time.perf_counter() - t0) * 1e3)
            reports.append(rep)
            assert num_cp_opt * 3 == len(drotv)
        t_latest = t_win_end - _dur_ns(1e-6)
        pose_latest = (t_latest, traj.evaluate(t_latest))
        # slideWindow
        t_win_beg += win_stride
        t_av_beg = t_win_end
        t_win_end += win_stride
        t_av_end = t_win_end
        count_window += 1
    if log:
        log("back end: %d windows, %.2f ms per window (set + solve + map upkeep)" % (len(reports), float(np.mean(be_ms))))
    return dict(ang_vel_t=av_t, ang_vel=av_w, traj=traj, dead_reckoning=dead_reckoning, IG=be.getIG(),
                reports=reports, fe_ms=fe_ms, be_ms=be_ms, windows=count_window)


def _quat_from_matrix(R):
    """Eigen::Quaterniond(R) for a rotation about Y (enough for R0)."""
    th = np.arctan2(R[0, 2], R[0, 0])
    return np.array([0.0, np.sin(th / 2), 0.0, np.cos(th / 2)])


def _angle_between(qa, qb):
    d = np.abs(np.sum(np.asarray(qa) * np.asarray(qb), axis=-1))
    return 2 * np.arccos(np.clip(d, 0, 1))


def _rel(q0, q):
    """q0^-1 * q for (x,y,z,w) arrays."""
    from scipy.spatial.transform import Rotation as Rot
    return (Rot.from_quat(q0).inv() * Rot.from_quat(q)).as_quat()


def evaluate_against_truth(stream, res):
    """Angular-velocity RMSE and orientation errors (both trajectories aligned to the truth at their first stamp)."""
    w_true = stream.omega_at(res["ang_vel_t"])
    w_err2 = np.sum((res["ang_vel"] - w_true) ** 2, axis=1)
    w_rmse = float(np.sqrt(np.mean(w_err2)))
    # the first solves start from omega = 0 (ang_vel_estimator.cpp:26) and need a few packets to lock on
    steady = res["ang_vel_t"] > res["ang_vel_t"][0] + 150_000_000
    w_rmse_steady = float(np.sqrt(np.mean(w_err2[steady]))) if steady.any() else w_rmse
    traj = res["traj"]
    t_lo = traj.t_beg_ns
    t_hi = t_lo + (traj.size() - traj.order + 1) * traj.dt_ns - 1
    ts = np.arange(t_lo, t_hi, 10_000_000, dtype=np.int64)
    q_est = np.array([traj.evaluate(int(tt)) for tt in ts])
    q_gt = stream.quat_at(ts)
    err_ba = _angle_between(_rel(q_est[0], q_est), _rel(q_gt[0], q_gt))
    dr_t = np.array([p[0] for p in res["dead_reckoning"]], np.int64)
    dr_q = np.array([p[1] for p in res["dead_reckoning"]])
    keep = dr_t <= t_hi
    dr_gt = stream.quat_at(dr_t[keep])
    err_dr = _angle_between(_rel(dr_q[0], dr_q[keep]), _rel(dr_gt[0], dr_gt))
    return dict(omega_rmse=w_rmse, omega_rmse_steady=w_rmse_steady, ba_err_deg_rms=float(np.rad2deg(np.sqrt(np.mean(err_ba ** 2)))),
                ba_err_deg_max=float(np.rad2deg(err_ba.max())),
                dr_err_deg_rms=float(np.rad2deg(np.sqrt(np.mean(err_dr ** 2)))),
                dr_err_deg_max=float(np.rad2deg(err_dr.max())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.2)
    ap.add_argument("--rate", type=float, default=2e6)
    ap.add_argument("--degree", type=int, default=1, choices=(1, 3))
    ap.add_argument("--host-events", action="store_true", help="re-upload events per packet/window (no EventStore)")
    ap.add_argument("--deterministic", action="store_true", help="bitwise reproducible evaluations (CMX_OPT_DETERMINISTIC)")
    a = ap.parse_args()
    stream = synth.event_stream(a.rate, a.seconds, 240, 180, 200.0, 200.0, 119.5, 89.5, omega_mean=(0.2, 1.8, 0.3),
                                omega_amp=(1.0, 0.8, 1.0))
    prm = Params()
    prm.spline_degree = a.degree
    prm.deterministic = a.deterministic
    t0 = time.perf_counter()
    res = run_pipeline(stream, prm, use_event_store=not a.host_events, log=print)
    wall = time.perf_counter() - t0
    m = evaluate_against_truth(stream, res)
    print("%.2f s of events (%d) processed in %.2f s wall" % (a.seconds, len(stream.x), wall))
    print("front end  |omega - truth| rmse      : %.4f rad/s (%.4f after the first 0.15 s)" % (m["omega_rmse"], m["omega_rmse_steady"]))
    print("dead reckoning orientation error    : rms %.3f deg, max %.3f deg" % (m["dr_err_deg_rms"], m["dr_err_deg_max"]))
    print("refined trajectory orientation error: rms %.3f deg, max %.3f deg" % (m["ba_err_deg_rms"], m["ba_err_deg_max"]))
    print("map: %d x %d, %.0f%% of pixels touched" % (res["IG"].shape[1], res["IG"].shape[0],
                                                       100.0 * float((res["IG"] > 0).mean())))


if __name__ == "__main__":
    main()
