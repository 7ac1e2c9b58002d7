"""-m gpu: one-process multi-GPU -- a GROUP of back-end contexts behind one handle (cmx_backend_create_group).

The reference's host is one process / one back-end thread / one GSL instance (src/cmax_slam.cpp:92,
src/backend/global_optim_contrast_gsl.cpp:23-33).  A group is driven exactly like a single context: ONE thread calls
set_window / eval / setupProblemAndOptimize on ONE handle; the library shards the window by whole batches, fans the
evaluation out, exchanges the partial planes between the members, adds the members' gradients (linear in their rows: no second
collective) and returns one contrast / gradient.

A one-GPU box runs the members on the SAME device (the direct transport: peer reduce-scatter + all-gather kernels ordered
by HIP events); the variants over 2 / 4 / 8 devices (RCCL through ncclCommInitAll, and the direct transport across
devices) skip by device count.  Everything is compared with the single-context evaluation of the whole window -- itself
compared with the oracle at these very sizes by test_gpu_baseline_configs.py / test_gpu_config5_full.py -- and, at a size
the oracle finishes in seconds, with the oracle directly."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, dist, synth
from util import RTOL, rel_img, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def _set(be, w, IG=None, n=None):
    n = len(w.x) if n is None else n
    be.set_window(w.x[:n], w.y[:n], w.t_ns[:n], w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns,
                  w.batch, w.sample_rate, w.sigma, _lib.VARIANCE, IG)


def _pair(hip, w, devices, IG=None, n=None, transport=0, fast=True):
    grp = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=devices, transport=transport)
    one = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    if fast:
        grp.set_fast_path()
        one.set_fast_path()
    _set(grp, w, IG, n)
    _set(one, w, IG, n)
    return grp, one


def _same(grp, one, seq, tol=1e-6):
    for i, (d, want) in enumerate(seq):
        c0, g0 = one.eval(d, want)
        c, g = grp.eval(d, want)
        assert rel_scalar(c, c0) < tol, (i, c, c0)
        if want:
            assert rel_vec(g, g0) < tol, (i, g, g0)


def test_group_of_two_on_one_gpu_config4_slab(hip):
    """BASELINE config 4's per-GPU slab (5M events, cubic K = 10, 1024^2) split over two members of one device: the shards are
    dist.batch_range's, the evaluations -- a sequence with a cost-only point, a repeated point (image reuse) and a parameter jump
    that leaves the exchange set -- equal the single context's; one handle, one thread."""
    w = synth.config4_slab(2, 8, 5_000_000)
    grp, one = _pair(hip, w, [0, 0])
    info = grp.group_info()
    assert info["members"] == 2 and info["devices"] == [0, 0] and info["transport"] == _lib.GROUP_DIRECT
    want = [dist.batch_range(len(w.x), w.batch, r, 2) for r in range(2)]
    assert info["events_per_member"] == [e - b for b, e in want]      # (sample rate 1: packed events = events of the shard)
    rng = np.random.default_rng(7)
    small = rng.normal(0, 0.004, w.P)
    jump = np.tile([0.5, 0.0, 0.0], w.P // 3)
    seq = [(np.zeros(w.P), True), (small, False), (small, True), (rng.normal(0, 0.004, w.P), True), (jump, True), (jump, False),
           (np.zeros(w.P), True)]
    _same(grp, one, seq)
    s = grp.stats()
    assert s["exchange_misses"] >= 1 and s["sharded_host_syncs"] == 0 and 0 < s["exchange_tiles"] < 1024 // 4, s
    assert s["comm_bytes"] < 2 * (1 << 20), s            # a set of tiles + the gradient rows, not 2 x 4 MB of planes
    # the planes a member holds after the exchange are the window's (what updateIG / publishEventImage read)
    # (fp32 sums in another order: two members' partial sums added, against one context's chunks)
    assert rel_img(grp.get_plane(_lib.PLANE_IL_OLD), one.get_plane(_lib.PLANE_IL_OLD)) < RTOL
    assert rel_img(grp.get_plane(_lib.PLANE_IWE), one.get_plane(_lib.PLANE_IWE)) < RTOL
    assert 0 < grp.group_info()["last_fanout_us"] < 5e5
    grp.close()
    one.close()


def test_group_of_two_on_one_gpu_config5_slab_with_a_map(hip):
    """BASELINE config 5's per-GPU slab (2.5M events, 1280x720, 4096x2048, linear K = 5) with a non-zero global map: alpha is
    formed by every member from the exchanged planes and equals the single context's; tile-set exchange on 32 MB planes."""
    w = synth.config5_slab(3, 8, 2_500_000)
    IG = np.zeros((w.Hp, w.Wp), np.float32)
    IG[900:1100, 1500:2600] = 0.7
    grp, one = _pair(hip, w, [0, 0], IG)
    rng = np.random.default_rng(8)
    seq = [(np.zeros(w.P), True), (rng.normal(0, 0.003, w.P), True), (rng.normal(0, 0.003, w.P), False),
           (np.tile([0.25, 0.0, 0.0], w.P // 3), True), (np.zeros(w.P), True)]
    _same(grp, one, seq)
    assert one.alpha > 0 and rel_scalar(grp.alpha, one.alpha) < 1e-7
    s = grp.stats()
    assert s["sharded_host_syncs"] == 0 and s["comm_bytes"] < 0.1 * 2 * w.Wp * w.Hp * 4, s
    # map upkeep acts on every member's replica; the next window (resident map) still agrees
    for ev in (grp, one):
        ev.eval(np.zeros(w.P), False)
        ev.setUpdateTimesIG(w.knots_init[0], 3)
        ev.updateIG(5)
    assert rel_img(grp.getIG(), one.getIG()) < RTOL
    for ev in (grp, one):
        _set(ev, w, "resident", n=1_000_000)
    _same(grp, one, [(np.zeros(w.P), True), (rng.normal(0, 0.003, w.P), True)])
    assert rel_scalar(grp.alpha, one.alpha) < 1e-7
    grp.close()
    one.close()


@pytest.mark.parametrize("members,n_events,batch", [(2, 30_001, 100), (3, 20_000, 128), (4, 150, 100), (4, 2_001, 1), (2, 1, 100)])
def test_group_vs_oracle_ragged_shards_and_empty_members(hip, oracle, members, n_events, batch):
    """Against the oracle directly, at sizes it finishes in seconds: a window whose last batch is the single trailing event the
    reference's loop never opens a batch for (event_pano_warper.cpp:188-196), members that hold NO events (150 events = 2
    batches for 4 members), batch size 1 (every member's last event would be dropped by a naive split), a one-event window."""
    w = synth.backend_window(n_events, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 8, 2, 0.25, seed=71 + members)
    w.batch = batch
    grp = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0] * members)
    grp.set_fast_path()
    _set(grp, w)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, w.batch, w.sample_rate, w.sigma, oracle.VARIANCE)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    per = grp.group_info()["events_per_member"]
    n_in_batches = n_events if (n_events - 1) % batch else n_events - 1      # a trailing single event is in no batch
    assert sum(per) == (n_in_batches if n_events > 1 else 0), (per, n_events)
    if members == 4 and n_events == 150:
        assert per[2] == 0 and per[3] == 0
    rng = np.random.default_rng(9)
    for k, want in enumerate([True, False, True]):
        d = rng.normal(0, 0.01, w.P) if k else np.zeros(w.P)
        c_ref, g_ref = ref.eval(d)
        c, g = grp.eval(d, want)
        assert rel_scalar(c, c_ref) < RTOL or (c_ref == 0 and c == 0), (k, c, c_ref)
        if want and np.abs(g_ref).max() > 0:
            assert rel_vec(g, g_ref) < RTOL, (k, g, g_ref)
    if n_events > 1:
        assert rel_img(grp.get_plane(_lib.PLANE_IL_OLD) + grp.get_plane(_lib.PLANE_IL_NEW), ref.IL_old + ref.IL_new) < RTOL
    grp.close()


def test_group_reference_shaped_path_and_small_planes(hip):
    """Derivative planes (CMX_GRAD_PLANES) through a group: 2 + P whole planes travel; 0.5 MB planes travel whole on the
    production path too."""
    w = synth.backend_window(30_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 10, 3, 0.35, seed=49)
    for fast in (False, True):
        grp, one = _pair(hip, w, [0, 0], fast=fast)
        _same(grp, one, [(np.zeros(w.P), True), (np.full(w.P, 0.003), False), (np.full(w.P, 0.003), True)])
        grp.close()
        one.close()


def test_group_solve_is_one_optimiser(hip):
    """cmx_backend_solve on the handle: ONE FR-CG driver whose every evaluation fans out -- the unchanged
    global_contrast_{f,df,fdf} bodies.  Same decisions as over the single context up to the summation order of the planes:
    compared by outcome (FR-CG's loose stopping rules amplify 1e-8 differences, see test_gpu_solver.py)."""
    w = synth.backend_window(200_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 1024, 512, 2, 5, 0, 0.2, seed=83, knot_sigma=0.02)
    grp, one = _pair(hip, w, [0, 0, 0])
    xg, rg = grp.setupProblemAndOptimize()
    x1, r1 = one.setupProblemAndOptimize()
    assert rg["initial_cost"] == pytest.approx(r1["initial_cost"], rel=1e-6)
    assert abs(rg["final_cost"] - r1["final_cost"]) < 2e-3 * abs(r1["final_cost"]), (rg, r1)
    assert rg["final_cost"] < rg["initial_cost"] and rg["n_f"] + rg["n_df"] >= 3
    assert np.abs(xg - x1).max() < 0.02, (xg, x1)
    # a second window on the same handle (workers woken again after idling)
    import time
    time.sleep(0.05)
    _set(grp, w, None, n=120_000)
    _set(one, w, None, n=120_000)
    _same(grp, one, [(xg, True), (np.zeros(w.P), True)])
    grp.close()
    one.close()


def test_group_errors_do_not_hang(hip):
    """A window one member rejects (an event outside the sensor in the SECOND member's shard) fails the call with that member's
    message and leaves the group without a window; bad arguments fail like on a plain context; the handle stays usable."""
    w = synth.backend_window(40_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 8, 2, 0.25, seed=91)
    grp = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0, 0])
    grp.set_fast_path()
    x_bad = w.x.copy()
    x_bad[30_000] = 1000
    with pytest.raises(hip.CmaxHipError) as e:
        grp.set_window(x_bad, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                       w.sample_rate, w.sigma, _lib.VARIANCE, None)
    assert e.value.status == _lib.ERR_EVENT_RANGE and "member 1" in str(e.value)
    grp.K, grp.num_fixed = w.K, w.num_fixed      # (the python mirror records them after a successful hand-over only)
    with pytest.raises(hip.CmaxHipError) as e:
        grp.eval(np.zeros(w.P))
    assert e.value.status == _lib.ERR_STATE
    with pytest.raises(hip.CmaxHipError):
        grp.accumulate(np.zeros(w.P))          # split-phase interface: not on a group
    with pytest.raises(hip.CmaxHipError):
        grp.set_window(w.x, w.y, w.t_ns, 3, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                       w.sample_rate, w.sigma, _lib.VARIANCE, None)     # spline order 3
    _set(grp, w)
    one = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    one.set_fast_path()
    _set(one, w)
    _same(grp, one, [(np.zeros(w.P), True)])
    grp.close()
    one.close()


def test_group_of_one_is_a_plain_context(hip):
    w = synth.backend_window(10_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 2, 5, 0, 0.2, seed=92)
    g1 = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0])
    assert g1.group_info()["members"] == 1
    g1.set_fast_path()
    _set(g1, w)
    c, g = g1.eval(np.zeros(w.P))
    assert g1.stats()["comm_bytes"] == 0 and np.isfinite(c)
    g1.close()


@pytest.mark.parametrize("n_dev", [2, 4, 8])
@pytest.mark.parametrize("transport", [_lib.GROUP_RCCL, _lib.GROUP_DIRECT])
def test_group_over_several_devices(hip, n_dev, transport):
    """The same over n_dev GPUs of one node (RCCL via ncclCommInitAll; the direct peer-to-peer transport) -- skips on boxes
    with fewer devices (every box the builder has seen)."""
    if _lib.lib().cmx_device_count() < n_dev:
        pytest.skip("needs %d GPUs" % n_dev)
    w = synth.config4_slab(1, 8, 2_000_000)
    grp, one = _pair(hip, w, list(range(n_dev)), transport=transport)
    rng = np.random.default_rng(17)
    seq = [(np.zeros(w.P), True), (rng.normal(0, 0.004, w.P), False), (rng.normal(0, 0.004, w.P), True),
           (np.tile([0.5, 0.0, 0.0], w.P // 3), True)]
    _same(grp, one, seq)
    xg, rg = grp.setupProblemAndOptimize()
    x1, r1 = one.setupProblemAndOptimize()
    assert abs(rg["final_cost"] - r1["final_cost"]) < 2e-3 * abs(r1["final_cost"])
    grp.close()
    one.close()
    # a panorama whose tile sort needs MORE than the default 64 KB of dynamic LDS (4096 x 2048: 2 x 8192 + 1 bins x 4 B) on EVERY member's
    # device: the raised limit is a per-device function attribute (it used to be set once per process)
    w5 = synth.config5_slab(3, 8, 600_000)
    grp, one = _pair(hip, w5, list(range(n_dev)), transport=transport)
    _same(grp, one, [(np.zeros(w5.P), True), (rng.normal(0, 0.004, w5.P), True)])
    grp.close()
    one.close()


# ---------------------------------------------------------------- round 5: the device event store behind a group
def _from(be, store, first, n, w, rate=None, IG=None):
    be.set_window_from(store, first, n, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                       w.sample_rate if rate is None else rate, w.sigma, _lib.VARIANCE, IG)


@pytest.mark.parametrize("members,first,n,batch,rate", [(2, 5_000, 50_001, 100, 1), (3, 0, 60_001, 128, 1), (4, 1_000, 150, 100, 1),
                                                        (2, 7, 40_000, 100, 3), (3, 0, 2_001, 1, 1)])
def test_group_window_cut_from_the_replicated_store_vs_oracle(hip, oracle, members, first, n, batch, rate):
    """cmx_backend_set_window_from on a group handle: every member cuts ITS batch range from the store's replica on its own
    device (pose_graph_optimizer.cpp:131-165 without the copy).  Against the oracle on the same slice: the window whose last batch is
    the single trailing event (50_001 / 60_001 / 2_001 events), members without events, sub-sampling restarting at every batch."""
    w = synth.backend_window(60_001, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 10, 3, 0.35, seed=52 + members)
    w.batch = batch
    store = hip.EventStore(w.W, w.H, capacity=100_000, devices=[0] * members)
    assert store.devices == [0]
    store.push(w.x[:20_000], w.y[:20_000], w.t_ns[:20_000])     # the stream arrives in chunks
    store.push(w.x[20_000:], w.y[20_000:], w.t_ns[20_000:])
    grp = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0] * members)
    grp.set_fast_path()
    _from(grp, store, first, n, w, rate)
    sl = slice(first, first + n)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, batch, rate, w.sigma, oracle.VARIANCE)
    ref.set_window(w.x[sl], w.y[sl], w.t_ns[sl], w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    # the same window handed over from host arrays: identical shards, identical bits
    host = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0] * members)
    host.set_fast_path()
    host.set_window(w.x[sl], w.y[sl], w.t_ns[sl], w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, batch,
                    rate, w.sigma, _lib.VARIANCE, None)
    assert grp.group_info()["events_per_member"] == host.group_info()["events_per_member"]
    rng = np.random.default_rng(3)
    for k, want in enumerate([True, False, True]):
        d = rng.normal(0, 0.005, w.P) if k else np.zeros(w.P)
        c_ref, g_ref = ref.eval(d)
        c, g = grp.eval(d, want)
        ch, gh = host.eval(d, want)
        assert rel_scalar(c, c_ref) < RTOL and rel_scalar(c, ch) < 1e-7, (k, c, c_ref, ch)
        if want:
            assert rel_vec(g, g_ref) < RTOL and rel_vec(g, gh) < 1e-6, (k, g, g_ref)
    assert rel_img(grp.get_plane(_lib.PLANE_IL_OLD) + grp.get_plane(_lib.PLANE_IL_NEW), ref.IL_old + ref.IL_new) < RTOL
    for e in (grp, host):
        e.close()
    store.close()


def test_group_store_sliding_windows_and_drop(hip):
    """The reference's flow on a group: the stream is pushed once, overlapping windows are cut by global index
    (pose_graph_optimizer.cpp:131-165), old events are dropped on every replica (ang_vel_estimator.cpp:149-173) and the indices keep
    their meaning; a single context on the same device cuts from the same store."""
    w = synth.backend_window(90_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 2, 8, 0, 0.3, seed=58)
    store = hip.EventStore(w.W, w.H, capacity=120_000, devices=[0, 0])
    store.push(w.x, w.y, w.t_ns)
    grp = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0, 0])
    one = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    for e in (grp, one):
        e.set_fast_path()
    d = np.random.default_rng(5).normal(0, 0.004, w.P)
    for k, (first, n) in enumerate([(0, 40_000), (20_000, 40_000), (40_000, 50_000)]):
        if k == 2:
            store.drop_before(30_000)
            assert store.begin == 30_000 and store.end == 90_000
        _from(grp, store, first, n, w)
        _from(one, store, first, n, w)
        (c, g), (c1, g1) = grp.eval(d), one.eval(d)
        assert rel_scalar(c, c1) < 1e-6 and rel_vec(g, g1) < 1e-6, (k, c, c1)
    with pytest.raises(hip.CmaxHipError) as e:
        _from(grp, store, 10_000, 30_000, w)       # dropped events: every member refuses, no half-installed window
    assert e.value.status == _lib.ERR_INVALID_ARG
    with pytest.raises(hip.CmaxHipError) as e:
        grp.eval(d)
    assert e.value.status == _lib.ERR_STATE
    _from(grp, store, 40_000, 50_000, w)
    assert np.isfinite(grp.eval(d)[0])
    for e in (grp, one):
        e.close()
    store.close()


def test_group_store_on_other_devices_is_refused(hip):
    """A store without a replica on a member's device is an argument error (not a peer read across the fabric)."""
    if _lib.lib().cmx_device_count() < 2:
        pytest.skip("needs 2 GPUs")
    w = synth.backend_window(10_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 2, 5, 0, 0.2, seed=59)
    store = hip.EventStore(w.W, w.H, capacity=20_000, device=0)
    store.push(w.x, w.y, w.t_ns)
    grp = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0, 1])
    with pytest.raises(hip.CmaxHipError) as e:
        _from(grp, store, 0, len(w.x), w)
    assert e.value.status == _lib.ERR_INVALID_ARG and "replica" in str(e.value)
    both = hip.EventStore(w.W, w.H, capacity=20_000, devices=[0, 1])
    assert both.devices == [0, 1]
    both.push(w.x, w.y, w.t_ns)
    _from(grp, both, 0, len(w.x), w)
    one = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, device=1)
    _from(one, both, 0, len(w.x), w)
    (c, g), (c1, g1) = grp.eval(np.zeros(w.P)), one.eval(np.zeros(w.P))
    assert rel_scalar(c, c1) < 1e-6 and rel_vec(g, g1) < 1e-6


# ---------------------------------------------------------------- round 5: advisor findings + the spin policy
def test_group_whose_setup_failed_refuses_calls_instead_of_hanging():
    """cmx_backend_create_group can fail AFTER the members exist (here: RCCL asked for two members on one device).  The handle
    comes back for cmx_last_error / cmx_destroy; every other call must fail with CMX_ERR_STATE -- no worker thread exists that
    would ever collect a fan-out."""
    import ctypes as C
    L = _lib.lib()
    w = synth.backend_window(2_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 2, 5, 0, 0.2, seed=93)
    lut = np.ascontiguousarray(w.lut, np.float64).reshape(-1)
    ctx = C.c_void_p()
    dv = (C.c_int * 2)(0, 0)
    rc = L.cmx_backend_create_group(C.byref(ctx), dv, 2, w.W, w.H, lut.ctypes.data_as(_lib.c_dp), w.Wp, w.Hp, _lib.GROUP_RCCL)
    assert rc == _lib.ERR_INVALID_ARG and ctx.value
    assert b"RCCL cannot place two ranks on one device" in L.cmx_last_error(ctx)
    assert L.cmx_set_option(ctx, _lib.OPT_REUSE_IMAGE, 0) == _lib.ERR_STATE
    assert b"was not set up" in L.cmx_last_error(ctx)
    assert L.cmx_set_sched_class(ctx, 0) == _lib.ERR_STATE
    x = np.zeros(w.P)
    c = C.c_double()
    assert L.cmx_backend_eval(ctx, x.ctypes.data_as(_lib.c_dp), C.byref(c), None) != _lib.OK
    L.cmx_destroy(ctx)


def test_group_pose_table_is_the_windows_and_leaves_the_live_table_alone(hip):
    """cmx_backend_get_pose_table on a group: the members' rows one after the other = the table of the single context on the whole
    window, at the LAST EVALUATION's parameters even after a prepare(hint) moved the live table -- and reading it does not disturb
    the evaluation that follows (the tile sort of the prepared window was built on the live table)."""
    w = synth.backend_window(40_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 8, 2, 0.25, seed=94)
    grp, one = _pair(hip, w, [0, 0, 0])
    d = np.random.default_rng(2).normal(0, 0.01, w.P)
    hint = np.random.default_rng(3).normal(0, 0.02, w.P)
    R0, J0, i0, t0 = one.get_pose_table()            # before any evaluation: zero increments
    Rg, Jg, ig, tg = grp.get_pose_table()
    assert len(tg) == len(t0) == (len(w.x) - 1 + w.batch - 1) // w.batch
    assert np.array_equal(tg, t0) and np.array_equal(ig, i0) and np.array_equal(Rg, R0) and np.array_equal(Jg, J0)
    for e in (grp, one):
        e.eval(d)
        e.prepare(hint)                              # the live table now stands at the hint
    R1, J1, i1, t1 = one.get_pose_table()
    Rg, Jg, ig, tg = grp.get_pose_table()
    assert np.array_equal(Rg, R1) and np.array_equal(Jg, J1) and np.array_equal(tg, t1)
    assert not np.array_equal(R1, R0)                # ... the last evaluation's parameters, not zero, not the hint
    fresh = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    fresh.set_fast_path()
    _set(fresh, w)
    fresh.eval(d)
    assert np.array_equal(fresh.get_pose_table()[0], R1)
    (c, g), (c1, g1) = grp.eval(hint), fresh.eval(hint)   # the prepared window evaluates at the hint as if nothing had been read
    assert rel_scalar(c, c1) < 1e-6 and rel_vec(g, g1) < 1e-6
    for e in (grp, one, fresh):
        e.close()


def test_group_prepare_with_an_empty_member_keeps_the_collectives_matched(hip):
    """eval(x); prepare(hint); eval(x) on a group where two of four members hold no events: the empty members must drop their
    'image is resident' state in prepare like their peers, or they skip the exchange the others enter (a 20 s barrier timeout)."""
    import time
    w = synth.backend_window(150, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 2, 5, 0, 0.2, seed=95)
    grp, one = _pair(hip, w, [0, 0, 0, 0])
    assert grp.group_info()["events_per_member"][2:] == [0, 0]
    x = np.full(w.P, 0.002)
    t0 = time.perf_counter()
    for e in (grp, one):
        e.eval(x)
        e.prepare(np.zeros(w.P))
    (c, g), (c1, g1) = grp.eval(x), one.eval(x)
    assert time.perf_counter() - t0 < 5.0
    assert rel_scalar(c, c1) < 1e-6 and rel_vec(g, g1) < 1e-6
    grp.close()
    one.close()


@pytest.mark.parametrize("spin", [0, 1, 30])
def test_spin_policy_changes_no_result_and_an_idle_group_gives_its_cores_back(hip, spin):
    """CMX_OPT_SPIN_WAIT (0 never spin / 1 default / n microseconds): same numbers; after the idle budget the workers of a group
    sleep -- the process burns no CPU while nothing is on the device."""
    import time
    w = synth.backend_window(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 2, 5, 0, 0.2, seed=96)
    grp, one = _pair(hip, w, [0, 0, 0])
    grp.set_option(_lib.OPT_SPIN_WAIT, spin)
    d = np.random.default_rng(4).normal(0, 0.004, w.P)
    _same(grp, one, [(np.zeros(w.P), True), (d, False), (d, True)])
    time.sleep(0.05)
    t_cpu, t_wall = time.process_time(), time.perf_counter()
    time.sleep(0.4)
    busy = (time.process_time() - t_cpu) / (time.perf_counter() - t_wall)
    assert busy < 0.25, "an idle group keeps %.2f cores busy" % busy      # (two spinning workers would read 2.0)
    _same(grp, one, [(d, True)])                                            # woken again
    grp.close()
    one.close()


def test_group_exchange_soak_short():
    """tools/soak_group.py at a size the suite can afford: wandering and jumping parameters, new windows in between, groups of 2 / 3 / 4
    members on one device, every 7th evaluation against a single context (60 000 evaluations of it ran clean on the round-5 build)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_group.py"), "700"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "soak ok" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])


@pytest.fixture
def forced_cross_device(hip):
    """cmax_hip_diag.h, CMX_DIAG_FORCE_CROSS_DEVICE: groups created inside take the multi-device code paths on ONE device."""
    L = _lib.lib()
    assert L.cmx_diag_set(_lib.DIAG_FORCE_CROSS_DEVICE, 1) == 0
    yield
    assert L.cmx_diag_set(_lib.DIAG_FORCE_CROSS_DEVICE, 0) == 0
    assert L.cmx_diag_set(99, 1) != 0  # unknown keys are refused


@pytest.mark.parametrize("members", [2, 3])
def test_cross_device_code_paths_on_one_device(hip, forced_cross_device, members):
    """no box with two visible devices exists (profiles/r06_partition_probe.txt: one render node, SPX, sysfs
    read-only), so the group's cross-device forms run HERE: with the switch on, a same-device group takes the system-scope-acquire
    (xdev) variants of every peer kernel -- whole-plane reduce-scatter + all-gather on the first evaluation, the one-shot
    exchange fused into the unpack afterwards, the miss exchange after a jump -- and its events release to system scope.  Same
    numbers as the single context (and as the same group without the switch, which the other tests hold against the oracle)."""
    w = synth.config4_slab(1, 8, 1_500_000)
    grp, one = _pair(hip, w, [0] * members, transport=_lib.GROUP_DIRECT)
    rng = np.random.default_rng(70 + members)
    small = rng.normal(0, 0.004, w.P)
    jump = np.tile([0.5, 0.0, 0.0], w.P // 3)
    seq = [(np.zeros(w.P), True), (small, False), (small, True), (rng.normal(0, 0.004, w.P), True), (jump, True), (np.zeros(w.P), True)]
    _same(grp, one, seq)
    s = grp.stats()
    assert s["exchange_misses"] >= 1 and s["sharded_host_syncs"] == 0, s
    # a solve through the handle, and a second window (whole planes again on its first evaluation)
    x, rep = grp.setupProblemAndOptimize()
    x1, rep1 = one.setupProblemAndOptimize()
    assert abs(rep["final_cost"] - rep1["final_cost"]) < 1e-3 * abs(rep1["final_cost"]), (rep, rep1)
    for ev in (grp, one):
        _set(ev, w, n=600_000)
    _same(grp, one, [(np.zeros(w.P), True), (small, True)])
    assert rel_img(grp.get_plane(_lib.PLANE_IWE), one.get_plane(_lib.PLANE_IWE)) < RTOL
    grp.close()
    one.close()


def test_cross_device_code_paths_small_planes_whole_plane_exchange(hip, oracle, forced_cross_device):
    """planes below 1 MB always travel whole (reduce-scatter + all-gather kernels, 16-byte peer accesses): the xdev variants of those
    two kernels against the oracle"""
    w = synth.backend_window(30_001, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 8, 2, 0.25, seed=171)
    grp = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0, 0, 0], transport=_lib.GROUP_DIRECT)
    _set(grp, w)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    rng = np.random.default_rng(3)
    for d in (np.zeros(w.P), rng.normal(0, 0.01, w.P)):
        c, g = grp.eval(d, True)
        cr, gr = ref.eval(d)
        assert rel_scalar(c, cr) < RTOL and rel_vec(g, gr) < RTOL
    grp.close()


def test_auto_transport_is_measured(hip):
    """CMX_GROUP_AUTO keeps the transport that moved a production-sized message fastest at creation.  On one device
    the only candidate is the direct transport (RCCL cannot place two ranks on a device): the calibration still runs -- 20 staged
    exchanges of 1 MB through the members' own threads -- and reports its time; an explicit transport is not timed; evaluations
    behind a calibrated group equal the single context's (and the members' contrasts agree bit for bit, or eval would have failed)."""
    w = synth.config4_slab(0, 8, 800_000)
    grp, one = _pair(hip, w, [0, 0], transport=_lib.GROUP_AUTO)
    ti = grp.group_transport_info()
    assert ti["chosen"] == _lib.GROUP_DIRECT and ti["measured"] and 1.0 < ti["us_direct"] < 5e4 and ti["us_rccl"] == -1.0, ti
    assert grp.group_info()["transport"] == _lib.GROUP_DIRECT
    rng = np.random.default_rng(31)
    _same(grp, one, [(np.zeros(w.P), True), (rng.normal(0, 0.004, w.P), True), (rng.normal(0, 0.004, w.P), False)])
    assert grp.stats()["comm_bytes"] < 2 * (1 << 20)   # (the calibration's traffic is not the evaluations')
    explicit = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0, 0], transport=_lib.GROUP_DIRECT)
    te = explicit.group_transport_info()
    assert te["chosen"] == _lib.GROUP_DIRECT and not te["measured"] and te["us_direct"] == -1.0, te
    assert one.group_transport_info() == {"chosen": _lib.GROUP_AUTO, "measured": False, "us_direct": -1.0, "us_rccl": -1.0}
    for ev in (grp, one, explicit):
        ev.close()
