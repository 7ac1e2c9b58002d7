"""-m gpu: the device-driven front-end solve (CMX_OPT_CHAIN_SOLVE, cmx_chain.cpp): the FR-CG state machine advances inside the
finalize steps on the device, the host replays it on the reported costs / gradients.  The result must be that of the host-driven
loop (src/frontend/local_optim_contrast_gsl.cpp:74-233 restated in cmx_frcg_sm.hpp) -- the evaluations themselves differ run to
run by the order of the fp32 atomics, so the two are compared by what they reach, as tests/test_gpu_baseline_configs.py does; the
hand-over paths (host takes over between two points / between a cost and its gradient) are forced through the test hooks."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, solver, synth

pytestmark = pytest.mark.gpu


def _fe(hip, p, chain, sigma=None, measure=_lib.VARIANCE):
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    fe.set_option(_lib.OPT_CHAIN_SOLVE, chain)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma if sigma is None else sigma, measure)
    return fe


def _close(a, b):
    """Two solves of the same problem by two drivers.  Their evaluations differ in the last bits (order of the fp32 atomics), and
    FR-CG's line search and stopping rules amplify that: over 2000 random starts two HOST-driven runs of one packet end more than
    0.2 % apart in 4-5 % of the cases, up to a third in the worst (profiles/r03_chain_vs_host_random.txt), and one of the packets
    below has two end points 1.4 % apart that either driver reaches (40 repetitions of this file: 5 landed on different ones).  So
    "the same result" is: the same start, both descended, end points within 5 % in cost and 0.2 rad/s in omega.  What pins the
    device's machine to the host's is elsewhere: every solve here must finish WITHOUT a hand-over, i.e. the host's replay of the
    machine on the reported costs / gradients agreed bit for bit with every point the device chose."""
    (xa, ra), (xb, rb) = a, b
    assert abs(ra["final_cost"] - rb["final_cost"]) < 5e-2 * abs(rb["final_cost"]), (ra, rb)
    assert np.abs(xa - xb).max() < 0.2, (xa, xb)
    assert ra["initial_cost"] == pytest.approx(rb["initial_cost"], rel=1e-6)
    assert ra["final_cost"] <= ra["initial_cost"] and rb["final_cost"] <= rb["initial_cost"]


@pytest.mark.parametrize("n_events,W,H,mode", [(100_000, 240, 180, 1), (400_000, 640, 480, 1), (100_000, 240, 180, 4)])
def test_chain_solve_reaches_what_the_host_driven_solve_reaches(hip, oracle, n_events, W, H, mode):
    """mode 1: self-gating slots (the image pass runs no finalize; the launch behind it decides whether it is the gradient pass);
    4: the first form (finalize behind the image pass, flag-gated gradient pass), which other blur radii always take."""
    p = synth.frontend_packet(n_events, W, H, 0.9 * W, 0.9 * W, (W - 1) / 2, (H - 1) / 2, seed=77)
    host = _fe(hip, p, 0).setupProblemAndOptimize(np.zeros(3))
    fe = _fe(hip, p, mode)
    dev = fe.setupProblemAndOptimize(np.zeros(3))
    st = fe.stats()
    assert st["chain_solves"] == 1 and st["chain_takeovers"] == 0 and st["chain_slots"] >= dev[1]["n_f"] + 1
    _close(dev, host)
    assert dev[1]["iterations"] >= 2 and host[1]["iterations"] >= 2
    # ... and what the same driver reaches over the CPU oracle
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, oracle.VARIANCE)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)

    def fdf(x, wg):
        c, g = ref.eval(x, wg)
        return -c, (-g if wg else None)
    x_ref, rep_ref = solver.frcg_minimize(fdf, np.zeros(3), **solver.FRONTEND)
    _close(dev, (x_ref, rep_ref))
    # the context is usable as before: a plain evaluation, a second solve (warm start), a new packet
    # (half-way to the solution: AT the solution the gradient is a 100x smaller difference of the same fp32 sums, and 1e-5 of it
    # is the reference arithmetic's own noise -- DESIGN.md section 2; that regime is tests/test_gpu_sweep300.py's, with its arbiter)
    c, g = fe.eval(0.5 * dev[0])
    c_ref, g_ref = ref.eval(0.5 * dev[0])
    assert abs(c - c_ref) < 1e-5 * abs(c_ref) and np.abs(g - g_ref).max() < 1e-5 * np.abs(g_ref).max()
    again = fe.setupProblemAndOptimize(dev[0])
    assert again[1]["final_cost"] <= dev[1]["final_cost"] + 1e-6 * abs(dev[1]["final_cost"])   # (costs are -contrast: a warm restart does not climb)
    assert fe.stats()["chain_solves"] == 2


@pytest.mark.parametrize("hook", [2, 3])
def test_host_takes_over_mid_solve(hip, hook):
    p = synth.frontend_packet(100_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=78)
    host = _fe(hip, p, 0).setupProblemAndOptimize(np.zeros(3))
    fe = _fe(hip, p, hook)
    dev = fe.setupProblemAndOptimize(np.zeros(3))
    assert fe.stats()["chain_takeovers"] == 1
    _close(dev, host)
    assert dev[1]["n_f"] + dev[1]["n_df"] >= 6   # the solve went on after the hand-over
    c0, _ = fe.eval(np.zeros(3), False)          # accumulators / counters were left clean
    assert c0 == pytest.approx(-dev[1]["initial_cost"], rel=1e-6)


@pytest.mark.parametrize("sigma,measure", [(0.0, _lib.VARIANCE), (0.5, _lib.VARIANCE), (2.0, _lib.VARIANCE), (1.0, _lib.MEAN_SQUARE)])
def test_chain_solve_other_image_passes(hip, sigma, measure):
    """Radius 0 / 2 / 8 take the general image kernels with a separate finalize launch (the machine's step runs there)."""
    p = synth.frontend_packet(80_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=79)
    host = _fe(hip, p, 0, sigma, measure).setupProblemAndOptimize(np.zeros(3))
    fe = _fe(hip, p, 1, sigma, measure)
    dev = fe.setupProblemAndOptimize(np.zeros(3))
    assert fe.stats()["chain_solves"] == 1 and fe.stats()["chain_takeovers"] == 0
    _close(dev, host)


def test_chain_solve_back_to_back_packets_and_ineligible_configurations(hip):
    fe = None
    for k in range(4):
        p = synth.frontend_packet(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=90 + k)
        if fe is None:
            fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
            fe.set_fast_path()
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
        dev = fe.setupProblemAndOptimize(np.zeros(3))
        host = _fe(hip, p, 0).setupProblemAndOptimize(np.zeros(3))
        _close(dev, host)
    st = fe.stats()
    assert st["chain_solves"] == 4 and st["chain_takeovers"] == 0
    assert st["chain_warm_starts"] == 3   # every solve after the first started without the initial copy / any clearing
    # deterministic mode and the reference-shaped path are solved host-driven (and bitwise reproducibly in the former)
    fe.set_deterministic(True)
    a = fe.setupProblemAndOptimize(np.zeros(3))
    b = fe.setupProblemAndOptimize(np.zeros(3))
    assert fe.stats()["chain_solves"] == 4 and np.array_equal(a[0], b[0])


def test_chain_solve_resorts_the_events_mid_solve_and_runs_beside_another_context(hip):
    """A fast rotation: the tile sort taken at omega = 0 no longer fits a few points into the solve (> 3 % of the votes leave
    their windows), so the chain re-sorts at the device's current point (the sort kernels read omega from device memory too).
    Two contexts solved from two host threads at once reach what they reach alone."""
    import threading
    p = synth.frontend_packet(150_000, 320, 240, 300.0, 300.0, 159.5, 119.5, T=0.12, omega_true=(2.5, -3.5, 1.5), seed=81)
    x0 = 0.3 * np.array(p.omega_true)   # (from omega = 0 this packet is a flat start: both drivers stop after 472 halvings)
    host = _fe(hip, p, 0).setupProblemAndOptimize(x0)
    fe = _fe(hip, p, 1)
    dev = fe.setupProblemAndOptimize(x0)
    st = fe.stats()
    assert st["chain_solves"] == 1 and st["chain_takeovers"] == 0 and st["rebins"] >= 2, st
    _close(dev, host)
    assert np.abs(dev[0][:2] - p.omega_true[:2]).max() < 0.5
    # the flat start: intermediate_point halves the step until a probe is a few ulps below the start or the point stops moving --
    # tens to hundreds of chained cost-only slots (how many is decided by the last bits of sums whose order varies run to run)
    flat = _fe(hip, p, 1).setupProblemAndOptimize(np.zeros(3))
    flat_h = _fe(hip, p, 0).setupProblemAndOptimize(np.zeros(3))
    for x, rep in (flat, flat_h):
        assert rep["n_f"] >= 10 and np.abs(x).max() < 1e-2 and abs(rep["final_cost"] - rep["initial_cost"]) < 1e-6 * abs(rep["initial_cost"])
    q = synth.frontend_packet(90_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=82)
    alone = [_fe(hip, pk, 1).setupProblemAndOptimize(np.zeros(3)) for pk in (p, q)]
    evs = [_fe(hip, pk, 1) for pk in (p, q)]
    out = [None, None]

    def work(i):
        for _ in range(5):
            out[i] = evs[i].setupProblemAndOptimize(np.zeros(3))
    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in range(2):
        _close(out[i], alone[i])
        assert evs[i].stats()["chain_solves"] == 5 and evs[i].stats()["chain_takeovers"] == 0


def test_warm_started_solves_between_other_uses_of_the_context(hip, oracle):
    """A device-driven solve that ends normally leaves the device-side state (end-of-solve flag, moment rows, ping-pong planes,
    accumulator rows) as the next solve's first slot expects it: that solve starts without the initial copy and without clearing
    anything (stats: chain_warm_starts).  Whatever happens in between -- plain evaluations, an image read-back, a new packet, a
    solve the host took over (cold start after it), a host-driven solve -- every solve must reach what the host-driven one does."""
    p = synth.frontend_packet(120_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=95)
    q = synth.frontend_packet(70_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=96)
    host_p = _fe(hip, p, 0).setupProblemAndOptimize(np.zeros(3))
    host_q = _fe(hip, q, 0).setupProblemAndOptimize(np.zeros(3))
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, oracle.VARIANCE)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    fe = _fe(hip, p, 1)
    warm = 0
    for k in range(6):
        dev = fe.setupProblemAndOptimize(np.zeros(3))
        _close(dev, host_p)
        assert fe.stats()["chain_warm_starts"] == warm, (k, fe.stats())
        warm += 1
        if k == 1:      # plain evaluations between two solves (they use the ping-pong planes and the accumulator rows)
            for x in (np.zeros(3), dev[0]):
                c, g = fe.eval(x)
                c_ref, g_ref = ref.eval(x)
                assert abs(c - c_ref) < 1e-5 * abs(c_ref) and np.abs(g - g_ref).max() < 1e-5 * np.abs(g_ref).max()
        if k == 2:      # the image of the last evaluated point
            iwe = fe.computeImageOfWarpedEvents(dev[0])
            assert np.isfinite(iwe).all() and iwe.sum() > 0
        if k == 3:      # odd and even numbers of cost-only evaluations in between (which ping-pong plane is the clean one)
            fe.eval(0.5 * dev[0], False)
        if k == 4:
            fe.eval(0.5 * dev[0], False)
            fe.eval(0.25 * dev[0], False)
    # a new packet on the same context: still warm
    fe.set_packet(q.x, q.y, q.t_ns, q.t_ref_ns, q.fx, q.fy, q.cx, q.cy, q.batch, q.sigma, _lib.VARIANCE)
    _close(fe.setupProblemAndOptimize(np.zeros(3)), host_q)
    assert fe.stats()["chain_warm_starts"] == warm
    warm += 1
    # a solve the host takes over leaves nothing to rely on: the next one starts cold, the one after it warm again
    fe.set_option(_lib.OPT_CHAIN_SOLVE, 2)
    _close(fe.setupProblemAndOptimize(np.zeros(3)), host_q)
    assert fe.stats()["chain_takeovers"] == 1 and fe.stats()["chain_warm_starts"] == warm   # (the taken-over solve itself started warm)
    fe.set_option(_lib.OPT_CHAIN_SOLVE, 1)
    _close(fe.setupProblemAndOptimize(np.zeros(3)), host_q)
    assert fe.stats()["chain_warm_starts"] == warm       # cold
    _close(fe.setupProblemAndOptimize(np.zeros(3)), host_q)
    warm += 1
    assert fe.stats()["chain_warm_starts"] == warm       # warm again
    # a host-driven solve in between does not disturb the state
    fe.set_option(_lib.OPT_CHAIN_SOLVE, 0)
    _close(fe.setupProblemAndOptimize(np.zeros(3)), host_q)
    fe.set_option(_lib.OPT_CHAIN_SOLVE, 1)
    _close(fe.setupProblemAndOptimize(np.zeros(3)), host_q)
    assert fe.stats()["chain_warm_starts"] == warm + 1 and fe.stats()["chain_takeovers"] == 1


@pytest.mark.parametrize("fused", [0, 1])
def test_flat_start_beside_a_busy_context_keeps_the_accumulators_clean(hip, oracle, fused):
    """in a self-gating slot whose outcome is 'cost only', workgroup 0 runs the finalize and the machine's step
    while other workgroups of the SAME launch may not have started; they used to read the gate from the machine that step had just
    rewritten -- a late workgroup could then take the gradient path, add to the accumulator rows and bump the arrival tickets
    with no finalize to consume them (the next slot's gradient polluted, its last-arriver detection early).  The gate now lives
    in a slot-parity copy (ChainDev::gate) the launch's own finalize never writes.  Stress: the halving loop of a flat start
    (hundreds of chained cost-only slots whose next request often passes for the same contrast) while a second context keeps
    the compute units busy with large launches, so that workgroups of the one-block-wide decisions arrive late; after every solve
    a plain gradient evaluation on the same context must equal the oracle's (it reads the accumulator rows and tickets the solve
    left behind), and the warm-started solve after it must reach what the host-driven one reaches."""
    import threading
    p = synth.frontend_packet(150_000, 320, 240, 300.0, 300.0, 159.5, 119.5, T=0.12, omega_true=(2.5, -3.5, 1.5), seed=81)
    big = synth.frontend_packet(1_000_000, 640, 480, 570.0, 570.0, 319.5, 239.5, seed=84)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, oracle.VARIANCE)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    x_chk = 0.3 * np.array(p.omega_true)
    c_ref, g_ref = ref.eval(x_chk)
    host_flat = _fe(hip, p, 0).setupProblemAndOptimize(np.zeros(3))
    host_good = _fe(hip, p, 0).setupProblemAndOptimize(x_chk)
    noisy = _fe(hip, big, 0)
    stop = threading.Event()

    def hammer():
        xs = np.random.default_rng(1).normal(0, 1.0, (64, 3))
        while not stop.is_set():
            noisy.eval_many(xs, True)        # 64 evaluations of 1M events queued back to back: the GPU never idles
    th = threading.Thread(target=hammer)
    th.start()
    try:
        fe = _fe(hip, p, 1)
        fe.set_option(_lib.OPT_FUSED_IMAGE, fused)  # round 6: slots of two launches (image pass inside the splat launch) / of three
        for k in range(6):
            flat = fe.setupProblemAndOptimize(np.zeros(3))
            assert flat[1]["n_f"] >= 10 and np.abs(flat[0]).max() < 1e-2
            assert abs(flat[1]["final_cost"] - host_flat[1]["final_cost"]) < 1e-5 * abs(host_flat[1]["final_cost"])
            c, g = fe.eval(x_chk)
            assert abs(c - c_ref) < 1e-5 * abs(c_ref) and np.abs(g - g_ref).max() < 1e-5 * np.abs(g_ref).max(), (k, g, g_ref)
            good = fe.setupProblemAndOptimize(x_chk)
            _close(good, host_good)
        st = fe.stats()
        # no take-over by a workgroup-vs-machine disagreement.  With fused slots the solves from 0.3 omega_true (a 57-pixel journey
        # at the ends of this 120 ms packet, beyond the 56 px the tiles' arrival counts cover) hand over to the host when a slot
        # reports votes out of reach -- every take-over must be one of those
        assert st["chain_solves"] == 12 and st["chain_takeovers"] <= (st["fused_redos"] if fused else 0), st
    finally:
        stop.set()
        th.join()
