"""CPU, world_size 2, gloo: the multi-GPU exchange (cmax_slam_amd/dist.py).

The HIP evaluator cannot run here, so the split-phase evaluator handed to ShardedEvaluator is a stand-in built
from the CPU oracle (test infrastructure): accumulate = raw IWE of the rank's batch range, finish = blur + contrast
on the all-reduced planes.  What is under test is the product's sharding + all-reduce logic: whole-batch ranges,
the exchange between splat and blur, and equality with the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleSplitPhase:
    """accumulate/finish stand-in with the same contract as FrontendEvaluator's split-phase API."""

    def __init__(self, po, p, sl, accum):
        self.po, self.p, self.accum = po, p, accum
        self.fe = po.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
        self.fe.set_packet(p.x[sl], p.y[sl], p.t_ns[sl], p.t_ref_ns)
        self.np_ = p.W * p.H
        self.count = 0

    def accumulate(self, om, want_grad):
        import torch
        if want_grad:
            iwe, d = self.fe.iwe(om, deriv=True, blur=False)
            planes = np.concatenate([iwe.reshape(1, -1), np.moveaxis(d, 2, 0).reshape(3, -1)])
        else:
            planes = self.fe.iwe(om, blur=False).reshape(1, -1)
        self.count = planes.size
        self.accum[:self.count] = torch.from_numpy(planes.reshape(-1))

    def accum_count(self):
        return self.count

    def finish(self, want_grad):
        po, p = self.po, self.p
        planes = self.accum[:self.count].numpy().reshape(-1, p.H, p.W).copy()
        blurred = np.stack([po.gaussian_blur(pl, p.sigma) for pl in planes])
        return po.contrast(blurred[0], blurred[1:] if want_grad else None, 0, want_grad)


class OracleSplitPhaseAdjoint(OracleSplitPhase):
    """Stand-in with the adjoint flavour's contract: only the I plane is exchanged; finish_begin leaves this rank's
    partial gradient sums  sum_px blur(D_k of MY events) * (B - mu)  in `gsum`; finish_end scales by 2/N."""

    def __init__(self, po, p, sl, accum, gsum):
        super().__init__(po, p, sl, accum)
        self.gsum = gsum
        self.n_g = 0

    def accumulate(self, om, want_grad):
        import torch
        self.om = om
        self.count = self.np_
        self.accum[:self.count] = torch.from_numpy(self.fe.iwe(om, blur=False).reshape(-1))

    def finish_begin(self, want_grad):
        import torch
        po, p = self.po, self.p
        B = po.gaussian_blur(self.accum[:self.count].numpy().reshape(p.H, p.W).copy(), p.sigma).astype(np.float64)
        self.B = B
        self.n_g = 0
        if want_grad:
            _, d = self.fe.iwe(self.om, deriv=True, blur=True)  # blurred derivative images of MY events only
            z = B - B.mean()
            self.gsum[:3] = torch.from_numpy(np.array([(d[..., k].astype(np.float64) * z).sum() for k in range(3)]))
            self.n_g = 3

    def grad_count(self):
        return self.n_g

    def finish_end(self, want_grad):
        N = self.B.size
        c = self.B.var()
        return c, (2.0 * self.gsum[:3].numpy().copy() / N if want_grad else None)


def _worker(rank, world, port, q, adjoint=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from cmax_slam_amd import synth
    from cmax_slam_amd.dist import ShardedEvaluator, batch_range
    from oracle import pyoracle as po
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    p = synth.frontend_packet(6_050, 96, 72, 80.0, 80.0, 47.5, 35.5, seed=13)
    beg, end = batch_range(len(p.x), p.batch, rank, world)
    accum = torch.zeros(4 * p.W * p.H, dtype=torch.float32)
    if adjoint:
        gsum = torch.zeros(8, dtype=torch.float64)
        sh = ShardedEvaluator(OracleSplitPhaseAdjoint(po, p, slice(beg, end), accum, gsum), accum, gsum)
    else:
        sh = ShardedEvaluator(OracleSplitPhase(po, p, slice(beg, end), accum), accum)
    om = (0.5, -0.7, 0.3)
    c, g = sh.eval(om, True)
    c_only, _ = sh.eval(om, False)
    if rank == 0:
        q.put((c, g, c_only, beg, end))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("adjoint", [False, True])
def test_sharded_eval_equals_single_process(oracle, adjoint):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, adjoint)) for r in range(2)]
    for pr in procs:
        pr.start()
    c, g, c_only, beg, end = q.get(timeout=180)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    from cmax_slam_amd import synth
    p = synth.frontend_packet(6_050, 96, 72, 80.0, 80.0, 47.5, 35.5, seed=13)
    assert beg == 0 and end == 3100  # 61 batches -> 31 + 30, whole batches per rank
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    c_ref, g_ref = ref.eval((0.5, -0.7, 0.3))
    # the only difference is fp32 summation order across the two partial images
    assert abs(c - c_ref) < 1e-6 * abs(c_ref) and abs(c_only - c_ref) < 1e-6 * abs(c_ref)
    assert np.abs(g - g_ref).max() < 1e-5 * np.abs(g_ref).max()  # adjoint: two exchanges (I plane, 3 doubles)


# ------------------------------------------------------------------------------------------------ back end
# north_star's sharded path is the back-end window (event_pano_warper.cpp:188-196 is the loop being split): two vote
# planes (IL_old / IL_new), per-rank knot support, the num_fixed column rule (:316-332).  The stand-ins below hold the
# contract of BackendEvaluator's split-phase API over the CPU oracle's vote loop.
BE_ARGS = dict(N=9_061, W=96, H=72, f=80.0, Wp=256, Hp=128, order=4, K=10, nf=3, T=0.35, seed=21)


def _be_window():
    from cmax_slam_amd import synth
    a = BE_ARGS
    return synth.backend_window(a["N"], a["W"], a["H"], a["f"], a["f"], (a["W"] - 1) / 2, (a["H"] - 1) / 2, a["Wp"], a["Hp"],
                                a["order"], a["K"], a["nf"], a["T"], seed=a["seed"])


def _be_prior_map(w):
    yy, xx = np.mgrid[0:w.Hp, 0:w.Wp]
    IG = (3.0 * np.exp(-((xx - 0.55 * w.Wp) ** 2 + (yy - 0.5 * w.Hp) ** 2) / 300.0)).astype(np.float32)
    IG[IG < 0.05] = 0
    return IG


class OracleBackendSplitPhase:
    """Faithful flavour: IL_old, IL_new and the P derivative planes of the rank's events are exchanged unblurred."""

    def __init__(self, po, w, sl, accum, IG):
        self.po, self.w, self.accum = po, w, accum
        self.be = po.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, w.batch, w.sample_rate, w.sigma, 0)
        self.be.set_window(w.x[sl], w.y[sl], w.t_ns[sl], w.knots_init, w.start_ns, w.dt_ns, w.num_fixed,
                           w.t_next_win_beg_ns, IG)
        self.np_ = w.Wp * w.Hp
        self.alpha = None   # frozen by the window's first evaluation (event_pano_warper.cpp:201-210)
        self.count = 0

    def accumulate(self, d, want_grad):
        import torch
        self.d = d
        old, new, pl = self.be.accumulate_raw(d, planes=want_grad)
        self.touched = None if pl is None else np.array([np.any(p) for p in pl])
        planes = [old.reshape(1, -1), new.reshape(1, -1)] + ([pl.reshape(pl.shape[0], -1)] if want_grad else [])
        planes = np.concatenate(planes)
        self.count = planes.size
        self.accum[:self.count] = torch.from_numpy(planes.reshape(-1))

    def accum_count(self):
        return self.count

    def _compose(self, old, new):
        po, w = self.po, self.w
        IL = old + new                                                          # :199 cv::add
        if self.alpha is None:
            self.alpha = po.lib().orc_be_alpha(po._fp(self.be.IG), po._fp(np.ascontiguousarray(IL)), IL.size)
        return (self.be.IG * np.float32(self.alpha) + IL).astype(np.float32)      # :213 cv::scaleAdd, fp32

    def finish(self, want_grad):
        po, w = self.po, self.w
        planes = self.accum[:self.count].numpy().reshape(-1, w.Hp, w.Wp).copy()
        iwe = po.gaussian_blur(self._compose(planes[0], planes[1]), w.sigma)
        D = np.stack([po.gaussian_blur(p, w.sigma) for p in planes[2:]]) if want_grad else None
        return po.contrast(iwe, D, 0, want_grad)


class OracleBackendSplitPhaseAdjoint(OracleBackendSplitPhase):
    """Adjoint flavour: only IL_old / IL_new travel; each rank then forms  sum_px blur(D_k of MY events) * (B - mu)
    for the P columns and those P doubles are summed across ranks (cmx_comm.cpp's second exchange)."""

    def __init__(self, po, w, sl, accum, IG, gsum):
        super().__init__(po, w, sl, accum, IG)
        self.gsum, self.n_g = gsum, 0

    def accumulate(self, d, want_grad):
        import torch
        self.d = d
        old, new, pl = self.be.accumulate_raw(d, planes=want_grad)
        self.my_planes = pl
        self.touched = None if pl is None else np.array([np.any(p) for p in pl])
        self.count = 2 * self.np_
        self.accum[:self.count] = torch.from_numpy(np.concatenate([old.reshape(-1), new.reshape(-1)]))

    def finish_begin(self, want_grad):
        import torch
        po, w = self.po, self.w
        planes = self.accum[:self.count].numpy().reshape(2, w.Hp, w.Wp).copy()
        self.B = po.gaussian_blur(self._compose(planes[0], planes[1]), w.sigma).astype(np.float64)
        self.n_g = 0
        if want_grad:
            z = self.B - self.B.mean()
            s = [(po.gaussian_blur(p, w.sigma).astype(np.float64) * z).sum() for p in self.my_planes]
            self.n_g = len(s)
            self.gsum[:self.n_g] = torch.from_numpy(np.array(s))

    def grad_count(self):
        return self.n_g

    def finish_end(self, want_grad):
        return self.B.var(), (2.0 * self.gsum[:self.n_g].numpy().copy() / self.B.size if want_grad else None)


def _be_worker(rank, world, port, q, adjoint):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from cmax_slam_amd.dist import ShardedEvaluator, batch_range
    from oracle import pyoracle as po
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    w = _be_window()
    IG = _be_prior_map(w)
    beg, end = batch_range(len(w.x), w.batch, rank, world)
    accum = torch.zeros((2 + w.P) * w.Wp * w.Hp, dtype=torch.float32)
    if adjoint:
        gsum = torch.zeros(64, dtype=torch.float64)
        ev = OracleBackendSplitPhaseAdjoint(po, w, slice(beg, end), accum, IG, gsum)
        sh = ShardedEvaluator(ev, accum, gsum)
    else:
        ev = OracleBackendSplitPhase(po, w, slice(beg, end), accum, IG)
        sh = ShardedEvaluator(ev, accum)
    d0 = np.zeros(w.P)
    d1 = np.random.default_rng(5).normal(0, 0.01, w.P)
    c0, g0 = sh.eval(d0, True)       # the window's first evaluation fixes alpha on the SUMMED IL (every rank the same)
    touched = ev.touched.copy()
    c1, g1 = sh.eval(d1, True)
    c1f, _ = sh.eval(d1, False)
    q.put((rank, c0, g0, c1, g1, c1f, ev.alpha, touched, beg, end))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("adjoint", [False, True])
def test_sharded_backend_window_equals_single_process(oracle, adjoint):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_be_worker, args=(r, 2, port, q, adjoint)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    w = _be_window()
    IG = _be_prior_map(w)
    assert (w.order, w.K, w.num_fixed, w.P) == (4, 10, 3, 21)
    # whole batches per rank; 9061 = 90 batches + a trailing ONE-event batch that the reference's loop skips (:188) --
    # it lands on the last rank, whose own loop skips it just the same
    (b0, e0), (b1, e1) = res[0][8:10], res[1][8:10]
    assert (b0, e0, b1, e1) == (0, 4600, 4600, 9061)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, w.batch, w.sample_rate, w.sigma, 0)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, IG)
    d1 = np.random.default_rng(5).normal(0, 0.01, w.P)
    c0_ref, g0_ref = ref.eval(np.zeros(w.P))
    c1_ref, g1_ref = ref.eval(d1)
    assert ref.alpha > 0
    for rank, c0, g0, c1, g1, c1f, alpha, touched, _, _ in res:
        assert abs(alpha - ref.alpha) < 1e-6 * ref.alpha
        assert abs(c0 - c0_ref) < 1e-6 * abs(c0_ref) and abs(c1 - c1_ref) < 1e-6 * abs(c1_ref)
        assert abs(c1f - c1_ref) < 1e-6 * abs(c1_ref)
        assert np.abs(g0 - g0_ref).max() < 1e-5 * np.abs(g0_ref).max()
        assert np.abs(g1 - g1_ref).max() < 1e-5 * np.abs(g1_ref).max()
    # both ranks return the same bits (replicated FR-CG drivers must take identical decisions)
    assert res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][4], res[1][4])
    # per-rank knot support: a rank's time range reaches only some control poses.  The first half of the window sits on
    # segments 0..3 = knots 0..6: its votes for the three FIXED knots are dropped by the j >= 0 rule (:318) and it never
    # touches the columns of knots 7..9; the second half reaches them -- every column is non-zero after the exchange
    t0, t1 = res[0][7].reshape(-1, 3).any(axis=1), res[1][7].reshape(-1, 3).any(axis=1)
    assert t0[0] and not t0[-1] and not t0[-2] and t1[-1] and (t0 | t1).all()
    assert np.all(np.abs(g1_ref.reshape(-1, 3)).max(axis=1) > 0)
