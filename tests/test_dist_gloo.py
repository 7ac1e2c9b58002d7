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
