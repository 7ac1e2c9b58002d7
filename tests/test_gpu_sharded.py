"""-m gpu: the multi-GPU split-phase path on ONE GPU.

(1) two evaluators, each holding half of the event batches, exchange their partial planes (and, in adjoint mode,
    their partial gradient sums) through plain tensor adds instead of RCCL -- exactly what the all-reduce would
    deliver -- and must reproduce the single-evaluator result and the oracle;
(2) the real ShardedEvaluator over a world_size-1 NCCL(=RCCL) process group: torch-owned accumulation planes,
    torch stream, in-place all_reduce on the evaluator's buffers."""
import os
import socket

import numpy as np
import pytest

from cmax_slam_amd import synth
from cmax_slam_amd.dist import ShardedEvaluator, attach_torch_accum, batch_range
from util import RTOL, rel_img, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def _exchange(torch, bufs, n):
    tot = bufs[0][:n] + bufs[1][:n]
    for b in bufs:
        b[:n] = tot


@pytest.mark.parametrize("fast", [False, True])
def test_frontend_two_shards_on_one_gpu(hip, oracle, fast):
    import torch
    p = synth.frontend_packet(50_050, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=41)
    dev = torch.device("cuda", 0)
    evs, accs, gss = [], [], []
    for r in range(2):
        beg, end = batch_range(len(p.x), p.batch, r, 2)
        fe = hip.reference_shaped.FrontendEvaluator(p.W, p.H, p.lut)
        if fast:
            fe.set_fast_path()
        fe.set_packet(p.x[beg:end], p.y[beg:end], p.t_ns[beg:end], p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
        acc = torch.zeros(fe.accum_capacity(), dtype=torch.float32, device=dev)
        gs = torch.zeros(8, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        fe.set_accum_buffer(acc.data_ptr(), acc.numel())
        fe.set_grad_buffer(gs.data_ptr(), gs.numel())
        evs.append(fe); accs.append(acc); gss.append(gs)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    for om in ((0.3, -0.5, 0.2), (0.6, -0.9, 0.4)):
        for want in (True, False):
            for fe in evs:
                fe.accumulate(om, want)
            torch.cuda.synchronize()
            assert evs[0].accum_count() == evs[1].accum_count()
            _exchange(torch, accs, evs[0].accum_count())
            torch.cuda.synchronize()
            for fe in evs:
                fe.finish_begin(want)
            torch.cuda.synchronize()
            n = evs[0].grad_count()
            assert n == (6 if (fast and want) else 0)  # S1 and S2 (border term) per parameter
            if n:
                _exchange(torch, gss, n)
                torch.cuda.synchronize()
            outs = [fe.finish_end(want) for fe in evs]
            c_ref, g_ref = ref.eval(om, want)
            for c, g in outs:
                assert rel_scalar(c, c_ref) < RTOL
                if want:
                    assert rel_vec(g, g_ref) < RTOL
            assert outs[0][0] == outs[1][0]  # both ranks finish on identical planes: bit-identical contrast


def test_backend_two_shards_on_one_gpu(hip, oracle):
    import torch
    w = synth.backend_window(40_040, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 10, 3, 0.35, seed=42)
    dev = torch.device("cuda", 0)
    for fast in (False, True):
        evs, accs, gss = [], [], []
        for r in range(2):
            beg, end = batch_range(len(w.x), w.batch, r, 2)
            be = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
            if fast:
                be.set_fast_path()
            be.set_window(w.x[beg:end], w.y[beg:end], w.t_ns[beg:end], w.order, w.knots_init, w.start_ns, w.dt_ns,
                          w.num_fixed, w.t_next_win_beg_ns)
            acc = torch.zeros(be.accum_capacity(), dtype=torch.float32, device=dev)
            gs = torch.zeros(64, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            be.set_accum_buffer(acc.data_ptr(), acc.numel())
            be.set_grad_buffer(gs.data_ptr(), gs.numel())
            evs.append(be); accs.append(acc); gss.append(gs)
        ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order)
        ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
        d = np.random.default_rng(3).normal(0, 0.01, w.P)
        for be in evs:
            be.accumulate(d, True)
        torch.cuda.synchronize()
        _exchange(torch, accs, evs[0].accum_count())
        torch.cuda.synchronize()
        for be in evs:
            be.finish_begin(True)
        torch.cuda.synchronize()
        n = evs[0].grad_count()
        assert n == (2 * w.P if fast else 0)
        if n:
            _exchange(torch, gss, n)
            torch.cuda.synchronize()
        c_ref, g_ref = ref.eval(d)
        for be in evs:
            c, g = be.finish_end(True)
            assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL


def test_sharded_evaluator_over_nccl_world1(hip, oracle):
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        p = synth.frontend_packet(30_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=43)
        fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
        fe.set_fast_path()
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
        accum, gsum, stream = attach_torch_accum(fe, dev)
        sh = ShardedEvaluator(fe, accum, gsum, force_collectives=True)
        ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
        ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        for om in ((0.3, -0.5, 0.2), (0.1, 0.2, 0.3)):
            with torch.cuda.stream(stream):
                c, g = sh.eval(om, True)
                c0, _ = sh.eval(om, False)
            c_ref, g_ref = ref.eval(om)
            assert rel_scalar(c, c_ref) < RTOL and rel_scalar(c0, c_ref) < RTOL
            assert rel_vec(g, g_ref) < RTOL
    finally:
        dist.destroy_process_group()


def test_native_rccl_communicator_world1(hip, oracle):
    """cmx_comm_attach with a 1-rank RCCL communicator: every evaluation (and the C++ solver) issues its ncclAllReduce
    calls from inside the evaluator; results must be unchanged."""
    p = synth.frontend_packet(30_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=43)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    for fast in (True, False):
        fe = hip.reference_shaped.FrontendEvaluator(p.W, p.H, p.lut)
        if fast:
            fe.set_fast_path()
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
        fe.comm_attach(fe.comm_unique_id(), 0, 1)
        for om in ((0.3, -0.5, 0.2), (0.1, 0.2, 0.3)):
            c_ref, g_ref = ref.eval(om)
            assert rel_scalar(fe.eval(om, want_grad=False)[0], c_ref) < RTOL
            c, g = fe.eval(om)   # df right after f at the same point: image reuse + gradient-sum exchange
            assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL
        x, rep = fe.setupProblemAndOptimize(np.zeros(3))
        assert rep["final_cost"] < rep["initial_cost"]
        fe.comm_detach()
        c, g = fe.eval((0.3, -0.5, 0.2))
        assert rel_scalar(c, ref.eval((0.3, -0.5, 0.2))[0]) < RTOL

    w = synth.backend_window(30_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 256, 128, 4, 10, 3, 0.35, seed=44)
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    be.comm_attach(be.comm_unique_id(), 0, 1)
    rb = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order)
    rb.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    d = np.full(w.P, 0.002)
    c, g = be.eval(d)
    c_ref, g_ref = rb.eval(d)
    assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL


@pytest.mark.parametrize("Wp,Hp", [(2048, 1024), (1100, 565)])
def test_tile_set_exchange_on_a_panorama_world1(hip, Wp, Hp):
    """Planes of 1 MB and more are exchanged as the set of tiles any rank voted into in the previous evaluation (with the
    occupancy map behind them).  With one rank the collectives are identities, so every result must equal the run without a
    communicator -- across a sequence of evaluations that moves the votes (also out of the set: repaired), alternates cost-only
    and gradient calls (ping-pong buffers, image reuse) and includes a non-zero global map.  1100 x 565: partial tiles at the
    right and bottom edges, a tile count that is no multiple of 16."""
    w = synth.backend_window(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5, Wp, Hp, 2, 5, 1, 0.2, seed=45)
    IG = np.zeros((w.Hp, w.Wp), np.float32)
    IG[300 * Hp // 1024:340 * Hp // 1024, 900 * Wp // 2048:1100 * Wp // 2048] = 1.5   # map content where this window's events do not reach
    IG[500 * Hp // 1024:520 * Hp // 1024, 1000 * Wp // 2048:1040 * Wp // 2048] = 0.7
    plain = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    shard = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    for ev in (plain, shard):
        ev.set_fast_path()
        ev.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns,
                      w.batch, w.sample_rate, w.sigma, 0, IG)
    shard.comm_attach(shard.comm_unique_id(), 0, 1)
    rng = np.random.default_rng(7)
    big = np.tile([0.2, 0.0, 0.0], w.P // 3)   # a pitch of 11 degrees: the votes move to other tile rows
    for i, d in enumerate([np.zeros(w.P), big, rng.normal(0, 0.01, w.P), -big, np.zeros(w.P), np.zeros(w.P)]):
        want = i % 2 == 0 or i == 5
        c0, g0 = plain.eval(d, want)
        c1, g1 = shard.eval(d, want)
        assert rel_scalar(c1, c0) < 1e-7, (i, c0, c1)
        if want:
            assert rel_vec(g1, g0) < 1e-6, i
        assert rel_img(shard.get_plane(_lib_plane("IL_OLD")), plain.get_plane(_lib_plane("IL_OLD"))) < 1e-6
    assert rel_scalar(shard.alpha, plain.alpha) < 1e-7 and plain.alpha > 0
    st = shard.stats()
    assert st["exchange_tiles"] > 0 and st["exchange_misses"] >= 1 and st["sharded_host_syncs"] == 0, st
    x0, r0 = plain.setupProblemAndOptimize()
    x1, r1 = shard.setupProblemAndOptimize()
    assert abs(r1["final_cost"] - r0["final_cost"]) < 1e-3 * abs(r0["final_cost"])


def test_split_phase_finish_on_a_tile_list_panorama_world1(hip):
    """4096x2048 (8192 image tiles > 2048): the image passes walk the compacted tile list and their moment rows are
    compact; the split-phase finish (finish_begin / all-reduce / finish_end, what an attached communicator runs) must
    finalize from those same rows.  One rank: results equal the plain evaluation, including a non-zero global map."""
    w = synth.backend_window(80_000, 320, 240, 260.0, 260.0, 159.5, 119.5, 4096, 2048, 2, 5, 0, 0.2, seed=46)
    IG = np.zeros((w.Hp, w.Wp), np.float32)
    IG[900:960, 1900:2200] = 1.2
    plain = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    shard = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    for ev in (plain, shard):
        ev.set_fast_path()
        ev.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns,
                      w.batch, w.sample_rate, w.sigma, 0, IG)
    shard.comm_attach(shard.comm_unique_id(), 0, 1)
    rng = np.random.default_rng(8)
    for i, d in enumerate([np.zeros(w.P), rng.normal(0, 0.01, w.P), rng.normal(0, 0.01, w.P), np.zeros(w.P)]):
        want = i != 2
        c0, g0 = plain.eval(d, want)
        c1, g1 = shard.eval(d, want)
        assert rel_scalar(c1, c0) < 1e-7, (i, c0, c1)
        if want:
            assert rel_vec(g1, g0) < 1e-6, i
    assert rel_scalar(shard.alpha, plain.alpha) < 1e-7 and plain.alpha > 0


def _lib_plane(name):
    from cmax_slam_amd import _lib
    return getattr(_lib, "PLANE_" + name)
