"""-m gpu: the sharded evaluation with TWO ranks on ONE GPU.

The 8-GPU node is not available to the builder, and RCCL refuses two ranks on one device, so the exchange logic of
cmx_comm.cpp is driven here through cmx_comm_attach_custom: two contexts, two host threads, and an all-reduce that meets at
a host barrier (device -> host, reduce, host -> device).  What is checked is everything a real communicator relies on:

  * both ranks issue the SAME sequence of collectives (count, dtype, op) -- also when one rank's shard is EMPTY
    (a rank without events used to take the whole-plane exchange while the others took the band exchange);
  * contrast and gradient equal the one-context evaluation of the whole window, and are bit-identical across ranks;
  * the tile-set exchange of panoramas: set known after the first evaluation, no host synchronisation inside an
    evaluation, and a jump of the parameters that moves the votes out of the set is detected and repaired."""
import threading

import numpy as np
import pytest

from cmax_slam_amd import _lib, dist, synth
from util import rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


class _DevPtr:
    def __init__(self, ptr, count, dt):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": {0: "|u1", 1: "<f4", 2: "<f8"}[dt],
                                         "data": (int(ptr), False), "version": 2}


class HostAllreduce:
    """All-reduce for n ranks living in n threads of this process (test transport, deliberately simple)."""

    def __init__(self, n):
        import torch
        self.torch, self.n = torch, n
        self.barrier = threading.Barrier(n, timeout=120)
        self.slots = [None] * n
        self.calls = [[] for _ in range(n)]

    def rank_fn(self, rank):
        torch = self.torch

        def fn(buf, count, dt, op, stream):
            torch.cuda.synchronize()
            t = torch.as_tensor(_DevPtr(buf, count, dt), device="cuda")
            self.slots[rank] = t.cpu().numpy().copy()
            self.calls[rank].append((int(count), int(dt), int(op)))
            self.barrier.wait()
            stack = np.stack(self.slots)
            red = stack.max(axis=0) if op == _lib.OP_MAX else stack.sum(axis=0, dtype=stack.dtype)
            self.barrier.wait()  # everyone has read the slots before anyone overwrites its own
            t.copy_(torch.from_numpy(np.ascontiguousarray(red)))
            torch.cuda.synchronize()
            return 0
        return fn


def _run_ranks(fns):
    """Run fns[r]() in its own thread; re-raise the first failure."""
    out, err = [None] * len(fns), []

    def work(r):
        try:
            out[r] = fns[r]()
        except BaseException as e:  # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=work, args=(r,)) for r in range(len(fns))]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    if err:
        raise err[0]
    return out


def _backend_pair(hip, w, IG, ranges, world=2):
    ar = HostAllreduce(world)
    evs = []
    for r in range(world):
        beg, end = ranges[r]
        be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
        be.set_fast_path()
        be.set_window(w.x[beg:end], w.y[beg:end], w.t_ns[beg:end], w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed,
                      w.t_next_win_beg_ns, w.batch, w.sample_rate, w.sigma, 0, IG)
        be.comm_attach_custom(ar.rank_fn(r), r, world)
        evs.append(be)
    one = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    one.set_fast_path()
    one.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                   w.sample_rate, w.sigma, 0, IG)
    return ar, evs, one


def _check_sequence(ar, evs, one, seq):
    for i, (d, want) in enumerate(seq):
        c0, g0 = one.eval(d, want)
        res = _run_ranks([lambda e=e: e.eval(d, want) for e in evs])
        for r, (c, g) in enumerate(res):
            assert rel_scalar(c, c0) < 1e-6, (i, r, c, c0)
            if want:
                assert rel_vec(g, g0) < 1e-6, (i, r)
        assert res[0][0] == res[1][0]                    # identical planes -> identical bits on every rank
        if want:
            assert np.array_equal(res[0][1], res[1][1])
        assert ar.calls[0] == ar.calls[1], (i, ar.calls[0][-6:], ar.calls[1][-6:])   # matched collectives


def test_two_ranks_large_panorama_tile_set_exchange_and_recovery(hip):
    w = synth.backend_window(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 4096, 2048, 2, 5, 0, 0.2, seed=47)
    IG = np.zeros((w.Hp, w.Wp), np.float32)
    IG[700:740, 1800:2300] = 1.3
    ranges = [dist.batch_range(len(w.x), w.batch, r, 2) for r in range(2)]
    ar, evs, one = _backend_pair(hip, w, IG, ranges)
    rng = np.random.default_rng(3)
    big = np.tile([0.25, 0.0, 0.0], w.P // 3)  # a pitch of 14 degrees: the votes move ~160 rows = 10 tile rows
    seq = [(np.zeros(w.P), True), (rng.normal(0, 0.01, w.P), True), (rng.normal(0, 0.01, w.P), False), (big, True),
           (big, True), (big + rng.normal(0, 0.005, w.P), False), (np.zeros(w.P), True)]
    _check_sequence(ar, evs, one, seq[:3])
    s = evs[0].stats()
    ntiles = ((w.Wp + 63) // 64) * ((w.Hp + 15) // 16)
    assert 0 < s["exchange_tiles"] < ntiles // 8 and s["exchange_misses"] == 0 and s["sharded_host_syncs"] == 0
    # bytes the last (cost-only) evaluation exchanged: the tile flags + the set's tiles of both planes -- a few per cent of the
    # 2 x 32 MB the whole planes would have been; both ranks count the same
    plane_bytes = w.Wp * w.Hp * 4
    assert ntiles < s["comm_bytes"] < 0.1 * 2 * plane_bytes and s["comm_bytes"] == evs[1].stats()["comm_bytes"]
    n_before = len(ar.calls[0])
    _check_sequence(ar, evs, one, seq[3:4])            # the jump: detected, repaired, results still right
    s = evs[0].stats()
    assert s["exchange_misses"] == 1 and evs[1].stats()["exchange_misses"] == 1
    # the set with the occupancy map behind it (one staging buffer), gradient sums, then the uncovered tiles and the gradient sums again
    assert len(ar.calls[0]) - n_before == 1 + 1 + 1 + 1
    _check_sequence(ar, evs, one, seq[4:])
    assert evs[0].stats()["exchange_misses"] <= 3 and evs[0].stats()["sharded_host_syncs"] == 0
    # the first evaluation exchanged whole planes (no set yet); later ones a set of tiles
    floats = [c for c in ar.calls[0] if c[1] == _lib.DT_F32]
    assert floats[0][0] == 2 * w.Wp * w.Hp and max(c[0] for c in floats[1:]) < w.Wp * w.Hp // 4
    assert all((c[0] - ntiles) % (2 * 64 * 16) == 0 or c[0] % (2 * 64 * 16) == 0 for c in floats[1:])
    assert rel_scalar(evs[0].alpha, one.alpha) < 1e-7 and one.alpha > 0


def test_two_ranks_config4_sized_window_exchanges_a_set_of_tiles(hip):
    """planes below 8 MB used to travel whole (BASELINE config 4: 2 x 4 MB per evaluation for ~3 % occupied
    tiles).  The exchange set applies from 1 MB planes on: a 1024 x 1024 panorama's evaluations exchange the flags and a few
    dozen tiles.  Results equal the single-GPU ones; a jump of the parameters is repaired like on the large panoramas."""
    w = synth.backend_window(200_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 1024, 1024, 4, 10, 1, 0.3, seed=53)
    ranges = [dist.batch_range(len(w.x), w.batch, r, 2) for r in range(2)]
    ar, evs, one = _backend_pair(hip, w, None, ranges)
    rng = np.random.default_rng(5)
    seq = [(np.zeros(w.P), True)] + [(rng.normal(0, 0.004, w.P), k % 2 == 0) for k in range(6)]
    _check_sequence(ar, evs, one, seq)
    s = evs[0].stats()
    assert s["exchange_misses"] == 0 and 0 < s["exchange_tiles"] < 1024 // 4, s
    assert s["comm_bytes"] < 2 * (1 << 20), s        # flags + staged tiles (+ gradient rows when asked for): a quarter of 2 x 4 MB
    assert s["comm_bytes"] == evs[1].stats()["comm_bytes"]
    jump = np.tile([0.6, 0.0, 0.0], w.P // 3)        # 34 degrees about the camera's x axis: other tiles
    _check_sequence(ar, evs, one, [(jump, True), (jump, False), (np.zeros(w.P), True)])
    assert 1 <= evs[0].stats()["exchange_misses"] <= 2 and evs[0].stats()["sharded_host_syncs"] == 0, evs[0].stats()


def test_two_ranks_votes_across_the_panorama_seam(hip):
    """The camera looks backwards: the votes sit at both ends of the panorama's rows (longitude +-180 degrees).  The exchange
    set dilates across the seam (tile column 0 is the neighbour of the last one), and a yaw that carries the votes over it is
    followed without a miss; results equal the one-context evaluation throughout."""
    w = synth.backend_window(150_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 2048, 1024, 4, 10, 0, 0.3, seed=61)
    ranges = [dist.batch_range(len(w.x), w.batch, r, 2) for r in range(2)]
    ar, evs, one = _backend_pair(hip, w, None, ranges)
    rng = np.random.default_rng(6)

    def yaw(deg):
        return np.tile([0.0, np.deg2rad(deg), 0.0], w.P // 3)
    seq = [(yaw(176.0), True)]
    for k, deg in enumerate([176.5, 177.5, 178.5, 179.5, 180.5, 181.5, 182.5, 183.5]):   # 1 degree = 5.7 pixels per step
        seq.append((yaw(deg) + rng.normal(0, 0.002, w.P), k % 2 == 0))
    _check_sequence(ar, evs, one, seq)
    s = evs[0].stats()
    assert s["exchange_misses"] == 0 and s["exchange_tiles"] > 0 and s["sharded_host_syncs"] == 0, s
    # both ends of the rows are occupied: the planes of one rank carry votes in the first and in the last tile column
    il = evs[0].get_plane(_lib.PLANE_IL_NEW) + evs[0].get_plane(_lib.PLANE_IL_OLD)
    assert il[:, :64].sum() > 0 and il[:, -64:].sum() > 0


def test_two_ranks_with_an_empty_shard_issue_the_same_collectives(hip):
    w = synth.backend_window(20_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 4096, 2048, 4, 6, 1, 0.15, seed=48)
    ranges = [(0, len(w.x)), (len(w.x), len(w.x))]     # rank 1 holds nothing (dist.batch_range does this when nb < world*per)
    ar, evs, one = _backend_pair(hip, w, None, ranges)
    rng = np.random.default_rng(4)
    seq = [(np.zeros(w.P), True), (rng.normal(0, 0.01, w.P), False), (rng.normal(0, 0.01, w.P), True)]
    _check_sequence(ar, evs, one, seq)
    assert any(c[1] == _lib.DT_U8 for c in ar.calls[1])   # the empty rank took the tile-set exchange like the other one


def test_two_ranks_small_panorama_whole_plane_and_solver(hip):
    w = synth.backend_window(30_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 10, 3, 0.35, seed=49)
    ranges = [dist.batch_range(len(w.x), w.batch, r, 2) for r in range(2)]
    ar, evs, one = _backend_pair(hip, w, None, ranges)
    _check_sequence(ar, evs, one, [(np.zeros(w.P), True), (np.full(w.P, 0.003), False), (np.full(w.P, 0.003), True)])
    assert all(c[1] != _lib.DT_U8 for c in ar.calls[0])   # 0.5 MB planes travel whole: no flag exchange
    # the replicated FR-CG drivers see identical numbers, so they take identical decisions and stay in lockstep
    res = _run_ranks([lambda e=e: e.setupProblemAndOptimize() for e in evs])
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    x1, r1 = one.setupProblemAndOptimize()
    assert abs(res[0][1]["final_cost"] - r1["final_cost"]) < 2e-3 * abs(r1["final_cost"])
    assert ar.calls[0] == ar.calls[1]


def test_two_ranks_frontend(hip):
    p = synth.frontend_packet(50_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=50)
    ar = HostAllreduce(2)
    evs = []
    for r in range(2):
        beg, end = dist.batch_range(len(p.x), p.batch, r, 2)
        fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
        fe.set_fast_path()
        fe.set_packet(p.x[beg:end], p.y[beg:end], p.t_ns[beg:end], p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
        fe.comm_attach_custom(ar.rank_fn(r), r, 2)
        evs.append(fe)
    one = hip.FrontendEvaluator(p.W, p.H, p.lut)
    one.set_fast_path()
    one.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    _check_sequence(ar, evs, one, [(np.zeros(3), True), (np.array([0.3, -0.5, 0.2]), False), (np.array([0.3, -0.5, 0.2]), True)])
    res = _run_ranks([lambda e=e: e.setupProblemAndOptimize(np.zeros(3)) for e in evs])
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
