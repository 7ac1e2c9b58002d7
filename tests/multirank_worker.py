"""Worker of tests/test_gpu_multirank.py: one process per GPU (torch.distributed.run), native RCCL communicator inside the
evaluator.  Every rank evaluates its batch-range shard of one back-end window (a large panorama, so the tile-set exchange
runs, and a small one) and of one front-end packet; rank 0 also evaluates the whole problem on its own GPU and compares.
Prints MULTIRANK_OK on success; any failure raises (non-zero exit)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from cmax_slam_amd import _lib, evaluator, synth
    from cmax_slam_amd.dist import batch_range
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    def attach(ev):
        idt = torch.zeros(128, dtype=torch.uint8, device=device)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(ev.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
        ev.comm_attach(bytes(idt.cpu().numpy().tobytes()), rank, world)

    def same_everywhere(c, g):
        t = torch.tensor([c] + list(g), dtype=torch.float64, device=device)
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "ranks disagree on contrast / gradient"

    rng = np.random.default_rng(5)
    for Wp, Hp, order, K, nf, T in ((4096, 2048, 2, 5, 0, 0.2), (512, 256, 4, 10, 3, 0.35)):
        w = synth.backend_window(80_000, 240, 180, 200.0, 200.0, 119.5, 89.5, Wp, Hp, order, K, nf, T, seed=51)
        beg, end = batch_range(len(w.x), w.batch, rank, world)
        be = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, device=local)
        be.set_window(w.x[beg:end], w.y[beg:end], w.t_ns[beg:end], w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed,
                      w.t_next_win_beg_ns, w.batch, w.sample_rate, w.sigma, _lib.VARIANCE)
        attach(be)
        one = None
        if rank == 0:
            one = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, device=local)
            one.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                           w.sample_rate, w.sigma, _lib.VARIANCE)
        big = np.tile([0.25, 0.0, 0.0], w.P // 3)
        for i, d in enumerate([np.zeros(w.P), rng.normal(0, 0.01, w.P), big, big, np.zeros(w.P)]):
            d = np.ascontiguousarray(d)
            t = torch.from_numpy(d).to(device)
            dist.broadcast(t, src=0)          # the same parameters on every rank
            d = t.cpu().numpy()
            want = i != 3
            c, g = be.eval(d, want)
            same_everywhere(c, g if want else [])
            if rank == 0:
                c1, g1 = one.eval(d, want)
                assert abs(c - c1) <= 1e-6 * abs(c1), (Wp, i, c, c1)
                if want:
                    assert np.abs(g - g1).max() <= 1e-6 * np.abs(g1).max(), (Wp, i)
        st = be.stats()
        assert st["sharded_host_syncs"] == 0
        if Wp == 4096:
            assert st["exchange_tiles"] > 0 and st["exchange_misses"] >= 1, st
        x, rep = be.setupProblemAndOptimize()
        same_everywhere(rep["final_cost"], x)
        be.close()
    p = synth.frontend_packet(100_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=52)
    beg, end = batch_range(len(p.x), p.batch, rank, world)
    fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut, device=local)
    fe.set_packet(p.x[beg:end], p.y[beg:end], p.t_ns[beg:end], p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    attach(fe)
    c, g = fe.eval(np.array([0.3, -0.5, 0.2]), True)
    same_everywhere(c, g)
    if rank == 0:
        one = evaluator.FrontendEvaluator(p.W, p.H, p.lut, device=local)
        one.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
        c1, g1 = one.eval(np.array([0.3, -0.5, 0.2]), True)
        assert abs(c - c1) <= 1e-6 * abs(c1) and np.abs(g - g1).max() <= 1e-6 * np.abs(g1).max()
    x, rep = fe.setupProblemAndOptimize(np.zeros(3))
    same_everywhere(rep["final_cost"], x)
    fe.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("MULTIRANK_OK world=%d" % world, flush=True)


if __name__ == "__main__":
    main()
