"""-m gpu: contexts own all their device memory (the reference keeps scratch in function-local statics): repeated
create / hand-over / evaluate / destroy cycles must not grow the device's memory footprint, whatever paths were used."""
import numpy as np
import pytest

from cmax_slam_amd import synth

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_create_destroy_cycles_do_not_leak(hip):
    p = synth.frontend_packet(50_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=85)
    w = synth.backend_window(50_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 2048, 1024, 4, 10, 3, 0.35, seed=86)
    IG = np.ones((w.Hp, w.Wp), np.float32)

    def cycle(k):
        store = hip.EventStore(p.W, p.H, capacity=len(p.x))
        store.push(p.x, p.y, p.t_ns)
        fe = hip.reference_shaped.FrontendEvaluator(p.W, p.H, p.lut)
        if k % 2:
            fe.set_fast_path()
        fe.set_packet_from(store, 0, len(p.x), p.t_ref_ns, p.fx, p.fy, p.cx, p.cy)
        fe.eval((0.1, 0.2, 0.3))
        fe.setupProblemAndOptimize(np.zeros(3))
        fe.computeImageOfWarpedEvents((0.1, 0.2, 0.3), want_deriv=True)
        be = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
        if k % 2 == 0:
            be.set_fast_path()
        be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns,
                      100, 1, 1.0, 0, IG if k % 3 == 0 else None)
        be.eval(np.zeros(w.P))
        be.eval(np.zeros(w.P), False)
        be.updateIG(10)
        if k % 4 == 0:
            be.comm_attach(be.comm_unique_id(), 0, 1)
            be.eval(np.full(w.P, 1e-3))
        for obj in (fe, be, store):
            obj.close()

    for k in range(4):          # warm the allocator / code objects / RCCL
        cycle(k)
    before = _free_bytes()
    for k in range(24):
        cycle(k)
    after = _free_bytes()
    assert before - after < 64 << 20, (before, after)  # allocator granularity, not a per-cycle leak (a cycle holds ~300 MB)
