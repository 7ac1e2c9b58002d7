"""CPU: the seeded synthetic generators are deterministic and well-formed."""
import numpy as np

from cmax_slam_amd import synth
from cmax_slam_amd.dist import batch_range


def test_frontend_packet_is_deterministic_sorted_in_range():
    a = synth.frontend_packet(5000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=1)
    b = synth.frontend_packet(5000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=1)
    np.testing.assert_array_equal(a.x, b.x)
    np.testing.assert_array_equal(a.t_ns, b.t_ns)
    assert len(a.x) == 5000 and a.x.max() < 240 and a.y.max() < 180
    assert np.all(np.diff(a.t_ns) >= 0) and a.t_ns[0] >= synth.T0_NS
    assert a.t_ns[0] < a.t_ref_ns < a.t_ns[-1]


def test_backend_window_shapes():
    w = synth.backend_window(4000, 120, 90, 100.0, 100.0, 59.5, 44.5, 256, 128, 4, 10, 3, 0.35, seed=2)
    assert w.K == 10 and w.P == 21 and len(w.x) == 4000
    assert np.all(np.diff(w.t_ns) >= 0)
    assert w.t_ns[-1] < w.start_ns + (w.K - w.order + 1) * w.dt_ns
    np.testing.assert_allclose(np.linalg.norm(w.knots_init, axis=1), 1.0, atol=1e-12)
    np.testing.assert_array_equal(w.knots_init[:3], w.knots_true[:3])  # fixed knots are not perturbed


def test_batch_range_partitions_whole_batches():
    for n, B, world in ((1000, 100, 2), (1001, 100, 3), (99, 100, 8), (40_000_000, 100, 8), (0, 100, 2)):
        spans = [batch_range(n, B, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            assert a1 == b0 and a0 <= a1
        for a0, a1 in spans[:-1]:
            assert (a0 % B == 0 or a0 == n) and (a1 % B == 0 or a1 == n)


def test_bench_cpu_baseline_leg_runs_without_a_gpu():
    """bench.py's cpu_baseline object (the oracle timed on host cores; the all-cores figure nested beside it) is
    pure CPU work: its keys and its bookkeeping are checked here on a small packet."""
    import bench

    p = synth.frontend_packet(20_000, 120, 90, 100.0, 100.0, 59.5, 44.5, seed=3)
    out = bench.cpu_baseline("frontend", p, np.array([0.3, -0.5, 0.2]), 0.4)
    assert out["kind"] == "port" and out["cores"] == 1 and out["unit"] == "events/s" and out["value"] > 0
    assert "full fdf evaluations" in out["sample"]
    ac = out["allcores"]
    assert "error" not in ac, ac
    assert 1 <= ac["cores"] <= ac["usable_cores"] == bench.usable_cores() and ac["value"] > 0
    # the byte models behind the roofline objects: SURVEY 8(d)'s algorithmic bytes, and mandatory bytes <= algorithmic
    for kind, order, P in (("frontend", 0, 3), ("backend", 2, 15), ("backend", 4, 21)):
        m = bench.byte_models(kind, order, 1_000_000, 640 * 480, 10_000, P, True, 50_000)
        assert m["splat"][0] == 1_000_000 * 60 and m["gather"][0] == 1_000_000 * 44
        assert all(mand <= alg for alg, mand in m.values())
    assert bench.byte_models("frontend", 0, 1, 1, 0, 3, False, 0)["splat"][0] == 156
    assert bench.byte_models("backend", 4, 1, 1, 1, 21, False, 0)["splat"][0] == 444
    assert bench.byte_models("backend", 2, 1, 1, 1, 15, False, 0)["splat"][0] == 252
    assert bench.whole_eval_bytes_8d("frontend", 0, 1_000_000, 640 * 480, 3) == 185_491_200   # 185.5 MB / eval
    assert abs(bench.whole_eval_bytes_8d("backend", 4, 5_000_000, 1024 * 1024, 21) - 2.77e9) < 0.01e9


def test_config4_slabs_concatenate_into_one_window_and_match_batch_range():
    from cmax_slam_amd import dist
    world, per = 4, 3000
    slabs = [synth.config4_slab(r, world, per) for r in range(world)]
    w = synth.concat_slabs(slabs)
    assert len(w.x) == world * per and np.all(np.diff(w.t_ns) >= 0)
    for r in range(world):
        assert dist.batch_range(len(w.x), w.batch, r, world) == (r * per, (r + 1) * per)
        assert np.array_equal(slabs[r].knots_init, slabs[0].knots_init)      # one trajectory, one window description
        assert slabs[r].t_next_win_beg_ns == slabs[0].t_next_win_beg_ns
        T = 0.35e9
        assert slabs[r].t_ns.min() >= w.start_ns + int(T * r / world) and slabs[r].t_ns.max() < w.start_ns + T * (r + 1) / world + 1
    # without slabs the generator is unchanged (the committed regression vectors depend on it)
    a, b = synth.config3(2000), synth.config3(2000)
    assert np.array_equal(a.x, b.x) and np.array_equal(a.t_ns, b.t_ns)
