"""-m gpu: CMX_OPT_FOLD_BATCH -- the per-batch pass of the back-end gradient folded into the per-event gather kernel (which
then also finalizes).  Folded and separate forms against each other and against the CPU oracle: both spline orders, batch
sizes at the edge of the fold's condition (runs per wave pass x columns <= 64 lanes), fixed knots (negative columns),
a prior map with border votes (the S2 sums), ragged last batch."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("order,K,nf,batch", [(4, 8, 2, 100), (4, 10, 0, 128), (4, 7, 3, 88), (4, 7, 1, 64), (2, 6, 1, 100), (2, 5, 0, 48),
                                              (2, 4, 2, 32)])
def test_folded_batch_pass(hip, oracle, order, K, nf, batch):
    W, H, Wp, Hp = 160, 120, 512, 256
    f = 1.2 * W
    T = 0.05 * (K - order + 1)
    w = synth.backend_window(30_011, W, H, f, f, (W - 1) / 2, (H - 1) / 2, Wp, Hp, order, K, nf, T, seed=500 + K + batch)
    rng = np.random.default_rng(batch)
    IG = np.zeros((Hp, Wp), np.float32)
    IG[Hp // 2 - 30:Hp // 2 + 30, 100:400] = rng.uniform(0.2, 2.0, (60, 300)).astype(np.float32)
    evs = []
    for fold in (1, 0):
        be = hip.BackendEvaluator(W, H, w.lut, Wp, Hp)
        be.set_fast_path()
        be.set_option(_lib.OPT_FOLD_BATCH, fold)
        be.set_window(w.x, w.y, w.t_ns, order, w.knots_init, w.start_ns, w.dt_ns, nf, w.t_next_win_beg_ns, batch, 1, 1.0,
                      _lib.VARIANCE, IG)
        evs.append(be)
    ref = oracle.Backend(W, H, w.lut, Wp, Hp, order, batch, 1, 1.0, _lib.VARIANCE)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, nf, w.t_next_win_beg_ns, IG)
    P = 3 * (K - nf)
    for x in (np.zeros(P), rng.normal(0, 0.01, P), rng.normal(0, 0.05, P)):
        c_ref, g_ref = ref.eval(x)
        ca, ga = evs[0].eval(x)
        cb, gb = evs[1].eval(x)
        assert rel_scalar(ca, c_ref) < RTOL and rel_vec(ga, g_ref) < RTOL, (order, K, nf, batch)
        assert rel_scalar(ca, cb) < 1e-7 and rel_vec(ga, gb) < 1e-6, (order, K, nf, batch, ga, gb)
        # gradient after a cost-only evaluation at the same point (image reuse) goes through the same kernel
        evs[0].eval(x * 0.5, False)
        _, g2 = evs[0].eval(x * 0.5, True)
        assert rel_vec(g2, ref.eval(x * 0.5)[1]) < RTOL
