"""-m gpu: dataset-level validation.  A synthetic rotating-camera event stream with known ground truth goes through
the whole chain -- front-end solves, angular-velocity integration, control-pose fitting, back-end window solves with
the device-resident global map -- as wired by examples/rotation_pipeline.py (the reference's sliding-window control
logic replayed around the C ABI).  The refined trajectory must beat dead reckoning and stay within a fraction of a
degree of the truth."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))

from cmax_slam_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stream():
    return synth.event_stream(2e6, 0.8, 240, 180, 200.0, 200.0, 119.5, 89.5, omega_mean=(0.2, 1.8, 0.3),
                              omega_amp=(1.0, 0.8, 1.0), seed=77)


@pytest.mark.parametrize("degree", [1, 3])
def test_events_to_trajectory_and_map(hip, stream, degree):
    """The order of the fp32 atomic votes differs from run to run and the FR-CG drivers' loose stopping rules amplify
    that into visibly different estimates (dead reckoning alone varies 0.6-1.2 deg rms between runs on the same
    events; about 3 % of the runs end with a refined error above 0.45 deg): the accuracy claims are made on the median
    of three runs, the structural ones on every run."""
    import rotation_pipeline as rp
    prm = rp.Params()
    prm.spline_degree = degree
    ms = []
    for _ in range(3):
        res = rp.run_pipeline(stream, prm)
        m = rp.evaluate_against_truth(stream, res)
        ms.append(m)
        assert res["windows"] >= 5 and len(res["reports"]) == res["windows"]
        assert m["omega_rmse"] < 0.4 and m["omega_rmse_steady"] < 0.2, m  # front end tracks the angular velocity [rad/s]
        # every window solve lowered its cost and the control-pose layout follows the B-note of SURVEY.md section 8
        assert all(r["final_cost"] <= r["initial_cost"] for r in res["reports"])
        n_first = 5 if degree == 1 else 7
        assert res["traj"].size() == n_first + 2 * (res["windows"] - 1)
        IG = res["IG"]
        assert IG.shape == (prm.pano_height, 2 * prm.pano_height) and np.isfinite(IG).all() and (IG > 0).mean() > 0.02
        assert m["ba_err_deg_rms"] < 1.5 and np.isfinite(m["ba_err_deg_max"]), m
    ba = float(np.median([m["ba_err_deg_rms"] for m in ms]))
    dr = float(np.median([m["dr_err_deg_rms"] for m in ms]))
    assert ba < dr, ms            # bundle adjustment improves on dead reckoning
    assert ba < 0.5, ms           # and stays within half a degree of the truth


def test_event_store_and_host_upload_agree(hip, stream):
    """Packets cut on the device and packets uploaded from the host see the same events; the FR-CG driver's loose
    stopping rules (tolfun 1e-4, local_optim_contrast_gsl.cpp:119-122) turn 1e-8 differences in the fp32 sums into
    different stopping points, so the two runs are compared on accuracy, not packet by packet."""
    import rotation_pipeline as rp
    a = rp.run_pipeline(stream, rp.Params(), use_event_store=True)
    b = rp.run_pipeline(stream, rp.Params(), use_event_store=False)
    np.testing.assert_array_equal(a["ang_vel_t"], b["ang_vel_t"])
    np.testing.assert_allclose(a["ang_vel"][0], b["ang_vel"][0], rtol=0, atol=0.05)  # same start, same first packet
    ma, mb = rp.evaluate_against_truth(stream, a), rp.evaluate_against_truth(stream, b)
    for m in (ma, mb):
        assert m["omega_rmse_steady"] < 0.2 and m["ba_err_deg_rms"] < 1.0, m


def test_deterministic_pipeline_repeats_bit_for_bit(hip, stream):
    """With CMX_OPT_DETERMINISTIC on both contexts the whole chain -- 78 packet solves, the control-pose fits, the window
    solves on the resident map -- returns the same bits on every run, and also when the packets are uploaded from the
    host instead of being cut from the device-resident store (same events, same arithmetic)."""
    import rotation_pipeline as rp
    prm = rp.Params()
    prm.deterministic = True
    a = rp.run_pipeline(stream, prm, use_event_store=True)
    b = rp.run_pipeline(stream, prm, use_event_store=True)
    c = rp.run_pipeline(stream, prm, use_event_store=False)
    for other in (b, c):
        np.testing.assert_array_equal(a["ang_vel"], other["ang_vel"])
        np.testing.assert_array_equal(a["IG"], other["IG"])
        assert [r["final_cost"] for r in a["reports"]] == [r["final_cost"] for r in other["reports"]]
    m = rp.evaluate_against_truth(stream, a)
    assert m["omega_rmse_steady"] < 0.2 and m["ba_err_deg_rms"] < 1.0, m
