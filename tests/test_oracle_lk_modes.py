"""The HIP LK kernel sums products exactly (integers), i.e. it equals the oracle's ORC_ACC_EXACT
mode bit for bit.  This CPU test measures how far that mode is from the reference accumulation
order (OpenCV's scalar float32 raster order, ORC_ACC_FLOAT_SEQ): the north-star tolerance
(<= 0.01 px, identical found/lost status) must hold between the two on every point tried."""
import numpy as np
import pytest

TOL_PX = 0.01


@pytest.mark.parametrize("seed,w,h,win,ml,eps", [(1, 752, 480, 21, 3, 0.01), (2, 640, 360, 15, 3, 0.01),
                                                 (3, 480, 320, 24, 4, 0.1), (4, 400, 300, 9, 2, 0.03)])
def test_exact_vs_reference_order(oracle, synth, seed, w, h, win, ml, eps):
    frames, _ = synth.sequence(seed, w, h, 3)
    rng = np.random.default_rng(seed)
    n = 1500
    pts = np.stack([rng.uniform(-4, w + 4, n), rng.uniform(-4, h + 4, n)], 1).astype(np.float32)
    tot = over = 0
    for (a, b) in ((0, 1), (1, 2), (2, 0)):
        r0, s0, _ = oracle.calc_optical_flow_pyr_lk(frames[a], frames[b], pts, win, ml, 30, eps, acc_mode=0)
        r1, s1, _ = oracle.calc_optical_flow_pyr_lk(frames[a], frames[b], pts, win, ml, 30, eps, acc_mode=1)
        assert np.array_equal(s0, s1)
        ok = s0 == 1
        assert ok.sum() > n // 3
        d = np.abs(r0[ok] - r1[ok]).max(1)
        tot += int(ok.sum())
        over += int((d > TOL_PX).sum())
        assert np.median(d) < 1e-4
    assert over <= 0.005 * tot, (over, tot)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_detected_corners_within_tolerance(oracle, synth, seed):
    w, h = 752, 480
    frames, _ = synth.sequence(seed, w, h, 3)
    tab, det = oracle.FeatureTable(400), oracle.Detector(4, 4, 300, 10.0)
    det.update_detection(tab, frames[0])
    pts = tab.xy[:tab.n.value].astype(np.float32)
    assert len(pts) > 100
    for b in (1, 2):
        r0, s0, _ = oracle.calc_optical_flow_pyr_lk(frames[0], frames[b], pts, 21, 3, 30, 0.01, acc_mode=0)
        r1, s1, _ = oracle.calc_optical_flow_pyr_lk(frames[0], frames[b], pts, 21, 3, 30, 0.01, acc_mode=1)
        assert np.array_equal(s0, s1)
        ok = s0 == 1
        assert np.abs(r0[ok] - r1[ok]).max() <= TOL_PX


def test_frontend_sequence_modes(oracle, synth):
    frames, _ = synth.sequence(9, 480, 320, 5)
    tabs = []
    for mode in (0, 1):
        det, trk, tab = oracle.Detector(3, 3, 180, 6.0), oracle.Tracker(30, 0.01, 21, 3, 1.0, mode), oracle.FeatureTable(256)
        for f in frames:
            trk.track(tab, f)
            det.update_detection(tab, f)
        tabs.append(tab.as_dict())
    a, b = tabs
    assert a.keys() == b.keys() and len(a) > 50
    assert max(max(abs(a[i][0] - b[i][0]), abs(a[i][1] - b[i][1])) for i in a) <= TOL_PX
