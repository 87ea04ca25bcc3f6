"""Parity of the HIP path against the CPU oracle, through the C ABI (include/vsf.h).

Bars (BASELINE.json north_star): integer stages and the detector bit-exact; tracked positions
within 0.01 px with identical kept/lost status.  The LK kernel sums products exactly, so against
the oracle's ORC_ACC_EXACT mode it is also checked bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_PX = 0.01  # north_star tolerance for tracked positions


def _seq(synth, seed, w, h, n):
    frames, mot = synth.sequence(seed, w, h, n)
    return frames


@pytest.mark.parametrize("w,h,win,ml", [(752, 480, 21, 3), (321, 243, 15, 4), (97, 61, 21, 3)])
def test_pyramid_bit_exact(vsf, oracle, synth, w, h, win, ml):
    img = _seq(synth, 3, w, h, 1)[0]
    ctx = vsf.Context(vsf.default_params(width=w, height=h, klt_win=win, klt_max_level=ml))
    got = ctx.debug_pyramid(img)
    ref = oracle.build_pyramid(img, win, ml)
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        assert np.array_equal(a, b)
    ctx.close()


@pytest.mark.parametrize("w,h,win,ml,eps", [(752, 480, 21, 3, 0.01), (640, 360, 15, 3, 0.01),
                                            (480, 320, 24, 4, 0.1), (400, 300, 31, 2, 0.03),
                                            (320, 240, 9, 2, 0.01)])
def test_lk_points(vsf, oracle, synth, w, h, win, ml, eps):
    frames = _seq(synth, 11, w, h, 2)
    rng = np.random.default_rng(5)
    n = 600
    pts = np.stack([rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)], 1).astype(np.float32)
    pts[:50] = np.round(pts[:50])  # integer positions: zero fractional weights
    ctx = vsf.Context(vsf.default_params(width=w, height=h, klt_win=win, klt_max_level=ml, klt_eps=eps))
    got, gst = ctx.debug_lk(frames[0], frames[1], pts)
    # bit-exact against the exact-sum mode
    ref, rst, _ = oracle.calc_optical_flow_pyr_lk(frames[0], frames[1], pts, win, ml, 30, eps,
                                                  acc_mode=oracle.ACC_EXACT)
    assert np.array_equal(gst, rst)
    assert np.array_equal(got[gst == 1].view(np.uint32), ref[rst == 1].view(np.uint32))
    # reference accumulation order (OpenCV scalar path): identical status, and the points beyond the tolerance are
    # EXACTLY the ones recorded in tests/golden/lk_seq_outliers.json (generated on the CPU from the oracle's two
    # modes by tests/golden/make_lk_outliers.py): none, in every case here -- the strict north-star form, no
    # exception rate.  (tests/test_oracle_lk_modes.py bounds the rate on texture-poor random windows.)
    import json
    import os
    fix = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_seq_outliers.json")))
    ref2, rst2, _ = oracle.calc_optical_flow_pyr_lk(frames[0], frames[1], pts, win, ml, 30, eps,
                                                    acc_mode=oracle.ACC_FLOAT_SEQ)
    assert np.array_equal(gst, rst2)
    d = np.abs(got - ref2).max(1)
    beyond = [int(i) for i in np.nonzero((gst == 1) & (d > TOL_PX))[0]]
    assert beyond == fix["%dx%d_win%d_ml%d_eps%g" % (w, h, win, ml, eps)]["outliers"]
    assert np.median(d[gst == 1]) < 1e-4
    assert gst.sum() > 100
    ctx.close()


@pytest.mark.parametrize("w,h,rows,cols", [(752, 480, 4, 4), (322, 242, 3, 5), (200, 100, 1, 1)])
def test_min_eig_bit_exact(vsf, oracle, synth, w, h, rows, cols):
    img = _seq(synth, 21, w, h, 1)[0]
    ctx = vsf.Context(vsf.default_params(width=w, height=h, grid_rows=rows, grid_cols=cols))
    rects, _ = oracle.generate_grid(w, h, cols, rows)
    for c, rect in enumerate(rects):
        got, grect = ctx.debug_min_eig(img, c)
        assert tuple(int(v) for v in rect) == grect
        ref = oracle.corner_min_eigen_val(img, rect)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    ctx.close()


def _run_both(vsf, oracle, frames, det_args, trk_args, cap, steps, mask=None, exact=True):
    h, w = frames[0].shape
    det = vsf.FeatureDetectorShiTomasiBucketing(*det_args, w, h, capacity=cap)
    trk = vsf.FeatureTrackerLucasKanade(*trk_args, w, h, capacity=cap)
    db = vsf.FeatureDatabase(cap)
    odet = oracle.Detector(*det_args)
    otrk = oracle.Tracker(*trk_args, oracle.ACC_EXACT if exact else oracle.ACC_FLOAT_SEQ)
    otab = oracle.FeatureTable(cap)
    for k in range(steps):
        trk.Track(db, frames[k])
        rc, ost = otrk.track(otab, frames[k])
        assert rc == 0
        a, b = db.as_dict(), otab.as_dict()
        assert a.keys() == b.keys(), "step %d: kept sets differ" % k
        if k > 0:
            assert tuple(trk.last_stats) == tuple(ost)
        for i in a:
            if exact:
                assert a[i] == b[i]
            else:
                assert abs(a[i][0] - b[i][0]) <= TOL_PX and abs(a[i][1] - b[i][1]) <= TOL_PX
                assert a[i][2] == b[i][2]
        if not exact:  # keep the two pipelines on the same inputs
            otab.set(db.ids[:db.n], db.xy[:db.n], db.age[:db.n])
        det.UpdateDetection(db, frames[k], mask)
        assert odet.update_detection(otab, frames[k], mask) == 0
        assert db.as_dict() == otab.as_dict(), "step %d: detector output differs" % k
        assert det.max_feature_index_ == odet.max_feature_index
    return db


@pytest.mark.parametrize("w,h,det_args,trk_args", [
    (752, 480, (4, 4, 300, 10.0), (30, 0.01, 21, 3, 1.0)),     # BASELINE C2
    (752, 480, (4, 4, 300, 2.0), (30, 0.01, 15, 3, 1.0)),      # example_feature_detection_and_tracking.cpp:19-23
    (640, 360, (2, 2, 200, 1.0), (30, 0.1, 24, 4, 1.0)),       # example_kimera_backend.cpp:67-80
    (333, 251, (3, 5, 150, 0.5), (20, 0.03, 9, 2, 0.5)),       # ragged cells, minDistance < 1
])
def test_frontend_sequence(vsf, oracle, synth, w, h, det_args, trk_args):
    frames = _seq(synth, 31, w, h, 5)
    db = _run_both(vsf, oracle, frames, det_args, trk_args, 512, 5, exact=True)
    assert db.n > 20
    _run_both(vsf, oracle, frames, det_args, trk_args, 512, 3, exact=False)


def test_detector_mask_and_quirks(vsf, oracle, synth):
    w, h = 400, 300
    frames = _seq(synth, 41, w, h, 2)
    mask = np.zeros((h, w), np.uint8)
    mask[40:260, 30:370] = 255
    mask[100:140, 100:200] = 0
    _run_both(vsf, oracle, frames, (2, 3, 90, 6.0), (30, 0.01, 21, 3, 1.0), 256, 2, mask=mask)
    # empty image: nothing detected, counter decrements and wraps (Q6)
    det = vsf.FeatureDetectorShiTomasiBucketing(2, 2, 40, 5.0, w, h)
    odet = oracle.Detector(2, 2, 40, 5.0)
    db, otab = vsf.FeatureDatabase(128), oracle.FeatureTable(128)
    flat = np.full((h, w), 77, np.uint8)
    det.UpdateDetection(db, flat)
    odet.update_detection(otab, flat)
    assert db.n == 0 and det.max_feature_index_ == odet.max_feature_index == 2 ** 64 - 1
    det.UpdateDetection(db, frames[0])
    odet.update_detection(otab, frames[0])
    assert db.as_dict() == otab.as_dict() and db.n > 0
    assert det.max_feature_index_ == odet.max_feature_index
