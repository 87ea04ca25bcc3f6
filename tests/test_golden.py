"""The oracle against the committed golden vectors (tests/golden/*.npz, made by
tests/golden/make_golden.py).  Parity unpinned by the reference (it holds no vectors for this path);
these fixtures pin the build's own restatement so it cannot drift."""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
CAP = 256


def _u32(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_matches_golden(oracle, path):
    g = np.load(path)
    frames = g["frames"]
    det_args = tuple(g["det_args"])
    det_args = (int(det_args[0]), int(det_args[1]), int(det_args[2]), float(det_args[3]))
    t = g["trk_args"]
    trk_args = (int(t[0]), float(t[1]), int(t[2]), int(t[3]), float(t[4]))
    h, w = frames[0].shape
    win, ml = trk_args[2], trk_args[3]
    pyr = oracle.build_pyramid(frames[0], win, ml)
    for l, lv in enumerate(pyr):
        assert np.array_equal(lv, g["pyr%d" % l])
    assert "pyr%d" % len(pyr) not in g.files
    assert np.array_equal(oracle.scharr(pyr[-1]), g["scharr0"])
    rects, origin = oracle.generate_grid(w, h, det_args[1], det_args[0])
    assert np.array_equal(rects, g["grid_rects"]) and np.array_equal(origin, g["grid_origin"])
    assert np.array_equal(_u32(oracle.corner_min_eigen_val(frames[0], rects[0])), _u32(g["mineig_cell0"]))
    assert np.array_equal(oracle.good_features_to_track(frames[0], 25, 0.1, det_args[3], roi=rects[0]),
                          g["gftt_cell0"])
    for mode, tag in ((oracle.ACC_FLOAT_SEQ, "seq"), (oracle.ACC_EXACT, "exact")):
        nxt, st, err = oracle.calc_optical_flow_pyr_lk(frames[0], frames[1], g["lk_pts"], win, ml, trk_args[0],
                                                       trk_args[1], acc_mode=mode)
        assert np.array_equal(st, g["lk_status_" + tag])
        assert np.array_equal(_u32(nxt), _u32(g["lk_next_" + tag]))
        det, trk, tab = oracle.Detector(*det_args), oracle.Tracker(*trk_args, mode), oracle.FeatureTable(CAP)
        for k in range(len(frames)):
            rc, stats = trk.track(tab, frames[k])
            n = tab.n.value
            assert rc == 0 and np.array_equal(tab.ids[:n], g["%s_trk%d_ids" % (tag, k)])
            assert np.array_equal(tab.xy[:n], g["%s_trk%d_xy" % (tag, k)])
            assert tuple(stats) == tuple(g["%s_trk%d_stats" % (tag, k)])
            assert det.update_detection(tab, frames[k]) == 0
            n = tab.n.value
            assert np.array_equal(tab.ids[:n], g["%s_det%d_ids" % (tag, k)])
            assert np.array_equal(tab.xy[:n], g["%s_det%d_xy" % (tag, k)])
            assert np.array_equal(tab.age[:n], g["%s_det%d_age" % (tag, k)])
            assert det.max_feature_index == int(g["%s_det%d_maxidx" % (tag, k)][0])


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_anms_matches_golden(oracle, path):
    g = np.load(path)
    frames = g["frames"]
    t = g["trk_args"]
    trk_args = (int(t[0]), float(t[1]), int(t[2]), int(t[3]), float(t[4]))
    amax, adist = int(g["anms_args"][0]), float(g["anms_args"][1])
    h, w = frames[0].shape
    det, trk, tab = oracle.AnmsDetector(amax, adist), oracle.Tracker(*trk_args, oracle.ACC_EXACT), oracle.FeatureTable(2 * CAP)
    kp0 = oracle.good_features_to_track(frames[0], 2 * amax, 0.001, 1.0,
                                        mask=det.build_mask(w, h, np.zeros((0, 2))), block_size=4)
    assert np.array_equal(kp0, g["anms_kp0"])
    for k in range(len(frames)):
        rc, _ = trk.track(tab, frames[k])
        n = tab.n.value
        assert rc == 0 and np.array_equal(np.packbits(det.build_mask(w, h, tab.xy[:n])), g["anms_mask%d" % k])
        assert det.update_detection(tab, frames[k]) in (0, 1)
        n = tab.n.value
        assert np.array_equal(tab.ids[:n], g["anms_det%d_ids" % k])
        assert np.array_equal(tab.xy[:n], g["anms_det%d_xy" % k])
        assert np.array_equal(tab.age[:n], g["anms_det%d_age" % k])
        assert det.max_feature_index == int(g["anms_det%d_maxidx" % k][0])


def test_golden_fixtures_exist():
    assert len(GOLD) >= 3


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_lssd_matches_golden(oracle, path):
    g = np.load(path)
    frames = g["frames"]
    d = tuple(g["det_args"])
    det_args = (int(d[0]), int(d[1]), int(d[2]), float(d[3]))
    lp = oracle.lssd_build_pyramid(frames[0], 3)
    for l in range(1, 4):
        assert np.array_equal(lp[l], g["lssd_pyr%d" % l])
    for mode, tag in ((oracle.LSSD_SEQ, "seq"), (oracle.LSSD_TREE, "tree")):
        nxt, ok = oracle.lssd_track_points(frames[0], frames[1], g["lssd_pts"], 3, 5, mode)
        assert np.array_equal(ok, g["lssd_valid_" + tag])
        v = ok.astype(bool)
        assert np.array_equal(_u32(nxt[v]), _u32(g["lssd_next_" + tag][v]))
    det, trk, tab = oracle.Detector(*det_args), oracle.LssdTracker(mode=oracle.LSSD_TREE), oracle.FeatureTable(CAP)
    for k in range(len(frames)):
        assert trk.track(tab, frames[k]) == 0
        n = tab.n.value
        assert np.array_equal(tab.ids[:n], g["lssd_trk%d_ids" % k]) and np.array_equal(tab.xy[:n], g["lssd_trk%d_xy" % k])
        assert det.update_detection(tab, frames[k]) == 0
        n = tab.n.value
        assert np.array_equal(tab.ids[:n], g["lssd_det%d_ids" % k]) and np.array_equal(tab.xy[:n], g["lssd_det%d_xy" % k])
        assert np.array_equal(tab.age[:n], g["lssd_det%d_age" % k])
