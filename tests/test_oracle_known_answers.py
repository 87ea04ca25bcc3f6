"""Known-answer tests that pin the CPU oracle WITHOUT any other implementation of the path:
analytic values, invariances and hand-built cases (SURVEY.md 8(c) "what pins the build then").
They run on the CPU (no GPU, no reference needed)."""
import numpy as np
import pytest


def test_pyrdown_constant_and_ramp(oracle):
    c = np.full((37, 53), 91, np.uint8)
    assert (oracle.pyr_down(c) == 91).all()
    # [1 4 6 4 1]/16 reproduces linear functions: dst(x') = src(2x') away from the border
    ramp = np.tile(np.arange(120, dtype=np.uint8), (40, 1))
    d = oracle.pyr_down(ramp)
    assert d.shape == (20, 60)
    assert (d[:, 1:-1] == (2 * np.arange(60))[None, 1:-1]).all()
    # odd sizes: (w+1)/2, (h+1)/2
    assert oracle.pyr_down(np.zeros((31, 47), np.uint8)).shape == (16, 24)


def test_pyrdown_matches_direct_formula(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (23, 31), dtype=np.uint8)
    k = np.array([1, 4, 6, 4, 1], np.int64)

    def refl(p, n):
        return -p if p < 0 else (2 * n - 2 - p if p >= n else p)
    h, w = img.shape
    exp = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    for y in range(exp.shape[0]):
        for x in range(exp.shape[1]):
            s = 0
            for j in range(5):
                for i in range(5):
                    s += k[i] * k[j] * int(img[refl(2 * y + j - 2, h), refl(2 * x + i - 2, w)])
            exp[y, x] = (s + 128) >> 8
    assert np.array_equal(oracle.pyr_down(img), exp)


def test_pyramid_stop_rule(oracle):
    img = np.zeros((480, 752), np.uint8)
    assert [l.shape for l in oracle.build_pyramid(img, 21, 3)] == [(480, 752), (240, 376), (120, 188), (60, 94)]
    # cv::buildOpticalFlowPyramid stops when the NEXT level would be <= the window
    assert len(oracle.build_pyramid(img, 21, 10)) == 5   # 47x30 built, 24x15 not
    assert len(oracle.build_pyramid(np.zeros((60, 94), np.uint8), 21, 3)) == 2


def test_scharr_ramps(oracle):
    w, h = 40, 30
    rx = np.tile(np.arange(w, dtype=np.uint8) * 3, (h, 1))
    d = oracle.scharr(rx)
    assert (d[:, 1:-1, 0] == 32 * 3).all() and (d[..., 1] == 0).all()
    assert (d[:, 0, 0] == 0).all() and (d[:, -1, 0] == 0).all()  # REFLECT_101: I(-1) == I(1)
    ry = np.tile((np.arange(h, dtype=np.uint8) * 5)[:, None], (1, w))
    d = oracle.scharr(ry)
    assert (d[1:-1, :, 1] == 32 * 5).all() and (d[..., 0] == 0).all()


def _smooth(w, h, dx=0.0, dy=0.0):
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    xs, ys = xs + dx, ys + dy
    v = (128 + 40 * np.sin(xs * 0.21) * np.cos(ys * 0.17) + 30 * np.sin(xs * 0.09 + ys * 0.13)
         + 25 * np.cos(xs * 0.31 - ys * 0.27))
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("mode", [0, 1])
def test_lk_identity_and_translation(oracle, mode):
    w, h = 320, 240
    a = _smooth(w, h)
    ys, xs = np.mgrid[40:200:20, 40:280:24]
    pts = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float32) + 0.37
    nxt, st, _ = oracle.calc_optical_flow_pyr_lk(a, a, pts, 21, 3, 30, 0.01, acc_mode=mode)
    assert (st == 1).all()
    assert np.abs(nxt - pts).max() < 1e-3          # identical images: zero flow
    for (dx, dy) in [(1.3, -0.7), (-2.25, 1.5), (0.5, 0.5)]:
        b = _smooth(w, h, dx, dy)                   # b(x) = a(x + d)  =>  flow = -d
        nxt, st, _ = oracle.calc_optical_flow_pyr_lk(a, b, pts, 21, 3, 30, 0.01, acc_mode=mode)
        assert (st == 1).all()
        assert np.abs(nxt - (pts - np.array([dx, dy], np.float32))).max() < 0.05


def test_lk_lost_points(oracle):
    a = _smooth(200, 150)
    flat = np.full_like(a, 100)
    pts = np.array([[50.0, 50.0], [-30.0, 20.0], [100.0, 400.0]], np.float32)
    _, st, _ = oracle.calc_optical_flow_pyr_lk(flat, flat, pts[:1], 21, 3, 30, 0.01)
    assert st[0] == 0                               # no texture: min eigenvalue below 1e-4
    _, st, _ = oracle.calc_optical_flow_pyr_lk(a, a, pts, 21, 3, 30, 0.01)
    assert list(st) == [1, 0, 0]                    # far outside the image: window rule at level 0


def test_min_eig_invariants_and_numpy_crosscheck(oracle):
    assert (oracle.corner_min_eigen_val(np.full((20, 30), 9, np.uint8)) == 0).all()
    # gradient along x only: rank-1 tensor, lambda_min == 0 exactly
    edge = np.tile(np.r_[np.zeros(10), np.full(10, 200)].astype(np.uint8), (16, 1))
    assert (oracle.corner_min_eigen_val(edge) == 0).all()
    # float64 numpy restatement of the same formulas on a random image (no ROI)
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (24, 28)).astype(np.uint8)
    p = np.pad(img.astype(np.float64), 1, mode="reflect")
    sc = 1.0 / (4 * 3 * 255.0)
    dx = ((p[:-2, 2:] - p[:-2, :-2]) + 2 * (p[1:-1, 2:] - p[1:-1, :-2]) + (p[2:, 2:] - p[2:, :-2])) * sc
    dy = ((p[2:, :-2] - p[:-2, :-2]) + 2 * (p[2:, 1:-1] - p[:-2, 1:-1]) + (p[2:, 2:] - p[:-2, 2:])) * sc

    def box(a):
        q = np.pad(a, 1, mode="reflect")
        return sum(q[j:j + a.shape[0], i:i + a.shape[1]] for j in range(3) for i in range(3))
    a, b, c = box(dx * dx) * 0.5, box(dx * dy), box(dy * dy) * 0.5
    lam = (a + c) - np.sqrt((a - c) ** 2 + b * b)
    got = oracle.corner_min_eigen_val(img)
    assert np.allclose(got, lam, rtol=2e-3, atol=1e-7)


def test_harris_invariants_and_numpy_crosscheck(oracle):
    """cv::cornerHarris (FeatureDetectorANMS's detection_use_harris_corner_): 0 on a flat image, negative on a
    straight edge (det 0, trace > 0), and a float64 restatement of a c - b^2 - k (a + c)^2 elsewhere."""
    assert (oracle.corner_harris(np.full((20, 30), 9, np.uint8)) == 0).all()
    edge = np.tile(np.r_[np.zeros(10), np.full(10, 200)].astype(np.uint8), (16, 1))
    r = oracle.corner_harris(edge, k=0.04)
    assert (r <= 0).all() and (r[:, 8:12] < 0).all() and (r[:, :7] == 0).all()
    rng = np.random.default_rng(7)
    for shape, bs in (((24, 28), 3), ((25, 31), 4)):   # 31: three pixels a row take the scalar (double k) form
        img = rng.integers(0, 256, shape).astype(np.uint8)
        p = np.pad(img.astype(np.float64), 1, mode="reflect")
        sc = 1.0 / (4 * bs * 255.0)
        dx = ((p[:-2, 2:] - p[:-2, :-2]) + 2 * (p[1:-1, 2:] - p[1:-1, :-2]) + (p[2:, 2:] - p[2:, :-2])) * sc
        dy = ((p[2:, :-2] - p[:-2, :-2]) + 2 * (p[2:, 1:-1] - p[:-2, 1:-1]) + (p[2:, 2:] - p[:-2, 2:])) * sc
        hl = bs // 2   # cv's anchor: offsets -hl .. bs - 1 - hl

        def box(a):
            q = np.pad(a, ((hl, bs - 1 - hl), (hl, bs - 1 - hl)), mode="reflect")
            return sum(q[j:j + a.shape[0], i:i + a.shape[1]] for j in range(bs) for i in range(bs))
        a, b, c = box(dx * dx), box(dx * dy), box(dy * dy)
        for k in (0.04, 0.15):
            want = a * c - b * b - k * (a + c) ** 2
            got = oracle.corner_harris(img, block_size=bs, k=k)
            assert np.allclose(got, want, rtol=5e-3, atol=1e-9 + 2e-6 * np.abs(want).max())
    # the ANMS detector's struct switches goodFeaturesToTrack to the Harris map
    img = rng.integers(0, 256, (60, 80)).astype(np.uint8)
    k0 = oracle.good_features_to_track(img, 50, 0.01, 3.0, block_size=4)
    k1 = oracle.good_features_to_track(img, 50, 0.01, 3.0, block_size=4, use_harris=True, k=0.04)
    assert len(k1) > 10 and not np.array_equal(k0, k1)
    resp = oracle.corner_harris(img, block_size=4, k=0.04)
    x, y = k1[0].astype(int)
    assert resp[y, x] == resp[1:-1, 1:-1].max()


def test_gftt_min_distance_is_greedy_in_response_order(oracle):
    """cv's minDistance rule for values the bucketing detector never uses (FeatureDetectorANMS.cpp:24): corners
    come out in descending response, no two closer than minDistance, and none that was dropped could be added."""
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (90, 120)).astype(np.uint8)
    lam = oracle.corner_min_eigen_val(img, block_size=4)
    allk = oracle.good_features_to_track(img, 100000, 0.01, 1.0, block_size=4)
    for md in (1.0, 1.2, 2.0, 4.5, 9.0):
        kp = oracle.good_features_to_track(img, 100000, 0.01, md, block_size=4).astype(int)
        v = lam[kp[:, 1], kp[:, 0]]
        assert (np.diff(v) <= 0).all()
        d2 = ((kp[:, None, :] - kp[None, :, :]) ** 2).sum(-1) + np.eye(len(kp), dtype=int) * 10 ** 6
        assert (d2 >= md * md).all()
        if md == 1.0:
            assert len(kp) == len(allk)
        else:
            kept = {tuple(p) for p in kp}
            for p in allk.astype(int):
                if tuple(p) not in kept:   # dropped: a kept corner with at least its response lies within md
                    near = ((kp - p) ** 2).sum(-1) < md * md
                    assert (lam[kp[near, 1], kp[near, 0]] >= lam[p[1], p[0]]).any()


def test_min_eig_roi_reads_parent_pixels(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 60)).astype(np.uint8)
    full = oracle.corner_min_eigen_val(img)
    roi = oracle.corner_min_eigen_val(img, (10, 8, 30, 20))
    # away from the ROI border the box filter sees the same covariances
    assert np.array_equal(roi[1:-1, 1:-1], full[9:27, 11:39])
    # on the ROI border the box filter reflects inside the ROI, so values differ from the full map
    assert not np.array_equal(roi[0], full[8, 10:40])


def test_gftt_isolated_squares(oracle):
    img = np.full((80, 100), 20, np.uint8)
    for (x, y) in [(20, 20), (60, 20), (20, 55), (70, 60)]:
        img[y:y + 8, x:x + 8] = 220
    pts = oracle.good_features_to_track(img, 100, 0.1, 3.0)
    assert len(pts) >= 8
    d = np.linalg.norm(pts[:, None] - pts[None], axis=2) + np.eye(len(pts)) * 1e9
    assert d.min() >= 3.0                           # greedy minimum distance
    # every corner sits at a square corner (within the 3x3 block)
    corners = np.array([(x + i, y + j) for (x, y) in [(20, 20), (60, 20), (20, 55), (70, 60)]
                        for i in (0, 7) for j in (0, 7)], np.float32)
    assert (np.abs(pts[:, None] - corners[None]).max(2).min(1) <= 2).all()
    assert len(oracle.good_features_to_track(img, 5, 0.1, 3.0)) == 5
    assert np.array_equal(oracle.good_features_to_track(img, 5, 0.1, 3.0), pts[:5])  # strongest first
    mask = np.zeros_like(img)
    mask[:, :50] = 1
    pm = oracle.good_features_to_track(img, 100, 0.1, 3.0, mask=mask)
    assert len(pm) > 0 and (pm[:, 0] < 50).all()
    assert len(oracle.good_features_to_track(img, 100, 0.1, 3.0, mask=np.zeros_like(img))) == 0


def test_generate_grid(oracle):
    rects, origin = oracle.generate_grid(752, 480, 4, 4)
    assert [tuple(r) for r in rects[:5]] == [(0, 0, 188, 120), (0, 120, 188, 120), (0, 240, 188, 120),
                                            (0, 360, 188, 120), (188, 0, 188, 120)]  # column-major (Q1)
    rects, origin = oracle.generate_grid(100, 50, 3, 1)
    assert [tuple(r) for r in rects] == [(0, 0, 33, 50), (33, 0, 33, 50), (67, 0, 33, 50)]  # cvRound per field
    assert origin[1][0] == np.float32(100.0 / 3.0)  # the unrounded Rect2f origin is kept


def _textured(synth, w=320, h=240, seed=5):
    return synth.sequence(seed, w, h, 2)[0]


def test_detector_ids_ages_and_counter(oracle, synth):
    f = _textured(synth)
    det = oracle.Detector(2, 2, 40, 6.0)
    tab = oracle.FeatureTable(128)
    assert det.update_detection(tab, f[0]) == 0
    n1 = tab.n.value
    d = tab.as_dict()
    assert n1 > 10 and sorted(d) == list(range(n1))          # Q6: first ids are 0..n-1
    assert all(v[2] == 1 for v in d.values())                # Q7: age 1
    assert det.max_feature_index == n1 - 1
    # second call: first cell is full -> early break (Q2), nothing added, counter decrements (Q6)
    assert det.update_detection(tab, f[0]) == 0
    assert tab.n.value == n1 and det.max_feature_index == n1 - 2
    # empty first cell again: ids restart at the stored counter and overwrite an existing id
    keep = [i for i, v in d.items() if not (v[0] < 160 and v[1] < 120)]
    tab.set(np.array(keep, np.uint64), np.array([d[i][:2] for i in keep]), np.full(len(keep), 3, np.uint32))
    assert det.update_detection(tab, f[0]) == 0
    d2 = tab.as_dict()
    start = n1 - 2
    new_ids = [i for i in d2 if d2[i][2] == 1]
    assert min(new_ids) == start
    assert new_ids == list(range(start, start + len(new_ids))) or set(new_ids) == set(range(start, start + len(new_ids)))


def test_detector_early_break_is_break_not_continue(oracle, synth):
    f = _textured(synth)
    det = oracle.Detector(2, 2, 40, 6.0)   # cap 10 per cell; cells column-major: (0,0),(0,1),(1,0),(1,1)
    tab = oracle.FeatureTable(128)
    # 10 distinct pixels in cell index 1 (x<160, y>=120): loop must stop there, cells 2 and 3 stay empty
    xy = np.array([[10.5 + 3 * i, 130.5] for i in range(10)])
    tab.set(np.arange(1000, 1010, dtype=np.uint64), xy, np.full(10, 5, np.uint32))
    assert det.update_detection(tab, f[0]) == 0
    new = [v for k, v in tab.as_dict().items() if k < 1000]
    assert len(new) > 0
    assert all(v[0] < 160 and v[1] < 120 for v in new)
    # two features in ONE pixel count once (countNonZero of a flag image, Q4)
    det2 = oracle.Detector(1, 1, 2, 6.0)
    tab2 = oracle.FeatureTable(16)
    tab2.set(np.array([7, 8], np.uint64), np.array([[50.2, 60.2], [50.7, 60.9]]), np.array([1, 1], np.uint32))
    assert det2.update_detection(tab2, f[0]) == 0
    assert tab2.n.value == 3                 # occupancy 1 of 2 -> one more detected


def test_detector_nn_rejection_is_strict(oracle, synth):
    f = _textured(synth)
    det = oracle.Detector(1, 1, 1, 4.0)
    tab = oracle.FeatureTable(16)
    det.update_detection(tab, f[0])
    (cx, cy, _), = tab.as_dict().values()     # strongest corner of the frame
    for off, expect_added in ((4.0, False), (4.0 + 1e-9, True)):
        d = oracle.Detector(1, 1, 2, 4.0)
        t = oracle.FeatureTable(16)
        # an existing feature OUTSIDE the image still counts for the NN test (Q5) but not for occupancy
        t.set(np.array([99], np.uint64), np.array([[cx + off, cy]]), np.array([2], np.uint32))
        d.update_detection(t, f[0])
        got = [(v[0], v[1]) for k, v in t.as_dict().items() if k != 99]
        assert ((cx, cy) in got) == expect_added


def test_tracker_wrapper_semantics(oracle, synth):
    f = _textured(synth)
    det = oracle.Detector(2, 2, 60, 8.0)
    tab = oracle.FeatureTable(128)
    det.update_detection(tab, f[0])
    before = tab.as_dict()
    trk = oracle.Tracker(30, 0.01, 21, 3, 1.0)
    rc, st = trk.track(tab, f[0])
    assert rc == 0 and tab.as_dict() == before            # first call only stores the frame
    rc, st = trk.track(tab, f[0])                         # identical frame: everything verified
    after = tab.as_dict()
    assert st == (len(before), len(before), len(before), len(before))
    assert after.keys() == before.keys()
    assert max(abs(after[i][0] - before[i][0]) + abs(after[i][1] - before[i][1]) for i in after) < 1e-3
    assert all(after[i][2] == before[i][2] for i in after)  # ages untouched (Q7)
    empty = oracle.FeatureTable(8)
    rc, st = trk.track(empty, f[1])
    assert rc == 0 and empty.n.value == 0
    # stateless 4-argument form gives the same result as the stateful one
    t1, t2 = oracle.FeatureTable(128), oracle.FeatureTable(128)
    ids = np.array(sorted(before), np.uint64)
    xy = np.array([before[int(i)][:2] for i in ids])
    for t in (t1, t2):
        t.set(ids, xy, np.ones(len(ids), np.uint32))
    a = oracle.Tracker(30, 0.01, 21, 3, 1.0)
    a.track(t1, f[0])
    a.track(t1, f[1])
    oracle.Tracker(30, 0.01, 21, 3, 1.0).track_pair(t2, f[0], f[1])
    assert t1.as_dict() == t2.as_dict()


def test_bgr2gray_known_answers(oracle):
    """cv::cvtColor(CV_BGR2GRAY), 8-bit: the grey levels OpenCV documents for the primaries (B 29, G 150, R 76),
    grey stays grey (the three weights sum to 2^14), and an independent numpy statement of the fixed-point rule."""
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [17, 17, 17], [1, 2, 3]]], np.uint8)
    g = oracle.bgr2gray(px)[0]
    assert list(g[:6]) == [29, 150, 76, 255, 0, 17]
    rng = np.random.default_rng(3)
    for cn in (3, 4):
        img = rng.integers(0, 256, (37, 53, cn), dtype=np.uint8)
        ref = ((img[..., 0].astype(np.int64) * 1868 + img[..., 1].astype(np.int64) * 9617 +
                img[..., 2].astype(np.int64) * 4899 + 8192) >> 14).astype(np.uint8)
        assert np.array_equal(oracle.bgr2gray(img), ref)
