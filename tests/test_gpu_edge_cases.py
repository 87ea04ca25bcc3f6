"""Edge cases through the C ABI against the oracle: empty tables, texture-free frames, features on
and beyond the borders, every LK window size the kernels are instantiated for, error returns."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _table(cap):
    return np.zeros(cap, np.uint64), np.zeros((cap, 2), np.float64), np.zeros(cap, np.uint32)


def test_empty_table_and_flat_frames(vsf, oracle):
    w, h = 200, 150
    flat = np.full((h, w), 97, np.uint8)
    for kind in (vsf.DETECTOR_BUCKETING, vsf.DETECTOR_ANMS):
        ctx = vsf.Context(vsf.default_params(width=w, height=h, max_features=40, grid_rows=2, grid_cols=2,
                                             min_distance=5.0, detector_kind=kind))
        ids, xy, age = _table(ctx.capacity)
        n, stats = ctx.track(flat, ids, xy, age, 0)          # first call: stores the frame
        assert n == 0
        n, stats = ctx.track(flat, ids, xy, age, 0)          # empty table: nothing to track
        assert n == 0 and tuple(stats) == (0, 0, 0, 0)
        n = ctx.detect(flat, ids, xy, age, 0)                 # no texture: max eigenvalue 0, no corner
        assert n == 0
        # every feature lost on a flat frame pair (min-eigenvalue test fails at level 0)
        ids[:3] = [5, 6, 7]
        xy[:3] = [[50.5, 60.25], [100.0, 75.0], [150.75, 20.5]]
        age[:3] = 2
        n, stats = ctx.track(flat, ids, xy, age, 3)
        assert n == 0 and stats[0] == 3 and stats[3] == 0
        ctx.close()
    # oracle agrees on the bucketing counter quirk for a call that adds nothing (Q6: wraps to 2^64 - 1)
    ctx = vsf.Context(vsf.default_params(width=w, height=h, max_features=40, grid_rows=2, grid_cols=2))
    ids, xy, age = _table(ctx.capacity)
    assert ctx.detect(flat, ids, xy, age, 0) == 0
    det = oracle.Detector(2, 2, 40, 2.0)
    T = oracle.FeatureTable(ctx.capacity)
    det.update_detection(T, flat)
    assert ctx.get_max_feature_index() == det.max_feature_index == 2 ** 64 - 1


@pytest.mark.parametrize("win", [3, 5, 7, 8, 9, 13, 16, 17, 21, 22, 24, 27, 31])
def test_lk_every_window_size(vsf, oracle, synth, win):
    w, h = 260, 200
    frames, _ = synth.sequence(50 + win, w, h, 2)
    rng = np.random.default_rng(win)
    n = 257
    pts = np.stack([rng.uniform(-win, w + win, n), rng.uniform(-win, h + win, n)], 1).astype(np.float32)
    pts[:16] = [[0, 0], [w - 1, h - 1], [0.5, h - 0.5], [w - 0.5, 0.5], [-win, -win], [w + win - 1, h + win - 1],
                [win / 2, win / 2], [w - win / 2, h - win / 2], [1e-3, 1e-3], [w / 2, -0.25], [-0.25, h / 2],
                [w - 1e-3, h / 2], [w / 2, h - 1e-3], [10, 10], [10.5, 10.5], [w - 10.25, h - 10.75]]
    for ml in (0, 3):
        ctx = vsf.Context(vsf.default_params(width=w, height=h, klt_win=win, klt_max_level=ml))
        got, gst = ctx.debug_lk(frames[0], frames[1], pts)
        ref, rst, _ = oracle.calc_optical_flow_pyr_lk(frames[0], frames[1], pts, win, ml, 30, 0.01,
                                                      acc_mode=oracle.ACC_EXACT)
        assert np.array_equal(gst, rst)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        ctx.close()


def test_features_outside_and_on_the_border(vsf, oracle, synth):
    w, h = 320, 240
    frames, _ = synth.sequence(9, w, h, 2)
    ctx = vsf.Context(vsf.default_params(width=w, height=h, max_features=64, grid_rows=2, grid_cols=2))
    cap = ctx.capacity
    ids, xy, age = _table(cap)
    pre = np.array([[-30.0, 50.0], [-20.9, -20.9], [0.0, 0.0], [319.999, 239.999], [340.5, 100.0], [100.0, 262.0],
                    [160.3, 120.7], [5.0, 5.0], [315.0, 236.0], [1e6, 1e6], [-1e6, 30.0]])
    k = len(pre)
    ids[:k] = np.arange(10, 10 + k)
    xy[:k] = pre
    age[:k] = 3
    trk = oracle.Tracker(30, 0.01, 21, 3, 1.0, acc_mode=oracle.ACC_EXACT)
    T = oracle.FeatureTable(cap)
    T.set(ids[:k].copy(), pre.copy(), age[:k].copy())
    n, stats = ctx.track_pair(frames[0], frames[1], ids, xy, age, k)
    rc, rstats = trk.track_pair(T, frames[0], frames[1])
    assert n == T.n.value and tuple(stats) == tuple(rstats)
    assert np.array_equal(ids[:n], T.ids[:n]) and np.array_equal(xy[:n], T.xy[:n])
    # detection with those (partly outside) features present: occupancy and NN rules (Q4, Q5)
    ids[:k] = np.arange(10, 10 + k)
    xy[:k] = pre
    T.set(ids[:k].copy(), pre.copy(), age[:k].copy())
    det = oracle.Detector(2, 2, 64, 2.0)
    n = ctx.detect(frames[0], ids, xy, age, k)
    assert det.update_detection(T, frames[0]) == 0
    assert n == T.n.value and np.array_equal(ids[:n], T.ids[:n]) and np.array_equal(xy[:n], T.xy[:n])


def test_error_returns(vsf, synth):
    with pytest.raises(vsf.VsfError):
        vsf.Context(vsf.default_params(width=16, height=16))                       # frame not larger than the window
    with pytest.raises(vsf.VsfError):
        vsf.Context(vsf.default_params(klt_win=33))
    with pytest.raises(vsf.VsfError):
        vsf.Context(vsf.default_params(detector_kind=7))
    with pytest.raises(vsf.VsfError):
        vsf.Context(vsf.default_params(detector_kind=vsf.DETECTOR_ANMS, max_features=4100))  # 8200 > 8192 keypoints
    with pytest.raises(vsf.VsfError):
        vsf.Context(vsf.default_params(detector_kind=vsf.DETECTOR_ANMS, min_distance=40.0))  # disc radius > 31
    ctx = vsf.Context(vsf.default_params(width=200, height=150))
    ids, xy, age = _table(ctx.capacity)
    # a tracker that holds a previous frame refuses another size (cv::calcOpticalFlowPyrLK would throw) ...
    ctx.track(np.zeros((150, 200), np.uint8), ids, xy, age, 0)
    with pytest.raises(vsf.VsfError) as e:
        ctx.track(np.zeros((100, 100), np.uint8), ids, xy, age, 0)
    assert e.value.code == -4  # VSF_E_GEOMETRY
    # ... a frame no larger than the LK window cannot be served at all ...
    with pytest.raises(vsf.VsfError) as e:
        ctx.detect(np.zeros((16, 16), np.uint8), ids, xy, age, 0)
    assert e.value.code == -4
    # ... and a context of several streams keeps its size
    bctx = vsf.Context(vsf.default_params(width=200, height=150, streams=2))
    with pytest.raises(vsf.VsfError) as e:
        bctx.detect(np.zeros((100, 100), np.uint8), ids, xy, age, 0)
    assert e.value.code == -4
    bctx.close()
    with pytest.raises(vsf.VsfError) as e:
        ctx.track(np.zeros((150, 200), np.uint8), ids, xy, age, ctx.capacity + 1)
    assert e.value.code == -3  # VSF_E_CAPACITY


def test_table_larger_than_the_lk_grid(vsf, oracle, synth):
    """More table entries than workgroups per stream (max_features + 64): workgroups loop over features."""
    w, h = 320, 240
    frames, _ = synth.sequence(31, w, h, 2)
    ctx = vsf.Context(vsf.default_params(width=w, height=h, max_features=10, capacity=700))
    cap = ctx.capacity
    ids, xy, age = _table(cap)
    rng = np.random.default_rng(1)
    k = 613
    ids[:k] = np.arange(1, k + 1)
    xy[:k] = np.stack([rng.uniform(5, w - 5, k), rng.uniform(5, h - 5, k)], 1)
    age[:k] = 1
    trk = oracle.Tracker(30, 0.01, 21, 3, 1.0, acc_mode=oracle.ACC_EXACT)
    T = oracle.FeatureTable(cap)
    T.set(ids[:k].copy(), xy[:k].copy(), age[:k].copy())
    n, stats = ctx.track_pair(frames[0], frames[1], ids, xy, age, k)
    rc, rstats = trk.track_pair(T, frames[0], frames[1])
    assert n == T.n.value and tuple(stats) == tuple(rstats)
    assert np.array_equal(ids[:n], T.ids[:n]) and np.array_equal(xy[:n], T.xy[:n])


def test_strided_host_frames_equal_packed_ones(vsf, synth):
    """cv::Mat::step > cols (an ROI of a wider image): frames, previous frames and masks with a row stride."""
    w, h = 200, 150
    frames, _ = synth.sequence(17, w, h, 3)
    wide = np.full((3, h, w + 37), 7, np.uint8)
    wide[:, :, 5:5 + w] = frames
    mask = np.full((h, w), 255, np.uint8)
    mask[:, :60] = 0
    wmask = np.zeros((h, w + 11), np.uint8)
    wmask[:, 3:3 + w] = mask

    def run(fr, mk):
        det = vsf.FeatureDetectorShiTomasiBucketing(2, 2, 80, 5.0, w, h, capacity=256)
        trk = vsf.FeatureTrackerLucasKanade(30, 0.01, 21, 3, 1.0, w, h, capacity=256)
        trk2 = vsf.FeatureTrackerLucasKanade(30, 0.01, 21, 3, 1.0, w, h, capacity=256)
        db, db2 = vsf.FeatureDatabase(256), vsf.FeatureDatabase(256)
        out = []
        for k in range(3):
            trk.Track(db, fr(k))
            det.UpdateDetection(db, fr(k), mk)
            out.append(db.as_dict())
        db2.set(db.ids[:db.n].copy(), db.xy[:db.n].copy(), db.age[:db.n].copy())
        trk2.Track(db2, fr(1), fr(2))
        out.append(db2.as_dict())
        return out

    packed = run(lambda k: frames[k], mask)
    strided = run(lambda k: wide[k, :, 5:5 + w], wmask[:, 3:3 + w])
    assert not wide[0, :, 5:5 + w].flags["C_CONTIGUOUS"]
    assert packed == strided and len(packed[-1]) > 10


def test_candidate_overflow_path_is_exact(vsf, oracle, synth):
    """More thresholded local maxima in a cell than candidate slots: select_kernel re-derives the candidates from
    the lambda map.  The slot count scales with the cell size, so the test forces a small one."""
    w, h = 320, 240
    frames, _ = synth.sequence(23, w, h, 2)
    rng = np.random.default_rng(1)
    noisy = np.clip(frames.astype(np.int32) + rng.integers(-25, 26, frames.shape), 0, 255).astype(np.uint8)
    det = vsf.FeatureDetectorShiTomasiBucketing(1, 2, 160, 3.0, w, h, capacity=512)
    det.ctx.debug_set_cand_cap(64)
    db = vsf.FeatureDatabase(512)
    odet, T = oracle.Detector(1, 2, 160, 3.0), oracle.FeatureTable(512)
    for k in range(2):
        det.UpdateDetection(db, noisy[k])
        odet.update_detection(T, noisy[k])
        assert db.as_dict() == T.as_dict() and db.n >= 60
        db.n = T.n.value = db.n // 2   # make room so that the second frame detects again


def test_refused_configurations(vsf):
    """Configurations whose kernels could not launch are refused at context creation, not at the first call."""
    import ctypes as C
    lib = vsf.lib()
    h = C.c_void_p()
    for kw in ({"max_features": 60000, "grid_rows": 1, "grid_cols": 8},    # append_kernel's LDS slots
               {"max_features": 70000, "grid_rows": 3, "grid_cols": 3}):   # per-cell quota above 8192
        p = vsf.default_params(**kw)
        assert lib.vsf_create(C.byref(p), C.byref(h)) == vsf.VSF_E_INVALID, kw
    # a large context and then a small one: the earlier context still launches with its larger dynamic LDS
    # (select_kernel keeps 15104 keys = 120 KB for a 376x240 cell, 2048 keys for the small context's)
    big = vsf.Context(vsf.default_params(width=752, height=480, grid_rows=2, grid_cols=2, max_features=400))
    small = vsf.Context(vsf.default_params(width=64, height=48, grid_rows=1, grid_cols=1, max_features=8, klt_win=9))
    from _pkg import load_synth
    img = load_synth().sequence(3, 752, 480, 1)[0][0]
    ids, xy, age = np.zeros(512, np.uint64), np.zeros((512, 2)), np.zeros(512, np.uint32)
    assert big.detect(img, ids, xy, age, 0) > 200
    small.close()
    big.close()


def test_vsf_step_equals_track_then_detect(vsf, oracle, synth):
    """vsf_step (one upload, one table round trip) against vsf_track + vsf_detect on separate contexts and against
    the oracle, with a mask and a strided frame, over a sequence."""
    w, h = 400, 300
    frames, _ = synth.sequence(61, w, h, 5)
    mask = np.ones((h, w), np.uint8)
    mask[:, :60] = 0
    wide = np.zeros((5, h, w + 16), np.uint8)
    wide[:, :, 3:3 + w] = frames
    kw = dict(grid_rows=3, grid_cols=2, max_features=150, min_distance=6.0, klt_win=15, klt_max_level=2, width=w, height=h,
              capacity=256)
    one = vsf.Context(vsf.default_params(**kw))
    det = vsf.FeatureDetectorShiTomasiBucketing(3, 2, 150, 6.0, w, h, capacity=256)
    trk = vsf.FeatureTrackerLucasKanade(30, 0.01, 15, 2, 1.0, w, h, capacity=256)
    a, b = vsf.FeatureDatabase(256), vsf.FeatureDatabase(256)
    odet, otrk, T = oracle.Detector(3, 2, 150, 6.0), oracle.Tracker(30, 0.01, 15, 2, 1.0, oracle.ACC_EXACT), oracle.FeatureTable(256)
    for k in range(5):
        a.n, stats = one.step(wide[k, :, 3:3 + w], a.ids, a.xy, a.age, a.n, mask)
        trk.Track(b, frames[k])
        det.UpdateDetection(b, frames[k], mask)
        otrk.track(T, frames[k])
        odet.update_detection(T, frames[k], mask)
        assert a.as_dict() == b.as_dict() == T.as_dict(), "frame %d" % k
        if k:
            assert tuple(stats) == tuple(trk.last_stats)
    assert a.n > 50
    one.close()


@pytest.mark.parametrize("w,h,rows,cols", [
    (63, 15, 1, 1), (64, 16, 1, 1), (65, 17, 1, 1), (66, 18, 1, 1),     # one tile, one tile + 1, 2, 3 positions
    (125, 29, 1, 1), (126, 30, 1, 1), (127, 31, 1, 1), (190, 44, 1, 1),  # around two and three tile steps (62 x 14)
    (9, 7, 1, 1), (5, 130, 1, 1), (130, 5, 1, 1), (4, 4, 1, 1),          # cells only a few pixels across
    (26, 22, 8, 8), (40, 9, 3, 13),                                     # cells of 2..4 pixels: no mirrored halo fits
])
def test_detector_tile_geometry(vsf, oracle, w, h, rows, cols):
    """The score kernel's tiling (64 x 16 lambda blocks stepping 62 x 14 from cell position 0, REFLECT_101 products
    mirrored at the cell border, the generic path of cells under 4 pixels): lambda map and detector output of
    cells whose sizes sit on and next to every tile boundary, against the oracle."""
    rng = np.random.default_rng(w * 1000 + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[h // 3: h // 3 + 3, :] = 200          # flat bands: skipped square roots, equal neighbours
    p = vsf.default_params(width=w, height=h, grid_rows=rows, grid_cols=cols, max_features=rows * cols * 6,
                           min_distance=1.5, klt_win=3, klt_max_level=0)
    ctx = vsf.Context(p)
    rects, _ = oracle.generate_grid(w, h, cols, rows)
    for c, rect in enumerate(rects):
        got, grect = ctx.debug_min_eig(img, c)
        assert tuple(int(v) for v in rect) == grect
        ref = oracle.corner_min_eigen_val(img, rect)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "cell %d" % c
    cap = ctx.capacity
    ids, xy, age = _table(cap)
    n = ctx.detect(img, ids, xy, age, 0)
    det = oracle.Detector(rows, cols, rows * cols * 6, 1.5)
    T = oracle.FeatureTable(cap)
    assert det.update_detection(T, img) == 0
    assert n == T.n.value
    assert np.array_equal(ids[:n], T.ids[:n]) and np.array_equal(xy[:n], T.xy[:n]) and np.array_equal(age[:n], T.age[:n])
    ctx.close()


@pytest.mark.parametrize("anms", [False, True])
def test_lambda_map_of_low_contrast_images(vsf, oracle, anms):
    """The smaller eigenvalue needs a correctly rounded square root; the kernel takes v_sqrt_f32 plus two residual
    tests and sends arguments under 2^-96 (where the residuals underflow) to sqrtf.  Images whose gradients are a
    grey level or two -- a - c of the order of one ulp, b = 0 -- put the argument in and around that range."""
    w, h = 192, 64
    rng = np.random.default_rng(17)
    imgs = []
    img = np.full((h, w), 100, np.uint8)
    img[rng.random((h, w)) < 0.03] += 1          # isolated single-level steps
    imgs.append(img)
    img = np.full((h, w), 7, np.uint8)
    img[:, ::2] += 1                              # one-level stripes: b = 0, a != c
    img[20:40, 50:90] += rng.integers(0, 2, (20, 40)).astype(np.uint8)
    imgs.append(img)
    ramp = (np.arange(w)[None, :] // 16 + np.arange(h)[:, None] // 16).astype(np.uint8)
    imgs.append(ramp)                             # one-level steps every 16 pixels, crossing
    bs = 4 if anms else 3
    kw = dict(detector_kind=vsf.DETECTOR_ANMS, anms_ignore_edge=0) if anms else {}
    ctx = vsf.Context(vsf.default_params(width=w, height=h, grid_rows=1, grid_cols=1, max_features=40, min_distance=3.0,
                                         klt_win=3, klt_max_level=0, **kw))
    det = oracle.AnmsDetector(40, 3.0, detection_ignore_edge=0) if anms else oracle.Detector(1, 1, 40, 3.0)
    for img in imgs:
        ref = oracle.corner_min_eigen_val(img, block_size=bs)
        if not anms:
            got, _ = ctx.debug_min_eig(img, 0)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        ids, xy, age = _table(ctx.capacity)
        n = ctx.detect(img, ids, xy, age, 0)
        T = oracle.FeatureTable(ctx.capacity)   # (the id counter runs on across the images, in both)
        det.update_detection(T, img)
        assert n == T.n.value and np.array_equal(xy[:n], T.xy[:n]) and np.array_equal(ids[:n], T.ids[:n])
    ctx.close()


def test_select_with_more_survivors_than_lds_keys(vsf, oracle):
    """select_kernel keeps 1024 candidate keys in LDS; a noisy cell leaves more above the threshold, and the greedy
    selection then runs on the cell's global list (keys under the threshold zeroed in place).  Also with the list
    itself overflowing (the lambda-map path) right after, on the same context."""
    w, h = 400, 300
    rng = np.random.default_rng(23)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    lam = oracle.corner_min_eigen_val(img)
    inner = lam[1:-1, 1:-1]
    assert (inner > 0.1 * lam.max()).sum() > 4000   # plenty above the threshold; a ninth or so are local maxima
    for rows, cols, md in ((1, 1, 2.0), (1, 2, 6.5)):
        ctx = vsf.Context(vsf.default_params(width=w, height=h, grid_rows=rows, grid_cols=cols, max_features=600,
                                             min_distance=md, klt_win=9, klt_max_level=1))
        det, T = oracle.Detector(rows, cols, 600, md), oracle.FeatureTable(ctx.capacity)
        ids, xy, age = _table(ctx.capacity)
        n = ctx.detect(img, ids, xy, age, 0)
        det.update_detection(T, img)
        assert n == T.n.value > 300 and np.array_equal(xy[:n], T.xy[:n]) and np.array_equal(ids[:n], T.ids[:n])
        ctx.close()
