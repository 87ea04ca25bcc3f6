"""GPU parity of the LSSD SE(2) patch tracker (FeatureTrackerLSSDLucasKanade) against the oracle in
its LSSD_TREE mode (half-wavefront butterfly sums, polynomial sin/cos), through the C ABI.
Bar: bit-exact (u16 pyramid, float positions, validity, table order).  The tolerance to the
pattern-order restatement (LSSD_SEQ) is checked beside it: median 1e-3 px, as the oracle test
states.  PARITY UNPINNED against the reference itself (no vectors, basalt absent)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(vsf, w, h, **kw):
    return vsf.default_params(tracker_kind=vsf.TRACKER_LSSD, width=w, height=h, **kw)


@pytest.mark.parametrize("w,h,levels", [(752, 480, 3), (321, 243, 4), (64, 48, 2), (130, 67, 3)])
def test_pyramid_bit_exact(vsf, oracle, w, h, levels):
    rng = np.random.default_rng(w + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ctx = vsf.Context(_params(vsf, w, h, lssd_levels=levels))
    got = ctx.debug_lssd_pyramid(img)
    ref = oracle.lssd_build_pyramid(img, levels)
    for l in range(1, levels + 1):
        assert np.array_equal(got[l - 1], ref[l]), "level %d" % l
    ctx.close()


@pytest.mark.parametrize("w,h,levels,iters", [(752, 480, 3, 5), (320, 240, 3, 5), (321, 243, 4, 8), (160, 120, 0, 5),
                                              (200, 150, 1, 1)])
def test_points_bit_exact(vsf, oracle, synth, w, h, levels, iters):
    frames, _ = synth.sequence(900 + w, w, h, 2)
    kp = oracle.good_features_to_track(frames[0], 301, 0.005, 5.0)
    rng = np.random.default_rng(1)
    extra = np.array([[0.0, 0.0], [w - 1.0, h - 1.0], [-3.0, 10.0], [w + 5.0, 10.0], [5.9, 5.9], [6.0, 6.0],
                      [w - 8.0, h - 8.0], [w / 2 + 0.25, h / 2 + 0.75], [1e6, 1e6], [2.5, h / 2]], np.float32)
    pts = np.concatenate([kp + rng.uniform(-0.5, 0.5, kp.shape).astype(np.float32), extra]).astype(np.float32)
    ctx = vsf.Context(_params(vsf, w, h, lssd_levels=levels, lssd_max_iterations=iters))
    for a, b in ((0, 1), (1, 0), (0, 0)):
        got, ok = ctx.debug_lssd(frames[a], frames[b], pts)
        ref, rok = oracle.lssd_track_points(frames[a], frames[b], pts, levels, iters, oracle.LSSD_TREE)
        assert np.array_equal(ok, rok)
        v = ok.astype(bool)
        assert np.array_equal(got[v], ref[v])
        seq, sok = oracle.lssd_track_points(frames[a], frames[b], pts, levels, iters, oracle.LSSD_SEQ)
        both = v & sok.astype(bool)
        if both.sum() > 20:
            assert np.median(np.abs(got - seq)[both]) < 1e-3
    # n = 1, n = 0
    got, ok = ctx.debug_lssd(frames[0], frames[1], pts[:1])
    ref, rok = oracle.lssd_track_points(frames[0], frames[1], pts[:1], levels, iters, oracle.LSSD_TREE)
    assert np.array_equal(ok, rok) and (not ok[0] or np.array_equal(got, ref))
    got, ok = ctx.debug_lssd(frames[0], frames[1], np.zeros((0, 2), np.float32))
    assert len(ok) == 0
    ctx.close()


@pytest.mark.parametrize("w,h,maxn", [(752, 480, 300), (320, 240, 150), (401, 303, 255)])
def test_sequence_bit_exact(vsf, oracle, synth, w, h, maxn):
    """FeatureDetectorShiTomasiBucketing + FeatureTrackerLSSDLucasKanade over a sequence."""
    frames, _ = synth.sequence(41 + w, w, h, 6)
    P = vsf.FeatureTrackerLSSDLucasKanade.Parameter()
    trk = vsf.FeatureTrackerLSSDLucasKanade(P, w, h, capacity=512)
    det = vsf.FeatureDetectorShiTomasiBucketing(3, 3, maxn, 6.0, w, h, capacity=512)
    db = vsf.FeatureDatabase(512)
    otrk = oracle.LssdTracker(P.optical_flow_max_recovered_dist_, P.optical_flow_levels_,
                              P.optical_flow_max_iterations_, oracle.LSSD_TREE)
    odet = oracle.Detector(3, 3, maxn, 6.0)
    T = oracle.FeatureTable(512)
    for t, fr in enumerate(frames):
        trk.Track(db, fr)
        assert otrk.track(T, fr) == 0
        assert db.n == T.n.value, "step %d" % t
        assert np.array_equal(db.ids[:db.n], T.ids[:db.n])
        assert np.array_equal(db.xy[:db.n], T.xy[:db.n])
        assert np.array_equal(db.age[:db.n], T.age[:db.n])
        if t:
            n_in = trk.last_stats[0]
            assert n_in >= db.n > 0
        det.UpdateDetection(db, fr)
        odet.update_detection(T, fr)
        assert db.n == T.n.value and np.array_equal(db.xy[:db.n], T.xy[:db.n])


def test_batch_matches_per_stream_oracle(vsf, oracle, synth):
    import torch
    w, h, B, T = 320, 240, 11, 4
    frames = np.stack([synth.sequence(500 + b, w, h, T)[0] for b in range(B)], 1)  # [T][B][H][W]
    p = _params(vsf, w, h, grid_rows=2, grid_cols=3, max_features=120, min_distance=7.0, streams=B, capacity=192)
    ctx = vsf.Context(p)
    dev = torch.from_numpy(frames).cuda()
    dets = [oracle.Detector(2, 3, 120, 7.0) for _ in range(B)]
    trks = [oracle.LssdTracker(mode=oracle.LSSD_TREE) for _ in range(B)]
    tabs = [oracle.FeatureTable(192) for _ in range(B)]
    for t in range(T):
        ctx.batch_step(dev[t].data_ptr(), h * w, w, torch.cuda.current_stream().cuda_stream)
        ids, xy, age, n, stats = ctx.batch_download()
        for b in range(B):
            trks[b].track(tabs[b], frames[t, b])
            dets[b].update_detection(tabs[b], frames[t, b])
            e = tabs[b].as_dict()
            got = {int(i): (float(q[0]), float(q[1]), int(a)) for i, q, a in zip(ids[b, :n[b]], xy[b, :n[b]], age[b, :n[b]])}
            assert got == e, "step %d stream %d" % (t, b)
    assert (ctx.batch_status() == 0).all()
    ctx.close()


def test_ldlt_inverse_every_pivot_order_bit_exact(vsf, oracle):
    """Eigen's pivoted 3x3 LDLT + solve(Identity): image data nearly always pivots on the rotation entry first,
    so the other five pivot orders (and ties, zero and negative pivots) are driven here directly."""
    import ctypes as C
    rng = np.random.default_rng(8)
    mats = []
    for scale in ([1, 1, 1], [1, 5, 30], [30, 5, 1], [5, 30, 1], [5, 1, 30], [1, 30, 5], [30, 1, 5]):
        for _ in range(300):
            J = (rng.normal(size=(24, 3)) * np.array(scale)).astype(np.float32)
            mats.append((J.T @ J).astype(np.float32))
    mats += [np.eye(3, dtype=np.float32) * v for v in (1.0, 2.5, 0.0)]                      # ties / singular
    mats += [np.diag(np.array(dd, np.float32)) for dd in ([1, 1, 2], [2, 1, 1], [1, 2, 2], [0, 3, 0], [-4, 1, 2])]
    sym = rng.normal(size=(200, 3, 3)).astype(np.float32)                                   # indefinite symmetric
    mats += list(sym + sym.transpose(0, 2, 1))
    H = np.stack(mats).astype(np.float32)
    ctx = vsf.Context(_params(vsf, 64, 48))
    got = ctx.debug_lssd_inverse3(H)
    ref = np.zeros_like(H)
    fn = oracle.lib().orc_lssd_debug_inverse3
    for i in range(len(H)):
        fn(H[i].ctypes.data_as(C.c_void_p), ref[i].ctypes.data_as(C.c_void_p))
    same = got.view(np.uint32) == ref.view(np.uint32)
    both_nan = np.isnan(got) & np.isnan(ref)
    assert (same | both_nan).all()
    d = np.abs(np.diagonal(H, axis1=1, axis2=2))
    assert len({int(k) for k in d[:2100].argmax(1)}) == 3     # every first pivot occurs
    ctx.close()


def test_caller_rewrites_table_between_calls(vsf, oracle, synth):
    """The tracker's own id -> translation map (transform_): rewritten positions of surviving ids are
    ignored, new ids start from the table, and a reordered table is matched by id."""
    w, h = 320, 240
    frames, _ = synth.sequence(12, w, h, 4)
    kp = oracle.good_features_to_track(frames[0], 90, 0.01, 8.0).astype(np.float64)
    n = len(kp)
    ctx = vsf.Context(_params(vsf, w, h, capacity=128))
    ids, xy, age = np.zeros(128, np.uint64), np.zeros((128, 2)), np.zeros(128, np.uint32)
    ids[:n], xy[:n], age[:n] = np.arange(1, n + 1), kp, 1
    T = oracle.FeatureTable(128)
    T.set(ids[:n].copy(), kp.copy(), age[:n].copy())
    ot = oracle.LssdTracker(mode=oracle.LSSD_TREE)
    rng = np.random.default_rng(0)
    for t, fr in enumerate(frames):
        n, _ = ctx.track(fr, ids, xy, age, n)
        ot.track(T, fr)
        assert n == T.n.value and n > 5
        assert np.array_equal(ids[:n], T.ids[:n]) and np.array_equal(xy[:n], T.xy[:n])
        # reorder the table, rewrite two positions, rename one id, append a new feature
        perm = rng.permutation(n)
        ids[:n], xy[:n], age[:n] = ids[perm], xy[perm], age[perm]
        xy[0] = [40.0 + t, 50.0]
        xy[3] += 2.5
        ids[5] = 1000 + t
        for k in range(20):
            ids[n], xy[n], age[n] = 2000 + 100 * t + k, kp[(7 * t + k) % len(kp)] + 0.5, 1
            n += 1
        T.set(ids[:n].copy(), xy[:n].copy(), age[:n].copy())
    ctx.close()


def test_errors_and_edges(vsf, oracle, synth):
    w, h = 160, 120
    with pytest.raises(vsf.VsfError):
        vsf.Context(_params(vsf, w, h, lssd_levels=8))
    with pytest.raises(vsf.VsfError):
        vsf.Context(_params(vsf, w, h, lssd_levels=-1))
    with pytest.raises(vsf.VsfError):
        vsf.Context(vsf.default_params(tracker_kind=7, width=w, height=h))
    ctx = vsf.Context(_params(vsf, w, h))
    frames, _ = synth.sequence(3, w, h, 2)
    ids, xy, age = np.zeros(ctx.capacity, np.uint64), np.zeros((ctx.capacity, 2)), np.zeros(ctx.capacity, np.uint32)
    with pytest.raises(vsf.VsfError):   # the LSSD tracker has no frame-pair Track
        ctx.track_pair(frames[0], frames[1], ids, xy, age, 0)
    # empty table through both calls; flat frames drop everything that leaves or cannot be solved
    n, _ = ctx.track(frames[0], ids, xy, age, 0)
    assert n == 0
    n, _ = ctx.track(frames[1], ids, xy, age, 0)
    assert n == 0
    flat = np.full((h, w), 90, np.uint8)
    xy[:3] = [[50, 50], [80.5, 60.25], [20, 100]]
    ids[:3] = [1, 2, 3]
    T = oracle.FeatureTable(8)
    T.set(ids[:3].copy(), xy[:3].copy(), age[:3].copy())
    ot = oracle.LssdTracker(mode=oracle.LSSD_TREE)
    ot.track(T, frames[1])
    ot.track(T, flat)
    n, _ = ctx.track(flat, ids, xy, age, 3)
    assert n == T.n.value and np.array_equal(xy[:n], T.xy[:n])
    # a context without the LSSD tracker refuses the LSSD debug calls
    lk = vsf.Context(vsf.default_params(width=w, height=h))
    with pytest.raises(vsf.VsfError):
        lk.debug_lssd(frames[0], frames[1], np.zeros((2, 2), np.float32))
    lk.close()
    ctx.close()


import glob  # noqa: E402
import os  # noqa: E402

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_golden_on_gpu(vsf, path):
    """The committed fixtures (tests/golden/make_golden.py) against the HIP path; no oracle involved."""
    g = np.load(path)
    frames = g["frames"]
    h, w = frames[0].shape
    d = g["det_args"]
    ctx = vsf.Context(_params(vsf, w, h))
    got = ctx.debug_lssd_pyramid(frames[0])
    for l in range(1, 4):
        assert np.array_equal(got[l - 1], g["lssd_pyr%d" % l])
    nxt, ok = ctx.debug_lssd(frames[0], frames[1], g["lssd_pts"])
    assert np.array_equal(ok, g["lssd_valid_tree"])
    v = ok.astype(bool)
    assert np.array_equal(nxt[v], g["lssd_next_tree"][v])
    ctx.close()
    trk = vsf.FeatureTrackerLSSDLucasKanade(vsf.FeatureTrackerLSSDLucasKanade.Parameter(), w, h, capacity=256)
    det = vsf.FeatureDetectorShiTomasiBucketing(int(d[0]), int(d[1]), int(d[2]), float(d[3]), w, h, capacity=256)
    db = vsf.FeatureDatabase(256)
    for k in range(len(frames)):
        trk.Track(db, frames[k])
        assert np.array_equal(db.ids[:db.n], g["lssd_trk%d_ids" % k]) and np.array_equal(db.xy[:db.n], g["lssd_trk%d_xy" % k])
        det.UpdateDetection(db, frames[k])
        assert np.array_equal(db.ids[:db.n], g["lssd_det%d_ids" % k]) and np.array_equal(db.xy[:db.n], g["lssd_det%d_xy" % k])
        assert np.array_equal(db.age[:db.n], g["lssd_det%d_age" % k])


def test_anms_detector_with_lssd_tracker_in_one_context(vsf, oracle, synth):
    """Both "next" components behind one context (what KimeraFrontend instantiates): batch_step on 3 streams."""
    import torch
    w, h, B, T = 320, 240, 3, 4
    frames = np.stack([synth.sequence(300 + b, w, h, T)[0] for b in range(B)], 1)
    p = _params(vsf, w, h, detector_kind=vsf.DETECTOR_ANMS, max_features=100, min_distance=9.0, streams=B)
    ctx = vsf.Context(p)
    dev = torch.from_numpy(frames).cuda()
    dets = [oracle.AnmsDetector(100, 9.0) for _ in range(B)]
    trks = [oracle.LssdTracker(mode=oracle.LSSD_TREE) for _ in range(B)]
    tabs = [oracle.FeatureTable(ctx.capacity) for _ in range(B)]
    for t in range(T):
        ctx.batch_step(dev[t].data_ptr(), h * w, w, torch.cuda.current_stream().cuda_stream)
        ids, xy, age, n, _ = ctx.batch_download()
        for b in range(B):
            trks[b].track(tabs[b], frames[t, b])
            assert dets[b].update_detection(tabs[b], frames[t, b]) in (0, 1)
            got = {int(i): (float(q[0]), float(q[1]), int(a)) for i, q, a in zip(ids[b, :n[b]], xy[b, :n[b]], age[b, :n[b]])}
            assert got == tabs[b].as_dict(), "step %d stream %d" % (t, b)
    assert (ctx.batch_status() == 0).all()
    ctx.close()


def test_tiny_frames_and_deep_pyramids(vsf, oracle, synth):
    """Levels smaller than the pattern: every point is out of bounds there, the level loop ends with an invalid
    patch, the table empties -- and nothing reads outside a level."""
    for w, h, levels in ((40, 24, 3), (18, 18, 1), (64, 48, 5)):
        frames, _ = synth.sequence(5, w, h, 3)
        ctx = vsf.Context(_params(vsf, w, h, lssd_levels=levels))
        ot = oracle.LssdTracker(0.09, levels, 5, oracle.LSSD_TREE)
        ids, xy, age = np.zeros(ctx.capacity, np.uint64), np.zeros((ctx.capacity, 2)), np.zeros(ctx.capacity, np.uint32)
        pts = np.array([[w / 2, h / 2], [8.0, 8.0], [w - 9.0, h - 9.0], [w / 3, h / 2]])
        n = len(pts)
        ids[:n], xy[:n], age[:n] = np.arange(1, n + 1), pts, 2
        T = oracle.FeatureTable(ctx.capacity)
        T.set(ids[:n].copy(), pts.copy(), age[:n].copy())
        for fr in frames:
            n, _ = ctx.track(fr, ids, xy, age, n)
            ot.track(T, fr)
            assert n == T.n.value and np.array_equal(xy[:n], T.xy[:n]) and np.array_equal(ids[:n], T.ids[:n])
        got = ctx.debug_lssd_pyramid(frames[0])
        ref = oracle.lssd_build_pyramid(frames[0], levels)
        for l in range(1, levels + 1):
            assert np.array_equal(got[l - 1], ref[l])
        ctx.close()


@pytest.mark.parametrize("seed", [41, 77, 5])
def test_against_pattern_order_restatement_752x480(vsf, oracle, synth, seed):
    """The HIP path against the oracle mode that restates the REFERENCE's evaluation (LSSD_SEQ: sums in pattern
    order, libm sinf / cosf as Sophus::SE2::exp calls them) on detector-produced features at 752x480 over a
    7-frame sequence, every step started from the same table: identical kept / lost sets after the
    forward-backward test, and a MAXIMUM deviation of the kept positions of 0.01 px (measured: < 5e-4 px)."""
    w, h = 752, 480
    frames, _ = synth.sequence(seed, w, h, 7)
    P = vsf.FeatureTrackerLSSDLucasKanade.Parameter()
    trk = vsf.FeatureTrackerLSSDLucasKanade(P, w, h, capacity=512)
    det = vsf.FeatureDetectorShiTomasiBucketing(4, 4, 300, 10.0, w, h, capacity=512)
    db = vsf.FeatureDatabase(512)
    oseq = oracle.LssdTracker(P.optical_flow_max_recovered_dist_, P.optical_flow_levels_,
                              P.optical_flow_max_iterations_, oracle.LSSD_SEQ)
    worst, total = 0.0, 0
    for t, fr in enumerate(frames):
        S = oracle.FeatureTable(512)
        S.set(db.ids[:db.n], db.xy[:db.n], db.age[:db.n])
        trk.Track(db, fr)
        assert oseq.track(S, fr) == 0
        a, b = db.as_dict(), S.as_dict()
        assert a.keys() == b.keys(), "kept sets differ at step %d: %s" % (t, sorted(a.keys() ^ b.keys())[:8])
        if a:
            worst = max(worst, max(max(abs(a[i][0] - b[i][0]), abs(a[i][1] - b[i][1])) for i in a))
            total += len(a)
        det.UpdateDetection(db, fr)
    assert total > 500
    assert worst <= 0.01, "worst deviation %.6f px over %d tracks" % (worst, total)


def test_points_validity_and_max_error_against_pattern_order(vsf, oracle, synth):
    """Forward TrackPoint on detector corners at 752x480: validity identical for >= 99 % of the points and the
    maximum error on the commonly valid ones <= 0.01 px, against LSSD_SEQ."""
    w, h = 752, 480
    ctx = vsf.Context(_params(vsf, w, h))
    for seed in (1652, 9, 23):
        frames, _ = synth.sequence(seed, w, h, 2)
        pts = oracle.good_features_to_track(frames[0], 600, 0.005, 5.0).astype(np.float32)
        got, ok = ctx.debug_lssd(frames[0], frames[1], pts)
        seq, sok = oracle.lssd_track_points(frames[0], frames[1], pts, 3, 5, oracle.LSSD_SEQ)
        assert (ok == sok).mean() >= 0.99
        both = ok.astype(bool) & sok.astype(bool)
        assert both.sum() > 200 and np.abs(got - seq)[both].max() <= 0.01
    ctx.close()
