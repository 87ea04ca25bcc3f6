"""GPU parity of the batched path WHERE THE BENCH RUNS IT (BASELINE.json configs[1] / configs[4]): vsf_batch_step
on more than 8 streams of 752x480 with the C2 parameters -- lk_fb_kernel's XCD map (stream = grp * 8 + xcd)
reaches grp > 0 only then -- for several frames against per-stream oracles, LK and LSSD; a bounded draw of the
randomised batch sweep (scratch/fuzz_batch.py); and the north-star LK tolerance against the reference
accumulation order at full size over a longer sequence.  All through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C2_DET = (4, 4, 300, 10.0)             # rows, cols, max features, min distance
C2_TRK = (30, 0.01, 21, 3, 1.0)        # iterations, eps, window, max level, back-track threshold


def _tables(ids, xy, age, n, b):
    return {int(i): (float(q[0]), float(q[1]), int(a)) for i, q, a in zip(ids[b, :n[b]], xy[b, :n[b]], age[b, :n[b]])}


def _occlude(frames, rng, frac=0.06):
    """The bench's moving occluders: 32x32 blocks of foreign texture over a fraction of every frame, so that
    tracks are lost and every step detects (frames: [T][B][H][W], modified in place)."""
    T, B, H, W = frames.shape
    for t in range(T):
        for b in range(B):
            occ = rng.random(((H + 31) // 32, (W + 31) // 32)) < frac
            alt = np.roll(frames[(t + 1) % T, (b + 1) % B], (int(rng.integers(H)), int(rng.integers(W))), (0, 1))
            m = np.kron(occ, np.ones((32, 32), bool))[:H, :W]
            frames[t, b][m] = alt[m]
    return frames


@pytest.mark.parametrize("tracker,B", [("lk", 19), ("lssd", 17)])
def test_c5_share_batch_752x480(vsf, oracle, synth, tracker, B):
    import torch
    w, h, T = 752, 480, 5
    rng = np.random.default_rng(5)
    frames = np.stack([synth.sequence(3000 + b, w, h, T)[0] for b in range(B)], 1).copy()   # [T][B][H][W]
    frames = _occlude(frames, rng)
    lssd = tracker == "lssd"
    p = vsf.default_params(grid_rows=C2_DET[0], grid_cols=C2_DET[1], max_features=C2_DET[2], min_distance=C2_DET[3],
                           klt_max_iter=C2_TRK[0], klt_eps=C2_TRK[1], klt_win=C2_TRK[2], klt_max_level=C2_TRK[3],
                           backtrack_thr=C2_TRK[4], width=w, height=h, streams=B,
                           tracker_kind=vsf.TRACKER_LSSD if lssd else vsf.TRACKER_LK)
    ctx = vsf.Context(p)
    dets = [oracle.Detector(*C2_DET) for _ in range(B)]
    trks = [oracle.LssdTracker(0.09, 3, 5, oracle.LSSD_TREE) if lssd else oracle.Tracker(*C2_TRK, oracle.ACC_EXACT)
            for _ in range(B)]
    tabs = [oracle.FeatureTable(ctx.capacity) for _ in range(B)]
    dev = torch.from_numpy(frames).cuda()
    lost = 0
    for t in range(T):
        ctx.batch_step(dev[t].data_ptr(), h * w, w, torch.cuda.current_stream().cuda_stream)
        ids, xy, age, n, stats = ctx.batch_download()
        for b in range(B):
            trks[b].track(tabs[b], frames[t, b])
            dets[b].update_detection(tabs[b], frames[t, b])
            assert _tables(ids, xy, age, n, b) == tabs[b].as_dict(), "step %d stream %d" % (t, b)
        if t:
            lost += int((stats[:, 0] - stats[:, 3]).sum())
    assert (ctx.batch_status() == 0).all()
    assert lost > B * (T - 1) * 3          # the occluders do cost tracks: detection runs in every step
    for b in range(B):
        assert ctx.get_max_feature_index(b) == dets[b].max_feature_index
    ctx.close()


# a bounded draw of scratch/fuzz_batch.py: random sizes, grids, windows, levels, detector x tracker kinds
@pytest.mark.parametrize("seed", range(12))
def test_batch_fuzz_draw(vsf, oracle, synth, seed):
    import torch
    rng = np.random.default_rng(1000 + seed)
    while True:
        w = int(rng.integers(16, 100)) * 4 if rng.random() < 0.6 else int(rng.integers(64, 400))
        h = int(rng.integers(48, 300))
        B, T = int(rng.integers(9, 21)), int(rng.integers(2, 4))
        lssd, anms = rng.random() < 0.5, rng.random() < 0.4
        maxn, mind = int(rng.integers(10, 150)), float(rng.choice([1.0, 2.0, 3.7, 6.0, 10.0]))
        rows, cols = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        win, lvl, llev = int(rng.choice([9, 15, 21, 31])), int(rng.integers(0, 4)), int(rng.integers(0, 4))
        if min(w, h) >> max(lvl, llev) < 16 or (anms and (w <= 48 or h <= 48)):
            continue
        kw = dict(width=w, height=h, streams=B, max_features=maxn, min_distance=mind, klt_win=win, klt_max_level=lvl,
                  tracker_kind=vsf.TRACKER_LSSD if lssd else vsf.TRACKER_LK, lssd_levels=llev)
        if anms:
            kw.update(detector_kind=vsf.DETECTOR_ANMS, anms_detection_features=2 * maxn)
        else:
            kw.update(grid_rows=rows, grid_cols=cols)
        try:
            ctx = vsf.Context(vsf.default_params(**kw))
        except vsf.VsfError as e:
            if e.code == vsf.VSF_E_GEOMETRY:   # a rounded grid cell leaves the image: refused, as documented
                continue
            raise
        break
    frames = np.stack([synth.sequence(int(rng.integers(0, 100000)), w, h, T)[0] for _ in range(B)], 1)
    dev = torch.from_numpy(frames).cuda()
    dets = [oracle.AnmsDetector(maxn, mind, detection_feature_number=2 * maxn) if anms
            else oracle.Detector(rows, cols, maxn, mind) for _ in range(B)]
    trks = [oracle.LssdTracker(0.09, llev, 5, oracle.LSSD_TREE) if lssd
            else oracle.Tracker(30, 0.01, win, lvl, 1.0, oracle.ACC_EXACT) for _ in range(B)]
    tabs = [oracle.FeatureTable(ctx.capacity) for _ in range(B)]
    for t in range(T):
        ctx.batch_step(dev[t].data_ptr(), h * w, w, torch.cuda.current_stream().cuda_stream)
        ids, xy, age, n, _ = ctx.batch_download()
        for b in range(B):
            trks[b].track(tabs[b], frames[t, b])
            dets[b].update_detection(tabs[b], frames[t, b])
            assert _tables(ids, xy, age, n, b) == tabs[b].as_dict(), "step %d stream %d of %r" % (t, b, kw)
    assert (ctx.batch_status() == 0).all()
    ctx.close()


@pytest.mark.parametrize("w,h,levels,grid,maxn,mind,T", [
    (752, 480, 4, (4, 4), 300, 10.0, 8),       # C2, 8 frames
    (1280, 720, 4, (5, 8), 1000, 8.0, 4),      # C3
])
def test_north_star_tolerance_against_reference_order(vsf, oracle, synth, w, h, levels, grid, maxn, mind, T):
    """North star: tracked positions within 0.01 px with identical lost/kept status, against the oracle mode that
    restates OpenCV's scalar float32 accumulation order (ORC_ACC_FLOAT_SEQ), on the features the detector
    produces, over a sequence at the bench's sizes.  No exception rate: every feature of every step."""
    from test_gpu_batch import _big_sequence
    if w == 752:
        frames = synth.sequence(77, w, h, T)[0]
    else:
        frames = np.concatenate([_big_sequence(synth, w, h, 13, 3)[0], _big_sequence(synth, w, h, 13, 3)[0][1::-1]])[:T]
    cap = maxn + 256
    det = vsf.FeatureDetectorShiTomasiBucketing(grid[0], grid[1], maxn, mind, w, h, capacity=cap)
    trk = vsf.FeatureTrackerLucasKanade(30, 0.01, 21, levels - 1, 1.0, w, h, capacity=cap)
    db = vsf.FeatureDatabase(cap)
    otrk = oracle.Tracker(30, 0.01, 21, levels - 1, 1.0, oracle.ACC_FLOAT_SEQ)
    worst, total = 0.0, 0
    for k in range(T):
        # both sides start every step from the SAME table (the GPU's), so the comparison is per step
        tab = oracle.FeatureTable(cap)
        tab.set(db.ids[:db.n], db.xy[:db.n], db.age[:db.n])
        trk.Track(db, frames[k])
        otrk.track(tab, frames[k])
        a, b = db.as_dict(), tab.as_dict()
        assert a.keys() == b.keys(), "kept/lost sets differ at step %d" % k
        if a:
            worst = max(worst, max(max(abs(a[i][0] - b[i][0]), abs(a[i][1] - b[i][1])) for i in a))
            total += len(a)
        det.UpdateDetection(db, frames[k])
    assert total > 0.5 * maxn * (T - 1)
    assert worst <= 0.01, "worst deviation %.5f px over %d tracks" % (worst, total)


def test_template_cache_invalidation(vsf, oracle, synth):
    """The LK template cache (vsf_lk.hpp) must never serve an entry whose position, slot or image is not the one it
    was built for: tables rewritten from the host between steps (moved, dropped and reordered entries), a tracker
    reset, track-only steps, a frame passed twice -- every table bit for bit against per-stream oracles."""
    import torch
    w, h, B, T = 400, 300, 9, 9
    frames = np.stack([synth.sequence(5000 + b, w, h, T)[0] for b in range(B)], 1)
    det_args, trk_args = (2, 3, 150, 7.0), (30, 0.01, 21, 3, 1.0)
    p = vsf.default_params(grid_rows=2, grid_cols=3, max_features=150, min_distance=7.0, klt_win=21, klt_max_level=3,
                           width=w, height=h, streams=B, capacity=256)
    ctx = vsf.Context(p)
    dev = torch.from_numpy(frames).cuda()
    dets = [oracle.Detector(*det_args) for _ in range(B)]
    trks = [oracle.Tracker(*trk_args, oracle.ACC_EXACT) for _ in range(B)]
    tabs = [oracle.FeatureTable(256) for _ in range(B)]
    rng = np.random.default_rng(8)
    st = torch.cuda.current_stream().cuda_stream

    def check(tag):
        ids, xy, age, n, _ = ctx.batch_download()
        for b in range(B):
            assert _tables(ids, xy, age, n, b) == tabs[b].as_dict(), "%s stream %d" % (tag, b)
        return ids, xy, age, n

    order = [0, 1, 2, 2, 3, 4, 5, 6, 7, 8]          # frame 2 twice: identical images, zero flow
    for k, t in enumerate(order):
        detect = k not in (5, 6)                    # two track-only steps
        if k == 7:                                  # forget the previous frame: the next call only stores
            ctx.reset_tracker()
            trks = [oracle.Tracker(*trk_args, oracle.ACC_EXACT) for _ in range(B)]
        if detect:
            ctx.batch_step(dev[t].data_ptr(), h * w, w, st)
        else:
            ctx.batch_track(dev[t].data_ptr(), h * w, w, st)
        for b in range(B):
            trks[b].track(tabs[b], frames[t, b])
            if detect:
                dets[b].update_detection(tabs[b], frames[t, b])
        ids, xy, age, n = check("step %d" % k)
        if k in (3, 8):                             # the host rewrites the tables: move some, drop some, reverse the order
            for b in range(B):
                m = int(n[b])
                keep = np.sort(rng.choice(m, size=max(1, m - 7), replace=False))[::-1]
                nid, nxy, nag = ids[b, keep].copy(), xy[b, keep].copy(), age[b, keep].copy()
                moved = rng.random(len(keep)) < 0.3
                nxy[moved] += rng.uniform(-0.75, 0.75, (int(moved.sum()), 2))
                ids[b, :len(keep)], xy[b, :len(keep)], age[b, :len(keep)] = nid, nxy, nag
                n[b] = len(keep)
                tabs[b].set(nid, nxy, nag)
            ctx.batch_upload(ids, xy, age, n)
    assert (ctx.batch_status() == 0).all()
    ctx.close()


@pytest.mark.parametrize("name,w,h,det,trk,B,D,T", [
    ("C2", 752, 480, C2_DET, C2_TRK, 1024, 8, 4),
    ("C3", 1280, 720, (5, 8, 1000, 8.0), (30, 0.01, 21, 3, 1.0), 256, 4, 3),
    ("C4", 3840, 2160, (10, 16, 4000, 12.0), (30, 0.01, 21, 4, 1.0), 32, 2, 3),
    # 512 streams: the middle workgroup size of the per-stream kernels (1024 threads below 512 streams, 256 from 1024)
    ("512 streams", 320, 240, (3, 4, 120, 8.0), (30, 0.01, 15, 2, 1.0), 512, 4, 4),
])
def test_full_size_batch_by_replication(vsf, oracle, synth, name, w, h, det, trk, B, D, T):
    """BASELINE configs[1..3] at FULL size (the bench's batches: 1024 streams of 752x480, 256 of 1280x720, 32 of
    3840x2160) through a size-independent property: the batch is B / D interleaved copies of D distinct sequences, so
    stream s must produce the table of stream s mod D wherever it sits in the launch grid (XCD map, template cache
    slots, compaction, per-stream counters), and the D distinct streams are held to the oracle bit for bit."""
    import torch
    rng = np.random.default_rng(9)
    base = np.stack([synth.sequence(4100 + b, w, h, T)[0] for b in range(D)], 1).copy()   # [T][D][H][W]
    base = _occlude(base, rng)
    p = vsf.default_params(grid_rows=det[0], grid_cols=det[1], max_features=det[2], min_distance=det[3],
                           klt_max_iter=trk[0], klt_eps=trk[1], klt_win=trk[2], klt_max_level=trk[3],
                           backtrack_thr=trk[4], width=w, height=h, streams=B)
    ctx = vsf.Context(p)
    dev = torch.from_numpy(base).cuda().repeat(1, B // D, 1, 1).contiguous()             # stream s = sequence s % D
    assert dev.shape == (T, B, h, w)
    dets = [oracle.Detector(*det) for _ in range(D)]
    trks = [oracle.Tracker(*trk, oracle.ACC_EXACT) for _ in range(D)]
    tabs = [oracle.FeatureTable(ctx.capacity) for _ in range(D)]
    for t in range(T):
        ctx.batch_step(dev[t].data_ptr(), h * w, w, torch.cuda.current_stream().cuda_stream)
        ids, xy, age, n, stats = ctx.batch_download()
        for b in range(D):
            trks[b].track(tabs[b], base[t, b])
            dets[b].update_detection(tabs[b], base[t, b])
            assert _tables(ids, xy, age, n, b) == tabs[b].as_dict(), "%s step %d stream %d" % (name, t, b)
        # every replica equals its original: whole rows of the table arrays, counts and the logged statistics
        n8 = n.reshape(B // D, D)
        assert (n8 == n8[0]).all(), "step %d: feature counts differ between replicas" % t
        for b in range(D):
            k = int(n[b])
            assert (ids[b::D, :k] == ids[b, :k]).all() and (age[b::D, :k] == age[b, :k]).all()
            assert (xy[b::D, :k] == xy[b, :k]).all(), "step %d: positions differ between replicas of %d" % (t, b)
        assert (stats.reshape(B // D, D, -1) == stats.reshape(B // D, D, -1)[0]).all()
    assert (ctx.batch_status() == 0).all()
    ctx.close()
