"""GPU parity of the ANMS detector (FeatureDetectorANMS) against the oracle, through the C ABI.
Bit-exact bar: mask, detected keypoint list (cv order), selected set, ids, ages, counter."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(vsf, w, h, maxn, mind, **kw):
    return vsf.default_params(detector_kind=vsf.DETECTOR_ANMS, max_features=maxn, min_distance=mind,
                              width=w, height=h, **kw)


def _table(cap):
    return np.zeros(cap, np.uint64), np.zeros((cap, 2), np.float64), np.zeros(cap, np.uint32)


@pytest.mark.parametrize("w,h,maxn,mind", [(752, 480, 300, 10.0), (320, 240, 100, 10.0), (401, 303, 150, 7.5),
                                           (640, 480, 500, 4.0),
                                           (1280, 720, 2000, 6.0),   # 4000 detected keypoints
                                           (3840, 2160, 4000, 12.0)])  # BASELINE C4: 8000 keypoints (limit 8192)
def test_anms_sequence_bit_exact(vsf, oracle, synth, w, h, maxn, mind):
    frames, _ = synth.sequence(21, w, h, 4 if w < 3000 else 3)
    ctx = vsf.Context(_params(vsf, w, h, maxn, mind))
    cap = ctx.capacity
    ids, xy, age = _table(cap)
    n = 0
    det = oracle.AnmsDetector(maxn, mind)
    T = oracle.FeatureTable(cap)
    rng = np.random.default_rng(3)
    for t, fr in enumerate(frames):
        n = ctx.detect(fr, ids, xy, age, n)
        rc = det.update_detection(T, fr)
        assert rc in (0, 1)
        # intermediates
        flag, kp = ctx.debug_anms()
        ref_flag = det.build_mask(w, h, T_prev_xy) if t > 0 else det.build_mask(w, h, np.zeros((0, 2)))
        assert np.array_equal(flag, ref_flag)
        ref_kp = oracle.good_features_to_track(fr, 2 * maxn, 0.001, 1.0, mask=ref_flag, block_size=4)
        assert np.array_equal(kp.astype(np.float32), ref_kp)
        # table
        assert n == T.n.value
        assert np.array_equal(ids[:n], T.ids[:n])
        assert np.array_equal(xy[:n], T.xy[:n])
        assert np.array_equal(age[:n], T.age[:n])
        assert ctx.get_max_feature_index() == det.max_feature_index
        # thin the table (as a tracker would), jitter the survivors to sub-pixel positions
        keep = np.sort(rng.choice(n, size=int(n * 0.55), replace=False))
        k = len(keep)
        for arr, oarr in ((ids, T.ids), (xy, T.xy), (age, T.age)):
            arr[:k] = arr[keep]
            oarr[:k] = arr[:k]
        xy[:k] += rng.uniform(-1.5, 1.5, (k, 2))
        T.xy[:k] = xy[:k]
        n = k
        T.n.value = k
        T_prev_xy = xy[:k].copy()


def test_anms_user_mask_borders_and_full_table(vsf, oracle, synth):
    w, h, maxn = 320, 240, 60
    frames, _ = synth.sequence(5, w, h, 1)
    fr = frames[0]
    mask = np.full((h, w), 255, np.uint8)
    mask[:, :100] = 2      # even value: wiped by the bitwise AND with the edge mask of ones
    mask[100:140, :] = 0
    mask[200:, 150:] = 1
    ctx = vsf.Context(_params(vsf, w, h, maxn, 12.0))
    cap = ctx.capacity
    ids, xy, age = _table(cap)
    # features at the borders / outside: discs clipped or skipped (strict 0 < x < w, 0 < y < h)
    pre = np.array([[0.5, 0.5], [319.5, 239.5], [0.0, 100.0], [160.0, 0.0], [330.0, 100.0], [150.2, 60.9], [25.0, 25.0]])
    k = len(pre)
    ids[:k] = np.arange(1000, 1000 + k)
    xy[:k] = pre
    age[:k] = 7
    det = oracle.AnmsDetector(maxn, 12.0)
    T = oracle.FeatureTable(cap)
    T.set(ids[:k].copy(), pre, age[:k].copy())
    n = ctx.detect(fr, ids, xy, age, k, mask)
    assert det.update_detection(T, fr, mask) == 0
    flag, kp = ctx.debug_anms()
    assert np.array_equal(flag, det.build_mask(w, h, pre, mask))
    assert n == T.n.value and np.array_equal(ids[:n], T.ids[:n]) and np.array_equal(xy[:n], T.xy[:n])
    assert np.array_equal(age[:n], T.age[:n]) and (xy[k:n, 0] >= 100).all()
    # table already at max_feature_number: nothing is added, the counter stays
    idx0 = ctx.get_max_feature_index()
    ids2, xy2, age2 = _table(cap)
    ids2[:maxn] = np.arange(1, maxn + 1)
    xy2[:maxn] = 50.0
    n2 = ctx.detect(fr, ids2, xy2, age2, maxn)
    assert n2 == maxn and ctx.get_max_feature_index() == idx0


def test_anms_fewer_detections_than_needed(vsf, oracle, synth):
    w, h = 320, 240
    frames, _ = synth.sequence(8, w, h, 1)
    ctx = vsf.Context(_params(vsf, w, h, 300, 10.0, anms_detection_features=40))
    cap = ctx.capacity
    ids, xy, age = _table(cap)
    n = ctx.detect(frames[0], ids, xy, age, 0)
    det = oracle.AnmsDetector(300, 10.0, detection_feature_number=40)
    T = oracle.FeatureTable(cap)
    assert det.update_detection(T, frames[0]) == 0
    assert n == T.n.value == 40
    assert np.array_equal(ids[:n], T.ids[:n]) and np.array_equal(xy[:n], T.xy[:n])


def test_anms_batch_matches_single(vsf, oracle, synth):
    import torch
    w, h, B, maxn = 320, 240, 5, 80
    seqs = [synth.sequence(40 + s, w, h, 3)[0] for s in range(B)]
    ctx = vsf.Context(_params(vsf, w, h, maxn, 10.0, streams=B))
    cap = ctx.capacity
    dets = [oracle.AnmsDetector(maxn, 10.0) for _ in range(B)]
    trks = [oracle.Tracker(30, 0.01, 21, 3, 1.0, acc_mode=oracle.ACC_EXACT) for _ in range(B)]
    tabs = [oracle.FeatureTable(cap) for _ in range(B)]
    for t in range(3):
        batch = torch.from_numpy(np.stack([seqs[s][t] for s in range(B)])).cuda()
        ctx.batch_step(batch.data_ptr(), w * h, w, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ids, xy, age, n, _ = ctx.batch_download()
        for s in range(B):
            trks[s].track(tabs[s], seqs[s][t])
            dets[s].update_detection(tabs[s], seqs[s][t])
            m = tabs[s].n.value
            assert n[s] == m
            assert np.array_equal(ids[s, :m], tabs[s].ids[:m])
            assert np.array_equal(xy[s, :m], tabs[s].xy[:m])
            assert np.array_equal(age[s, :m], tabs[s].age[:m])
    assert (ctx.batch_status() == 0).all()


@pytest.mark.parametrize("B", [512, 1024])
def test_anms_many_streams_by_replication(vsf, oracle, synth, B):
    """The top-K and SSC kernels run one workgroup per stream whose size the library picks from the stream count
    (1024 threads below 512 streams -- every other test here --, 512 below 1024, 256 from 1024 on: the bench's batch).
    B interleaved copies of 4 distinct sequences: the 4 are held to the oracle, every replica to its original."""
    import torch
    w, h, maxn, D, T = 240, 176, 120, 4, 4
    base = np.stack([synth.sequence(700 + b, w, h, T)[0] for b in range(D)], 1).copy()   # [T][D][H][W]
    ctx = vsf.Context(_params(vsf, w, h, maxn, 8.0, streams=B))
    cap = ctx.capacity
    dev = torch.from_numpy(base).cuda().repeat(1, B // D, 1, 1).contiguous()
    dets = [oracle.AnmsDetector(maxn, 8.0) for _ in range(D)]
    trks = [oracle.Tracker(30, 0.01, 21, 3, 1.0, acc_mode=oracle.ACC_EXACT) for _ in range(D)]
    tabs = [oracle.FeatureTable(cap) for _ in range(D)]
    for t in range(T):
        ctx.batch_step(dev[t].data_ptr(), w * h, w, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ids, xy, age, n, _ = ctx.batch_download()
        for b in range(D):
            trks[b].track(tabs[b], base[t, b])
            dets[b].update_detection(tabs[b], base[t, b])
            m = tabs[b].n.value
            assert n[b] == m, "step %d stream %d" % (t, b)
            assert np.array_equal(ids[b, :m], tabs[b].ids[:m]) and np.array_equal(xy[b, :m], tabs[b].xy[:m])
            assert np.array_equal(age[b, :m], tabs[b].age[:m])
            assert (n[b::D] == m).all()
            assert (ids[b::D, :m] == ids[b, :m]).all() and (xy[b::D, :m] == xy[b, :m]).all()
            assert (age[b::D, :m] == age[b, :m]).all()
    assert (ctx.batch_status() == 0).all()
    ctx.close()


def test_anms_golden_fixtures_on_gpu(vsf):
    """The committed golden vectors (tests/golden/*.npz, generated from the oracle) through the C ABI."""
    import glob
    import os
    paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
    assert paths
    for path in paths:
        g = np.load(path)
        frames = g["frames"]
        t = g["trk_args"]
        amax, adist = int(g["anms_args"][0]), float(g["anms_args"][1])
        h, w = frames[0].shape
        ctx = vsf.Context(_params(vsf, w, h, amax, adist, klt_max_iter=int(t[0]), klt_eps=float(t[1]),
                                  klt_win=int(t[2]), klt_max_level=int(t[3]), backtrack_thr=float(t[4])))
        ids, xy, age = _table(ctx.capacity)
        n = 0
        for k in range(len(frames)):
            n, _ = ctx.track(frames[k], ids, xy, age, n)
            n = ctx.detect(frames[k], ids, xy, age, n)
            flag, kp = ctx.debug_anms()
            assert np.array_equal(np.packbits(flag), g["anms_mask%d" % k])
            if k == 0:
                assert np.array_equal(kp.astype(np.float32), g["anms_kp0"])
            assert np.array_equal(ids[:n], g["anms_det%d_ids" % k])
            assert np.array_equal(xy[:n], g["anms_det%d_xy" % k])
            assert np.array_equal(age[:n], g["anms_det%d_age" % k])
            assert ctx.get_max_feature_index() == int(g["anms_det%d_maxidx" % k][0])
        ctx.close()
