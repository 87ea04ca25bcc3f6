"""The rest of FeatureDetectorANMS::Parameter on the GPU (FeatureDetectorANMS.cpp:24, :27-28, passed to
cv::GFTTDetector::create at :41-47): Harris scoring (detection_use_harris_corner_ / detection_k_) and
detection_min_feature_distance_ above 1 (cv's greedy minDistance rule).  Bit-exact against the oracle, through
the C ABI: the GFTT keypoint list in cv's order, then the table after FeatureDetectorANMS::Detect.

The oracle's Harris formula restates OpenCV 3.4's calcHarris from memory (parity unpinned, like the rest of the
OpenCV arithmetic: the reference holds no Harris fixture)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _table(cap):
    return np.zeros(cap, np.uint64), np.zeros((cap, 2), np.float64), np.zeros(cap, np.uint32)


def _run_sequence(vsf, oracle, frames, w, h, maxn, mind, gpu_kw, orc_kw, gftt_kw, min_kp=20, mask=None):
    ctx = vsf.Context(vsf.default_params(detector_kind=vsf.DETECTOR_ANMS, max_features=maxn, min_distance=mind,
                                         width=w, height=h, **gpu_kw))
    cap = ctx.capacity
    ids, xy, age = _table(cap)
    det = oracle.AnmsDetector(maxn, mind, **orc_kw)
    T = oracle.FeatureTable(cap)
    rng = np.random.default_rng(11)
    n = 0
    prev_xy = np.zeros((0, 2))
    nfeat = orc_kw.get("detection_feature_number", 2 * maxn)
    for fr in frames:
        n = ctx.detect(fr, ids, xy, age, n, mask)
        assert det.update_detection(T, fr, mask) in (0, 1)
        _, kp = ctx.debug_anms()
        ref_flag = det.build_mask(w, h, prev_xy, mask)
        ref_kp = oracle.good_features_to_track(fr, nfeat, det.s.detection_quality_level,
                                               det.s.detection_min_feature_distance, mask=ref_flag,
                                               block_size=det.s.detection_block_size, **gftt_kw)
        assert len(ref_kp) >= min_kp
        assert np.array_equal(kp.astype(np.float32), ref_kp)
        assert n == T.n.value
        assert np.array_equal(ids[:n], T.ids[:n]) and np.array_equal(xy[:n], T.xy[:n])
        assert np.array_equal(age[:n], T.age[:n])
        assert ctx.get_max_feature_index() == det.max_feature_index
        keep = np.sort(rng.choice(n, size=int(n * 0.5), replace=False))
        k = len(keep)
        for arr, oarr in ((ids, T.ids), (xy, T.xy), (age, T.age)):
            arr[:k] = arr[keep]
            oarr[:k] = arr[:k]
        xy[:k] += rng.uniform(-1.0, 1.0, (k, 2))
        T.xy[:k] = xy[:k]
        n = T.n.value = k
        prev_xy = xy[:k].copy()
    ctx.close()


@pytest.mark.parametrize("w,h,bs,k", [(752, 480, 4, 0.04), (401, 303, 3, 0.04), (322, 241, 4, 0.06),
                                      (323, 240, 3, 0.15)])
def test_anms_harris_sequence_bit_exact(vsf, oracle, synth, w, h, bs, k):
    """Widths with every remainder mod 4: the last width % 4 pixels of a row take cv's scalar form (k in double)."""
    frames, _ = synth.sequence(31, w, h, 3)
    _run_sequence(vsf, oracle, frames, w, h, 200, 8.0,
                  dict(anms_use_harris=1, anms_harris_k=k, anms_block_size=bs),
                  dict(detection_use_harris_corner=1, detection_k=k, detection_block_size=bs),
                  dict(use_harris=True, k=k))


def test_anms_harris_response_map_differs_from_min_eig(vsf, oracle, synth):
    """Guards against a Harris switch that is silently ignored: the two detectors pick different corners."""
    w, h = 320, 240
    fr = synth.sequence(5, w, h, 1)[0][0]
    out = []
    for hr in (0, 1):
        ctx = vsf.Context(vsf.default_params(detector_kind=vsf.DETECTOR_ANMS, max_features=100, min_distance=8.0,
                                             width=w, height=h, anms_use_harris=hr))
        ids, xy, age = _table(ctx.capacity)
        ctx.detect(fr, ids, xy, age, 0)
        out.append(ctx.debug_anms()[1].copy())
        ctx.close()
    assert out[0].shape != out[1].shape or not np.array_equal(out[0], out[1])


def test_anms_harris_negative_maximum(vsf, oracle):
    """An image of edges only: every unmasked Harris response is negative.  cv's threshold max * quality (quality in
    [0, 1]) is then at or above the maximum and nothing passes: no corner, the table stays as it was, no id is used."""
    w, h = 160, 120
    rng = np.random.default_rng(5)
    x = np.arange(w)[None, :]
    img = (128 + 100 * np.sin(x * 0.9) + rng.integers(-1, 2, (h, w))).clip(0, 255).astype(np.uint8)
    resp = oracle.corner_harris(img, block_size=4, k=0.2)
    assert resp[20:-20, 20:-20].max() < 0
    for q in (0.0, 0.001, 1.0):
        ctx = vsf.Context(vsf.default_params(detector_kind=vsf.DETECTOR_ANMS, max_features=50, min_distance=6.0,
                                             width=w, height=h, anms_use_harris=1, anms_harris_k=0.2,
                                             anms_quality_level=q))
        ids, xy, age = _table(ctx.capacity)
        ids[:2], xy[:2], age[:2] = [7, 9], [[30.0, 40.0], [100.5, 60.25]], [3, 4]
        det = oracle.AnmsDetector(50, 6.0, detection_use_harris_corner=1, detection_k=0.2, detection_quality_level=q)
        T = oracle.FeatureTable(ctx.capacity)
        T.set(ids[:2].copy(), xy[:2].copy(), age[:2].copy())
        det.update_detection(T, img)
        n = ctx.detect(img, ids, xy, age, 2)
        assert n == T.n.value == 2 and len(ctx.debug_anms()[1]) == 0
        assert ctx.get_max_feature_index() == det.max_feature_index
        ctx.close()


@pytest.mark.parametrize("w,h,md,nfeat,noise", [
    (752, 480, 1.2, 600, 0), (752, 480, 3.0, 600, 0), (401, 303, 7.3, 400, 0),
    (752, 480, 20.0, 600, 0),     # fewer corners than wanted exist: the whole list is swept
    (320, 240, 31.0, 300, 0),
    (752, 480, 5.0, 3000, 25),    # 17755 candidates, the last accepted one is the 5543rd: two chunks of 4096 keys
    (752, 480, 9.0, 2000, 25),    # ... the 10522nd: three chunks
    (640, 480, 3.0, 8000, 25)])   # chunks of 8192 keys
def test_anms_detection_min_distance_bit_exact(vsf, oracle, synth, w, h, md, nfeat, noise):
    frames, _ = synth.sequence(37, w, h, 3 if not noise else 2)
    if noise:
        rng = np.random.default_rng(2)
        frames = np.clip(frames.astype(np.int32) + rng.integers(-noise, noise + 1, frames.shape), 0, 255).astype(np.uint8)
    _run_sequence(vsf, oracle, frames, w, h, nfeat // 2, 9.0,
                  dict(anms_detection_min_distance=md, anms_detection_features=nfeat),
                  dict(detection_min_feature_distance=md, detection_feature_number=nfeat), {})


def test_anms_detection_min_distance_noise_and_overflow(vsf, oracle):
    """Noise: a local maximum every few pixels, plateaus included.  Once with the candidate list, once with the
    list forced to overflow (keys re-derived from the response map on every pass)."""
    w, h = 200, 150
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (h, w)).astype(np.uint8)
    img[40:80, 50:120] = 77   # a flat patch: response 0, never a corner
    for cand_cap in (None, 64):
        for md, hr in ((4.5, 0), (2.0, 1)):
            ctx = vsf.Context(vsf.default_params(detector_kind=vsf.DETECTOR_ANMS, max_features=150, min_distance=5.0,
                                                 width=w, height=h, anms_detection_min_distance=md,
                                                 anms_use_harris=hr, anms_quality_level=0.01))
            if cand_cap:
                ctx.debug_set_cand_cap(cand_cap)
            ids, xy, age = _table(ctx.capacity)
            n = ctx.detect(img, ids, xy, age, 0)
            _, kp = ctx.debug_anms()
            det = oracle.AnmsDetector(150, 5.0, detection_min_feature_distance=md, detection_use_harris_corner=hr,
                                      detection_quality_level=0.01)
            T = oracle.FeatureTable(ctx.capacity)
            det.update_detection(T, img)
            ref_kp = oracle.good_features_to_track(img, 300, 0.01, md, mask=det.build_mask(w, h, np.zeros((0, 2))),
                                                   block_size=4, use_harris=bool(hr))
            assert len(ref_kp) > 100 and np.array_equal(kp.astype(np.float32), ref_kp)
            assert n == T.n.value and np.array_equal(xy[:n], T.xy[:n]) and np.array_equal(ids[:n], T.ids[:n])
            ctx.close()


@pytest.mark.parametrize("B", [5, 1024])
def test_anms_batch_with_min_distance_and_harris(vsf, oracle, synth, B):
    """The batch entry (one workgroup per stream; 256 threads a workgroup from 1024 streams on) against the oracle."""
    import torch
    w, h, maxn, U = 320, 240, 80, 5
    seqs = [synth.sequence(50 + s, w, h, 2)[0] for s in range(U)]
    ctx = vsf.Context(vsf.default_params(detector_kind=vsf.DETECTOR_ANMS, max_features=maxn, min_distance=7.0,
                                         width=w, height=h, streams=B, anms_detection_min_distance=5.0,
                                         anms_use_harris=1))
    cap = ctx.capacity
    dets = [oracle.AnmsDetector(maxn, 7.0, detection_min_feature_distance=5.0, detection_use_harris_corner=1)
            for _ in range(U)]
    trks = [oracle.Tracker(30, 0.01, 21, 3, 1.0, acc_mode=oracle.ACC_EXACT) for _ in range(U)]
    tabs = [oracle.FeatureTable(cap) for _ in range(U)]
    for t in range(2):
        batch = torch.from_numpy(np.stack([seqs[s % U][t] for s in range(B)])).cuda()
        ctx.batch_step(batch.data_ptr(), w * h, w, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ids, xy, age, n, _ = ctx.batch_download()
        for s in range(U):
            trks[s].track(tabs[s], seqs[s][t])
            dets[s].update_detection(tabs[s], seqs[s][t])
        for s in range(B):
            m = tabs[s % U].n.value
            assert n[s] == m and m > 40
            assert np.array_equal(ids[s, :m], tabs[s % U].ids[:m])
            assert np.array_equal(xy[s, :m], tabs[s % U].xy[:m])
    assert (ctx.batch_status() == 0).all()
    ctx.close()


def test_anms_params_with_user_mask_and_bgr_input(vsf, oracle, synth):
    """Harris + minimum distance together, behind a user mask, on a 3-channel frame (converted on the device)."""
    w, h = 320, 240
    frames, _ = synth.sequence(61, w, h, 3)
    mask = np.full((h, w), 255, np.uint8)
    mask[:, 200:] = 0
    mask[60:90, :] = 2     # even: wiped by the AND with the edge mask of ones
    gpu_kw = dict(anms_use_harris=1, anms_harris_k=0.05, anms_detection_min_distance=3.3)
    orc_kw = dict(detection_use_harris_corner=1, detection_k=0.05, detection_min_feature_distance=3.3)
    _run_sequence(vsf, oracle, frames, w, h, 120, 8.0, gpu_kw, orc_kw, dict(use_harris=True, k=0.05), mask=mask)
    # the same frame as BGR (B = G = R: the fixed-point gray weights sum to 2^14, the conversion is the identity)
    ctx = vsf.Context(vsf.default_params(detector_kind=vsf.DETECTOR_ANMS, max_features=120, min_distance=8.0,
                                         width=w, height=h, **gpu_kw))
    ids, xy, age = _table(ctx.capacity)
    bgr = np.repeat(frames[0][:, :, None], 3, axis=2)
    n = ctx.detect(bgr, ids, xy, age, 0, mask)
    det, T = oracle.AnmsDetector(120, 8.0, **orc_kw), oracle.FeatureTable(ctx.capacity)
    det.update_detection(T, frames[0], mask)
    assert n == T.n.value > 30 and np.array_equal(xy[:n], T.xy[:n]) and (xy[:n, 0] < 200).all()
    ctx.close()


def test_anms_min_distance_up_to_one_changes_nothing_and_limits_are_refused(vsf, synth):
    import ctypes as C
    w, h = 200, 150
    fr = synth.sequence(62, w, h, 1)[0][0]
    outs = []
    for md in (1.0, 0.0, 0.5, -3.0):     # cv: minDistance < 1 runs no distance test, 1 never fails it
        ctx = vsf.Context(vsf.default_params(detector_kind=vsf.DETECTOR_ANMS, max_features=60, min_distance=6.0,
                                             width=w, height=h, anms_detection_min_distance=md))
        ids, xy, age = _table(ctx.capacity)
        n = ctx.detect(fr, ids, xy, age, 0)
        outs.append((n, xy[:n].copy()))
        ctx.close()
    assert all(o[0] == outs[0][0] > 20 and np.array_equal(o[1], outs[0][1]) for o in outs[1:])
    hnd = C.c_void_p()
    for kw in (dict(anms_detection_min_distance=31.5), dict(anms_detection_min_distance=float("nan")),
               dict(anms_use_harris=1, anms_harris_k=float("inf"))):
        p = vsf.default_params(detector_kind=vsf.DETECTOR_ANMS, max_features=60, min_distance=6.0, width=w, height=h, **kw)
        assert vsf.lib().vsf_create(C.byref(p), C.byref(hnd)) == vsf.VSF_E_INVALID, kw
