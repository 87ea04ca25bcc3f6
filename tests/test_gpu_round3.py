"""Round-3 additions of the boundary, through the C ABI against the CPU oracle: multi-channel detector input
(cv::cvtColor BGR2GRAY, FeatureDetectorShiTomasiBucketing.cpp:105-110), a frame size per call on a single-stream
context (the reference takes any cv::Mat), the caller-capacity rule of vsf_step / vsf_detect, the optional LK
template cache (opt-out, reporting), the native gather entry (world of one) and the shader clock probe."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _table(cap):
    return np.zeros(cap, np.uint64), np.zeros((cap, 2), np.float64), np.zeros(cap, np.uint32)


def _bgr(synth, seed, w, h, cn):
    frames, _ = synth.sequence(seed, w, h, cn)
    img = np.stack([frames[k] for k in range(cn)], 2)  # every channel its own texture
    return np.ascontiguousarray(img)


@pytest.mark.parametrize("w,h,cn,kind", [(752, 480, 3, "bucketing"), (321, 243, 4, "bucketing"), (402, 300, 3, "anms")])
def test_detect_bgr_matches_oracle(vsf, oracle, synth, w, h, cn, kind):
    img = _bgr(synth, 41, w, h, cn)
    gray = oracle.bgr2gray(img)
    if kind == "anms":
        par = vsf.FeatureDetectorANMS.Parameter()
        det = vsf.FeatureDetectorANMS(par, w, h)
        odet = oracle.AnmsDetector(par.max_feature_number_, par.min_feature_distance_)
        cap = det.ctx.capacity
    else:
        det = vsf.FeatureDetectorShiTomasiBucketing(4, 4, 300, 5.0, w, h)
        odet = oracle.Detector(4, 4, 300, 5.0)
        cap = det.ctx.capacity
    db, tab = vsf.FeatureDatabase(cap), oracle.FeatureTable(cap)
    for _ in range(2):  # second call: existing features, id counter
        det.UpdateDetection(db, img)
        assert odet.update_detection(tab, gray) == 0
        assert db.n == tab.n.value and db.n > 50
        assert np.array_equal(db.ids[:db.n], tab.ids[:db.n]) and np.array_equal(db.xy[:db.n], tab.xy[:db.n])
    # strided rows, and the same result as the grey frame through vsf_detect
    wide = np.zeros((h, w + 5, cn), np.uint8)
    wide[:, :w] = img
    ids, xy, age = _table(cap)
    n = C.c_int32(0)
    ctx2 = vsf.Context(vsf.default_params(width=w, height=h, grid_rows=4, grid_cols=4, max_features=300, min_distance=5.0))
    rc = vsf.lib().vsf_detect_bgr(ctx2._h, wide.ctypes.data_as(C.c_void_p), w, h, (w + 5) * cn, cn, None, 0,
                                  ids.ctypes.data_as(C.c_void_p), xy.ctypes.data_as(C.c_void_p),
                                  age.ctypes.data_as(C.c_void_p), C.byref(n), len(ids))
    assert rc == 0
    ids2, xy2, age2 = _table(cap)
    ctx3 = vsf.Context(vsf.default_params(width=w, height=h, grid_rows=4, grid_cols=4, max_features=300, min_distance=5.0))
    n2 = ctx3.detect(gray, ids2, xy2, age2, 0)
    assert n.value == n2 and np.array_equal(xy[:n2], xy2[:n2]) and np.array_equal(ids[:n2], ids2[:n2])
    assert vsf.lib().vsf_detect_bgr(ctx2._h, wide.ctypes.data_as(C.c_void_p), w, h, (w + 5) * cn, 2, None, 0,
                                    ids.ctypes.data_as(C.c_void_p), xy.ctypes.data_as(C.c_void_p),
                                    age.ctypes.data_as(C.c_void_p), C.byref(n), len(ids)) == vsf.VSF_E_INVALID


def test_batch_bgr_to_gray(vsf, oracle, synth):
    import torch
    w, h, B = 322, 200, 3
    imgs = np.stack([_bgr(synth, 50 + s, w, h, 3) for s in range(B)])
    ctx = vsf.Context(vsf.default_params(width=w, height=h, streams=B))
    d_in = torch.from_numpy(imgs).cuda()
    d_out = torch.zeros((B, h, w), dtype=torch.uint8, device="cuda")
    ctx.batch_bgr_to_gray(d_in.data_ptr(), h * w * 3, w * 3, 3, d_out.data_ptr(), h * w, w,
                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    for s in range(B):
        assert np.array_equal(got[s], oracle.bgr2gray(imgs[s]))


def test_frame_size_per_call(vsf, oracle, synth):
    """One detector object, frames of several sizes in a row (the reference takes any cv::Mat): tables and the id
    counter match the oracle's detector, whose state likewise carries across sizes."""
    det = vsf.FeatureDetectorShiTomasiBucketing(3, 3, 90, 4.0, 320, 240)
    odet = oracle.Detector(3, 3, 90, 4.0)
    cap = det.ctx.capacity
    for k, (w, h) in enumerate([(320, 240), (401, 303), (320, 240), (752, 480), (97, 61)]):
        img = synth.sequence(60 + k, w, h, 1)[0][0]
        db, tab = vsf.FeatureDatabase(cap), oracle.FeatureTable(cap)
        det.UpdateDetection(db, img)
        assert odet.update_detection(tab, img) == 0
        assert db.as_dict() == tab.as_dict() and db.n > 10, (w, h)
        assert det.max_feature_index_ == odet.max_feature_index
        i = det.ctx.info()
        assert (i.width, i.height) == (w, h)
    # the tracker follows a new size once its previous frame is forgotten; the pair form takes any size
    trk = vsf.FeatureTrackerLucasKanade(30, 0.01, 21, 3, 1.0, 320, 240, capacity=cap)
    otrk = oracle.Tracker(30, 0.01, 21, 3, 1.0, oracle.ACC_EXACT)
    fr, _ = synth.sequence(70, 480, 360, 3)
    db, tab = vsf.FeatureDatabase(cap), oracle.FeatureTable(cap)
    det.UpdateDetection(db, fr[0])
    odet.update_detection(tab, fr[0])
    trk.Track(db, fr[0], fr[1])   # 4-argument form on 480x360 with a context created for 320x240
    otrk.track_pair(tab, fr[0], fr[1])
    assert db.as_dict() == tab.as_dict() and db.n > 30
    trk.Track(db, fr[1])          # 3-argument form: stores
    trk.Track(db, fr[2])
    otrk.track(tab, fr[1])
    otrk.track(tab, fr[2])
    assert db.as_dict() == tab.as_dict() and db.n > 30


def test_step_checks_the_callers_capacity_before_anything_runs(vsf, oracle, synth):
    w, h = 320, 240
    frames, _ = synth.sequence(12, w, h, 3)
    ctx = vsf.Context(vsf.default_params(width=w, height=h, grid_rows=2, grid_cols=2, max_features=120, min_distance=5.0))
    cap = ctx.capacity
    ids, xy, age = _table(cap)
    n, _ = ctx.step(frames[0], ids, xy, age, 0)
    assert n > 60
    before = (ctx.get_max_feature_index(), ids.copy(), xy.copy(), n)
    short = (ids[:n + 3].copy(), xy[:n + 3].copy(), age[:n + 3].copy())   # too short for what detection may add
    with pytest.raises(vsf.VsfError) as e:
        ctx.step(frames[1], short[0], short[1], short[2], n)
    assert e.value.code == vsf.VSF_E_CAPACITY
    assert ctx.get_max_feature_index() == before[0]            # nothing ran: id counter and previous frame unchanged
    with pytest.raises(vsf.VsfError):
        ctx.detect(frames[1], short[0], short[1], short[2], n)
    assert ctx.get_max_feature_index() == before[0]
    # the retry with full-size arrays gives what an undisturbed run gives
    n1, st1 = ctx.step(frames[1], ids, xy, age, n)
    ref = vsf.Context(vsf.default_params(width=w, height=h, grid_rows=2, grid_cols=2, max_features=120, min_distance=5.0))
    rids, rxy, rage = _table(cap)
    rn, _ = ref.step(frames[0], rids, rxy, rage, 0)
    rn, rst = ref.step(frames[1], rids, rxy, rage, rn)
    assert (n1, tuple(st1)) == (rn, tuple(rst)) and np.array_equal(ids[:n1], rids[:n1]) and np.array_equal(xy[:n1], rxy[:n1])


def test_template_cache_opt_out_and_report(vsf, synth):
    import torch
    w, h, B = 320, 240, 4
    frames = np.stack([synth.sequence(80 + s, w, h, 4)[0] for s in range(B)], 1)  # [T][B][h][w]
    d = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    res = []
    for limit in (0, -1, 1 << 20):
        ctx = vsf.Context(vsf.default_params(width=w, height=h, streams=B, max_features=100, grid_rows=2, grid_cols=2,
                                             min_distance=5.0, lk_cache_max_bytes=limit))
        info = ctx.info()
        if limit == 0:
            assert info.lk_cache_bytes == 2 * B * ctx.capacity * (info.levels * (160 * 16 + 16) + 16)
        else:
            assert info.lk_cache_bytes == 0   # disabled / above the caller's limit
        for t in range(4):
            ctx.batch_step(d[t].data_ptr(), h * w, w, torch.cuda.current_stream().cuda_stream)
        res.append(ctx.batch_download())
        ctx.close()
    # same tables with and without the cache (entries beyond a stream's count are unspecified: not compared)
    ids0, xy0, age0, n0 = res[0][:4]
    for ids, xy, age, n in (r[:4] for r in res[1:]):
        assert np.array_equal(n, n0)
        for s in range(B):
            m = int(n0[s])
            assert np.array_equal(ids[s, :m], ids0[s, :m]) and np.array_equal(xy[s, :m], xy0[s, :m])
            assert np.array_equal(age[s, :m], age0[s, :m])
    assert n0.min() > 40


def test_native_gather_world_of_one(vsf, synth):
    import importlib
    import torch
    mg = importlib.import_module("vi_slam_amd.multigpu")
    w, h, B = 320, 240, 3
    frames = np.stack([synth.sequence(90 + s, w, h, 2)[0] for s in range(B)], 1)
    d = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    ctx = vsf.Context(vsf.default_params(width=w, height=h, streams=B, max_features=80, grid_rows=2, grid_cols=2))
    s = torch.cuda.current_stream().cuda_stream
    for t in range(2):
        ctx.batch_step(d[t].data_ptr(), h * w, w, s)
    rec = torch.zeros(B * ctx.capacity * 24, dtype=torch.uint8, device="cuda")
    ctx.batch_export_tracks(rec.data_ptr(), s)
    g = mg.NativeGather(vsf, ctx, 1, 0, 0)
    assert g.ok
    out = g.gather(s)
    torch.cuda.synchronize()
    assert len(out) == 1 and torch.equal(out[0], rec)
    tracks = mg.unpack_tracks(out, 1, B, ctx.capacity)
    ids, xy, age, n, _ = ctx.batch_download()
    for st in range(B):
        assert sorted(tracks[st]) == sorted(int(i) for i in ids[st, :n[st]]) and n[st] > 30
    # a communicator object for a world of one needs no librccl either
    c = vsf.Comm(None, 0, 1, 0)
    ctx.gather_tracks(c, g.recv.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(g.recv, rec)
    c.close()
    # the RCCL library itself can be opened and hands out a unique id (what rank 0 of a real job does)
    ident = vsf.comm_unique_id()
    assert len(ident) == vsf.COMM_ID_BYTES and any(ident)
    g.close()


def test_sclk_probe(vsf):
    ctx = vsf.Context(vsf.default_params(width=320, height=240))
    mhz, rt = ctx.sclk_probe(3000)
    assert 10.0 < rt < 1000.0 and 50.0 < mhz < 4000.0


def test_profile_class_mask(vsf, synth):
    """vsf_profile_enable with a class mask: only the asked kernel classes sit between events."""
    import torch
    w, h, B = 160, 120, 2
    frames = np.stack([synth.sequence(90 + s, w, h, 4)[0] for s in range(B)], 1)
    d = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    ctx = vsf.Context(vsf.default_params(width=w, height=h, streams=B, max_features=40, grid_rows=2, grid_cols=2,
                                         min_distance=4.0, klt_win=9, klt_max_level=1))
    s = torch.cuda.current_stream().cuda_stream
    ctx.batch_step(d[0].data_ptr(), h * w, w, s)
    lk = vsf.KERNEL_NAMES.index("lk")
    ctx.profile_enable(True, kernels=[lk])
    ctx.batch_step(d[1].data_ptr(), h * w, w, s)
    ctx.batch_step(d[2].data_ptr(), h * w, w, s)
    ctx.profile_enable(False)
    prof = ctx.profile_read()
    assert prof["lk"][1] == 2 and prof["lk"][0] > 0 and all(v[1] == 0 for k, v in prof.items() if k != "lk")
    ctx.profile_enable(True)
    ctx.batch_step(d[3].data_ptr(), h * w, w, s)
    ctx.profile_enable(False)
    prof = ctx.profile_read()
    assert prof["lk"][1] == 1 and prof["score"][1] >= 1 and prof["pyramid"][1] >= 1
    ctx.close()
