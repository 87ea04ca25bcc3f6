"""CPU checks of the LSSD SE(2) patch-tracker restatement (oracle/vsf_oracle.c, PARITY UNPINNED: the
reference holds no vectors for FeatureTrackerLSSDLucasKanade and basalt/Eigen/Sophus are absent).
What can be checked without the reference: the u16 mip pyramid against an independent numpy
statement of basalt's subsample, the 3x3 LDLT inverse against numpy, tracking accuracy against the
known synthetic motion, the tracker's table semantics, and that the two summation orders
(LSSD_SEQ = pattern order / LSSD_TREE = half-wavefront butterfly) agree to float rounding."""
import ctypes as C

import numpy as np
import pytest


def _np_subsample(a):
    """basalt Image::subsample restated with numpy: [1 4 6 4 1]^2, reflect-101 borders, one rounding."""
    h, w = a.shape
    dh, dw = h // 2, w // 2
    k = np.array([1, 4, 6, 4, 1], np.int64)

    def refl(i, n):
        i = np.abs(i)
        return np.where(i >= n, 2 * n - 2 - i, i)

    ys = refl(2 * np.arange(dh)[:, None] + np.arange(-2, 3)[None, :], h)  # [dh][5]
    xs = refl(2 * np.arange(dw)[:, None] + np.arange(-2, 3)[None, :], w)
    a = a.astype(np.int64)
    rows = (a[ys] * k[None, :, None]).sum(1)          # [dh][w]
    out = (rows[:, xs] * k[None, None, :]).sum(2)     # [dh][dw]
    return ((out + 128) >> 8).astype(np.uint16)


@pytest.mark.parametrize("w,h,levels", [(752, 480, 3), (321, 243, 4), (64, 48, 2), (37, 29, 3)])
def test_pyramid_matches_numpy_restatement(oracle, w, h, levels):
    rng = np.random.default_rng(w * 7 + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    pyr = oracle.lssd_build_pyramid(img, levels)
    assert len(pyr) == levels + 1
    assert np.array_equal(pyr[0], img.astype(np.uint16) << 8)
    for l in range(1, levels + 1):
        assert pyr[l].shape == (h >> l, w >> l)
        assert np.array_equal(pyr[l], _np_subsample(pyr[l - 1]))


def test_ldlt_inverse_matches_numpy(oracle):
    rng = np.random.default_rng(5)
    fn = oracle.lib().orc_lssd_debug_inverse3
    for _ in range(200):
        J = rng.normal(size=(24, 3)).astype(np.float32) * rng.uniform(0.1, 50)
        H = (J.T @ J).astype(np.float32)
        Hi = np.zeros((3, 3), np.float32)
        fn(H.ctypes.data_as(C.c_void_p), Hi.ctypes.data_as(C.c_void_p))
        ref = np.linalg.inv(H.astype(np.float64))
        assert np.allclose(Hi, ref, rtol=2e-3 * np.linalg.cond(H.astype(np.float64)) ** 0.5, atol=1e-7)


@pytest.mark.parametrize("mode", [0, 1])
def test_tracks_known_translation(oracle, synth, mode):
    w, h = 320, 240
    frames, flows = synth.sequence(77, w, h, 2)
    kp = oracle.good_features_to_track(frames[0], 150, 0.01, 8.0)
    kp = kp[(kp[:, 0] > 40) & (kp[:, 0] < w - 40) & (kp[:, 1] > 40) & (kp[:, 1] < h - 40)]
    out, ok = oracle.lssd_track_points(frames[0], frames[1], kp, 3, 5, mode)
    assert ok.mean() > 0.8
    d = (out - kp)[ok.astype(bool)]
    med = np.median(d, axis=0)
    # the synthetic sequence moves by a few pixels per frame; the tracked displacement field is tight
    assert np.abs(d - med).max(axis=1).mean() < 1.5
    # identical frames: zero motion to float rounding
    out0, ok0 = oracle.lssd_track_points(frames[0], frames[0], kp, 3, 5, mode)
    assert ok0.all() and np.abs(out0 - kp).max() < 1e-3


def test_seq_and_tree_sums_agree_to_rounding(oracle, synth):
    w, h = 320, 240
    frames, _ = synth.sequence(31, w, h, 2)
    kp = oracle.good_features_to_track(frames[0], 200, 0.01, 6.0)
    a, oka = oracle.lssd_track_points(frames[0], frames[1], kp, 3, 5, oracle.LSSD_SEQ)
    b, okb = oracle.lssd_track_points(frames[0], frames[1], kp, 3, 5, oracle.LSSD_TREE)
    both = (oka & okb).astype(bool)
    assert (oka == okb).mean() > 0.97
    assert np.median(np.abs(a - b)[both]) < 1e-3


def test_tracker_semantics(oracle, synth):
    w, h = 320, 240
    frames, _ = synth.sequence(12, w, h, 3)
    kp = oracle.good_features_to_track(frames[0], 120, 0.01, 8.0)
    n0 = len(kp)
    T = oracle.FeatureTable(256)
    T.set(np.arange(10, 10 + n0, dtype=np.uint64), kp.astype(np.float64), np.full(n0, 3, np.uint32))
    trk = oracle.LssdTracker()
    assert trk.track(T, frames[0]) == 0
    # first call: pyramid stored, table untouched (FeatureTrackerLSSDLucasKanade.cpp:57-68)
    assert T.n.value == n0 and np.array_equal(T.xy[:n0], kp.astype(np.float64))
    assert trk.track(T, frames[1]) == 0
    n1 = T.n.value
    assert 0 < n1 <= n0
    # survivors keep id order and age; positions are floats widened to double
    assert np.all(np.diff(T.ids[:n1].astype(np.int64)) > 0) and (T.age[:n1] == 3).all()
    assert np.array_equal(T.xy[:n1], T.xy[:n1].astype(np.float32).astype(np.float64))
    # features out of the frame are dropped
    T2 = oracle.FeatureTable(16)
    T2.set(np.array([1, 2, 3], np.uint64), np.array([[-5.0, 10.0], [w + 3.0, 20.0], [1.0, 1.0]]), np.ones(3, np.uint32))
    assert trk.track(T2, frames[2]) == 0
    assert T2.n.value == 0
    # empty table
    T3 = oracle.FeatureTable(4)
    assert trk.track(T3, frames[2]) == 0 and T3.n.value == 0


def test_tracker_keeps_its_own_translation_per_id(oracle, synth):
    """transform_ is refreshed from the database only for ids new to it (:73-80): a position the caller
    rewrites for a surviving id is ignored, a new id starts from the table."""
    w, h = 320, 240
    frames, _ = synth.sequence(12, w, h, 3)
    kp = oracle.good_features_to_track(frames[0], 60, 0.01, 10.0).astype(np.float64)
    n0 = len(kp)
    ids = np.arange(1, n0 + 1, dtype=np.uint64)

    def run(mutate):
        T = oracle.FeatureTable(128)
        T.set(ids.copy(), kp.copy(), np.ones(n0, np.uint32))
        trk = oracle.LssdTracker()
        trk.track(T, frames[0])
        trk.track(T, frames[1])
        mutate(T)
        trk.track(T, frames[2])
        return T.as_dict()

    base = run(lambda T: None)

    def rewrite(T):  # same id, another position: ignored
        T.xy[0] = [17.0, 23.0]

    seen = {}

    def rename(T):   # a new id at another tracked position: starts from the table
        seen["a"], seen["b"] = int(T.ids[0]), int(T.ids[1])
        T.ids[0] = 999
        T.xy[0] = T.xy[1]

    assert run(rewrite) == base
    got = run(rename)
    assert seen["a"] not in got
    if seen["b"] in got:
        assert got[999][:2] == got[seen["b"]][:2]


def test_se2_exp_polynomial_against_libm(oracle):
    """SE2::exp's sin / cos: the reference calls libm (LSSD_SEQ); the wavefront-exact mode the HIP kernel is held
    to (LSSD_TREE) evaluates Taylor polynomials (degree 9 / 10).  Over |theta| <= 1 rad -- an increment of 57
    degrees in ONE Gauss-Newton step; the rotation increments the tracker produces on the test sequences stay
    below 0.2 rad -- the polynomial is within 1.5 float ulp of the correctly rounded value and within 1e-7 of
    libm's sinf / cosf; beyond |theta| = 2 it degrades (5e-5 at 2 rad), where a patch has long failed its
    validity test."""
    import ctypes as C
    fn = oracle.lib().orc_lssd_debug_sincos
    fn.argtypes = [C.c_float, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    th = np.concatenate([np.linspace(-1, 1, 20001), np.float32(10.0) ** np.linspace(-8, 0, 400),
                         -(np.float32(10.0) ** np.linspace(-8, 0, 400))]).astype(np.float32)
    worst_ulp = worst_abs = 0.0
    for t in th:
        s, c, s2, c2 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        fn(float(t), oracle.LSSD_TREE, C.byref(s), C.byref(c))
        fn(float(t), oracle.LSSD_SEQ, C.byref(s2), C.byref(c2))
        for got, lib, true in ((s.value, s2.value, np.sin(np.float64(t))), (c.value, c2.value, np.cos(np.float64(t)))):
            ulp = np.spacing(np.float32(abs(true))) if true != 0 else np.float32(1e-45)
            worst_ulp = max(worst_ulp, abs(got - true) / float(ulp))
            worst_abs = max(worst_abs, abs(got - lib))
    assert worst_ulp <= 1.5 and worst_abs <= 1.2e-7, (worst_ulp, worst_abs)
    s, c = C.c_float(), C.c_float()
    fn(2.0, oracle.LSSD_TREE, C.byref(s), C.byref(c))
    assert abs(s.value - np.sin(2.0)) < 1e-4
