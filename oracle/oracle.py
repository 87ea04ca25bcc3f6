"""ctypes binding of the CPU oracle (oracle/vsf_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package (vi-slam_amd/) never does.  See oracle/vsf_oracle.h for the "parity unpinned"
statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

ACC_FLOAT_SEQ = 0  # OpenCV scalar accumulation order ("restated reference")
ACC_EXACT = 1      # exact integer sums (order independent)


def build():
    """Compile the oracle with gcc (no-op when up to date)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load(name):
    path = os.path.join(_BUILD, name)
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


class _Detector(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("max_features", C.c_int32),
                ("min_distance", C.c_double), ("max_feature_index", C.c_uint64)]


class _Tracker(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("eps", C.c_double), ("win", C.c_int32),
                ("max_level", C.c_int32), ("backtrack_thr", C.c_double), ("acc_mode", C.c_int),
                ("old_image", C.c_void_p), ("old_w", C.c_int), ("old_h", C.c_int)]


class _Stats(C.Structure):
    _fields_ = [("input", C.c_int32), ("forward", C.c_int32), ("backward", C.c_int32),
                ("verified", C.c_int32)]


class _Level(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("data", C.POINTER(C.c_uint8))]


class _Pyramid(C.Structure):
    _fields_ = [("max_level", C.c_int), ("lv", _Level * 16)]


_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
_lib_cache = {}


def lib(omp=False):
    key = "omp" if omp else "st"
    if key not in _lib_cache:
        L = _load("libvsf_oracle_omp.so" if omp else "libvsf_oracle.so")
        L.orc_num_threads.restype = C.c_int
        _lib_cache[key] = L
    return _lib_cache[key]


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


def pyr_down(img):
    img = _img(img)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyr_down_u8(img.ctypes.data_as(C.c_void_p), w, h, w, out.ctypes.data_as(C.c_void_p))
    return out


def build_pyramid(img, win, max_level):
    img = _img(img)
    h, w = img.shape
    P = _Pyramid()
    ml = lib().orc_build_pyramid(img.ctypes.data_as(C.c_void_p), w, h, w, win, max_level,
                                 C.byref(P))
    levels = []
    for i in range(ml + 1):
        L = P.lv[i]
        levels.append(np.ctypeslib.as_array(L.data, shape=(L.h, L.w)).copy())
    lib().orc_free_pyramid(C.byref(P))
    return levels


def scharr(img):
    img = _img(img)
    h, w = img.shape
    out = np.empty((h, w, 2), np.int16)
    lib().orc_scharr_s16(img.ctypes.data_as(C.c_void_p), w, h, w, out.ctypes.data_as(C.c_void_p))
    return out


def calc_optical_flow_pyr_lk(prev, nxt, pts, win=21, max_level=3, max_count=30, eps=0.01,
                             min_eig_thr=1e-4, acc_mode=ACC_FLOAT_SEQ):
    prev, nxt = _img(prev), _img(nxt)
    h, w = prev.shape
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = len(pts)
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    lib().orc_calc_optical_flow_pyr_lk(
        prev.ctypes.data_as(C.c_void_p), w, nxt.ctypes.data_as(C.c_void_p), w, w, h,
        pts.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
        st.ctypes.data_as(C.c_void_p), err.ctypes.data_as(C.c_void_p), n, win, max_level,
        max_count, C.c_double(eps), C.c_double(min_eig_thr), acc_mode)
    return out, st, err


def corner_min_eigen_val(img, roi=None, block_size=3):
    img = _img(img)
    h, w = img.shape
    rx, ry, rw, rh = [int(v) for v in (roi if roi is not None else (0, 0, w, h))]
    out = np.empty((rh, rw), np.float32)
    lib().orc_corner_min_eigen_val(img.ctypes.data_as(C.c_void_p), w, h, w, rx, ry, rw, rh,
                                   block_size, out.ctypes.data_as(C.c_void_p))
    return out


def corner_harris(img, roi=None, block_size=3, k=0.04):
    img = _img(img)
    h, w = img.shape
    rx, ry, rw, rh = [int(v) for v in (roi if roi is not None else (0, 0, w, h))]
    out = np.empty((rh, rw), np.float32)
    lib().orc_corner_harris(img.ctypes.data_as(C.c_void_p), w, h, w, rx, ry, rw, rh, block_size, C.c_double(k),
                            out.ctypes.data_as(C.c_void_p))
    return out


def good_features_to_track(img, max_corners, quality, min_dist, roi=None, mask=None,
                           block_size=3, use_harris=False, k=0.04):
    img = _img(img)
    h, w = img.shape
    rx, ry, rw, rh = [int(v) for v in (roi if roi is not None else (0, 0, w, h))]
    cap = max(rw * rh, 1)
    out = np.zeros((cap, 2), np.float32)
    mptr, mstride = None, 0
    if mask is not None:
        mask = _img(mask)
        mstride = mask.shape[1]
        mptr = C.c_void_p(mask.ctypes.data + ry * mstride + rx)
    n = lib().orc_good_features_to_track_ex(img.ctypes.data_as(C.c_void_p), w, h, w, rx, ry, rw, rh,
                                            max_corners, C.c_double(quality), C.c_double(min_dist),
                                            mptr, mstride, block_size, 1 if use_harris else 0, C.c_double(k),
                                            out.ctypes.data_as(C.c_void_p), cap)
    return out[:n].copy()


def bgr2gray(img):
    """cv::cvtColor(img, CV_BGR2GRAY) of an [h][w][3 or 4] uint8 array."""
    img = np.ascontiguousarray(img, np.uint8)
    assert img.ndim == 3 and img.shape[2] in (3, 4)
    h, w, cn = img.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray(img.ctypes.data_as(C.c_void_p), w, h, w * cn, cn, out.ctypes.data_as(C.c_void_p), w)
    return out


def generate_grid(w, h, cols, rows):
    rects = np.zeros((cols * rows, 4), np.int32)
    origin = np.zeros((cols * rows, 2), np.float32)
    lib().orc_generate_grid(w, h, cols, rows, rects.ctypes.data_as(C.c_void_p),
                            origin.ctypes.data_as(C.c_void_p))
    return rects, origin


class FeatureTable:
    """SoA image of the reference's FeaturePositionDatabase + FeatureAgeDatabase
    (src/vslam/type_defines.hpp:95-99): same key set in both maps."""

    def __init__(self, capacity):
        self.capacity = capacity
        self.ids = np.zeros(capacity, np.uint64)
        self.xy = np.zeros((capacity, 2), np.float64)
        self.age = np.zeros(capacity, np.uint32)
        self.n = C.c_int(0)

    def set(self, ids, xy, age):
        n = len(ids)
        assert n <= self.capacity
        self.ids[:n] = ids
        self.xy[:n] = xy
        self.age[:n] = age
        self.n = C.c_int(n)

    def as_dict(self):
        n = self.n.value
        return {int(i): (float(x), float(y), int(a))
                for i, (x, y), a in zip(self.ids[:n], self.xy[:n], self.age[:n])}

    def _ptrs(self):
        return (self.ids.ctypes.data_as(C.c_void_p), self.xy.ctypes.data_as(C.c_void_p),
                self.age.ctypes.data_as(C.c_void_p), C.byref(self.n))


class Detector:
    """FeatureDetectorShiTomasiBucketing (ref: FeatureDetectorShiTomasiBucketing.cpp:38-239)."""

    def __init__(self, rows, cols, max_features, min_distance):
        self.s = _Detector()
        lib().orc_detector_init(C.byref(self.s), rows, cols, max_features,
                                C.c_double(min_distance))

    @property
    def max_feature_index(self):
        return int(self.s.max_feature_index)

    def update_detection(self, table, img, mask=None):
        img = _img(img)
        h, w = img.shape
        mptr, mstride = None, 0
        if mask is not None:
            mask = _img(mask)
            mptr, mstride = mask.ctypes.data_as(C.c_void_p), mask.shape[1]
        return lib().orc_detector_update(C.byref(self.s), img.ctypes.data_as(C.c_void_p), w, h, w,
                                         mptr, mstride, *table._ptrs(), table.capacity)


class _AnmsDetector(C.Structure):
    _fields_ = [("max_feature_number", C.c_int32), ("min_feature_distance", C.c_double),
                ("detection_feature_number", C.c_int32),
                ("detection_min_feature_distance", C.c_double),
                ("detection_quality_level", C.c_double), ("detection_block_size", C.c_int32),
                ("detection_ignore_edge", C.c_int32), ("non_max_suppression_tolerance", C.c_double),
                ("max_feature_index", C.c_uint64), ("detection_use_harris_corner", C.c_int32),
                ("detection_k", C.c_double)]


def circle_filled(img, cx, cy, radius, color=0):
    """cv::circle(img, (cx, cy), radius, color, FILLED), in place."""
    assert img.dtype == np.uint8 and img.flags["C_CONTIGUOUS"]
    h, w = img.shape
    lib().orc_circle_filled(img.ctypes.data_as(C.c_void_p), w, h, w, int(cx), int(cy), int(radius),
                            C.c_uint8(color))
    return img


def circle_halfwidths(radius):
    out = np.zeros(radius + 1, np.int32)
    lib().orc_circle_halfwidths(int(radius), out.ctypes.data_as(C.c_void_p))
    return out


def sort_idx_descending(keys):
    keys = np.ascontiguousarray(keys, np.int32)
    idx = np.zeros(len(keys), np.int32)
    lib().orc_sort_idx_descending(keys.ctypes.data_as(C.c_void_p), len(keys),
                                  idx.ctypes.data_as(C.c_void_p))
    return idx


def ssc(kxy, num_ret, tolerance, cols, rows):
    kxy = np.ascontiguousarray(kxy, np.float32).reshape(-1, 2)
    res = np.zeros(max(len(kxy), 1), np.int32)
    n = lib().orc_ssc(kxy.ctypes.data_as(C.c_void_p), len(kxy), int(num_ret), C.c_float(tolerance),
                      int(cols), int(rows), res.ctypes.data_as(C.c_void_p))
    return res[:max(n, 0)].copy(), n


class AnmsDetector:
    """FeatureDetectorANMS (ref: FeatureDetectorANMS.cpp:13-212)."""

    def __init__(self, max_features=300, min_distance=10.0, **kw):
        self.s = _AnmsDetector()
        lib().orc_anms_detector_init(C.byref(self.s), max_features, C.c_double(min_distance))
        for k, v in kw.items():
            setattr(self.s, k, v)

    @property
    def max_feature_index(self):
        return int(self.s.max_feature_index)

    def build_mask(self, w, h, xy, mask=None):
        xy = np.ascontiguousarray(xy, np.float64).reshape(-1, 2)
        flag = np.zeros((h, w), np.uint8)
        mptr, mstride = None, 0
        if mask is not None:
            mask = _img(mask)
            mptr, mstride = mask.ctypes.data_as(C.c_void_p), mask.shape[1]
        lib().orc_anms_build_mask(C.byref(self.s), w, h, mptr, mstride,
                                  xy.ctypes.data_as(C.c_void_p), len(xy),
                                  flag.ctypes.data_as(C.c_void_p))
        return flag

    def update_detection(self, table, img, mask=None):
        img = _img(img)
        h, w = img.shape
        mptr, mstride = None, 0
        if mask is not None:
            mask = _img(mask)
            mptr, mstride = mask.ctypes.data_as(C.c_void_p), mask.shape[1]
        return lib().orc_anms_detector_update(C.byref(self.s), img.ctypes.data_as(C.c_void_p), w, h,
                                              w, mptr, mstride, *table._ptrs(), table.capacity)


LSSD_SEQ = 0   # sums in pattern index order, libm sin/cos ("restated reference")
LSSD_TREE = 1  # 32-slot butterfly sums + polynomial sin/cos (what a half-wavefront reproduces)


class _Level16(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("data", C.POINTER(C.c_uint16))]


class _Pyramid16(C.Structure):
    _fields_ = [("levels", C.c_int), ("lv", _Level16 * 16)]


class _LssdTracker(C.Structure):
    _fields_ = [("max_recovered_dist", C.c_double), ("max_iterations", C.c_int32), ("levels", C.c_int32),
                ("mode", C.c_int), ("old_pyr", _Pyramid16), ("has_old", C.c_int),
                ("map_ids", C.c_void_p), ("map_xy", C.c_void_p), ("map_n", C.c_int), ("map_cap", C.c_int)]


def lssd_build_pyramid(img, levels=3):
    """u8 -> u16 (<< 8) -> basalt ManagedImagePyr::setFromImage: list of levels+1 uint16 images."""
    img = _img(img)
    h, w = img.shape
    P = _Pyramid16()
    rc = lib().orc_lssd_build_pyramid(img.ctypes.data_as(C.c_void_p), w, h, w, levels, C.byref(P))
    assert rc == 0
    out = [np.ctypeslib.as_array(P.lv[i].data, shape=(P.lv[i].h, P.lv[i].w)).copy() for i in range(levels + 1)]
    lib().orc_lssd_free_pyramid(C.byref(P))
    return out


def lssd_track_points(prev, cur, pts, levels=3, max_iterations=5, mode=LSSD_SEQ):
    """FrameToFrameOpticalFlow::TrackPoint (forward only) on a point list: (next [n][2] f32, valid [n])."""
    prev, cur = _img(prev), _img(cur)
    h, w = prev.shape
    P0, P1 = _Pyramid16(), _Pyramid16()
    L = lib()
    L.orc_lssd_build_pyramid(prev.ctypes.data_as(C.c_void_p), w, h, w, levels, C.byref(P0))
    L.orc_lssd_build_pyramid(cur.ctypes.data_as(C.c_void_p), w, h, w, levels, C.byref(P1))
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    out = pts.copy()
    valid = np.zeros(len(pts), np.uint8)
    for i in range(len(pts)):
        valid[i] = L.orc_lssd_track_point(C.byref(P0), C.byref(P1), pts[i].ctypes.data_as(C.c_void_p),
                                          out[i].ctypes.data_as(C.c_void_p), levels, max_iterations, mode)
    L.orc_lssd_free_pyramid(C.byref(P0))
    L.orc_lssd_free_pyramid(C.byref(P1))
    return out, valid


class LssdTracker:
    """FeatureTrackerLSSDLucasKanade (ref: FeatureTrackerLSSDLucasKanade.cpp:13-119)."""

    def __init__(self, max_recovered_dist=0.09, levels=3, max_iterations=5, mode=LSSD_SEQ):
        self.s = _LssdTracker()
        lib().orc_lssd_tracker_init(C.byref(self.s), C.c_double(max_recovered_dist), levels, max_iterations, mode)

    def __del__(self):
        try:
            lib().orc_lssd_tracker_free(C.byref(self.s))
        except Exception:
            pass

    def track(self, table, cur):
        cur = _img(cur)
        h, w = cur.shape
        return lib().orc_lssd_tracker_track(C.byref(self.s), cur.ctypes.data_as(C.c_void_p), w, h, w,
                                            *table._ptrs())


class CameraDS(C.Structure):
    """DoubleSphereCameraModel intrinsics (ref: src/vslam/data/Camera.hpp:134)."""
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("xi", C.c_double), ("alpha", C.c_double)]


def ds_unproject(cam, xy):
    xy = np.ascontiguousarray(xy, np.float64).reshape(-1, 2)
    out = np.zeros((len(xy), 3))
    ok = np.zeros(len(xy), bool)
    L = lib()
    L.orc_ds_unproject.restype = C.c_int
    for i, (u, v) in enumerate(xy):
        b = (C.c_double * 3)()
        ok[i] = bool(L.orc_ds_unproject(C.byref(cam), C.c_double(u), C.c_double(v), b))
        out[i] = list(b)
    return out, ok


def frontend_post_track(cam, xy, age, counting_feature_age, low_keyframe_feature_number,
                        minimum_keyframe_interval, keyframe_interval, dt_since_keyframe):
    """Returns (keyframe, bearings [n][3], aged count); `age` (uint32 array) is incremented in place
    for a keyframe."""
    xy = np.ascontiguousarray(xy, np.float64).reshape(-1, 2)
    assert age.dtype == np.uint32 and age.flags["C_CONTIGUOUS"]
    n = len(xy)
    bearings = np.zeros((n, 3))
    aged = C.c_uint32(0)
    kf = lib().orc_frontend_post_track(C.byref(cam), xy.ctypes.data_as(C.c_void_p),
                                       age.ctypes.data_as(C.c_void_p), n, int(counting_feature_age),
                                       C.c_uint32(low_keyframe_feature_number),
                                       C.c_double(minimum_keyframe_interval), C.c_double(keyframe_interval),
                                       C.c_double(dt_since_keyframe), bearings.ctypes.data_as(C.c_void_p),
                                       C.byref(aged))
    return bool(kf), bearings, aged.value


class Tracker:
    """FeatureTrackerLucasKanade (ref: FeatureTrackerLucasKanade.cpp:8-240)."""

    def __init__(self, max_iter, eps, win, max_level, backtrack_thr, acc_mode=ACC_FLOAT_SEQ):
        self.s = _Tracker()
        lib().orc_tracker_init(C.byref(self.s), max_iter, C.c_double(eps), win, max_level,
                               C.c_double(backtrack_thr), acc_mode)

    def __del__(self):
        try:
            lib().orc_tracker_free(C.byref(self.s))
        except Exception:
            pass

    def track(self, table, cur):
        cur = _img(cur)
        h, w = cur.shape
        st = _Stats()
        rc = lib().orc_tracker_track(C.byref(self.s), cur.ctypes.data_as(C.c_void_p), w, h, w,
                                     *table._ptrs(), C.byref(st))
        return rc, (st.input, st.forward, st.backward, st.verified)

    def track_pair(self, table, prev, cur):
        prev, cur = _img(prev), _img(cur)
        h, w = cur.shape
        st = _Stats()
        rc = lib().orc_tracker_track_pair(C.byref(self.s), prev.ctypes.data_as(C.c_void_p), w,
                                          cur.ctypes.data_as(C.c_void_p), w, w, h,
                                          *table._ptrs(), C.byref(st))
        return rc, (st.input, st.forward, st.backward, st.verified)


class StreamsBaseline:
    """Timed CPU baseline: `streams` independent (tracker, detector) pairs stepped together
    (orc_step_streams), OpenMP over streams."""

    def __init__(self, streams, w, h, capacity, det_params, trk_params, omp=True, threads=0):
        self.L = lib(omp)
        self.L.orc_set_threads(int(threads))  # 0 = every core
        self.streams, self.w, self.h, self.capacity = streams, w, h, capacity
        self.trk = (_Tracker * streams)()
        self.det = (_Detector * streams)()
        for s in range(streams):
            self.L.orc_tracker_init(C.byref(self.trk[s]), trk_params[0], C.c_double(trk_params[1]),
                                    trk_params[2], trk_params[3], C.c_double(trk_params[4]),
                                    ACC_FLOAT_SEQ)
            self.L.orc_detector_init(C.byref(self.det[s]), det_params[0], det_params[1],
                                     det_params[2], C.c_double(det_params[3]))
        self.ids = np.zeros((streams, capacity), np.uint64)
        self.xy = np.zeros((streams, capacity, 2), np.float64)
        self.age = np.zeros((streams, capacity), np.uint32)
        self.n = np.zeros(streams, np.int32)

    @property
    def threads(self):
        # several streams: one thread per stream at most; ONE stream: the row / point loops take every thread
        return self.L.orc_num_threads() if self.streams == 1 else min(self.L.orc_num_threads(), self.streams)

    def step(self, frames):
        frames = np.ascontiguousarray(frames, np.uint8)
        assert frames.shape == (self.streams, self.h, self.w)
        return self.L.orc_step_streams(
            self.trk, self.det, self.streams, frames.ctypes.data_as(C.c_void_p),
            C.c_long(self.h * self.w), self.w, self.h, self.w,
            self.ids.ctypes.data_as(C.c_void_p), self.xy.ctypes.data_as(C.c_void_p),
            self.age.ctypes.data_as(C.c_void_p), self.n.ctypes.data_as(C.c_void_p), self.capacity)

    def close(self):
        for s in range(self.streams):
            self.L.orc_tracker_free(C.byref(self.trk[s]))
