"""vi-slam_amd: MI355X-native vision-frontend hot path (pyramid -> Shi-Tomasi grid detect -> KLT).

This package is a thin ctypes binding of the C ABI in include/vsf.h (libvsf_hip.so, hand-written
HIP kernels for gfx950) plus Python mirrors of the two reference classes the path sits behind
(vslam::feature::FeatureDetectorShiTomasiBucketing / FeatureTrackerLucasKanade).  There is no CPU
fallback: loading fails loudly when the HIP library is missing, and creating a context fails
without a GPU.  The directory name carries a '-', so import it with `load_package()` from
repo-root helpers (tests/conftest.py, bench.py, __graft_entry__.py) under the name `vi_slam_amd`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvsf_hip.so")

VSF_OK = 0
VSF_E_INVALID, VSF_E_HIP, VSF_E_CAPACITY, VSF_E_GEOMETRY, VSF_E_NOMEM = -1, -2, -3, -4, -5
KERNEL_NAMES = ["pyramid", "lk", "compact", "occupancy", "score", "candidates", "select", "append"]

# every symbol include/vsf.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "vsf_default_params", "vsf_create", "vsf_destroy", "vsf_last_error", "vsf_version",
    "vsf_track", "vsf_track_pair", "vsf_detect", "vsf_step", "vsf_get_max_feature_index",
    "vsf_set_max_feature_index", "vsf_batch_track", "vsf_batch_detect", "vsf_batch_step",
    "vsf_batch_upload", "vsf_batch_download", "vsf_batch_device_tables", "vsf_batch_status",
    "vsf_batch_export_tracks",
    "vsf_reset_tracker", "vsf_debug_pyramid", "vsf_debug_lk", "vsf_debug_min_eig", "vsf_debug_anms",
    "vsf_debug_set_cand_cap",
    "vsf_batch_frontend_post_track", "vsf_unproject", "vsf_debug_lssd", "vsf_debug_lssd_pyramid", "vsf_debug_lssd_inverse3",
    "vsf_profile_enable", "vsf_profile_read", "vsf_lk_points_noalloc", "vsf_get_info",
    "vsf_detect_bgr", "vsf_batch_bgr_to_gray",
    "vsf_comm_unique_id", "vsf_comm_create", "vsf_comm_wrap", "vsf_comm_destroy", "vsf_comm_rank",
    "vsf_comm_last_error", "vsf_gather_tracks", "vsf_debug_sclk_probe",
]


class VsfParams(C.Structure):
    _fields_ = [
        ("grid_rows", C.c_int32), ("grid_cols", C.c_int32), ("max_features", C.c_int32),
        ("min_distance", C.c_double), ("quality_level", C.c_double), ("block_size", C.c_int32),
        ("klt_max_iter", C.c_int32), ("klt_eps", C.c_double), ("klt_win", C.c_int32),
        ("klt_max_level", C.c_int32), ("min_eig_thr", C.c_double), ("backtrack_thr", C.c_double),
        ("width", C.c_int32), ("height", C.c_int32), ("streams", C.c_int32),
        ("capacity", C.c_int32), ("device_id", C.c_int32),
        ("detector_kind", C.c_int32), ("anms_detection_features", C.c_int32),
        ("anms_quality_level", C.c_double), ("anms_block_size", C.c_int32),
        ("anms_ignore_edge", C.c_int32), ("anms_tolerance", C.c_double),
        ("tracker_kind", C.c_int32), ("lssd_levels", C.c_int32), ("lssd_max_iterations", C.c_int32),
        ("lssd_max_recovered_dist", C.c_double), ("lk_cache_max_bytes", C.c_int64),
        ("anms_detection_min_distance", C.c_double), ("anms_use_harris", C.c_int32), ("anms_harris_k", C.c_double),
    ]


class VsfInfo(C.Structure):
    _fields_ = [("lk_cache_bytes", C.c_int64), ("width", C.c_int32), ("height", C.c_int32), ("streams", C.c_int32),
                ("capacity", C.c_int32), ("levels", C.c_int32), ("reserved", C.c_int32)]


DETECTOR_BUCKETING = 0
DETECTOR_ANMS = 1
TRACKER_LK = 0
TRACKER_LSSD = 1


class VsfCameraDS(C.Structure):
    """DoubleSphereCameraModel intrinsics (ref: src/vslam/data/Camera.cpp:101-122)."""
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("xi", C.c_double), ("alpha", C.c_double)]


class VsfKeyframeParams(C.Structure):
    """KimeraFrontend::Parameter fields used by the keyframe decision (KimeraFrontendParameter.cpp:11-14)."""
    _fields_ = [("keyframe_interval", C.c_double), ("minimum_keyframe_interval", C.c_double),
                ("low_keyframe_feature_number", C.c_uint32), ("counting_feature_age", C.c_int32)]


def keyframe_params(keyframe_interval=5.0, minimum_keyframe_interval=0.5, low_keyframe_feature_number=200,
                    counting_feature_age=2):
    return VsfKeyframeParams(keyframe_interval, minimum_keyframe_interval, low_keyframe_feature_number,
                             counting_feature_age)


class VsfTrackStats(C.Structure):
    _fields_ = [("input", C.c_int32), ("forward", C.c_int32), ("backward", C.c_int32),
                ("verified", C.c_int32)]


class VsfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vsf error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Loads libvsf_hip.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("HIP extension missing: %s (run __graft_entry__.build())" % LIB_PATH)
        # PyTorch wheels bundle their own libamdhip64.  Two HIP runtimes in one process leave the
        # second one without devices, so when torch is installed let it bring its runtime in first
        # (libvsf_hip.so then binds to that copy).  VSF_NO_TORCH=1 skips this for torch-free hosts.
        if os.environ.get("VSF_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(LIB_PATH)
        L.vsf_last_error.restype = C.c_char_p
        L.vsf_version.restype = C.c_char_p
        L.vsf_last_error.argtypes = [C.c_void_p]
        L.vsf_destroy.argtypes = [C.c_void_p]
        L.vsf_destroy.restype = None
        L.vsf_comm_destroy.argtypes = [C.c_void_p]
        L.vsf_comm_destroy.restype = None
        L.vsf_comm_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def default_params(**kw):
    p = VsfParams()
    lib().vsf_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """Owner of one vsf_ctx."""

    def __init__(self, params):
        self.params = params
        self._h = C.c_void_p()
        rc = lib().vsf_create(C.byref(params), C.byref(self._h))
        if rc != VSF_OK:
            self._h = None
            raise VsfError(rc, "vsf_create failed (no GPU / bad parameters)")
        self.streams = params.streams
        self.width, self.height = params.width, params.height
        cap = C.c_int32()  # the library's own answer (the default depends on the detector kind)
        self._chk(lib().vsf_batch_device_tables(self._h, None, None, None, None, C.byref(cap)))
        self.capacity = cap.value

    def info(self):
        i = VsfInfo()
        self._chk(lib().vsf_get_info(self._h, C.byref(i)))
        return i

    def close(self):
        if getattr(self, "_h", None):
            lib().vsf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != VSF_OK:
            raise VsfError(rc, lib().vsf_last_error(self._h).decode())

    # ---- single-stream host API ----
    @staticmethod
    def _frame(img):
        """A 2-D uint8 array whose rows are contiguous is passed as it is (row stride = cv::Mat::step, e.g. an
        ROI view of a wider image); anything else is copied to a packed array first."""
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1 or img.strides[0] < img.shape[1]:
            img = np.ascontiguousarray(img, np.uint8)
        return img, img.strides[0]

    def track(self, img, ids, xy, age, n):
        img, stride = self._frame(img)
        h, w = img.shape
        nn = C.c_int32(n)
        st = VsfTrackStats()
        self._chk(lib().vsf_track(self._h, _vp(img), w, h, stride, _vp(ids), _vp(xy), _vp(age),
                                  C.byref(nn), C.byref(st)))
        return nn.value, (st.input, st.forward, st.backward, st.verified)

    def track_pair(self, prev, cur, ids, xy, age, n):
        prev, pstride = self._frame(prev)
        cur, cstride = self._frame(cur)
        h, w = cur.shape
        nn = C.c_int32(n)
        st = VsfTrackStats()
        self._chk(lib().vsf_track_pair(self._h, _vp(prev), pstride, _vp(cur), cstride, w, h, _vp(ids), _vp(xy),
                                       _vp(age), C.byref(nn), C.byref(st)))
        return nn.value, (st.input, st.forward, st.backward, st.verified)

    def detect(self, img, ids, xy, age, n, mask=None):
        """UpdateDetection; an [h][w][3 or 4] array is a BGR(A) frame (cv::cvtColor on the device, vsf_detect_bgr)."""
        nn = C.c_int32(n)
        mptr, mstride = None, 0
        if mask is not None:
            mask, mstride = self._frame(mask)
            mptr = _vp(mask)
        if np.ndim(img) == 3 and np.shape(img)[2] != 1:
            img = np.ascontiguousarray(img, np.uint8)
            h, w, cn = img.shape
            self._chk(lib().vsf_detect_bgr(self._h, _vp(img), w, h, w * cn, cn, mptr, mstride, _vp(ids), _vp(xy),
                                           _vp(age), C.byref(nn), len(ids)))
        else:
            if np.ndim(img) == 3:
                img = np.asarray(img)[:, :, 0]
            img, stride = self._frame(img)
            h, w = img.shape
            self._chk(lib().vsf_detect(self._h, _vp(img), w, h, stride, mptr, mstride, _vp(ids), _vp(xy),
                                       _vp(age), C.byref(nn), len(ids)))
        self.width, self.height = w, h  # a single-stream context follows the caller's frame size
        return nn.value

    def step(self, img, ids, xy, age, n, mask=None):
        """Track, then UpdateDetection on the same frame in one call (vsf_step)."""
        img, stride = self._frame(img)
        h, w = img.shape
        nn = C.c_int32(n)
        st = VsfTrackStats()
        mptr, mstride = None, 0
        if mask is not None:
            mask, mstride = self._frame(mask)
            mptr = _vp(mask)
        self._chk(lib().vsf_step(self._h, _vp(img), w, h, stride, mptr, mstride, _vp(ids), _vp(xy), _vp(age),
                                 C.byref(nn), len(ids), C.byref(st)))
        return nn.value, (st.input, st.forward, st.backward, st.verified)

    def get_max_feature_index(self, stream=0):
        v = C.c_uint64()
        self._chk(lib().vsf_get_max_feature_index(self._h, stream, C.byref(v)))
        return v.value

    def set_max_feature_index(self, value, stream=0):
        self._chk(lib().vsf_set_max_feature_index(self._h, stream, C.c_uint64(value)))

    def reset_tracker(self):
        self._chk(lib().vsf_reset_tracker(self._h))

    # ---- batched device API (pointers are raw device addresses, e.g. torch tensor.data_ptr()) ----
    def batch_track(self, d_frames, frame_stride, row_stride, stream=0):
        self._chk(lib().vsf_batch_track(self._h, C.c_void_p(d_frames), C.c_int64(frame_stride),
                                        row_stride, C.c_void_p(stream)))

    def batch_detect(self, d_frames, frame_stride, row_stride, d_masks=0, mask_stride=0,
                     mask_row_stride=0, stream=0):
        self._chk(lib().vsf_batch_detect(self._h, C.c_void_p(d_frames), C.c_int64(frame_stride),
                                         row_stride, C.c_void_p(d_masks), C.c_int64(mask_stride),
                                         mask_row_stride, C.c_void_p(stream)))

    def batch_bgr_to_gray(self, d_bgr, bgr_frame_stride, bgr_row_stride, channels, d_gray, gray_frame_stride,
                          gray_row_stride, stream=0):
        self._chk(lib().vsf_batch_bgr_to_gray(self._h, C.c_void_p(d_bgr), C.c_int64(bgr_frame_stride), bgr_row_stride,
                                              channels, C.c_void_p(d_gray), C.c_int64(gray_frame_stride),
                                              gray_row_stride, C.c_void_p(stream)))

    def batch_step(self, d_frames, frame_stride, row_stride, stream=0):
        self._chk(lib().vsf_batch_step(self._h, C.c_void_p(d_frames), C.c_int64(frame_stride),
                                       row_stride, C.c_void_p(stream)))

    def batch_upload(self, ids, xy, age, n, first=0):
        count = len(n)
        ids = np.ascontiguousarray(ids, np.uint64)
        xy = np.ascontiguousarray(xy, np.float64)
        age = np.ascontiguousarray(age, np.uint32)
        n = np.ascontiguousarray(n, np.int32)
        assert ids.shape == (count, self.capacity) and xy.shape == (count, self.capacity, 2)
        self._chk(lib().vsf_batch_upload(self._h, first, count, _vp(ids), _vp(xy), _vp(age), _vp(n)))

    def batch_download(self, first=0, count=None):
        count = self.streams - first if count is None else count
        ids = np.zeros((count, self.capacity), np.uint64)
        xy = np.zeros((count, self.capacity, 2), np.float64)
        age = np.zeros((count, self.capacity), np.uint32)
        n = np.zeros(count, np.int32)
        stats = np.zeros((count, 4), np.int32)
        self._chk(lib().vsf_batch_download(self._h, first, count, _vp(ids), _vp(xy), _vp(age), _vp(n),
                                           _vp(stats)))
        return ids, xy, age, n, stats

    def batch_device_tables(self):
        p = [C.c_void_p() for _ in range(4)]
        cap = C.c_int32()
        self._chk(lib().vsf_batch_device_tables(self._h, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]),
                                                C.byref(p[3]), C.byref(cap)))
        return [x.value for x in p], cap.value

    def batch_export_tracks(self, d_records, stream=0):
        self._chk(lib().vsf_batch_export_tracks(self._h, C.c_void_p(d_records), C.c_void_p(stream)))

    def gather_tracks(self, comm, d_recv, stream=0):
        """vsf_gather_tracks: this rank's records into its slot of d_recv (device address of world x streams x capacity
        x 24 bytes), then one RCCL all-gather in place.  comm: a Comm, or None for a world of one."""
        self._chk(lib().vsf_gather_tracks(self._h, comm._h if comm is not None else None, C.c_void_p(d_recv),
                                          C.c_void_p(stream)))

    def batch_status(self):
        st = np.zeros(self.streams, np.int32)
        self._chk(lib().vsf_batch_status(self._h, _vp(st)))
        return st

    # ---- building blocks for parity tests ----
    def debug_pyramid(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        dims = [(h, w)]
        for _ in range(8):
            hh, ww = dims[-1]
            dims.append(((hh + 1) // 2, (ww + 1) // 2))
        bufs = [np.zeros(d, np.uint8) for d in dims[:8]]
        arr = (C.c_void_p * 8)(*[b.ctypes.data for b in bufs])
        ml = C.c_int32()
        self._chk(lib().vsf_debug_pyramid(self._h, _vp(img), w, h, w, arr, C.byref(ml)))
        return [img] + bufs[1:ml.value + 1]

    def debug_lk(self, prev, nxt, pts):
        prev = np.ascontiguousarray(prev, np.uint8)
        nxt = np.ascontiguousarray(nxt, np.uint8)
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        h, w = prev.shape
        n = len(pts)
        out = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        self._chk(lib().vsf_debug_lk(self._h, _vp(prev), _vp(nxt), w, h, _vp(pts), _vp(out), _vp(st), n))
        return out, st

    def debug_min_eig(self, img, cell):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros(h * w, np.float32)
        rect = (C.c_int32 * 4)()
        self._chk(lib().vsf_debug_min_eig(self._h, _vp(img), w, h, w, cell, _vp(out), rect))
        rx, ry, rw, rh = [int(v) for v in rect]
        return out[:rw * rh].reshape(rh, rw).copy(), (rx, ry, rw, rh)

    def debug_set_cand_cap(self, slots):
        self._chk(lib().vsf_debug_set_cand_cap(self._h, int(slots)))

    def debug_lssd(self, prev, nxt, pts):
        prev = np.ascontiguousarray(prev, np.uint8)
        nxt = np.ascontiguousarray(nxt, np.uint8)
        h, w = prev.shape
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        out = np.zeros_like(pts)
        ok = np.zeros(len(pts), np.uint8)
        self._chk(lib().vsf_debug_lssd(self._h, _vp(prev), _vp(nxt), w, h, _vp(pts), _vp(out), _vp(ok), len(pts)))
        return out, ok

    def debug_lssd_inverse3(self, H):
        H = np.ascontiguousarray(H, np.float32).reshape(-1, 3, 3)
        out = np.zeros_like(H)
        self._chk(lib().vsf_debug_lssd_inverse3(self._h, _vp(H), _vp(out), len(H)))
        return out

    def debug_lssd_pyramid(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = []
        for l in range(1, self.params.lssd_levels + 1):
            lv = np.zeros((h >> l, w >> l), np.uint16)
            self._chk(lib().vsf_debug_lssd_pyramid(self._h, _vp(img), w, h, w, l, _vp(lv)))
            out.append(lv)
        return out

    # ---- frontend step around the path ----
    def frontend_post_track(self, cam, kp, dt_since_keyframe, d_bearings, stream=0):
        """KimeraFrontend.cpp:247-304 on every stream: bearings into the device buffer `d_bearings`
        (streams*capacity*3 doubles), returns (is_keyframe [B], aged [B]); ages of keyframe streams
        are incremented on the device."""
        dt = np.ascontiguousarray(dt_since_keyframe, np.float64)
        assert dt.shape == (self.streams,)
        kf = np.zeros(self.streams, np.int32)
        aged = np.zeros(self.streams, np.uint32)
        self._chk(lib().vsf_batch_frontend_post_track(self._h, C.byref(cam), C.byref(kp), _vp(dt),
                                                      C.c_void_p(d_bearings), _vp(kf), _vp(aged),
                                                      C.c_void_p(stream)))
        return kf, aged

    def unproject(self, cam, xy):
        xy = np.ascontiguousarray(xy, np.float64).reshape(-1, 2)
        out = np.zeros((len(xy), 3), np.float64)
        ok = np.zeros(len(xy), np.uint8)
        self._chk(lib().vsf_unproject(self._h, C.byref(cam), _vp(xy), len(xy), _vp(out), _vp(ok)))
        return out, ok.astype(bool)

    def debug_anms(self, stream=0):
        """(mask GFTT saw [h][w], detected keypoints [n][2] in cv order) of the last ANMS detection."""
        i = self.info()
        w, h = i.width, i.height
        flag = np.zeros((h, w), np.uint8)
        kp = np.zeros((8192, 2), np.int16)
        n = C.c_int32(0)
        self._chk(lib().vsf_debug_anms(self._h, stream, _vp(flag), _vp(kp), C.byref(n)))
        return flag, kp[:n.value].copy()

    # ---- profiling ----
    def sclk_probe(self, spin_us=20000):
        """(shader clock MHz, constant counter MHz) measured over spin_us on a stream of its own (vsf_debug_sclk_probe)."""
        a, b = C.c_double(), C.c_double()
        self._chk(lib().vsf_debug_sclk_probe(self._h, int(spin_us), C.byref(a), C.byref(b)))
        return a.value, b.value

    def profile_enable(self, on=True, kernels=None):
        """kernels: indices of the kernel classes to time (K_* below); None = all of them."""
        v = 0 if not on else (1 if kernels is None else sum(1 << (1 + k) for k in kernels))
        self._chk(lib().vsf_profile_enable(self._h, v))

    def profile_read(self):
        ms = (C.c_float * len(KERNEL_NAMES))()
        cnt = (C.c_int32 * len(KERNEL_NAMES))()
        self._chk(lib().vsf_profile_read(self._h, ms, cnt))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(KERNEL_NAMES)}


COMM_ID_BYTES = 128


def comm_unique_id():
    """ncclGetUniqueId through the library (rank 0); hand the bytes to the other ranks by any means."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = lib().vsf_comm_unique_id(buf)
    if rc != VSF_OK:
        raise VsfError(rc, lib().vsf_comm_last_error().decode())
    return bytes(buf)


class Comm:
    """Owner of one vsf_comm (an RCCL communicator; none at all for a world of one)."""

    def __init__(self, unique_id, rank, nranks, device_id=0):
        self._h = C.c_void_p()
        idbuf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id) if unique_id is not None else None
        rc = lib().vsf_comm_create(idbuf, rank, nranks, device_id, C.byref(self._h))
        if rc != VSF_OK:
            self._h = None
            raise VsfError(rc, lib().vsf_comm_last_error().decode())
        self.rank, self.nranks = rank, nranks

    def close(self):
        if getattr(self, "_h", None):
            lib().vsf_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def lk_points_noalloc(params, d_prev, prev_stride, d_next, next_stride, d_prev_pts, d_next_pts, d_status, n, stream=0):
    """vsf_lk_points_noalloc: every pointer is a raw DEVICE address; allocates nothing (include/vsf.h)."""
    rc = lib().vsf_lk_points_noalloc(C.byref(params), C.c_void_p(d_prev), prev_stride, C.c_void_p(d_next), next_stride,
                                     C.c_void_p(d_prev_pts), C.c_void_p(d_next_pts), C.c_void_p(d_status), n,
                                     C.c_void_p(stream))
    if rc != VSF_OK:
        raise VsfError(rc, "vsf_lk_points_noalloc failed")


class FeatureDatabase:
    """FeaturePositionDatabase + FeatureAgeDatabase (ref: src/vslam/type_defines.hpp:95-99) as the
    SoA image that crosses the C ABI."""

    def __init__(self, capacity):
        self.capacity = capacity
        self.ids = np.zeros(capacity, np.uint64)
        self.xy = np.zeros((capacity, 2), np.float64)
        self.age = np.zeros(capacity, np.uint32)
        self.n = 0

    def set(self, ids, xy, age):
        n = len(ids)
        assert n <= self.capacity
        self.ids[:n], self.xy[:n], self.age[:n] = ids, xy, age
        self.n = n

    def as_dict(self):
        return {int(i): (float(p[0]), float(p[1]), int(a))
                for i, p, a in zip(self.ids[:self.n], self.xy[:self.n], self.age[:self.n])}


class FeatureDetectorShiTomasiBucketing:
    """Mirror of vslam::feature::FeatureDetectorShiTomasiBucketing
    (ref: FeatureDetectorShiTomasiBucketing.hpp:30-59): same ctor arguments, UpdateDetection."""

    def __init__(self, division_number_row, division_number_col, max_feature_number,
                 min_feature_distance, width, height, device_id=0, capacity=0):
        p = default_params(grid_rows=division_number_row, grid_cols=division_number_col,
                           max_features=max_feature_number, min_distance=min_feature_distance,
                           width=width, height=height, device_id=device_id, capacity=capacity)
        self.ctx = Context(p)

    def UpdateDetection(self, db, current_image, mask_image=None):
        db.n = self.ctx.detect(current_image, db.ids, db.xy, db.age, db.n, mask_image)

    @property
    def max_feature_index_(self):
        return self.ctx.get_max_feature_index()


class FeatureDetectorANMS:
    """Mirror of vslam::feature::FeatureDetectorANMS (ref: FeatureDetectorANMS.hpp / .cpp:13-67):
    Parameter fields by their reference names, UpdateDetection."""

    class Parameter:
        def __init__(self):  # FeatureDetectorANMS.cpp:13-35
            self.max_feature_number_ = 300
            self.min_feature_distance_ = 10.0
            self.detection_feature_number_ = self.max_feature_number_ * 2
            self.detection_min_feature_distance_ = 1.0
            self.detection_quality_level_ = 0.001
            self.detection_block_size_ = 4
            self.detection_use_harris_corner_ = False
            self.detection_k_ = 0.04
            self.detection_ignore_edge_ = 20
            self.non_max_suppression_tolerance_ = 0.1

    def __init__(self, parameter, width, height, device_id=0, capacity=0):
        p = default_params(detector_kind=DETECTOR_ANMS, max_features=parameter.max_feature_number_,
                           min_distance=parameter.min_feature_distance_,
                           anms_detection_features=parameter.detection_feature_number_,
                           anms_quality_level=parameter.detection_quality_level_,
                           anms_block_size=parameter.detection_block_size_,
                           anms_ignore_edge=parameter.detection_ignore_edge_,
                           anms_tolerance=parameter.non_max_suppression_tolerance_,
                           anms_detection_min_distance=parameter.detection_min_feature_distance_,
                           anms_use_harris=1 if parameter.detection_use_harris_corner_ else 0,
                           anms_harris_k=parameter.detection_k_,
                           width=width, height=height, device_id=device_id, capacity=capacity)
        self.parameter_ = parameter
        self.ctx = Context(p)

    def UpdateDetection(self, db, current_image, mask_image=None):
        db.n = self.ctx.detect(current_image, db.ids, db.xy, db.age, db.n, mask_image)

    @property
    def max_feature_index_(self):
        return self.ctx.get_max_feature_index()


class FeatureTrackerLSSDLucasKanade:
    """Mirror of vslam::feature::FeatureTrackerLSSDLucasKanade (ref: FeatureTrackerLSSDLucasKanade.cpp:13-119):
    Parameter fields by their reference names, 3-argument Track."""

    class Parameter:
        def __init__(self):  # FeatureTrackerLSSDLucasKanade.cpp:13-23
            self.optical_flow_type_ = "frame_to_frame"
            self.optical_flow_detection_grid_size_ = 32
            self.optical_flow_max_recovered_dist_ = 0.09
            self.optical_flow_pattern_ = 51  # unused by the frame-to-frame class, which fixes Pattern24
            self.optical_flow_max_iterations_ = 5
            self.optical_flow_levels_ = 3

    def __init__(self, parameter, width, height, device_id=0, capacity=1024):
        p = default_params(tracker_kind=TRACKER_LSSD, lssd_levels=parameter.optical_flow_levels_,
                           lssd_max_iterations=parameter.optical_flow_max_iterations_,
                           lssd_max_recovered_dist=parameter.optical_flow_max_recovered_dist_,
                           width=width, height=height, device_id=device_id, capacity=capacity)
        self.parameter_ = parameter
        self.ctx = Context(p)
        self.last_stats = None

    def Track(self, db, current_frame):
        db.n, self.last_stats = self.ctx.track(current_frame, db.ids, db.xy, db.age, db.n)


class FeatureTrackerLucasKanade:
    """Mirror of vslam::feature::FeatureTrackerLucasKanade
    (ref: FeatureTrackerLucasKanade.hpp:17-48): same ctor arguments, 3- and 4-argument Track."""

    def __init__(self, klt_max_iteration, klt_epsilon, klt_window_size, klt_max_level,
                 backtrack_distance_threshold, width, height, device_id=0, capacity=1024):
        p = default_params(klt_max_iter=klt_max_iteration, klt_eps=klt_epsilon,
                           klt_win=klt_window_size, klt_max_level=klt_max_level,
                           backtrack_thr=backtrack_distance_threshold, width=width, height=height,
                           device_id=device_id, capacity=capacity)
        self.ctx = Context(p)
        self.last_stats = None

    def Track(self, db, *images):
        if len(images) == 1:
            db.n, self.last_stats = self.ctx.track(images[0], db.ids, db.xy, db.age, db.n)
        else:
            db.n, self.last_stats = self.ctx.track_pair(images[0], images[1], db.ids, db.xy, db.age,
                                                        db.n)
