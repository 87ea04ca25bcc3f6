"""Host mirror of vslam::dataprovider::EurocKimeraDataProvider
(ref: src/vslam/io/EurocKimeraDataProvider.hpp:14-72, .cpp:21-228) over libvsf_io.so (include/vsf_io.h).

Same constructor arguments and calls (GetInput(), GetInput(index), GetSensorPose()); GetInput returns
None where the reference returns std::nullopt.  The decoded frame is a numpy uint8 array that goes
straight into FeatureTracker*.Track / FeatureDetector*.UpdateDetection or into a batch for
Context.batch_step.  The library is required: there is no Python fallback decoder."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

IO_OK, IO_E_INVALID, IO_E_FILE, IO_E_FORMAT, IO_E_UNSUPPORTED, IO_E_RANGE = 0, -1, -2, -3, -4, -5

IO_ABI_SYMBOLS = [
    "vsf_io_last_error", "vsf_png_info", "vsf_png_decode_gray8", "vsf_png_read_gray8", "vsf_euroc_open",
    "vsf_euroc_close", "vsf_euroc_size", "vsf_euroc_entry", "vsf_euroc_frame_size", "vsf_euroc_frame",
    "vsf_euroc_camera_model", "vsf_euroc_mask", "vsf_euroc_sensor_pose",
]


class VsfIoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vsf_io error %d: %s" % (code, msg))
        self.code = code


class EurocCamera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("xi", C.c_double), ("alpha", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


def io_lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "lib", "libvsf_io.so")
        if not os.path.exists(path):
            raise ImportError("libvsf_io.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        L.vsf_io_last_error.restype = C.c_char_p
        for name in IO_ABI_SYMBOLS[1:]:
            getattr(L, name).restype = C.c_int
        L.vsf_euroc_close.restype = None
        L.vsf_png_info.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int32)] * 4
        L.vsf_png_decode_gray8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32]
        L.vsf_png_read_gray8.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.vsf_euroc_open.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.vsf_euroc_close.argtypes = [C.c_void_p]
        L.vsf_euroc_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.vsf_euroc_entry.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_char_p)]
        L.vsf_euroc_frame_size.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.vsf_euroc_frame.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.c_void_p, C.c_int32,
                                      C.c_int32, C.c_int32]
        L.vsf_euroc_camera_model.argtypes = [C.c_void_p, C.POINTER(EurocCamera)]
        L.vsf_euroc_mask.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 3 + [C.c_void_p, C.c_int32]
        L.vsf_euroc_sensor_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _chk(rc):
    if rc != IO_OK:
        raise VsfIoError(rc, io_lib().vsf_io_last_error().decode())


def png_info(data):
    buf = np.frombuffer(data, np.uint8)
    w, h, d, c = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    _chk(io_lib().vsf_png_info(buf.ctypes.data, len(buf), C.byref(w), C.byref(h), C.byref(d), C.byref(c)))
    return w.value, h.value, d.value, c.value


def png_decode_gray8(data):
    """cv::imread(.., IMREAD_GRAYSCALE) of a grayscale PNG held in memory -> uint8 [h][w]."""
    buf = np.frombuffer(data, np.uint8)
    w, h, _, _ = png_info(data)
    out = np.empty((h, w), np.uint8)
    _chk(io_lib().vsf_png_decode_gray8(buf.ctypes.data, len(buf), out.ctypes.data, w))
    return out


class KimeraFrontendInput:
    """ref: src/vslam/frontend/KimeraFrontend.hpp (KimeraFrontendInput): timestamp_, frame_, mask_, camera_model_."""

    def __init__(self, timestamp, frame, mask, camera_model):
        self.timestamp_ = timestamp
        self.frame_ = frame
        self.mask_ = mask
        self.camera_model_ = camera_model


class EurocKimeraDataProvider:
    def __init__(self, path_to_dataset_root, path_to_calibration_file, path_to_mask_image=""):
        self._h = C.c_void_p()
        _chk(io_lib().vsf_euroc_open(os.fsencode(path_to_dataset_root), os.fsencode(path_to_calibration_file),
                                     os.fsencode(path_to_mask_image), C.byref(self._h)))
        self.last_index_ = 0
        n = C.c_uint64()
        _chk(io_lib().vsf_euroc_size(self._h, C.byref(n)))
        self.size = n.value
        cam = EurocCamera()
        _chk(io_lib().vsf_euroc_camera_model(self._h, C.byref(cam)))
        self.camera_model_ = cam
        has, mw, mh = C.c_int32(), C.c_int32(), C.c_int32()
        _chk(io_lib().vsf_euroc_mask(self._h, C.byref(has), C.byref(mw), C.byref(mh), None, 0))
        self.mask_image_ = None
        if has.value:
            self.mask_image_ = np.empty((mh.value, mw.value), np.uint8)
            _chk(io_lib().vsf_euroc_mask(self._h, C.byref(has), C.byref(mw), C.byref(mh),
                                         self.mask_image_.ctypes.data, mw.value))
        self._wh = None

    def close(self):
        if getattr(self, "_h", None):
            io_lib().vsf_euroc_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def frame_size(self):
        if self._wh is None:
            w, h = C.c_int32(), C.c_int32()
            _chk(io_lib().vsf_euroc_frame_size(self._h, C.byref(w), C.byref(h)))
            self._wh = (w.value, h.value)
        return self._wh

    def entry(self, index):
        ts, fn = C.c_uint64(), C.c_char_p()
        _chk(io_lib().vsf_euroc_entry(self._h, index, C.byref(ts), C.byref(fn)))
        return ts.value, fn.value.decode()

    def GetInput(self, index=None):
        """GetInput() advances last_index_ (.cpp:43-68); GetInput(index) does not (.cpp:70-91)."""
        advance = index is None
        if advance:
            index = self.last_index_
        if index >= self.size:
            return None
        w, h = self.frame_size()
        frame = np.empty((h, w), np.uint8)
        t = C.c_double()
        _chk(io_lib().vsf_euroc_frame(self._h, index, C.byref(t), frame.ctypes.data, w, w, h))
        if advance:
            self.last_index_ += 1
        return KimeraFrontendInput(t.value, frame, self.mask_image_, self.camera_model_)

    def GetSensorPose(self):
        R, t = np.zeros(9), np.zeros(3)
        _chk(io_lib().vsf_euroc_sensor_pose(self._h, R.ctypes.data, t.ctypes.data))
        return R.reshape(3, 3), t
