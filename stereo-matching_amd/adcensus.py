"""Host-side mirror of the reference matcher interface on top of the C-ABI (include/adcensus.h).

`ADCensusStereo` carries the reference's three methods -- Initialize / Match / Reset
(reference ADCensusStereo.h:20,28,36) -- with the same argument meaning and the same error
behaviour (they return False where the reference returns false). Everything runs in
lib/libadcensus_hip.so; there is no CPU fallback and importing this module fails loudly when the
library has not been built.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADC_LIB_PATH") or os.path.join(_HERE, "lib", "libadcensus_hip.so")  # override: A/B builds

ADC_OK, ADC_ERR_ARG, ADC_ERR_NOT_INIT, ADC_ERR_HIP, ADC_ERR_UNSUPPORTED, ADC_ERR_NOMEM = range(6)


class ADCensusOption(C.Structure):
    """Field-for-field ADCensusOption (reference adcensus_types.h:64-96); defaults from :86-93."""
    _fields_ = [
        ("min_disparity", C.c_int32), ("max_disparity", C.c_int32),
        ("lambda_ad", C.c_int32), ("lambda_census", C.c_int32),
        ("cross_L1", C.c_int32), ("cross_L2", C.c_int32),
        ("cross_t1", C.c_int32), ("cross_t2", C.c_int32),
        ("so_p1", C.c_float), ("so_p2", C.c_float),
        ("so_tso", C.c_int32), ("irv_ts", C.c_int32),
        ("irv_th", C.c_float), ("lrcheck_thres", C.c_float),
        ("do_lr_check", C.c_uint8), ("do_filling", C.c_uint8),
        ("do_discontinuity_adjustment", C.c_uint8), ("reserved_", C.c_uint8),
    ]

    def __init__(self, **kw):
        super().__init__()
        lib().adc_default_option(C.byref(self))
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


class BatchDesc(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("option", ADCensusOption),
        ("n_pairs", C.c_int32),
        ("left_bgr", C.POINTER(C.c_void_p)),
        ("right_bgr", C.POINTER(C.c_void_p)),
        ("disp_left", C.POINTER(C.c_void_p)),
        ("n_devices", C.c_int32),
        ("devices", C.POINTER(C.c_int32)),
        ("handles", C.POINTER(C.c_void_p)),
    ]


# adc_buffer_id (include/adcensus.h) -> (numpy dtype, shape kind)
BUFFERS = {
    "gray_left": (0, np.uint8, "px"), "gray_right": (1, np.uint8, "px"),
    "census_left": (2, np.uint64, "px"), "census_right": (3, np.uint64, "px"),
    "cost_init": (4, np.float32, "vol"),
    "arms": (5, np.uint8, "px4"),
    "sup_count_h": (6, np.uint16, "px"), "sup_count_v": (7, np.uint16, "px"),
    "cost_aggr": (8, np.float32, "vol"),
    "cost_scan": (9, np.float32, "vol"), "cost_scan_aux": (10, np.float32, "vol"),
    "disp_left_wta": (11, np.float32, "px"), "disp_right_wta": (12, np.float32, "px"),
    "disp_outlier": (13, np.float32, "px"),
    "mask_occ": (14, np.uint8, "px"), "mask_mis": (15, np.uint8, "px"),
    "disp_voting": (16, np.float32, "px"),
    "mask_occ_voting": (17, np.uint8, "px"), "mask_mis_voting": (18, np.uint8, "px"),
    "disp_interp": (19, np.float32, "px"),
    "edge_mask": (20, np.uint8, "px"),
    "disp_edge": (21, np.float32, "px"),
    "disp_final": (22, np.float32, "px"),
}
STAGES = ["pre", "cost", "aggr", "scan", "wta", "refine", "total"]
# adc_kernel_id: the volume kernels of one match in launch order
KERNELS = ["cost"] + ["aggr%d" % i for i in range(8)] + ["scan%d" % i for i in range(4)] + ["wta_right"]
STATIC_SCRATCH_BYTES = 128 * 1024 * 1024

# every symbol include/adcensus.h declares
EXPORTS = [
    "adc_default_option", "adc_create", "adc_destroy", "adc_initialize", "adc_reset", "adc_match",
    "adc_match_device", "adc_synchronize", "adc_set_debug_taps", "adc_set_profiling", "adc_stage_times",
    "adc_aggr_kernel_stats", "adc_buffer_bytes", "adc_get_buffer", "adc_match_batch", "adc_last_error",
    "adc_kernel_times", "adc_transfer_times", "adc_match_device_static",
]
# ... and include/adcensus_harness.h
HARNESS_EXPORTS = ["adc_png_read", "adc_image_free", "adc_png_write_gray", "adc_disparity_to_u8", "adc_eval_0103",
                   "adc_pfm_read", "adc_float_free", "adc_pfm_write", "adc_eval_pfm", "adc_disparity_to_depth",
                   "adc_calib_read"]

_lib = None


def lib():
    """Load libadcensus_hip.so. Raises if it is missing: there is no other implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `make -C %s` (hipcc, gfx950). "
                "There is no CPU fallback." % (LIB_PATH, _HERE))
        L = C.CDLL(LIB_PATH)
        L.adc_create.restype = C.c_void_p
        L.adc_create.argtypes = [C.c_int]
        L.adc_destroy.restype = None
        L.adc_destroy.argtypes = [C.c_void_p]
        L.adc_default_option.restype = None
        L.adc_initialize.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.adc_reset.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.adc_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.adc_match_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.adc_synchronize.argtypes = [C.c_void_p]
        L.adc_set_debug_taps.argtypes = [C.c_void_p, C.c_int]
        L.adc_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.adc_stage_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.adc_aggr_kernel_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.adc_kernel_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.adc_transfer_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.adc_match_device_static.argtypes = [C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p]
        L.adc_buffer_bytes.restype = C.c_size_t
        L.adc_buffer_bytes.argtypes = [C.c_void_p, C.c_int]
        L.adc_get_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.adc_match_batch.argtypes = [C.c_void_p]
        L.adc_last_error.restype = C.c_char_p
        L.adc_last_error.argtypes = [C.c_void_p]
        L.adc_png_read.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.adc_image_free.restype = None
        L.adc_image_free.argtypes = [C.c_void_p]
        L.adc_png_write_gray.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.adc_disparity_to_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.adc_eval_0103.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_double,
                                    C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.adc_pfm_read.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.adc_float_free.restype = None
        L.adc_float_free.argtypes = [C.c_void_p]
        L.adc_pfm_write.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.adc_eval_pfm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_double,
                                   C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.adc_disparity_to_depth.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                             C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.adc_calib_read.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib = L
    return _lib


class ADCensusStereo:
    """The reference's matcher class (ADCensusStereo.h:9-90) backed by the HIP library."""

    def __init__(self, device=0):
        self._h = lib().adc_create(int(device))
        if not self._h:
            raise MemoryError("adc_create failed")
        self.width = self.height = 0
        self.option = None
        self.status = ADC_OK

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        if getattr(self, "_h", None):
            lib().adc_destroy(self._h)
            self._h = None

    # -- reference interface ----------------------------------------------------------------
    def Initialize(self, width, height, option):
        self.status = lib().adc_initialize(self._h, int(width), int(height), C.byref(option))
        if self.status == ADC_OK:
            self.width, self.height, self.option = int(width), int(height), option
        return self.status == ADC_OK

    def Reset(self, width, height, option):
        self.status = lib().adc_reset(self._h, int(width), int(height), C.byref(option))
        if self.status == ADC_OK:
            self.width, self.height, self.option = int(width), int(height), option
        return self.status == ADC_OK

    def Match(self, img_left, img_right, disp_left):
        """img_left/img_right: BGR u8[H][W][3] numpy arrays (or None); disp_left: f32[H][W] output."""
        try:
            pl = self._host_ptr(img_left, np.uint8, 3)
            pr = self._host_ptr(img_right, np.uint8, 3)
            pd = self._host_ptr(disp_left, np.float32, 1)
        except ValueError:
            # the library copies width*height*3 bytes from each image and width*height floats to the map,
            # whatever the arrays hold: a wrong shape is refused here like the reference's `return false`
            self.status = ADC_ERR_ARG
            return False
        self.status = lib().adc_match(self._h, pl, pr, pd)
        return self.status == ADC_OK

    # -- conveniences -------------------------------------------------------------------------
    def match(self, img_left, img_right):
        disp = np.empty((self.height, self.width), np.float32)
        if not self.Match(np.ascontiguousarray(img_left, np.uint8), np.ascontiguousarray(img_right, np.uint8), disp):
            raise RuntimeError("Match failed (%d): %s" % (self.status, self.last_error()))
        return disp

    @staticmethod
    def match_device_static(device, width, height, option, d_left_ptr, d_right_ptr, d_disp_ptr, stream=None):
        """adc_match_device_static: no handle, nothing allocated; device pointers (ints), asynchronous on `stream`."""
        rc = lib().adc_match_device_static(int(device), int(width), int(height), C.byref(option), d_left_ptr, d_right_ptr,
                                           d_disp_ptr, stream)
        if rc != ADC_OK:
            raise RuntimeError("adc_match_device_static failed (%d): %s" % (rc, lib().adc_last_error(None).decode()))

    def match_device(self, d_left_ptr, d_right_ptr, d_disp_ptr, stream=None):
        """Device pointers (ints) in, asynchronous on `stream` (int hipStream_t or None)."""
        self.status = lib().adc_match_device(self._h, d_left_ptr, d_right_ptr, d_disp_ptr, stream)
        if self.status != ADC_OK:
            raise RuntimeError("adc_match_device failed (%d): %s" % (self.status, self.last_error()))

    def synchronize(self):
        lib().adc_synchronize(self._h)

    def set_debug_taps(self, enable=True):
        rc = lib().adc_set_debug_taps(self._h, int(bool(enable)))
        if rc != ADC_OK:
            raise RuntimeError("adc_set_debug_taps failed: %s" % self.last_error())

    def set_profiling(self, enable=True):
        lib().adc_set_profiling(self._h, int(bool(enable)))

    def stage_times(self):
        ms = (C.c_float * len(STAGES))()
        rc = lib().adc_stage_times(self._h, ms, len(STAGES))
        if rc != ADC_OK:
            raise RuntimeError("adc_stage_times failed: %s" % self.last_error())
        return dict(zip(STAGES, [float(v) for v in ms]))

    def aggr_kernel_stats(self):
        n = C.c_int(0)
        t = C.c_float(0)
        rc = lib().adc_aggr_kernel_stats(self._h, C.byref(n), C.byref(t))
        if rc != ADC_OK:
            raise RuntimeError("adc_aggr_kernel_stats failed: %s" % self.last_error())
        return n.value, t.value

    def kernel_times(self):
        """Per-launch device times (ms) of the last profiled match, by KERNELS name."""
        ms = (C.c_float * len(KERNELS))()
        rc = lib().adc_kernel_times(self._h, ms, len(KERNELS))
        if rc != ADC_OK:
            raise RuntimeError("adc_kernel_times failed: %s" % self.last_error())
        return dict(zip(KERNELS, [float(v) for v in ms]))

    def transfer_times(self):
        """Of the last profiled Match: host copy in, upload, download, host copy out (ms)."""
        ms = (C.c_float * 4)()
        rc = lib().adc_transfer_times(self._h, ms, 4)
        if rc != ADC_OK:
            raise RuntimeError("adc_transfer_times failed: %s" % self.last_error())
        return dict(zip(["host_in", "h2d", "d2h", "host_out"], [float(v) for v in ms]))

    def get_buffer(self, name):
        bid, dt, kind = BUFFERS[name]
        D = self.option.max_disparity - self.option.min_disparity
        shape = {"px": (self.height, self.width), "px4": (self.height, self.width, 4),
                 "vol": (self.height, self.width, D)}[kind]
        out = np.empty(shape, dt)
        need = lib().adc_buffer_bytes(self._h, bid)
        assert need == out.nbytes, (name, need, out.nbytes)
        rc = lib().adc_get_buffer(self._h, bid, out.ctypes.data, out.nbytes)
        if rc != ADC_OK:
            raise RuntimeError("adc_get_buffer(%s) failed (%d): %s" % (name, rc, self.last_error()))
        return out

    def last_error(self):
        return lib().adc_last_error(self._h).decode()

    def _host_ptr(self, a, dtype, ch):
        """Address of a contiguous array of the initialised frame's shape ((H, W, 3) images, (H, W) map)."""
        if a is None:
            return None
        want = (self.height, self.width, 3) if ch == 3 else (self.height, self.width)
        if not (isinstance(a, np.ndarray) and a.dtype == dtype and a.flags["C_CONTIGUOUS"]):
            raise ValueError("contiguous %s array expected" % np.dtype(dtype).name)
        if self.width and a.shape != want:
            raise ValueError("array of shape %s expected, got %s" % (want, a.shape))
        return a.ctypes.data


def match_batch(lefts, rights, option, devices, handles=None, out=None):
    """adc_match_batch: the pairs are spread over `devices` (each device takes the next unmatched pair when it
    is free). handles: ADCensusStereo objects, one per device, already initialised for this size and option,
    reused instead of a create/initialise/destroy per call. out: preallocated f32[H][W] arrays to fill.
    Returns the list of disparity maps."""
    n = len(lefts)
    if n == 0:
        return []
    if len(rights) != n:
        raise ValueError("as many right images as left images expected")
    H, W = lefts[0].shape[:2]
    for a in list(lefts) + list(rights):
        if a.shape != (H, W, 3):
            raise ValueError("every image must have shape %s, got %s" % ((H, W, 3), a.shape))
    lefts = [np.ascontiguousarray(a, np.uint8) for a in lefts]
    rights = [np.ascontiguousarray(a, np.uint8) for a in rights]
    disps = out if out is not None else [np.empty((H, W), np.float32) for _ in range(n)]
    if len(disps) != n or any(a.shape != (H, W) or a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"] for a in disps):
        raise ValueError("out must hold one contiguous f32 (H, W) array per pair")
    d = BatchDesc()
    d.width, d.height, d.option, d.n_pairs = W, H, option, n
    d.left_bgr = (C.c_void_p * n)(*[a.ctypes.data for a in lefts])
    d.right_bgr = (C.c_void_p * n)(*[a.ctypes.data for a in rights])
    d.disp_left = (C.c_void_p * n)(*[a.ctypes.data for a in disps])
    d.n_devices = len(devices)
    d.devices = (C.c_int32 * len(devices))(*devices)
    if handles is not None:
        if len(handles) != len(devices):
            raise ValueError("one handle per device expected")
        d.handles = (C.c_void_p * len(handles))(*[h._h for h in handles])
    rc = lib().adc_match_batch(C.byref(d))
    if rc != ADC_OK:
        raise RuntimeError("adc_match_batch failed (%d): %s" % (rc, lib().adc_last_error(None).decode()))
    return disps


# -- harness (include/adcensus_harness.h): what the reference's caller does around Match() --------------
def png_read(path, channels=3):
    """8-bit PNG -> BGR u8[H][W][3] (channels=3, what cv::imread hands the reference) or gray u8[H][W]."""
    p, w, h = C.c_void_p(), C.c_int(), C.c_int()
    rc = lib().adc_png_read(os.fsencode(path), channels, C.byref(p), C.byref(w), C.byref(h))
    if rc != 0:
        raise IOError("adc_png_read(%s) failed (%d)" % (path, rc))
    n = w.value * h.value * channels
    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).copy()
    lib().adc_image_free(p)
    return a.reshape((h.value, w.value, 3) if channels == 3 else (h.value, w.value))


def png_write_gray(path, gray):
    g = np.ascontiguousarray(gray, np.uint8)
    rc = lib().adc_png_write_gray(os.fsencode(path), g.ctypes.data, g.shape[1], g.shape[0])
    if rc != 0:
        raise IOError("adc_png_write_gray(%s) failed (%d)" % (path, rc))


def disparity_to_u8(disp):
    """The reference's SaveDisparityMap normalisation (main.cpp:443-468)."""
    d = np.ascontiguousarray(disp, np.float32)
    out = np.empty(d.shape, np.uint8)
    rc = lib().adc_disparity_to_u8(d.ctypes.data, d.shape[1], d.shape[0], out.ctypes.data)
    assert rc == 0
    return out


def eval_0103(disp, gt_u8, gt_div=4.0, delta=2.0, seconds=1.0):
    """The reference's PerformanceEval0103 (performance_eval.cpp:218-253): (RMS, PEP, Mde)."""
    d = np.ascontiguousarray(disp, np.float32)
    g = np.ascontiguousarray(gt_u8, np.uint8)
    assert d.shape == g.shape
    rms, pep, mde = C.c_float(), C.c_float(), C.c_float()
    rc = lib().adc_eval_0103(d.ctypes.data, g.ctypes.data, d.shape[1], d.shape[0], gt_div, delta, seconds,
                             C.byref(rms), C.byref(pep), C.byref(mde))
    assert rc == 0
    return rms.value, pep.value, mde.value


def pfm_read(path):
    """Single-channel little-endian PFM -> f32[H][W], top row first (LoadPFM, performance_eval.cpp:11-91)."""
    p, w, h = C.c_void_p(), C.c_int(), C.c_int()
    rc = lib().adc_pfm_read(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h))
    if rc != 0:
        raise IOError("adc_pfm_read(%s) failed (%d)" % (path, rc))
    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(h.value, w.value)).copy()
    lib().adc_float_free(p)
    return a


def pfm_write(path, data):
    d = np.ascontiguousarray(data, np.float32)
    rc = lib().adc_pfm_write(os.fsencode(path), d.ctypes.data, d.shape[1], d.shape[0])
    if rc != 0:
        raise IOError("adc_pfm_write(%s) failed (%d)" % (path, rc))


def eval_pfm(disp, gt, delta=2.0, seconds=1.0):
    """The reference's PerformanceEval against a PFM ground truth (performance_eval.cpp:184-215)."""
    d = np.ascontiguousarray(disp, np.float32)
    g = np.ascontiguousarray(gt, np.float32)
    assert d.shape == g.shape
    rms, pep, mde = C.c_float(), C.c_float(), C.c_float()
    rc = lib().adc_eval_pfm(d.ctypes.data, g.ctypes.data, d.shape[1], d.shape[0], delta, seconds,
                            C.byref(rms), C.byref(pep), C.byref(mde))
    assert rc == 0
    return rms.value, pep.value, mde.value


def disparity_to_depth(disp, focal, baseline, doffs=0.0):
    """The reference's Disp2Depth2 (main.cpp:499-551): (depth f32, depth u8, min, max)."""
    d = np.ascontiguousarray(disp, np.float32)
    depth = np.empty(d.shape, np.float32)
    u8 = np.empty(d.shape, np.uint8)
    lo, hi = C.c_float(), C.c_float()
    rc = lib().adc_disparity_to_depth(d.ctypes.data, d.shape[1], d.shape[0], focal, baseline, doffs,
                                      depth.ctypes.data, u8.ctypes.data, C.byref(lo), C.byref(hi))
    assert rc == 0
    return depth, u8, lo.value, hi.value


def calib_read(path):
    f, b, o = C.c_float(), C.c_float(), C.c_float()
    rc = lib().adc_calib_read(os.fsencode(path), C.byref(f), C.byref(b), C.byref(o))
    if rc != 0:
        raise IOError("adc_calib_read(%s) failed (%d)" % (path, rc))
    return f.value, b.value, o.value
