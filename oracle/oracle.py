"""ctypes binding of the CPU oracle (oracle/liboracle_adcensus.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the cpu_baseline
leg of bench.py as the checker / reported baseline. The shipped library never loads it.
Parity unpinned by the reference itself (it has no tests or recorded outputs); see
adcensus_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_adcensus.so")


class Option(C.Structure):
    """Field-for-field ADCensusOption (reference adcensus_types.h:64-96)."""
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


TAP_FIELDS = [
    ("gray_left", np.uint8, "px"), ("gray_right", np.uint8, "px"),
    ("census_left", np.uint64, "px"), ("census_right", np.uint64, "px"),
    ("cost_init", np.float32, "vol"),
    ("arms", np.uint8, "px4"),
    ("sup_count_h", np.uint16, "px"), ("sup_count_v", np.uint16, "px"),
    ("cost_aggr", np.float32, "vol"),
    ("cost_scan", np.float32, "vol"),
    ("cost_scan_aux", np.float32, "vol"),
    ("disp_left_wta", np.float32, "px"), ("disp_right_wta", np.float32, "px"),
    ("disp_outlier", np.float32, "px"),
    ("mask_occ", np.uint8, "px"), ("mask_mis", np.uint8, "px"),
    ("disp_voting", np.float32, "px"),
    ("mask_occ_voting", np.uint8, "px"), ("mask_mis_voting", np.uint8, "px"),
    ("disp_interp", np.float32, "px"),
    ("edge_mask", np.uint8, "px"),
    ("disp_edge", np.float32, "px"),
]


class Taps(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _, _ in TAP_FIELDS]


def build(force=False):
    """Compile the oracle with gcc if the shared object is missing (or force=True)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_match.restype = C.c_int
        _lib.orc_bad_fraction.restype = C.c_double
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_get_literal.restype = C.c_uint
        _lib.orc_literal_max3.restype = C.c_int
        _lib.orc_literal_min4.restype = C.c_float
        _lib.orc_literal_min4.argtypes = [C.c_float] * 4
    return _lib


LIT_HPASS_RETURN, LIT_MIN_MACRO, LIT_MAX_MACRO, LIT_INTERP_RETURN, LIT_LOW_SENTINEL = 1, 2, 4, 8, 16
LIT_ALL = 31


class literal:
    """Context manager: run the enclosed oracle calls with the given literal-defect switches
    (sensitivity study only; parity tests and baselines always run with 0)."""

    def __init__(self, flags):
        self.flags = int(flags)

    def __enter__(self):
        self.prev = lib().orc_get_literal()
        lib().orc_set_literal(C.c_uint(self.flags))
        return self

    def __exit__(self, *exc):
        lib().orc_set_literal(C.c_uint(self.prev))
        return False


def literal_max3(a, b, c):
    return lib().orc_literal_max3(C.c_int(a), C.c_int(b), C.c_int(c))


def literal_min4(l1, l2, l3, l4):
    return lib().orc_literal_min4(l1, l2, l3, l4)


def default_option(**kw):
    o = Option()
    lib().orc_default_option(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 3, "BGR u8[H][W][3] expected"
    return a


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(C.c_int(int(n)))


def match(left_bgr, right_bgr, opt=None, want=()):
    """Run the whole pipeline. Returns (disp f32[H][W], {tap name: array}).

    want: iterable of tap names (see TAP_FIELDS) or "all".
    """
    L, R = _img(left_bgr), _img(right_bgr)
    assert L.shape == R.shape
    H, W = L.shape[:2]
    opt = opt or default_option()
    D = opt.max_disparity - opt.min_disparity
    shapes = {"px": (H, W), "px4": (H, W, 4), "vol": (H, W, max(D, 0))}
    names = [n for n, _, _ in TAP_FIELDS] if want == "all" else list(want)
    taps = Taps()
    out = {}
    for name, dt, kind in TAP_FIELDS:
        if name in names:
            arr = np.empty(shapes[kind], dtype=dt)
            out[name] = arr
            setattr(taps, name, arr.ctypes.data)
    disp = np.empty((H, W), dtype=np.float32)
    rc = lib().orc_match(_p(L), _p(R), C.c_int(W), C.c_int(H), C.byref(opt), _p(disp), C.byref(taps))
    if rc != 0:
        raise RuntimeError("orc_match failed with code %d" % rc)
    return disp, out


# ---- stage-level entry points (used by the known-answer tests) -------------------------------

def gray(bgr):
    a = _img(bgr)
    H, W = a.shape[:2]
    g = np.empty((H, W), np.uint8)
    lib().orc_gray(_p(a), C.c_int(W), C.c_int(H), _p(g))
    return g


def census(gray_img):
    g = np.ascontiguousarray(gray_img, np.uint8)
    H, W = g.shape
    c = np.empty((H, W), np.uint64)
    lib().orc_census(_p(g), C.c_int(W), C.c_int(H), _p(c))
    return c


def cost_init(L, R, cenL, cenR, dmin, dmax, lambda_ad=10, lambda_census=30):
    L, R = _img(L), _img(R)
    H, W = L.shape[:2]
    cenL = np.ascontiguousarray(cenL, np.uint64)
    cenR = np.ascontiguousarray(cenR, np.uint64)
    c = np.empty((H, W, dmax - dmin), np.float32)
    lib().orc_cost_init(_p(L), _p(R), _p(cenL), _p(cenR), C.c_int(W), C.c_int(H), C.c_int(dmin),
                        C.c_int(dmax), C.c_int(lambda_ad), C.c_int(lambda_census), _p(c))
    return c


def build_arms(L, L1=34, L2=17, t1=20, t2=6):
    L = _img(L)
    H, W = L.shape[:2]
    a = np.empty((H, W, 4), np.uint8)
    lib().orc_build_arms(_p(L), C.c_int(W), C.c_int(H), C.c_int(L1), C.c_int(L2), C.c_int(t1),
                         C.c_int(t2), _p(a))
    return a


def sup_counts(arms):
    a = np.ascontiguousarray(arms, np.uint8)
    H, W = a.shape[:2]
    ch = np.empty((H, W), np.uint16)
    cv = np.empty((H, W), np.uint16)
    lib().orc_sup_counts(_p(a), C.c_int(W), C.c_int(H), _p(ch), _p(cv))
    return ch, cv


def aggregate(cost, arms, cnt_h, cnt_v, iters=4):
    c = np.array(cost, dtype=np.float32, order="C", copy=True)
    H, W, D = c.shape
    a = np.ascontiguousarray(arms, np.uint8)
    ch = np.ascontiguousarray(cnt_h, np.uint16)
    cv = np.ascontiguousarray(cnt_v, np.uint16)
    lib().orc_aggregate(_p(c), _p(a), _p(ch), _p(cv), C.c_int(W), C.c_int(H), C.c_int(D), C.c_int(iters))
    return c


def scanline_pass(L, R, src, dmin, dmax, p1=1.0, p2=3.0, tso=15, vertical=False, forward=True):
    L, R = _img(L), _img(R)
    H, W = L.shape[:2]
    s = np.ascontiguousarray(src, np.float32)
    d = np.empty_like(s)
    lib().orc_scanline_pass(_p(L), _p(R), _p(s), _p(d), C.c_int(W), C.c_int(H), C.c_int(dmin),
                            C.c_int(dmax), C.c_float(p1), C.c_float(p2), C.c_int(tso),
                            C.c_int(int(vertical)), C.c_int(int(forward)))
    return d


def wta_left(cost, dmin, dmax):
    c = np.ascontiguousarray(cost, np.float32)
    H, W, _ = c.shape
    d = np.empty((H, W), np.float32)
    lib().orc_wta_left(_p(c), C.c_int(W), C.c_int(H), C.c_int(dmin), C.c_int(dmax), _p(d))
    return d


def wta_right(cost, dmin, dmax):
    c = np.ascontiguousarray(cost, np.float32)
    H, W, _ = c.shape
    d = np.empty((H, W), np.float32)
    lib().orc_wta_right(_p(c), C.c_int(W), C.c_int(H), C.c_int(dmin), C.c_int(dmax), _p(d))
    return d


def outlier(disp_l, disp_r, thres=1.0):
    dl = np.ascontiguousarray(disp_l, np.float32)
    dr = np.ascontiguousarray(disp_r, np.float32)
    H, W = dl.shape
    out = np.empty_like(dl)
    occ = np.empty((H, W), np.uint8)
    mis = np.empty((H, W), np.uint8)
    lib().orc_outlier(_p(dl), _p(dr), C.c_int(W), C.c_int(H), C.c_float(thres), _p(out), _p(occ), _p(mis))
    return out, occ, mis


def region_voting_pass(disp, mask, arms, dmin, dmax, tau_s=20, tau_h=0.4):
    d = np.ascontiguousarray(disp, np.float32)
    m = np.array(mask, dtype=np.uint8, order="C", copy=True)
    a = np.ascontiguousarray(arms, np.uint8)
    H, W = d.shape
    out = np.empty_like(d)
    lib().orc_region_voting_pass(_p(d), _p(out), _p(m), _p(a), C.c_int(W), C.c_int(H), C.c_int(dmin),
                                 C.c_int(dmax), C.c_int(tau_s), C.c_float(tau_h))
    return out, m


def interpolate(L, disp, occ, mis, dmax):
    L = _img(L)
    d = np.ascontiguousarray(disp, np.float32)
    o = np.ascontiguousarray(occ, np.uint8)
    m = np.ascontiguousarray(mis, np.uint8)
    H, W = d.shape
    out = np.empty_like(d)
    lib().orc_interpolate(_p(L), _p(d), _p(out), _p(o), _p(m), C.c_int(W), C.c_int(H), C.c_int(dmax))
    return out


def edge_detect(disp, thres=5.0):
    d = np.ascontiguousarray(disp, np.float32)
    H, W = d.shape
    e = np.empty((H, W), np.uint8)
    lib().orc_edge_detect(_p(d), C.c_int(W), C.c_int(H), C.c_float(thres), _p(e))
    return e


def edge_adjust(disp, cost, edge, dmin, dmax):
    d = np.ascontiguousarray(disp, np.float32)
    c = np.ascontiguousarray(cost, np.float32)
    e = np.ascontiguousarray(edge, np.uint8)
    H, W = d.shape
    out = np.empty_like(d)
    lib().orc_edge_adjust(_p(d), _p(out), _p(c), _p(e), C.c_int(W), C.c_int(H), C.c_int(dmin), C.c_int(dmax))
    return out


def median3(disp):
    d = np.ascontiguousarray(disp, np.float32)
    H, W = d.shape
    out = np.empty_like(d)
    lib().orc_median3(_p(d), _p(out), C.c_int(W), C.c_int(H))
    return out


def bad_fraction(disp, gt_u8, gt_scale=4.0, delta=2.0):
    """bad-delta as the reference's PerformanceEval0103 (performance_eval.cpp:218-253)."""
    d = np.ascontiguousarray(disp, np.float32)
    g = np.ascontiguousarray(gt_u8, np.uint8)
    H, W = d.shape
    rms = C.c_double(0)
    bad = lib().orc_bad_fraction(_p(d), _p(g), C.c_int(W), C.c_int(H), C.c_float(gt_scale),
                                 C.c_float(delta), C.byref(rms))
    return bad, rms.value
