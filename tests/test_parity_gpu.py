"""GPU parity: the HIP path, called through the C-ABI, against the CPU oracle on the same inputs.

Bar (stricter than the 1e-5 / 99.9 % the north star asks): every tap -- integer stages, the fp32
volumes and the final disparity map -- is bit-identical to the oracle, because every kernel keeps
the reference's operation order and the library is built without FMA contraction.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, artroom1, fnv1a64, middlebury2003

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(GOLDEN, "golden.json")))

# pipeline order, so the first failing name is the first stage that went wrong
TAP_ORDER = [
    "gray_left", "gray_right", "census_left", "census_right", "arms", "sup_count_h", "sup_count_v",
    "cost_init", "cost_aggr", "cost_scan_aux", "cost_scan", "disp_left_wta", "disp_right_wta",
    "disp_outlier", "mask_occ", "mask_mis", "disp_voting", "mask_occ_voting", "mask_mis_voting",
    "disp_interp", "edge_mask", "disp_edge", "disp_final",
]


def same(a, b):
    """Bitwise equality that treats +inf == +inf (no NaNs are expected in either)."""
    return a.shape == b.shape and np.array_equal(a, b)


def describe(name, got, want):
    diff = got != want
    n = int(diff.sum())
    idx = np.argwhere(diff)[:5].tolist()
    msg = "%s: %d of %d elements differ; first at %s" % (name, n, diff.size, idx)
    if got.dtype.kind == "f" and n:
        fin = np.isfinite(got) & np.isfinite(want) & diff
        if fin.any():
            rel = np.abs(got[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1e-30)
            msg += "; max rel err %.3e" % rel.max()
        msg += "; got %s want %s" % (got[diff][:4], want[diff][:4])
    return msg


def gpu_match(pkg, L, R, taps=True, **opt):
    H, W = L.shape[:2]
    h = pkg.ADCensusStereo(0)
    if taps:
        h.set_debug_taps(True)
    o = pkg.ADCensusOption(**opt)
    assert h.Initialize(W, H, o), h.last_error()
    disp = h.match(L, R)
    out = {"disp_final": disp}
    if taps:
        for name in TAP_ORDER:
            out[name] = h.get_buffer(name)
        assert same(out["disp_final"], disp)
    h.close()
    return out


def check_against_oracle(pkg, orc, L, R, **opt):
    got = gpu_match(pkg, L, R, **opt)
    disp, want = orc.match(L, R, orc.default_option(**opt), want="all")
    want["disp_final"] = disp
    for name in TAP_ORDER:
        assert same(got[name], want[name]), describe(name, got[name], want[name])
    return got


def test_teddy_crop_golden_and_oracle(pkg, orc):
    z = np.load(os.path.join(GOLDEN, "teddy_crop_64x48x16.npz"))
    got = check_against_oracle(pkg, orc, z["left"], z["right"], max_disparity=16)
    for name in TAP_ORDER:
        assert same(got[name], z[name]), describe(name, got[name], z[name])


@pytest.mark.parametrize("name", ["teddy", "cones"])
def test_middlebury_2003_full(pkg, orc, name):
    L, R, gt = middlebury2003(name)
    got = check_against_oracle(pkg, orc, L, R, max_disparity=64)
    g = GOLD[name]
    for tap in TAP_ORDER:
        assert fnv1a64(got[tap]) == g["hash"][tap], tap
    bad, _ = orc.bad_fraction(got["disp_final"], gt, 4.0, 2.0)
    assert abs(bad - g["bad2"]) <= 0.0005          # within 0.05 pp (it is in fact identical)
    assert bad == g["bad2"]


@pytest.mark.parametrize("seed,W,H,dmin,dmax", [(7, 320, 200, 0, 64), (11, 203, 117, 0, 48), (13, 230, 150, 3, 133)])
def test_synthetic_ragged_sizes(pkg, orc, synth, seed, W, H, dmin, dmax):
    """Widths/heights that are not tile multiples, D not a multiple of 64, min_disparity != 0."""
    L, R, _ = synth.pair(seed, W, H, dmax - dmin)
    got = check_against_oracle(pkg, orc, L, R, min_disparity=dmin, max_disparity=dmax)
    g = GOLD["synth_%d" % seed]
    assert fnv1a64(got["disp_final"]) == g["hash"]["disp_final"]


@pytest.mark.parametrize("opt", [
    dict(max_disparity=40, do_lr_check=0),
    dict(max_disparity=40, do_filling=0),
    dict(max_disparity=40, do_discontinuity_adjustment=0),
    dict(max_disparity=40, do_lr_check=0, do_filling=0, do_discontinuity_adjustment=0),
    dict(max_disparity=24, cross_L1=9, cross_L2=4, cross_t1=30, cross_t2=10, lambda_ad=7, lambda_census=20),
    dict(max_disparity=24, so_p1=0.5, so_p2=2.0, so_tso=8, irv_ts=10, irv_th=0.3, lrcheck_thres=0.5),
    dict(max_disparity=20, cross_L1=60),            # ring of 256 slots
    dict(max_disparity=20, cross_L1=150),           # beyond the LDS ring: direct kernel
    dict(min_disparity=-4, max_disparity=28),
])
def test_option_variants(pkg, orc, synth, opt):
    L, R, _ = synth.pair(21, 160, 96, 32)
    check_against_oracle(pkg, orc, L, R, **opt)


@pytest.mark.parametrize("D", [1, 2, 3, 65, 200, 300])
def test_disparity_counts(pkg, orc, synth, D):
    """1..3 disparities (every WTA result at a range end), and NV = 2, 4, 8 lanes-per-pixel layouts."""
    L, R, _ = synth.pair(22, 150, 40, max(D, 8))
    check_against_oracle(pkg, orc, L, R, max_disparity=D)


@pytest.mark.parametrize("W,H,dmin,dmax", [(10, 9, -5, 251), (3, 5, 0, 256), (2, 4, -3, 253), (70, 20, 7, 263),
                                           (130, 12, -40, 216), (40, 11, -5, 195), (300, 7, 250, 506), (9, 6, -300, -44)])
def test_flag_strings_at_row_ends(pkg, orc, W, H, dmin, dmax):
    """More than 128 disparities: the scanline passes take their colour-step flags from the shifted row strings and
    patch the disparities whose flag is the pixel's own. Narrow frames and shifted disparity ranges make that most
    or all of a row (no usable right-image column at all, ranges far off the image on either side)."""
    rng = np.random.default_rng(1000 + W)
    L = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
    R = np.roll(L, -3, axis=1)
    R[::2] = rng.integers(0, 256, size=R[::2].shape).astype(np.uint8)
    check_against_oracle(pkg, orc, L, R, min_disparity=dmin, max_disparity=dmax)


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_configurations(pkg, orc, seed):
    """Seeded random frame sizes, disparity ranges and options (arm lengths and thresholds, penalties, refinement
    switches): every tap against the oracle."""
    rng = np.random.default_rng(7000 + seed)
    W, H = int(rng.integers(1, 180)), int(rng.integers(1, 120))
    D = int(rng.choice([1, 2, 5, 17, 64, 65, 100, 128, 129, 200, 256, 300]))
    dmin = int(rng.integers(-12, 13))
    L1 = int(rng.integers(1, 45))
    opt = dict(min_disparity=dmin, max_disparity=dmin + D, cross_L1=L1, cross_L2=int(rng.integers(1, L1 + 1)),
               cross_t1=int(rng.integers(5, 40)), cross_t2=int(rng.integers(2, 20)),
               so_p1=float(rng.uniform(0.2, 2.0)), so_p2=float(rng.uniform(1.0, 4.0)), so_tso=int(rng.integers(5, 30)),
               lrcheck_thres=float(rng.uniform(0.5, 2.0)), irv_ts=int(rng.integers(5, 40)), irv_th=float(rng.uniform(0.2, 0.6)),
               do_lr_check=bool(rng.integers(0, 2)), do_filling=bool(rng.integers(0, 2)),
               do_discontinuity_adjustment=bool(rng.integers(0, 2)))
    L = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
    # smooth the texture along x so that arms have room to grow, and shift it for the right view
    k = int(rng.integers(1, 9))
    L = (np.cumsum(np.pad(L.astype(np.int32), ((0, 0), (k, 0), (0, 0)), mode="edge"), axis=1)[:, k:] -
         np.cumsum(np.pad(L.astype(np.int32), ((0, 0), (k, 0), (0, 0)), mode="edge"), axis=1)[:, :-k]) // k
    L = np.ascontiguousarray(L.astype(np.uint8))
    R = np.ascontiguousarray(np.roll(L, -int(rng.integers(0, 9)), axis=1))
    check_against_oracle(pkg, orc, L, R, **opt)


def test_tiny_images(pkg, orc):
    rng = np.random.default_rng(5)
    for (H, W) in [(1, 1), (1, 7), (9, 1), (2, 2), (5, 3)]:
        L = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
        R = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
        check_against_oracle(pkg, orc, L, R, max_disparity=5)


@pytest.mark.parametrize("seed,W,H,dmin,dmax", [(51, 203, 117, 0, 48), (52, 230, 150, 3, 133), (53, 150, 40, 0, 300),
                                                 (54, 700, 64, -4, 60), (55, 5, 3, 0, 5), (57, 1920, 72, 5, 261)])
def test_cost_fused_first_pass(pkg, orc, synth, seed, W, H, dmin, dmax):
    """Without the debug taps the initial costs are computed inside the first aggregation pass; the final
    scanline volume and the final map (both readable without taps) must still be the oracle's bits."""
    L, R, _ = synth.pair(seed, max(W, 16), max(H, 16), max(dmax - dmin, 8))
    L, R = np.ascontiguousarray(L[:H, :W]), np.ascontiguousarray(R[:H, :W])
    h = pkg.ADCensusStereo(0)
    assert h.Initialize(W, H, pkg.ADCensusOption(min_disparity=dmin, max_disparity=dmax)), h.last_error()
    disp = h.match(L, R)
    vol = h.get_buffer("cost_scan")
    h.close()
    want, taps = orc.match(L, R, orc.default_option(min_disparity=dmin, max_disparity=dmax), want=["cost_scan"])
    assert same(vol, taps["cost_scan"]), describe("cost_scan", vol, taps["cost_scan"])
    assert same(disp, want), describe("disp_final", disp, want)


def test_aggregated_volume_midsize(pkg, orc, synth):
    """640x360x256: the aggregated and the final cost volume bit for bit (about a quarter of a billion
    support-count divisions and as many scanline updates against the CPU's)."""
    L, R, _ = synth.pair(71, 640, 360, 256)
    h = pkg.ADCensusStereo(0)
    h.set_debug_taps(True)
    assert h.Initialize(640, 360, pkg.ADCensusOption(max_disparity=256)), h.last_error()
    h.match(L, R)
    got = {k: h.get_buffer(k) for k in ("cost_aggr", "cost_scan")}
    h.close()
    _, want = orc.match(L, R, orc.default_option(max_disparity=256), want=["cost_aggr", "cost_scan"])
    for k in ("cost_aggr", "cost_scan"):
        assert same(got[k], want[k]), describe(k, got[k], want[k])


def _scan_volume_and_map_match(pkg, orc, L, R, D):
    H, W = L.shape[:2]
    h = pkg.ADCensusStereo(0)
    assert h.Initialize(W, H, pkg.ADCensusOption(max_disparity=D)), h.last_error()
    disp = h.match(L, R)
    vol = h.get_buffer("cost_scan")
    h.close()
    want, taps = orc.match(L, R, orc.default_option(max_disparity=D), want=["cost_scan"])
    assert same(vol, taps["cost_scan"]), describe("cost_scan", vol, taps["cost_scan"])
    assert same(disp, want), describe("disp_final", disp, want)


@pytest.mark.parametrize("W,H", [(64, 1080), (96, 1100), (33, 1500), (24, 1600)])
def test_more_rows_than_simds(pkg, orc, synth, W, H):
    """Tall, narrow frames with 256 disparities: horizontal passes over more rows than the chip has SIMDs take
    the mixed launch (one wave per row + four waves per row; at 1080 rows its one-wave rows run the lean loop, at
    1100 the general one) or, from 1.5x the SIMD count, two waves per row; the final volume must still be the
    oracle's bits."""
    L, R, _ = synth.pair(81, 256, H, 256)
    _scan_volume_and_map_match(pkg, orc, np.ascontiguousarray(L[:, :W]), np.ascontiguousarray(R[:, :W]), 256)


@pytest.mark.parametrize("mode", ["0", "2"])
def test_scanline_wave_layouts(pkg, orc, synth, monkeypatch, mode):
    """ADC_SCAN_SPLIT=0: one wave per scanline in every pass; 2: two waves per scanline in every pass that allows it."""
    monkeypatch.setenv("ADC_DEBUG_ENV", "1")   # a handle reads the ADC_* switches in adc_create, only with this set
    monkeypatch.setenv("ADC_SCAN_SPLIT", mode)
    L, R, _ = synth.pair(82, 200, 120, 256)
    _scan_volume_and_map_match(pkg, orc, L, R, 256)


def test_direct_aggregation_kernel(pkg, orc, synth, monkeypatch):
    monkeypatch.setenv("ADC_DEBUG_ENV", "1")
    monkeypatch.setenv("ADC_AGG_VARIANT", "1")
    L, R, _ = synth.pair(23, 170, 90, 70)
    check_against_oracle(pkg, orc, L, R, max_disparity=70)


def test_determinism_reuse_and_reset(pkg, orc, synth):
    L1, R1, _ = synth.pair(31, 200, 120, 64)
    L2, R2, _ = synth.pair(32, 200, 120, 64)
    h = pkg.ADCensusStereo(0)
    o = pkg.ADCensusOption(max_disparity=64)
    assert h.Initialize(200, 120, o)
    a = h.match(L1, R1)
    b = h.match(L2, R2)
    a2 = h.match(L1, R1)          # second Match on the same handle (the reference frees its census buffers here)
    assert same(a, a2)
    assert same(a, orc.match(L1, R1, orc.default_option(max_disparity=64))[0])
    assert same(b, orc.match(L2, R2, orc.default_option(max_disparity=64))[0])
    L3, R3, _ = synth.pair(33, 96, 64, 24)
    assert h.Reset(96, 64, pkg.ADCensusOption(max_disparity=24))
    c = h.match(L3, R3)
    assert same(c, orc.match(L3, R3, orc.default_option(max_disparity=24))[0])
    h.close()


def test_error_paths_on_device(pkg):
    h = pkg.ADCensusStereo(0)
    img = np.zeros((8, 8, 3), np.uint8)
    out = np.zeros((8, 8), np.float32)
    assert not h.Match(img, img, out)                       # not initialised
    assert h.Initialize(8, 8, pkg.ADCensusOption(max_disparity=4)), h.last_error()
    assert not h.Match(None, img, out) and not h.Match(img, None, out) and not h.Match(img, img, None)
    assert h.Match(img, img, out)
    with pytest.raises(RuntimeError):
        h.get_buffer("cost_init")                           # debug tap without set_debug_taps
    h.close()


def test_match_device_and_batch(pkg, orc, synth):
    import torch
    L, R, _ = synth.pair(41, 256, 128, 64)
    want = orc.match(L, R, orc.default_option(max_disparity=64))[0]
    dev = torch.device("cuda:0")
    tl = torch.from_numpy(L).to(dev)
    tr = torch.from_numpy(R).to(dev)
    td = torch.empty((128, 256), dtype=torch.float32, device=dev)
    h = pkg.ADCensusStereo(0)
    assert h.Initialize(256, 128, pkg.ADCensusOption(max_disparity=64))
    stream = torch.cuda.current_stream().cuda_stream
    h.set_profiling(True)
    h.match_device(tl.data_ptr(), tr.data_ptr(), td.data_ptr(), stream)
    torch.cuda.synchronize()
    assert same(td.cpu().numpy(), want)
    t = h.stage_times()
    assert t["total"] > 0 and abs(sum(t[k] for k in ("pre", "cost", "aggr", "scan", "wta", "refine")) - t["total"]) < 0.05 * t["total"] + 0.05
    h.close()
    # the batch path pipelines uploads, matches and downloads over two staging slots: 5 pairs on one handle
    # reuse each slot at least twice; [0, 0] runs two such pipelines side by side on the same GPU
    pairs = [synth.pair(50 + i, 128, 64, 32)[:2] for i in range(5)]
    wants = [orc.match(l, r, orc.default_option(max_disparity=32))[0] for l, r in pairs]
    for devices in ([0], [0, 0]):
        disps = pkg.match_batch([p[0] for p in pairs], [p[1] for p in pairs], pkg.ADCensusOption(max_disparity=32), devices)
        for d, w in zip(disps, wants):
            assert same(d, w)
    assert pkg.match_batch([], [], pkg.ADCensusOption(max_disparity=32), [0]) == []


def test_cpp_shim_program(pkg, orc, synth, tmp_path):
    """The reference caller's C++ sequence (Initialize, Match, Match(null), Match) through ADCensusStereo.hpp."""
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.__file__), "bin", "shim_check")
    assert os.path.exists(exe), "build bin/shim_check first (make -C stereo-matching_amd)"
    L, R, _ = synth.pair(61, 180, 100, 48)
    L.tofile(tmp_path / "l.raw")
    R.tofile(tmp_path / "r.raw")
    subprocess.check_call([exe, "180", "100", "2", "50", str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), str(tmp_path / "o.raw")])
    got = np.fromfile(tmp_path / "o.raw", dtype=np.float32).reshape(100, 180)
    assert same(got, orc.match(L, R, orc.default_option(min_disparity=2, max_disparity=50))[0])


def test_full_resolution_2021_scene(pkg, orc):
    """1920x1080, 256 disparities: the final map against the committed oracle output, and the
    refinement chain re-run by the oracle on the GPU's own WTA maps."""
    L, R = artroom1()
    H, W = L.shape[:2]
    h = pkg.ADCensusStereo(0)
    assert h.Initialize(W, H, pkg.ADCensusOption(max_disparity=256)), h.last_error()
    disp = h.match(L, R)
    g = GOLD["artroom1"]
    z = np.load(os.path.join(GOLDEN, "artroom1_sample.npz"))
    assert same(disp.ravel()[z["idx"]], z["disp_final"])
    assert fnv1a64(h.get_buffer("arms")) == g["hash"]["arms"]
    assert fnv1a64(h.get_buffer("sup_count_h")) == g["hash"]["sup_count_h"]
    assert fnv1a64(h.get_buffer("census_left")) == g["hash"]["census_left"]
    dr = h.get_buffer("disp_right_wta")
    assert fnv1a64(dr) == g["hash"]["disp_right_wta"]
    assert fnv1a64(disp) == g["hash"]["disp_final"]
    # size-independent property: WTA of the device's own final volume, recomputed on the host
    vol = h.get_buffer("cost_scan")
    assert same(orc.wta_right(vol, 0, 256), dr)
    h.close()


def test_full_resolution_synthetic_bench_pair(pkg, synth):
    """The pair bench.py times (seed 1000, 1920x1080x256) against the committed oracle hash."""
    L, R, _ = synth.pair(1000, 1920, 1080, 256)
    g = GOLD["synth_1000_1080p"]
    assert fnv1a64(L) + fnv1a64(R) == g["input_hash"]
    h = pkg.ADCensusStereo(0)
    assert h.Initialize(1920, 1080, pkg.ADCensusOption(max_disparity=256)), h.last_error()
    disp = h.match(L, R)
    assert fnv1a64(disp) == g["hash"]["disp_final"]
    assert fnv1a64(h.get_buffer("disp_right_wta")) == g["hash"]["disp_right_wta"]
    h.close()


def test_uhd_512_disparities(pkg, orc, synth):
    """3840x2160, 512 disparities (two 17 GB volumes resident in HBM; 8 disparities per lane in the
    scanline kernels): the final map against the oracle run on the same host, and run-to-run determinism.
    The oracle needs ~70 GB of host memory at this size, so the comparison is skipped on smaller hosts."""
    import psutil
    L, R, _ = synth.pair(2026, 3840, 2160, 512)
    h = pkg.ADCensusStereo(0)
    assert h.Initialize(3840, 2160, pkg.ADCensusOption(max_disparity=512)), h.last_error()
    a = h.match(L, R)
    dr = h.get_buffer("disp_right_wta")
    b = h.match(L, R)
    assert same(a, b)
    assert np.isfinite(a).mean() > 0.9
    h.close()
    if psutil.virtual_memory().available < 110 * 2**30:
        pytest.skip("host too small for the 4K x 512 oracle; determinism checked only")
    want, taps = orc.match(L, R, orc.default_option(max_disparity=512), want=["disp_right_wta"])
    assert same(dr, taps["disp_right_wta"]), describe("disp_right_wta", dr, taps["disp_right_wta"])
    assert same(a, want), describe("disp_final", a, want)


def test_two_handles_from_two_threads(pkg, orc, synth):
    """Two handles initialised and matching at the same time from two host threads (what adc_match_batch does with
    one thread per device): per-device kernel set-up must not depend on which thread or handle came first.
    256 disparities and 34-pixel arms take the kernels that need more than the default 64 KB of dynamic LDS."""
    import threading
    pairs = [synth.pair(95 + i, 224, 96, 256)[:2] for i in range(2)]
    wants = [orc.match(l, r, orc.default_option(max_disparity=256))[0] for l, r in pairs]
    outs, errs = [None, None], []
    gate = threading.Barrier(2)

    def run(i):
        try:
            h = pkg.ADCensusStereo(0)
            gate.wait()
            assert h.Initialize(224, 96, pkg.ADCensusOption(max_disparity=256)), h.last_error()
            gate.wait()
            for _ in range(3):
                outs[i] = h.match(*pairs[i])
            h.close()
        except Exception as e:          # noqa: BLE001 -- reported by the main thread
            errs.append(e)
            gate.abort()

    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for o, w in zip(outs, wants):
        assert same(o, w)


def test_support_count_wrapping_to_zero(pkg, orc):
    """128-pixel arms on flat images: the support region of pixel (127, 127) holds 256 x 256 pixels, its 16-bit
    count (the reference's vec_sup_count_ is uint16) wraps to 0 and the aggregated cost there is sum / 0 = +inf,
    as in the oracle."""
    H = W = 300
    L = np.full((H, W, 3), 90, np.uint8)
    R = np.full((H, W, 3), 120, np.uint8)
    opt = dict(max_disparity=8, cross_L1=128, cross_L2=128, do_lr_check=0, do_filling=0, do_discontinuity_adjustment=0)
    h = pkg.ADCensusStereo(0)
    h.set_debug_taps(True)
    assert h.Initialize(W, H, pkg.ADCensusOption(**opt)), h.last_error()
    h.match(L, R)
    got = {k: h.get_buffer(k) for k in ("sup_count_h", "sup_count_v", "cost_aggr")}
    h.close()
    _, want = orc.match(L, R, orc.default_option(**opt), want=["sup_count_h", "sup_count_v", "cost_aggr"])
    assert got["sup_count_h"][127, 127] == 0 and want["sup_count_h"][127, 127] == 0
    assert np.isinf(want["cost_aggr"][127, 127]).all()
    for k in got:
        assert same(got[k], want[k]), describe(k, got[k], want[k])


def test_batch_with_caller_handles(pkg, orc, synth):
    """adc_match_batch on handles the caller keeps: two calls on the same handles (nothing is created or
    initialised by the calls), two handles on the one GPU drawing pairs from the shared counter."""
    opt = pkg.ADCensusOption(max_disparity=32)
    pairs = [synth.pair(150 + i, 128, 64, 32)[:2] for i in range(7)]
    wants = [orc.match(l, r, orc.default_option(max_disparity=32))[0] for l, r in pairs]
    hs = [pkg.ADCensusStereo(0), pkg.ADCensusStereo(0)]
    for h in hs:
        assert h.Initialize(128, 64, opt), h.last_error()
    for _ in range(2):
        disps = pkg.match_batch([p[0] for p in pairs], [p[1] for p in pairs], opt, [0, 0], handles=hs)
        for d, w in zip(disps, wants):
            assert same(d, w)
    # a handle initialised for another size is refused, not used
    assert hs[1].Reset(96, 64, opt)
    with pytest.raises(RuntimeError, match="not initialised for this device, size and option"):
        pkg.match_batch([p[0] for p in pairs], [p[1] for p in pairs], opt, [0, 0], handles=hs)
    with pytest.raises(ValueError):
        pkg.match_batch([pairs[0][0], pairs[1][0][:32]], [pairs[0][1], pairs[1][1][:32]], opt, [0])
    for h in hs:
        h.close()


def test_match_from_pinned_memory_and_timers(pkg, orc, synth):
    """adc_match on pinned caller memory (the DMA source / target itself) and on pageable memory give the same map;
    the per-kernel and transfer timers of a profiled match are populated and consistent with the stage times."""
    import torch
    L, R, _ = synth.pair(160, 320, 160, 64)
    want = orc.match(L, R, orc.default_option(max_disparity=64))[0]
    h = pkg.ADCensusStereo(0)
    assert h.Initialize(320, 160, pkg.ADCensusOption(max_disparity=64))
    h.set_profiling(True)
    pl, pr = torch.from_numpy(L).pin_memory(), torch.from_numpy(R).pin_memory()
    pd = torch.empty((160, 320), dtype=torch.float32).pin_memory()
    assert h.Match(pl.numpy(), pr.numpy(), pd.numpy())
    assert same(pd.numpy(), want)
    x = h.transfer_times()
    assert x["host_in"] == 0.0 and x["host_out"] == 0.0 and x["h2d"] > 0 and x["d2h"] > 0
    assert same(h.match(L, R), want)
    x = h.transfer_times()
    assert x["host_in"] > 0 and x["host_out"] > 0
    k, st = h.kernel_times(), h.stage_times()
    # (a launch that runs two passes reports under the first of them: the second is a back-to-back event pair, 0 or more)
    launches, total = h.aggr_kernel_stats()
    assert launches in (5, 8)
    own = (0, 1, 3, 5, 7) if launches == 5 else range(8)
    assert all(k["aggr%d" % i] > 0 for i in own) and all(k["aggr%d" % i] >= 0 for i in range(8))
    assert all(k["scan%d" % i] > 0 for i in range(4)) and k["wta_right"] > 0
    assert sum(k["aggr%d" % i] for i in range(8)) <= st["aggr"] * 1.02 + 0.01
    assert abs(total - sum(k["aggr%d" % i] for i in range(8))) < 1e-3
    h.close()


def test_wrong_array_shapes_are_refused(pkg):
    h = pkg.ADCensusStereo(0)
    assert h.Initialize(16, 8, pkg.ADCensusOption(max_disparity=4))
    ok = np.zeros((8, 16, 3), np.uint8)
    out = np.zeros((8, 16), np.float32)
    assert h.Match(ok, ok, out)
    assert not h.Match(np.zeros((8, 16), np.uint8), ok, out)            # a gray image
    assert not h.Match(ok, np.zeros((4, 16, 3), np.uint8), out)         # a frame of another size
    assert not h.Match(ok, ok, np.zeros((8, 8), np.float32))            # a short output map
    assert not h.Match(ok, ok, np.zeros((8, 16), np.float64))
    assert h.status == pkg.ADC_ERR_ARG
    h.close()


def test_frame_batch_at_size_on_one_gpu(pkg, synth):
    """The frame-batch workload on one GPU: 256 pairs of 1920x1080 with 256 disparities through adc_match_batch
    on a handle the caller keeps (8 distinct synthetic pairs, seeds 1000..1007, taken 32 times each; the host
    holds the 256 output maps). Every map must equal the single-Match result of its pair, and the first pair's
    the committed oracle hash."""
    W, H, D, NP, ND = 1920, 1080, 256, 256, 8
    opt = pkg.ADCensusOption(max_disparity=D)
    pairs = [synth.pair(1000 + i, W, H, D)[:2] for i in range(ND)]
    h = pkg.ADCensusStereo(0)
    assert h.Initialize(W, H, opt), h.last_error()
    singles = [h.match(l, r) for l, r in pairs]
    assert fnv1a64(singles[0]) == GOLD["synth_1000_1080p"]["hash"]["disp_final"]
    assert len({fnv1a64(s) for s in singles}) == ND
    disps = pkg.match_batch([pairs[i % ND][0] for i in range(NP)], [pairs[i % ND][1] for i in range(NP)], opt, [0], handles=[h])
    h.close()
    for i, d in enumerate(disps):
        assert same(d, singles[i % ND]), "pair %d" % i


@pytest.mark.parametrize("fused", ["0", "1"])
@pytest.mark.parametrize("D,dmin", [(64, 0), (200, 0), (45, 3), (2, 0)])
def test_left_wta_fused_and_separate(pkg, orc, synth, monkeypatch, fused, D, dmin):
    """The left-view WTA as part of the last scanline pass (large frames) and as a kernel of its own (small
    frames) are the same bits; ADC_WTA_FUSED forces either on a small frame."""
    monkeypatch.setenv("ADC_DEBUG_ENV", "1")
    monkeypatch.setenv("ADC_WTA_FUSED", fused)
    L, R, _ = synth.pair(170, 180, 70, max(D, 8))
    got = gpu_match(pkg, L, R, min_disparity=dmin, max_disparity=dmin + D)
    disp, want = orc.match(L, R, orc.default_option(min_disparity=dmin, max_disparity=dmin + D), want=["disp_left_wta"])
    assert same(got["disp_left_wta"], want["disp_left_wta"]), describe("disp_left_wta", got["disp_left_wta"], want["disp_left_wta"])
    assert same(got["disp_final"], disp)


@pytest.mark.parametrize("W,H,dmin,dmax,opt", [
    (200, 120, 0, 256, {}),                       # 128 disparities per slot, both chunks full
    (170, 90, 0, 70, {}),                         # 64 per slot, second chunk partly empty
    (150, 130, 3, 133, {}),                       # 130 disparities: 128-wide slots, second chunk nearly empty
    (97, 75, -4, 60, dict(cross_L1=20, cross_L2=9)),
    (64, 33, 0, 16, dict(cross_L1=5, cross_L2=2)),   # segments shorter than a block
    (40, 300, 0, 40, dict(cross_L1=40)),          # long columns, arms beyond the default
])
@pytest.mark.parametrize("fuse", ["1", "0"])
def test_fused_aggregation_pairs(pkg, orc, synth, monkeypatch, W, H, dmin, dmax, opt, fuse):
    """The dividing aggregation pass and the same-axis pass behind it run as one launch wherever the kernel's rings
    fit the LDS (the default); ADC_AGG_FUSE=0 runs all eight passes on their own. Either way every tap must be the
    oracle's bits, with and without the taps' separate cost kernel."""
    monkeypatch.setenv("ADC_DEBUG_ENV", "1")
    monkeypatch.setenv("ADC_AGG_FUSE", fuse)
    L, R, _ = synth.pair(180, max(W, 16), max(H, 16), max(dmax - dmin, 8))
    L, R = np.ascontiguousarray(L[:H, :W]), np.ascontiguousarray(R[:H, :W])
    got = check_against_oracle(pkg, orc, L, R, min_disparity=dmin, max_disparity=dmax, **opt)
    h = pkg.ADCensusStereo(0)
    assert h.Initialize(W, H, pkg.ADCensusOption(min_disparity=dmin, max_disparity=dmax, **opt)), h.last_error()
    disp = h.match(L, R)              # no taps: the first pass computes the costs itself, other ping-pong order
    vol = h.get_buffer("cost_scan")
    aux = h.get_buffer("cost_scan_aux")
    h.close()
    assert same(disp, got["disp_final"]) and same(vol, got["cost_scan"]) and same(aux, got["cost_scan_aux"])


@pytest.mark.parametrize("mode", ["0", "1", "2"])
@pytest.mark.parametrize("W,H,dmin,dmax", [(200, 120, 0, 256), (61, 150, -3, 125), (33, 35, 0, 128)])
def test_fused_aggregation_fill_modes(pkg, orc, synth, monkeypatch, mode, W, H, dmin, dmax):
    """Full 128-float ring slots (D a multiple of 128): ring 1 of the fused aggregation launch is filled either by every
    wave for itself (ADC_AGG_MODE=0) or by a loader wave beside fifteen computing waves (1, the default; 30 positions per
    iteration instead of 32, other ring sizes). Both are the oracle's bits, lines shorter than one iteration included."""
    monkeypatch.setenv("ADC_DEBUG_ENV", "1")
    monkeypatch.setenv("ADC_AGG_MODE", mode)
    L, R, _ = synth.pair(190, max(W, 16), max(H, 16), dmax - dmin)
    L, R = np.ascontiguousarray(L[:H, :W]), np.ascontiguousarray(R[:H, :W])
    check_against_oracle(pkg, orc, L, R, min_disparity=dmin, max_disparity=dmax)


@pytest.mark.parametrize("split", ["0", "2", "3", "5"])
@pytest.mark.parametrize("W,H,dmin,dmax", [(300, 200, 0, 256), (131, 333, -2, 126), (257, 96, 0, 64)])
def test_fused_aggregation_line_pieces(pkg, orc, synth, monkeypatch, split, W, H, dmin, dmax):
    """The fused aggregation launch cuts the lines of its last, partly filled round over the CUs into pieces; a piece
    recomputes the first stage of the arm span in front of it. ADC_AGG_SPLIT=n cuts EVERY line into n pieces (0: none):
    the volumes and the map must not depend on where a line is cut."""
    monkeypatch.setenv("ADC_DEBUG_ENV", "1")
    monkeypatch.setenv("ADC_AGG_SPLIT", split)
    L, R, _ = synth.pair(191, W, H, dmax - dmin)
    check_against_oracle(pkg, orc, L, R, min_disparity=dmin, max_disparity=dmax)


@pytest.mark.parametrize("L1,W,H", [(70, 320, 200), (130, 300, 280), (255, 560, 300)])
def test_region_voting_long_arms(pkg, orc, synth, L1, W, H):
    """Arm caps far beyond the default 34 with thresholds that let the arms grow: region rows longer than a wave
    instruction's share of lanes, regions taller than 64 rows (several chunks of row arms per pixel) and more tiles
    under a region's bounding box than the two stamp loads per lane cover. Every tap against the oracle."""
    L, R, _ = synth.pair(300 + L1, W, H, 48)
    got = check_against_oracle(pkg, orc, L, R, max_disparity=48, cross_L1=L1, cross_L2=L1 // 2, cross_t1=60, cross_t2=40)
    a = got["arms"].astype(int).reshape(-1, 4)
    assert a.max() > min(L1, 254) * 0.9 and (a[:, 2] + a[:, 3] + 1).max() > 64      # long arms, tall regions did occur
    assert got["mask_occ"].sum() + got["mask_mis"].sum() > 100                       # and there was something to vote on
