"""Host-side harness (include/adcensus_harness.h): PNG decode/encode, SaveDisparityMap's normalisation and
PerformanceEval0103, against PIL and the numpy restatement in oracle/harness_oracle.py. CPU only, except
the end-to-end run of the command-line program."""
import os
import subprocess

import numpy as np
import pytest

from conftest import DATA, GOLDEN, ROOT, load_bgr, load_gray


@pytest.fixture(scope="module")
def horc():
    from oracle import harness_oracle
    return harness_oracle


@pytest.mark.parametrize("rel", ["teddy/im2.png", "cones/im6.png", "artroom1/im0.png"])
def test_png_read_bgr_matches_pil(pkg, rel):
    path = os.path.join(DATA, rel)
    assert np.array_equal(pkg.adcensus.png_read(path, 3), load_bgr(path))


def test_png_read_gray_and_reject_colour(pkg):
    path = os.path.join(DATA, "teddy", "disp2.png")
    assert np.array_equal(pkg.adcensus.png_read(path, 1), load_gray(path))
    with pytest.raises(IOError):
        pkg.adcensus.png_read(os.path.join(DATA, "teddy", "im2.png"), 1)   # imread(path, 0) of a colour file: not supported
    with pytest.raises(IOError):
        pkg.adcensus.png_read(os.path.join(DATA, "teddy", "missing.png"), 3)


def test_png_write_round_trip(pkg, tmp_path):
    rng = np.random.default_rng(3)
    for shape in [(1, 1), (7, 13), (375, 450)]:
        g = rng.integers(0, 256, size=shape).astype(np.uint8)
        p = str(tmp_path / "g.png")
        pkg.adcensus.png_write_gray(p, g)
        assert np.array_equal(load_gray(p), g)                # PIL reads what we wrote
        assert np.array_equal(pkg.adcensus.png_read(p, 1), g)
        assert np.array_equal(pkg.adcensus.png_read(p, 3), np.repeat(g[:, :, None], 3, axis=2))


def test_png_read_filters_palette_alpha(pkg, tmp_path):
    """Files PIL writes with adaptive filters, a palette, and alpha channels."""
    from PIL import Image
    rng = np.random.default_rng(4)
    rgb = rng.integers(0, 256, size=(33, 41, 3)).astype(np.uint8)
    rgb[:, :20] = rgb[:, :1]                                  # smooth part so sub/up/paeth filters get chosen
    cases = {"rgb": Image.fromarray(rgb), "rgba": Image.fromarray(np.dstack([rgb, rgb[:, :, :1]]), "RGBA"),
             "pal": Image.fromarray(rgb).quantize(64), "la": Image.fromarray(rgb[:, :, 0]).convert("LA")}
    for name, im in cases.items():
        p = str(tmp_path / (name + ".png"))
        im.save(p, optimize=True)
        want = np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1]
        assert np.array_equal(pkg.adcensus.png_read(p, 3), want), name


def test_disparity_to_u8(pkg, horc):
    z = np.load(os.path.join(GOLDEN, "teddy_disp.npz"))
    d = z["disp_final"]
    assert np.array_equal(pkg.adcensus.disparity_to_u8(d), horc.disparity_to_u8(d))
    rng = np.random.default_rng(5)
    r = (rng.random((40, 50), dtype=np.float32) * 60 - 5).astype(np.float32)      # includes negatives: |d|
    r[rng.random((40, 50)) < 0.2] = np.inf
    got = pkg.adcensus.disparity_to_u8(r)
    assert np.array_equal(got, horc.disparity_to_u8(r))
    assert got[np.isinf(r)].max() == 0 and got.max() == 255
    assert pkg.adcensus.disparity_to_u8(np.full((4, 4), np.inf, np.float32)).max() == 0
    assert pkg.adcensus.disparity_to_u8(np.full((4, 50), 7.0, np.float32)).max() == 0   # one distinct value: span 0


@pytest.mark.parametrize("name", ["teddy", "cones"])
def test_eval_0103(pkg, orc, horc, name):
    z = np.load(os.path.join(GOLDEN, name + "_disp.npz"))
    gt = load_gray(os.path.join(DATA, name, "disp2.png"))
    d = z["disp_final"]
    rms, pep, mde = pkg.adcensus.eval_0103(d, gt, 4.0, 2.0, 0.5)
    o_rms, o_pep, o_mde = horc.eval_0103(d, gt, 4.0, 2.0, 0.5)
    assert pep == o_pep and abs(rms - o_rms) <= 1e-6 * o_rms
    assert mde == o_mde or (np.isinf(mde) and np.isinf(o_mde))
    bad, rms64 = orc.bad_fraction(d, gt, 4.0, 2.0)            # the double-precision statement of the same numbers
    assert abs(pep - bad) < 1e-6 and abs(rms - rms64) < 1e-3


def test_pfm_round_trip_and_eval(pkg, horc, tmp_path):
    rng = np.random.default_rng(6)
    gt = (rng.random((37, 53), dtype=np.float32) * 200).astype(np.float32)
    gt[rng.random((37, 53)) < 0.1] = np.inf                  # unknown ground truth
    p = str(tmp_path / "disp0.pfm")
    pkg.adcensus.pfm_write(p, gt)
    assert np.array_equal(horc.pfm_read(p), gt)              # an independent reader agrees (bottom-up rows)
    assert np.array_equal(pkg.adcensus.pfm_read(p), gt)
    open(str(tmp_path / "c.pfm"), "wb").write(b"PF\n2 2\n-1.0\n" + bytes(48))
    with pytest.raises(IOError):
        pkg.adcensus.pfm_read(str(tmp_path / "c.pfm"))        # colour PFM: not a disparity map
    est = (gt + rng.normal(0, 1.5, gt.shape)).astype(np.float32)
    est[np.isinf(est)] = 17.0
    got, want = pkg.adcensus.eval_pfm(est, gt, 2.0, 0.25), horc.eval_pfm(est, gt, 2.0, 0.25)
    assert got[1] == want[1] and abs(got[0] - want[0]) <= 1e-6 * want[0] and got[2] == want[2]
    est[3, 4] = np.inf                                       # one invalid estimate: RMS and Mde go to infinity
    got = pkg.adcensus.eval_pfm(est, gt, 2.0, 0.25)
    assert np.isinf(got[0]) and np.isinf(got[2]) and got[1] == horc.eval_pfm(est, gt, 2.0, 0.25)[1]


def test_depth_and_calib(pkg, horc):
    f, b, doffs = pkg.adcensus.calib_read(os.path.join(DATA, "artroom1", "calib.txt"))
    assert (round(f, 2), round(b, 2), doffs) == (1733.74, 536.62, 0.0)   # the reference's table entry 0 (main.cpp:9-18)
    z = np.load(os.path.join(GOLDEN, "teddy_disp.npz"))
    d = z["disp_final"].copy()
    d[0, :5] = [np.inf, 0.0, -3.0, 1.0, 60.0]
    depth, u8, lo, hi = pkg.adcensus.disparity_to_depth(d, f, b, doffs)
    o_depth, o_u8, o_lo, o_hi = horc.depth(d, f, b, doffs)
    assert np.array_equal(depth, o_depth) and np.array_equal(u8, o_u8) and (lo, hi) == (o_lo, o_hi)
    assert depth[0, 0] == 0 and depth[0, 1] == 0 and depth[0, 2] == np.float32(np.float32(f) * np.float32(b) / np.float32(3.0))
    assert lo == 0.0                                          # a zero anywhere pulls the minimum to 0, as in the reference


@pytest.mark.gpu
def test_cli_end_to_end(pkg, orc, tmp_path):
    """adc_match: PNGs in -> disparity.png + evaluation block, against the library called directly."""
    import json
    exe = os.path.join(os.path.dirname(pkg.__file__), "bin", "adc_match")
    assert os.path.exists(exe), "build bin/adc_match first (make -C stereo-matching_amd)"
    d = os.path.join(DATA, "teddy")
    out, raw = str(tmp_path / "disparity.png"), str(tmp_path / "disp.f32")
    p = subprocess.run([exe, os.path.join(d, "im2.png"), os.path.join(d, "im6.png"), out, "--min", "0", "--max", "64",
                        "--gt", os.path.join(d, "disp2.png"), "--gt-div", "4", "--raw", raw],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    disp = np.fromfile(raw, np.float32).reshape(375, 450)
    z = np.load(os.path.join(GOLDEN, "teddy_disp.npz"))
    assert np.array_equal(disp, z["disp_final"])
    assert np.array_equal(load_gray(out), pkg.adcensus.disparity_to_u8(disp))
    gold = json.load(open(os.path.join(GOLDEN, "golden.json")))["teddy"]["bad2"]
    pep = float([l for l in p.stdout.splitlines() if l.startswith("PEP_exact")][0].split("=")[1])
    assert abs(pep - gold) < 1e-6
    assert "Evalution:" in p.stdout and "PEP = %.3f (bad 2.0)" % gold in p.stdout


@pytest.mark.gpu
def test_cli_2021_scene_pfm_eval_and_depth(pkg, horc, tmp_path):
    """adc_match on the 1920x1080 Middlebury 2021 scene, 256 disparities: the map against the committed oracle
    hash, the 2021 evaluation (PerformanceEval, performance_eval.cpp:184-215) against a PFM ground truth and the
    depth export (Disp2Depth2 with the scene's calib.txt, main.cpp:499-551), both against the host-side oracle of
    the harness. The reference checkout holds no 2021 ground-truth PFMs, so the ground truth is synthetic: a
    slanted plane with a band of unknown (+inf) pixels, written with the library's own PFM writer."""
    import json
    from conftest import fnv1a64
    exe = os.path.join(os.path.dirname(pkg.__file__), "bin", "adc_match")
    assert os.path.exists(exe), "build bin/adc_match first (make -C stereo-matching_amd)"
    d = os.path.join(DATA, "artroom1")
    H, W = 1080, 1920
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    gt = (40.0 + 0.05 * xx + 0.02 * yy).astype(np.float32)
    gt[100:140, :] = np.inf                                  # unknown ground truth (inf in the 2021 PFMs)
    gt_path = str(tmp_path / "disp0.pfm")
    pkg.adcensus.pfm_write(gt_path, gt)
    out, raw, depth_png = str(tmp_path / "disparity.png"), str(tmp_path / "disp.f32"), str(tmp_path / "depth.png")
    p = subprocess.run([exe, os.path.join(d, "im0.png"), os.path.join(d, "im1.png"), out, "--min", "0", "--max", "256",
                        "--gt-pfm", gt_path, "--calib", os.path.join(d, "calib.txt"), "--depth", depth_png, "--raw", raw],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    disp = np.fromfile(raw, np.float32).reshape(H, W)
    gold = json.load(open(os.path.join(GOLDEN, "golden.json")))["artroom1"]
    assert fnv1a64(disp) == gold["hash"]["disp_final"]
    assert np.array_equal(load_gray(out), horc.disparity_to_u8(disp))
    # evaluation block
    rms, pep, _ = horc.eval_pfm(disp, horc.pfm_read(gt_path), 2.0, 1.0)
    lines = {l.split("=")[0].strip(): l.split("=")[1].strip() for l in p.stdout.splitlines() if "=" in l}
    assert float(lines["PEP_exact"]) == np.float32(pep)
    got_rms = float(lines["RMS_exact"])
    assert (np.isinf(got_rms) and np.isinf(rms)) or abs(got_rms - rms) <= 1e-6 * abs(rms)
    # depth export
    f, b, doffs = pkg.adcensus.calib_read(os.path.join(d, "calib.txt"))
    _, o_u8, o_lo, o_hi = horc.depth(disp, f, b, doffs)
    assert np.array_equal(load_gray(depth_png), o_u8)
    assert "max_depth%g" % o_hi in p.stdout and "min_depth%g" % o_lo in p.stdout


@pytest.mark.gpu
def test_c_abi_under_host_sanitizers():
    """The library built with AddressSanitizer + UBSan on its host code (`make -C stereo-matching_amd san`), driven
    through every C-ABI entry point by tests/cpp/host_san_driver.cpp: no report, exit code 0."""
    exe = os.path.join(ROOT, "stereo-matching_amd", "bin", "host_san_driver")
    if not os.path.exists(exe):
        pytest.skip("bin/host_san_driver not built (make -C stereo-matching_amd san)")
    env = dict(os.environ, ASAN_OPTIONS="protect_shadow_gap=0:detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=240)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert "host sanitizer driver ok" in p.stdout
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
