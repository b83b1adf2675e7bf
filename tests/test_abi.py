"""CPU: the C-ABI library loads, exports every symbol include/adcensus.h declares, and its argument
checks behave like the reference's `return false` paths (ADCensusStereo.cu:26-35,66-71). No compute
runs here; with no GPU every device-touching call must fail loudly, never fall back."""
import ctypes as C
import os
import re

import numpy as np

from conftest import ROOT


def header_functions(name="adcensus.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(adc_[a-z0-9_]+)\s*\(", txt)))


def test_exports_match_header(pkg):
    L = pkg.lib()
    names = header_functions()
    assert names == sorted(pkg.EXPORTS)
    for n in names:
        assert hasattr(L, n), n
    harness = header_functions("adcensus_harness.h")
    assert harness == sorted(pkg.HARNESS_EXPORTS)
    for n in harness:
        assert hasattr(L, n), n


def test_option_defaults_and_layout(pkg, orc):
    o = pkg.ADCensusOption()
    oo = orc.default_option()
    assert C.sizeof(o) == C.sizeof(oo) == 60
    for f, _ in o._fields_:
        assert getattr(o, f) == getattr(oo, f), f
    # ADCensusOption() defaults, reference adcensus_types.h:86-93 and config.h:12-13,26-29
    assert (o.min_disparity, o.max_disparity, o.lambda_ad, o.lambda_census) == (0, 64, 10, 30)
    assert (o.cross_L1, o.cross_L2, o.cross_t1, o.cross_t2) == (34, 17, 20, 6)
    assert (o.so_p1, o.so_p2, o.so_tso, o.irv_ts) == (1.0, 3.0, 15, 20)
    assert abs(o.irv_th - 0.4) < 1e-7 and o.lrcheck_thres == 1.0
    assert (o.do_lr_check, o.do_filling, o.do_discontinuity_adjustment) == (1, 1, 1)


def test_argument_errors_without_device(pkg):
    h = pkg.ADCensusStereo(0)
    o = pkg.ADCensusOption()
    img = np.zeros((4, 5, 3), np.uint8)
    out = np.zeros((4, 5), np.float32)
    assert not h.Match(img, img, out) and h.status == pkg.adcensus.ADC_ERR_NOT_INIT
    assert not h.Initialize(0, 10, o) and h.status == pkg.adcensus.ADC_ERR_ARG
    assert not h.Initialize(10, -1, o) and h.status == pkg.adcensus.ADC_ERR_ARG
    assert not h.Initialize(10, 10, pkg.ADCensusOption(min_disparity=5, max_disparity=5)) and h.status == pkg.adcensus.ADC_ERR_ARG
    assert not h.Initialize(10, 10, pkg.ADCensusOption(max_disparity=2000)) and h.status == pkg.adcensus.ADC_ERR_UNSUPPORTED
    assert "512" in h.last_error()
    # 512 disparities is the widest layout the kernels are built for; one more is refused, not attempted
    assert not h.Initialize(10, 10, pkg.ADCensusOption(min_disparity=-3, max_disparity=510)) and h.status == pkg.adcensus.ADC_ERR_UNSUPPORTED
    assert pkg.lib().adc_buffer_bytes(h._h, 0) == 0
    h.close()


def test_no_cpu_fallback(pkg):
    """On a machine without a GPU, Initialize must fail with a HIP error instead of computing elsewhere."""
    import torch
    if torch.cuda.is_available():
        return
    h = pkg.ADCensusStereo(0)
    assert not h.Initialize(16, 16, pkg.ADCensusOption())
    assert h.status == pkg.adcensus.ADC_ERR_HIP and h.last_error()
    h.close()


def test_product_does_not_reference_oracle():
    pk = os.path.join(ROOT, "stereo-matching_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.replace("the CPU oracle", "").lower() or f == "Makefile", os.path.join(dp, f)


def test_cpp_shim_error_paths(pkg):
    """include/ADCensusStereo.hpp compiled as plain C++ against the C-ABI: the reference's `return false` paths."""
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.__file__), "bin", "shim_check")
    assert os.path.exists(exe), "build bin/shim_check first (make -C stereo-matching_amd)"
    assert subprocess.run([exe, "--errors"], capture_output=True).returncode == 0
