"""The entry point that allocates nothing (adc_match_device_static) against adc_initialize + adc_match_device.

Called through the C-ABI; the results must be the same bits. Two calls in a row share the GPU's static
scratch, the second on a smaller frame (so the array is laid out anew over the first call's leftovers),
then the first size again and a repeat without any re-layout.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reference(pkg, torch, L, R, **opt):
    """The existing entry point: a handle with its own hipMalloc'ed buffers."""
    H, W = L.shape[:2]
    dev = torch.device("cuda:0")
    tl, tr = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    td = torch.empty((H, W), dtype=torch.float32, device=dev)
    h = pkg.ADCensusStereo(0)
    assert h.Initialize(W, H, pkg.ADCensusOption(**opt)), h.last_error()
    h.match_device(tl.data_ptr(), tr.data_ptr(), td.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = td.cpu().numpy()
    h.close()
    return out


def _static(pkg, torch, L, R, **opt):
    H, W = L.shape[:2]
    dev = torch.device("cuda:0")
    tl, tr = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    td = torch.full((H, W), -7.0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    pkg.ADCensusStereo.match_device_static(0, W, H, pkg.ADCensusOption(**opt), tl.data_ptr(), tr.data_ptr(), td.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return td.cpu().numpy()


def test_static_scratch_entry_point(pkg, orc, synth):
    import torch
    big = synth.pair(91, 320, 200, 64)[:2]
    small = synth.pair(92, 150, 90, 40)[:2]
    want_big = _reference(pkg, torch, *big, max_disparity=64)
    want_small = _reference(pkg, torch, *small, min_disparity=2, max_disparity=42)
    # the existing entry point is itself the oracle's bits on these inputs
    assert np.array_equal(want_big, orc.match(*big, orc.default_option(max_disparity=64))[0])
    # two calls in a row, the second with the smaller input
    got_big = _static(pkg, torch, *big, max_disparity=64)
    got_small = _static(pkg, torch, *small, min_disparity=2, max_disparity=42)
    assert np.array_equal(got_big, want_big)
    assert np.array_equal(got_small, want_small)
    # back to the first size (re-layout over the small frame's leftovers), then a call that re-lays nothing
    assert np.array_equal(_static(pkg, torch, *big, max_disparity=64), want_big)
    assert np.array_equal(_static(pkg, torch, *big, max_disparity=64), want_big)
    # a different option on the same size is a new layout too (other tables)
    want_opt = _reference(pkg, torch, *big, max_disparity=64, lambda_ad=7, so_tso=9)
    assert np.array_equal(_static(pkg, torch, *big, max_disparity=64, lambda_ad=7, so_tso=9), want_opt)


def test_static_scratch_middlebury_fits(pkg):
    """A 450x375 frame with 64 disparities (the Middlebury 2003 size) fits the static scratch."""
    import torch
    from conftest import middlebury2003
    L, R, _ = middlebury2003("teddy")
    assert np.array_equal(_static(pkg, torch, L, R, max_disparity=64), _reference(pkg, torch, L, R, max_disparity=64))


def test_static_scratch_limits(pkg):
    import torch
    dev = torch.device("cuda:0")
    img = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device=dev)
    out = torch.empty((1080, 1920), dtype=torch.float32, device=dev)
    with pytest.raises(RuntimeError, match="static scratch"):
        pkg.ADCensusStereo.match_device_static(0, 1920, 1080, pkg.ADCensusOption(max_disparity=256), img.data_ptr(),
                                               img.data_ptr(), out.data_ptr(), None)
    with pytest.raises(RuntimeError):      # null pointer
        pkg.ADCensusStereo.match_device_static(0, 64, 64, pkg.ADCensusOption(max_disparity=16), None, img.data_ptr(),
                                               out.data_ptr(), None)
    with pytest.raises(RuntimeError):      # empty disparity range, as adc_initialize
        pkg.ADCensusStereo.match_device_static(0, 64, 64, pkg.ADCensusOption(min_disparity=5, max_disparity=5), img.data_ptr(),
                                               img.data_ptr(), out.data_ptr(), None)
    # a failed call leaves the entry point usable
    small = torch.zeros((64, 64, 3), dtype=torch.uint8, device=dev)
    o2 = torch.empty((64, 64), dtype=torch.float32, device=dev)
    pkg.ADCensusStereo.match_device_static(0, 64, 64, pkg.ADCensusOption(max_disparity=16), small.data_ptr(), small.data_ptr(),
                                           o2.data_ptr(), None)
    torch.cuda.synchronize()


def test_static_scratch_allocates_no_device_memory(pkg, synth):
    """What the entry point exists for: the GPU's free memory is the same before and after its calls, re-layouts for
    another size and another option included (hipMemGetInfo through torch; inputs and outputs are allocated first)."""
    import torch
    dev = torch.device("cuda:0")
    cases = []
    for seed, W, H, opt in [(93, 256, 160, dict(max_disparity=48)), (94, 130, 100, dict(max_disparity=32)),
                            (93, 256, 160, dict(max_disparity=48, lambda_ad=8))]:
        L, R = synth.pair(seed, W, H, 48)[:2]
        cases.append((W, H, pkg.ADCensusOption(**opt), torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev),
                      torch.empty((H, W), dtype=torch.float32, device=dev)))
    stream = torch.cuda.current_stream().cuda_stream

    def run_all():
        for W, H, o, tl, tr, td in cases + cases[:1]:          # three layouts, then the first one again
            pkg.ADCensusStereo.match_device_static(0, W, H, o, tl.data_ptr(), tr.data_ptr(), td.data_ptr(), stream)
            pkg.ADCensusStereo.match_device_static(0, W, H, o, tl.data_ptr(), tr.data_ptr(), td.data_ptr(), stream)   # no re-layout
        torch.cuda.synchronize()

    run_all()                                                   # code objects loaded, the runtime's own pools warm
    free0 = torch.cuda.mem_get_info()[0]
    run_all()
    assert torch.cuda.mem_get_info()[0] == free0


def test_static_scratch_refuses_a_capturing_stream(pkg, synth):
    """On a stream that is being captured into a graph the entry point enqueues nothing and says so (a re-layout would
    invalidate the capture; replaying a captured frame is not supported); the caller's capture stays usable and the
    entry point works as before afterwards."""
    import torch
    dev = torch.device("cuda:0")
    W, H, D = 200, 120, 40
    L, R = synth.pair(95, W, H, D)[:2]
    want = _reference(pkg, torch, L, R, max_disparity=D)
    opt = pkg.ADCensusOption(max_disparity=D)
    tl, tr = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    td = torch.zeros((H, W), dtype=torch.float32, device=dev)
    probe = torch.zeros(16, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        pkg.ADCensusStereo.match_device_static(0, W, H, opt, tl.data_ptr(), tr.data_ptr(), td.data_ptr(), s.cuda_stream)
    s.synchronize()
    assert np.array_equal(td.cpu().numpy(), want)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        probe.add_(1.0)                                           # the caller's own captured work
        for w in (W, W - 8):                                      # the same size (no re-layout needed) and another one
            with pytest.raises(RuntimeError, match="captured"):
                pkg.ADCensusStereo.match_device_static(0, w, H, opt, tl.data_ptr(), tr.data_ptr(), td.data_ptr(), s.cuda_stream)
    graph.replay()                                                # the capture was left intact: only the caller's node runs
    torch.cuda.synchronize()
    assert float(probe.sum()) == 16.0
    td.fill_(-3.0)
    with torch.cuda.stream(s):
        pkg.ADCensusStereo.match_device_static(0, W, H, opt, tl.data_ptr(), tr.data_ptr(), td.data_ptr(), s.cuda_stream)
    s.synchronize()
    assert np.array_equal(td.cpu().numpy(), want)
