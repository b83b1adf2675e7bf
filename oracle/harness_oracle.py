"""TEST INFRASTRUCTURE ONLY: numpy restatement of the host-side harness of the reference's caller
(PARITY UNPINNED like the rest of the oracle: the reference ships no expected outputs for it).

  disparity_to_u8  -- SaveDisparityMap, ADC/main.cpp:443-468
  eval_0103        -- PerformanceEval0103, ADC/performance_eval.cpp:218-253
  pfm_read         -- LoadPFM, ADC/performance_eval.cpp:11-91
  eval_pfm         -- PerformanceEval, ADC/performance_eval.cpp:184-215
  depth            -- Disp2Depth2, ADC/main.cpp:499-551
ADC/ = final_version_project/Stereo_Matching/ADCensus_CUDA/ in the reference."""
import numpy as np


def disparity_to_u8(disp):
    d = np.abs(np.asarray(disp, np.float32))
    valid = d != np.float32(np.inf)
    out = np.zeros(d.shape, np.uint8)
    if not valid.any():
        return out
    lo = np.float32(min(np.float32(d.shape[1]), d[valid].min()))
    hi = np.float32(max(-np.float32(d.shape[1]), d[valid].max()))
    span = np.float32(hi - lo)
    if not span > 0:
        return out
    v = ((d[valid] - lo) / span * np.float32(255)).astype(np.float32)
    out[valid] = np.trunc(v).astype(np.uint8)
    return out


def eval_0103(disp, gt_u8, gt_div=4.0, delta=2.0, seconds=1.0):
    """float32 accumulation in row-major order, as the reference does."""
    d = np.asarray(disp, np.float32).ravel()
    g = np.asarray(gt_u8, np.uint8).ravel().astype(np.float32)
    known = g != 0
    est = np.where(d != np.float32(np.inf), d, np.float32(0))
    err = (est - g / np.float32(gt_div)).astype(np.float32)[known]
    sq = np.float32(0)
    if err.size:
        sq = np.add.accumulate((err * err).astype(np.float32), dtype=np.float32)[-1]
    bad = np.float32(np.count_nonzero(np.abs(err) > np.float32(delta)))
    n = np.float32(d.size)
    dmax = np.float32(max(np.float32(0), d.max()))
    rms = np.sqrt(np.float32(sq / n))
    pep = np.float32(bad / n)
    with np.errstate(over="ignore", invalid="ignore"):
        mde = np.float32(np.float64(np.float32(n * dmax)) / seconds * 1e-6)
    return float(rms), float(pep), float(mde)


def pfm_read(path):
    raw = open(path, "rb").read()
    parts = raw.split(None, 4)
    assert parts[0] == b"Pf"
    w, h, scale = int(parts[1]), int(parts[2]), float(parts[3])
    assert scale < 0
    off = raw.index(parts[3]) + len(parts[3]) + 1
    a = np.frombuffer(raw, "<f4", count=w * h, offset=off).reshape(h, w)
    return np.ascontiguousarray(a[::-1])


def eval_pfm(disp, gt, delta=2.0, seconds=1.0):
    d = np.asarray(disp, np.float32).ravel()
    g = np.asarray(gt, np.float32).ravel()
    g = np.where(g == np.float32(np.inf), np.float32(0), g)
    with np.errstate(over="ignore", invalid="ignore"):
        err = (d - g).astype(np.float32)
        sq = np.add.accumulate((err * err).astype(np.float32), dtype=np.float32)[-1]
        bad = np.float32(np.count_nonzero(np.abs(err) > np.float32(delta)))
        n = np.float32(d.size)
        dmax = np.float32(max(np.float32(0), d.max()))
        rms = np.sqrt(np.float32(sq / n))
        pep = np.float32(bad / n)
        mde = np.float32(np.float64(np.float32(n * dmax)) * 1e-6 / seconds)
    return float(rms), float(pep), float(mde)


def depth(disp, focal, baseline, doffs=0.0):
    d = np.abs(np.asarray(disp, np.float32))
    d = np.where(d == np.float32(np.inf), np.float32(0), d)
    z = np.zeros(d.shape, np.float32)
    nz = d != 0
    z[nz] = (np.float32(focal) * np.float32(baseline) / d[nz] + np.float32(doffs)).astype(np.float32)
    lo = np.float32(min(np.float32(114514.0), z.min()))
    hi = np.float32(max(np.float32(0), z.max()))
    u8 = np.zeros(d.shape, np.uint8)
    if hi - lo > 0:
        u8 = np.trunc(((z - lo) / np.float32(hi - lo) * np.float32(255)).astype(np.float32)).astype(np.uint8)
    return z, u8, float(lo), float(hi)
