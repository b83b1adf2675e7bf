#!/usr/bin/env python3
"""A coarse tie between the CPU oracle and the one output of the reference's own binary that the reference holds:
panel (a) of figs/middleburry2021.png, the JET-coloured, min-max-normalised, quarter-size disparity map of the
Middlebury 2021 scene artroom2 (the bird statue sits in the lower left corner: artroom1 shows it further right;
ADC/main.cpp:443-476 writes such maps; MAX_DISPARITY was 130, ADC/config.h:3).

The panel is lossy (down-scaled, colour-mapped, JPEG-like edges in the collage), so this is a sanity check against
gross errors -- never a golden vector: the colours are mapped back to 0..255 through the JET table, the oracle's map
of Middlebury/2021/artroom2 (D = 130, the reference's defaults) is normalised the same way and reduced to the panel's size,
and the two are compared by Spearman rank correlation and, after the best affine map of one onto the other (the two
normalisations need not share their end points), by the share of pixels within 8 % of the range.
Runs on the CPU, in the container that has /root/reference; writes profiles/r03_figure_sanity.json."""
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
FIG = "/root/reference/figs/middleburry2021.png"


def jet_table():
    """OpenCV's COLORMAP_JET as 256 RGB triples (piecewise-linear red / green / blue ramps)."""
    x = np.arange(256) / 255.0
    r = np.clip(1.5 - np.abs(4 * x - 3), 0, 1)
    g = np.clip(1.5 - np.abs(4 * x - 2), 0, 1)
    b = np.clip(1.5 - np.abs(4 * x - 1), 0, 1)
    return np.stack([r, g, b], axis=1) * 255.0


def main():
    from oracle import oracle as orc
    from conftest import load_bgr
    fig = np.asarray(Image.open(FIG).convert("RGB")).astype(np.float32)
    panel = fig[9:273, 494:971]                         # panel (a), right half: the disparity map
    tab = jet_table()
    d2 = ((panel[:, :, None, :] - tab[None, None, :, :]) ** 2).sum(axis=3)
    ref = d2.argmin(axis=2).astype(np.float32)          # 0..255 as the reference's normalisation wrote it
    dist = np.sqrt(d2.min(axis=2))
    on_map = dist < 40.0                                # pixels whose colour lies on the JET curve at all
    scene = "/root/reference/Middlebury/2021/artroom2"
    L, R = load_bgr(os.path.join(scene, "im0.png")), load_bgr(os.path.join(scene, "im1.png"))
    disp, _ = orc.match(L, R, orc.default_option(min_disparity=0, max_disparity=130))
    a = np.abs(disp)
    fin = np.isfinite(a)
    lo, hi = a[fin].min(), a[fin].max()
    u8 = np.where(fin, (a - lo) / (hi - lo) * 255.0, 0.0).astype(np.float32)
    small = np.asarray(Image.fromarray(u8).resize((panel.shape[1], panel.shape[0]), Image.BILINEAR))
    m = on_map
    from scipy.stats import spearmanr
    rho = float(spearmanr(ref[m].ravel(), small[m].ravel()).correlation)
    # the two normalisations need not share their end points (a single outlier moves the minimum or the maximum):
    # compare after the best affine map of one onto the other
    A = np.stack([small[m].ravel(), np.ones(int(m.sum()), np.float32)], axis=1)
    (alpha, beta), *_ = np.linalg.lstsq(A, ref[m].ravel(), rcond=None)
    fitted = alpha * small + beta
    close = float((np.abs(ref[m] - fitted[m]) <= 0.08 * 255).mean())
    pearson = float(np.corrcoef(ref[m].ravel(), small[m].ravel())[0, 1])
    out = {"figure": "figs/middleburry2021.png panel (a) (artroom2), %dx%d pixels" % (panel.shape[1], panel.shape[0]),
           "oracle": "artroom2 1920x1080, disparities [0,130), defaults; normalised as ADC/main.cpp:443-476 and reduced to the panel",
           "pixels_on_jet_curve": float(m.mean()), "spearman_rho": rho, "pearson_r": pearson,
           "affine_fit": {"scale": float(alpha), "offset": float(beta)},
           "share_within_8pct_of_range_after_fit": close,
           "mean_abs_difference_of_255_after_fit": float(np.abs(ref[m] - fitted[m]).mean()),
           "note": "sanity against gross errors only; the shipped binary also carries the literal defects of the deviation "
                   "register (two-direction scanline, macro mis-expansions), so agreement is expected to be loose"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_figure_sanity.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
