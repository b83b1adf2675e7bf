"""Seeded synthetic stereo pairs (frozen definition; used by the parity tests and by bench.py).

pair(seed, W, H, D) -> (left_bgr u8[H][W][3], right_bgr u8[H][W][3], gt_disp f32[H][W])

  left  : per channel, the mean of 4 octaves of bilinearly up-sampled uniform u8 noise (cell sizes
          64, 32, 16, 8 px) blended with a fine 2 px octave, plus 64 flat-colour rectangles
          (long cross arms), plus +-2 uniform noise;
  gt    : piecewise-planar, background plane plus 8 fronto-parallel or slanted rectangles with
          disparities in [4, 0.75*D];
  right : left forward-warped by round(gt) with a z-buffer (larger disparity wins), holes filled
          from the left neighbour, plus its own +-2 noise.
All random numbers come from numpy's PCG64 seeded with `seed`.
"""
import numpy as np


def _upsampled_noise(rng, H, W, cell):
    gh, gw = H // cell + 2, W // cell + 2
    g = rng.integers(0, 256, size=(gh, gw)).astype(np.float32)
    ys = np.arange(H, dtype=np.float32) / cell
    xs = np.arange(W, dtype=np.float32) / cell
    y0 = ys.astype(np.int32)
    x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def pair(seed, W, H, D):
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    left = np.empty((H, W, 3), np.float32)
    for ch in range(3):
        acc = np.zeros((H, W), np.float32)
        for cell in (64, 32, 16, 8):
            acc += _upsampled_noise(rng, H, W, cell)
        fine = _upsampled_noise(rng, H, W, 2)
        left[:, :, ch] = 0.8 * (acc / 4.0) + 0.2 * fine
    for _ in range(64):
        w = int(rng.integers(max(8, W // 40), max(16, W // 6)))
        h = int(rng.integers(max(8, H // 40), max(16, H // 6)))
        x = int(rng.integers(0, max(1, W - w)))
        y = int(rng.integers(0, max(1, H - h)))
        left[y:y + h, x:x + w, :] = rng.integers(0, 256, size=3).astype(np.float32)
    # ground-truth disparity: background + 8 planes
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    lo, hi = 4.0, max(5.0, 0.75 * D)
    gt = np.full((H, W), lo + 0.15 * (hi - lo), np.float32) + 0.1 * (hi - lo) * (yy / H)
    for _ in range(8):
        w = int(rng.integers(W // 8, W // 2))
        h = int(rng.integers(H // 8, H // 2))
        x = int(rng.integers(0, W - w))
        y = int(rng.integers(0, H - h))
        d0 = float(rng.uniform(lo, hi))
        sx = float(rng.uniform(-0.03, 0.03)) if rng.integers(0, 2) else 0.0
        sy = float(rng.uniform(-0.03, 0.03)) if rng.integers(0, 2) else 0.0
        plane = d0 + sx * (xx[y:y + h, x:x + w] - x) + sy * (yy[y:y + h, x:x + w] - y)
        gt[y:y + h, x:x + w] = np.clip(plane, lo, hi)
    left_u8 = np.clip(np.rint(left + rng.integers(-2, 3, size=left.shape)), 0, 255).astype(np.uint8)
    # forward warp with z-buffer
    d_int = np.rint(gt).astype(np.int32)
    xs = np.arange(W, dtype=np.int32)[None, :].repeat(H, 0)
    xr = xs - d_int
    right = np.zeros((H, W, 3), np.uint8)
    zbuf = np.full((H, W), -1, np.int32)
    order = np.argsort(d_int, axis=1, kind="stable")  # ascending disparity: nearer surfaces written last
    rows = np.arange(H)[:, None].repeat(W, 1)
    xs_o = np.take_along_axis(xs, order, 1)
    xr_o = np.take_along_axis(xr, order, 1)
    d_o = np.take_along_axis(d_int, order, 1)
    ok = xr_o >= 0
    for j in range(W):  # column by column of the sorted order keeps "larger disparity wins" deterministic
        m = ok[:, j]
        r = rows[m, j]
        c = xr_o[m, j]
        right[r, c] = left_u8[r, xs_o[m, j]]
        zbuf[r, c] = d_o[m, j]
    hole = zbuf < 0
    for x in range(1, W):  # fill holes from the left neighbour
        m = hole[:, x]
        right[m, x] = right[m, x - 1]
    right = np.clip(right.astype(np.int32) + rng.integers(-2, 3, size=right.shape), 0, 255).astype(np.uint8)
    # BGR order is what the matcher consumes; the generator is channel-symmetric
    return np.ascontiguousarray(left_u8), np.ascontiguousarray(right), gt
