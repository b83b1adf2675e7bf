"""Known-answer tests that pin the CPU oracle (the reference ships no tests or golden vectors).

Each case is small enough to check by hand or against an independent pure-Python restatement of the
reference formula written here (different code from oracle/adcensus_oracle.c).
Reference files: final_version_project/Stereo_Matching/ADCensus_CUDA/ (ADC/).
"""
import numpy as np
import pytest

from conftest import middlebury2003

INF = np.float32(np.inf)
LARGE = np.float32(99999.0)


def rand_img(seed, H, W, smooth=False):
    rng = np.random.default_rng(seed)
    if smooth:
        base = rng.integers(0, 256, size=(H // 4 + 2, W // 4 + 2, 3))
        img = np.kron(base, np.ones((4, 4, 1)))[:H, :W] + rng.integers(-3, 4, size=(H, W, 3))
        return np.clip(img, 0, 255).astype(np.uint8)
    return rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)


# ---- gray (ADC/cost_computor.cu:83-86) ----------------------------------------------------------
def test_gray_formula(orc):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(256, 512, 3)).astype(np.uint8)
    img[0, 0] = (0, 0, 0)
    img[0, 1] = (255, 255, 255)
    img[0, 2] = (255, 0, 0)   # pure blue in BGR
    img[0, 3] = (0, 0, 255)   # pure red
    b, g, r = [img[..., i].astype(np.float64) for i in range(3)]
    want = ((r * 0.299 + g * 0.587) + b * 0.114).astype(np.uint8)  # numpy evaluates as written, truncation
    got = orc.gray(img)
    assert np.array_equal(got, want)
    assert got[0, 0] == 0 and got[0, 2] == 29 and got[0, 3] == 76
    assert got[0, 1] in (254, 255)


# ---- census (ADC/census_cost.cu:46-97) ------------------------------------------------------------
def census_py(g):
    H, W = g.shape
    G = lambda y, x: int(g[y, x]) if 0 <= y < H and 0 <= x < W else 0
    out = np.zeros((H, W), np.uint64)
    for y in range(H):
        for x in range(W):
            c = 0
            for k in range(3):
                for m in range(9):
                    if G(y - 3 + k, x - 4 + m) >= G(y + 3 - k, x + 4 - m):
                        c |= 1 << (k * 9 + m)
            for m in range(4):
                if G(y, x - 4 + m) >= G(y, x + 4 - m):
                    c |= 1 << (27 + m)
            out[y, x] = c
    return out


def test_census_constant_and_dot(orc):
    g = np.full((20, 24), 77, np.uint8)
    c = orc.census(g)
    assert c[10, 12] == 0x7FFFFFFF            # interior of a flat image: every comparison is >=
    assert np.all(c < (1 << 31))
    # a single bright dot clears exactly the bit whose mirrored partner it is
    g = np.zeros((20, 24), np.uint8)
    g[12, 14] = 200                           # dot at (+2 rows, +2 cols) from the centre (10,12)
    c = orc.census(g)
    # partner of (dy=+2, dx=+2) is (dy=-2, dx=-2): k = 1, m = 2 -> bit 11
    assert c[10, 12] == 0x7FFFFFFF & ~(1 << 11)
    # dot in the upper half changes nothing (e1 > e2 is still >=)
    g[:] = 0
    g[8, 10] = 200
    assert orc.census(g)[10, 12] == 0x7FFFFFFF


def test_census_vs_python(orc):
    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, size=(13, 17)).astype(np.uint8)
    assert np.array_equal(orc.census(g), census_py(g))


# ---- initial cost (ADC/cost_computor.cu:165-205) ----------------------------------------------------
def test_cost_identical_and_border(orc):
    img = rand_img(5, 12, 40)
    cen = orc.census(orc.gray(img))
    c = orc.cost_init(img, img, cen, cen, 0, 8)
    assert np.all(c[:, :, 0] == 0.0)                       # same pixel, same census
    for d in range(1, 8):
        assert np.all(c[:, :d, d] == 1.0)                  # x - d < 0
    # one value by the formula, in float32, left to right
    y, x, d = 5, 20, 3
    s = int(np.abs(img[y, x].astype(int) - img[y, x - d].astype(int)).sum())
    h = bin(int(cen[y, x]) ^ int(cen[y, x - d])).count("1")
    ad = np.float32(s) / np.float32(3.0)
    e1 = np.float32(np.exp(np.float64(-ad / np.float32(10))))
    e2 = np.float32(np.exp(np.float64(-np.float32(h) / np.float32(30))))
    want = (np.float32(1) - e1 + np.float32(1)) - e2
    assert c[y, x, d] == want


def test_cost_min_disparity_offset(orc):
    img = rand_img(6, 8, 30)
    cen = orc.census(orc.gray(img))
    a = orc.cost_init(img, img, cen, cen, 0, 10)
    b = orc.cost_init(img, img, cen, cen, 4, 10)
    assert np.array_equal(a[:, :, 4:], b)


# ---- cross arms (ADC/cost_aggregator.cu:130-260) -----------------------------------------------------
def arms_py(img, L1=34, L2=17, t1=20, t2=6):
    H, W, _ = img.shape
    I = img.astype(int)
    cd = lambda p, q: int(np.abs(p - q).max())
    out = np.zeros((H, W, 4), np.uint8)
    for y in range(H):
        for x in range(W):
            for idx, (dx, dy) in enumerate([(-1, 0), (1, 0), (0, -1), (0, 1)]):
                arm, last = 0, I[y, x]
                xn, yn = x + dx, y + dy
                for n in range(min(L1, 255)):
                    if not (0 <= xn < W and 0 <= yn < H):
                        break
                    p = I[yn, xn]
                    d1 = cd(p, I[y, x])
                    if d1 >= t1:
                        break
                    if n > 0 and cd(p, last) >= t1:
                        break
                    if n + 1 > L2 and d1 >= t2:
                        break
                    arm += 1
                    last = p
                    xn += dx
                    yn += dy
                out[y, x, idx] = arm
    return out


def test_arms_flat_step_ramp(orc):
    flat = np.full((80, 90, 3), 100, np.uint8)
    a = orc.build_arms(flat)
    assert tuple(a[40, 45]) == (34, 34, 34, 34)
    assert tuple(a[0, 0]) == (0, 34, 0, 34)
    assert tuple(a[3, 5]) == (5, 34, 3, 34)
    assert tuple(a[79, 89]) == (34, 0, 34, 0)
    step = flat.copy()
    step[:, 50:] = 160                       # colour step of 60 >= t1
    a = orc.build_arms(step)
    assert a[40, 45, 1] == 4 and a[40, 50, 0] == 0 and a[40, 55, 0] == 5
    ramp = np.zeros((5, 120, 3), np.uint8)
    ramp[:, :, 1] = np.arange(120)           # +1 per pixel in one channel
    a = orc.build_arms(ramp)
    # right arm from x=10: d1 = n+1; the 18th pixel (n = 17) needs d1 < t2 = 6 -> stops at 17
    assert a[2, 10, 1] == 17 and a[2, 60, 0] == 17
    ch = ramp.copy()
    ch[:, :, 1] = (np.arange(120) // 3) * 3  # plateaus: d1 grows by 3 every 3 px -> reaches 18 < 20 at n = 17
    assert orc.build_arms(ch)[2, 30, 1] == 17


def test_arms_vs_python(orc):
    img = rand_img(9, 24, 40, smooth=True)
    assert np.array_equal(orc.build_arms(img), arms_py(img))
    assert np.array_equal(orc.build_arms(img, 9, 4, 30, 10), arms_py(img, 9, 4, 30, 10))


# ---- support counts (ADC/cost_aggregator.cu:262-316) ----------------------------------------------------
def test_sup_counts(orc):
    flat = np.full((120, 130, 3), 9, np.uint8)
    arms = orc.build_arms(flat)
    ch, cv = orc.sup_counts(arms)
    assert ch[60, 65] == 69 * 69 and cv[60, 65] == 69 * 69
    assert ch[0, 0] == 35 * 35
    img = rand_img(11, 30, 36, smooth=True)
    arms = orc.build_arms(img).astype(int)
    ch, cv = orc.sup_counts(arms.astype(np.uint8))
    H, W = 30, 36
    for (y, x) in [(0, 0), (7, 9), (29, 35), (15, 20)]:
        l, r, t, b = arms[y, x]
        assert ch[y, x] == sum(arms[y + k, x, 0] + arms[y + k, x, 1] + 1 for k in range(-t, b + 1))
        assert cv[y, x] == sum(arms[y, x + k, 2] + arms[y, x + k, 3] + 1 for k in range(-l, r + 1))


# ---- aggregation (ADC/cost_aggregator.cu:329-392,440-454) ------------------------------------------------
def aggregate_py(cost, arms, ch, cv, iters=4):
    H, W, D = cost.shape
    A = cost.astype(np.float32).copy()
    arms = arms.astype(int)
    for it in range(iters):
        hfirst = it % 2 == 0
        T = np.zeros_like(A)
        for y in range(H):
            for x in range(W):
                l, r, t, b = arms[y, x]
                acc = np.zeros(D, np.float32)
                rng_ = [(y, x + k) for k in range(-l, r + 1)] if hfirst else [(y + k, x) for k in range(-t, b + 1)]
                for (yy, xx) in rng_:
                    acc = (acc + A[yy, xx]).astype(np.float32)
                T[y, x] = acc
        O = np.zeros_like(A)
        for y in range(H):
            for x in range(W):
                l, r, t, b = arms[y, x]
                acc = np.zeros(D, np.float32)
                rng_ = [(y + k, x) for k in range(-t, b + 1)] if hfirst else [(y, x + k) for k in range(-l, r + 1)]
                for (yy, xx) in rng_:
                    acc = (acc + T[yy, xx]).astype(np.float32)
                O[y, x] = acc / np.float32((ch if hfirst else cv)[y, x])
        A = O
    return A


def test_aggregate_constant_and_python(orc):
    img = rand_img(13, 20, 26, smooth=True)
    arms = orc.build_arms(img)
    ch, cv = orc.sup_counts(arms)
    ones = np.ones((20, 26, 3), np.float32)
    assert np.array_equal(orc.aggregate(ones, arms, ch, cv), ones)
    rng = np.random.default_rng(14)
    cost = rng.random((20, 26, 3), dtype=np.float32) * 2
    got = orc.aggregate(cost, arms, ch, cv)
    want = aggregate_py(cost, arms, ch, cv)
    assert np.array_equal(got, want)


# ---- scanline (ADC/scanline_optimizer.cu:68-140) ------------------------------------------------------------
def scan_py(L, R, src, dmin, dmax, p1, p2, tso, vertical, forward):
    H, W, _ = L.shape
    D = dmax - dmin
    f = np.float32
    Li, Ri = L.astype(int), R.astype(int)
    cd = lambda p, q: int(np.abs(p - q).max())
    P1t = [f(p1), f(p1) / f(4), f(p1) / f(10)]
    P2t = [f(p2), f(p2) / f(4), f(p2) / f(10)]
    dst = np.zeros_like(src)
    d = 1 if forward else -1
    lines = range(W) if vertical else range(H)
    for line in lines:
        n = H if vertical else W
        pos = [(i if forward else n - 1 - i) for i in range(n)]
        pts = [(p, line) if vertical else (line, p) for p in pos]
        y0, x0 = pts[0]
        dst[y0, x0] = src[y0, x0]
        last = np.concatenate([[LARGE], dst[y0, x0], [LARGE]]).astype(np.float32)
        minlast = min(LARGE, last.min())
        for i in range(1, n):
            y, x = pts[i]
            py, px = pts[i - 1]
            d1 = cd(Li[y, x], Li[py, px])
            d2 = d1
            mn = LARGE
            for k in range(D):
                xr = x - k - dmin
                if 0 < xr < W - 1:
                    prev = Ri[y - d, xr] if vertical else Ri[y, xr - d]
                    d2 = cd(Ri[y, xr], prev)
                sel = int(d1 >= tso) + int(d2 >= tso)
                l1 = last[k + 1]
                l2 = f(last[k] + P1t[sel])
                l3 = f(last[k + 2] + P1t[sel])
                l4 = f(minlast + P2t[sel])
                cs = f(f(src[y, x, k] + min(l1, l2, l3, l4)) / f(2))
                dst[y, x, k] = cs
                mn = min(mn, cs)
            minlast = mn
            last[1:D + 1] = dst[y, x]
    return dst


def test_scanline_toy_by_hand(orc):
    # one row, three pixels, three disparities, flat colours (d1 = d2 = 0 -> P1 = 1, P2 = 3)
    L = np.full((1, 3, 3), 50, np.uint8)
    src = np.array([[[1.0, 0.2, 0.9], [0.5, 0.6, 0.1], [0.3, 0.3, 0.3]]], np.float32)
    out = orc.scanline_pass(L, L, src, 0, 3, vertical=False, forward=True)
    f = np.float32
    assert np.array_equal(out[0, 0], src[0, 0])
    # pixel 1: Lp = [1.0, 0.2, 0.9], min = 0.2
    want1 = [f((f(0.5) + min(f(1.0), f(0.2) + f(1), f(0.2) + f(3))) / f(2)),          # l2 = 99999+1
             f((f(0.6) + min(f(0.2), f(1.0) + f(1), f(0.9) + f(1), f(0.2) + f(3))) / f(2)),
             f((f(0.1) + min(f(0.9), f(0.2) + f(1), f(0.2) + f(3))) / f(2))]
    assert np.array_equal(out[0, 1], np.array(want1, np.float32))
    assert out[0, 1, 1] == f(0.4)


@pytest.mark.parametrize("vertical,forward", [(False, True), (False, False), (True, True), (True, False)])
@pytest.mark.parametrize("dmin", [0, 2])
def test_scanline_vs_python(orc, vertical, forward, dmin):
    rng = np.random.default_rng(21)
    L = rand_img(22, 7, 12, smooth=True)
    R = rand_img(23, 7, 12, smooth=True)
    src = rng.random((7, 12, 6), dtype=np.float32) * 2
    got = orc.scanline_pass(L, R, src, dmin, dmin + 6, 1.0, 3.0, 15, vertical, forward)
    want = scan_py(L, R, src, dmin, dmin + 6, 1.0, 3.0, 15, vertical, forward)
    assert np.array_equal(got, want)


# ---- WTA (ADC/ADCensusStereo.cu:200-267,358-424) ----------------------------------------------------------------
def test_wta_left_right_by_hand(orc):
    f = np.float32
    c = np.full((1, 6, 4), 1.0, np.float32)
    c[0, 2] = [0.9, 0.2, 0.6, 0.8]       # best d=1: 1 + (0.9-0.6)/((0.9+0.6-0.4)*2)
    c[0, 3] = [0.1, 0.5, 0.5, 0.5]       # best at d=0 -> invalid
    c[0, 4] = [0.5, 0.5, 0.5, 0.1]       # best at d=3 -> invalid
    c[0, 5] = [0.7, 0.3, 0.3, 0.9]       # tie: first minimum wins (d=1)
    dl = orc.wta_left(c, 0, 4)
    assert dl[0, 2] == f(1) + (f(0.9) - f(0.6)) / ((f(0.9) + f(0.6) - 2 * f(0.2)) * f(2))
    assert np.isinf(dl[0, 3]) and np.isinf(dl[0, 4])
    assert dl[0, 5] == f(1) + (f(0.7) - f(0.3)) / ((f(0.7) + f(0.3) - 2 * f(0.3)) * f(2))
    assert np.isinf(dl[0, 0])            # all equal: strict '>' keeps d = 0 -> invalid
    dr = orc.wta_right(c, 0, 4)
    # right pixel 2: candidates cost(2+d, d) = [0.9, 0.5, 0.5, 0.9] -> first minimum d=1
    assert dr[0, 2] == f(1) + (f(0.9) - f(0.5)) / ((f(0.9) + f(0.5) - 2 * f(0.5)) * f(2))
    # right pixel 1: candidates [1.0, 0.2, 0.5, 0.1] -> best at the last disparity -> written as the integer
    assert dr[0, 1] == 3.0
    # right pixel 5: only d=0 in range -> best stays 0 -> written as the integer
    assert dr[0, 5] == 0.0
    # right pixel 4: candidates d=0 (0.5), d=1 (0.3); d=2,3 out of range hold Large_Float
    assert dr[0, 4] == f(1) + (f(0.5) - LARGE) / ((f(0.5) + LARGE - 2 * f(0.3)) * f(2))


# ---- outlier detection (ADC/multistep_refiner.cu:79-155) -------------------------------------------------------
def test_outlier_cases(orc):
    W = 12
    dl = np.full((1, W), 2.0, np.float32)
    dr = np.full((1, W), 2.0, np.float32)
    dl[0, 3] = INF            # already invalid -> mismatch
    dl[0, 0] = 2.0            # x - d = -2 -> outside -> mismatch, invalidated
    dl[0, 6] = 4.0            # -> right col 2, dr = 2 -> |4-2| > 1; col_rl = 4, dl[4] = 2 < 4 -> mismatch
    dl[0, 8] = 1.0            # -> right col 7, |1-2| = 1 not > 1 -> consistent
    dl[0, 9] = -0.5           # negative -> mismatch
    dr[0, 8] = 5.0
    dl[0, 10] = 2.0           # -> right col 8, dr = 5: col_rl = 13 >= W -> mismatch
    dl[0, 5] = 0.4            # -> right col 5 (lround(4.6)), dr = 2: col_rl = 7, dl[7] = 2 > 0.4 -> occlusion
    out, occ, mis = orc.outlier(dl, dr, 1.0)
    assert list(np.nonzero(mis[0])[0]) == [0, 1, 3, 6, 9, 10]
    assert list(np.nonzero(occ[0])[0]) == [5]
    assert np.all(np.isinf(out[0, [0, 1, 3, 5, 6, 9, 10]]))
    assert out[0, 8] == 1.0 and out[0, 4] == 2.0


# ---- region voting (ADC/multistep_refiner.cu:383-435) --------------------------------------------------------------
def test_region_voting(orc):
    H, W = 21, 21
    flat = np.full((H, W, 3), 10, np.uint8)
    arms = orc.build_arms(flat, 3, 17, 20, 6)      # arms of 3 -> 7x7 regions in the interior
    disp = np.full((H, W), 5.4, np.float32)
    disp[8:13, 8:13] = 9.0
    mask = np.zeros((H, W), np.uint8)
    disp[10, 10] = INF
    mask[10, 10] = 1
    disp[2, 2] = INF                                 # not in the mask: untouched
    out, m = orc.region_voting_pass(disp, mask, arms, 0, 16, 20, 0.4)
    # region 7x7 = 49 px: 24 valid at 9.0 (5x5 minus centre), 24 at 5.4, centre invalid -> first max wins: bin 5
    assert out[10, 10] == 5.0 and m[10, 10] == 0 and np.isinf(out[2, 2])
    disp[8:13, 8:13] = 9.0
    disp[7, 7:14] = 9.0
    disp[10, 10] = INF
    out, m = orc.region_voting_pass(disp, mask, arms, 0, 16, 20, 0.4)
    assert out[10, 10] == 9.0                        # 31 votes of 48 > 0.4
    out, m = orc.region_voting_pass(disp, mask, arms, 0, 16, 48, 0.4)
    assert np.isinf(out[10, 10]) and m[10, 10] == 1  # count 48 is not > tau_s = 48
    out, m = orc.region_voting_pass(disp, mask, arms, 0, 16, 20, 0.7)
    assert np.isinf(out[10, 10])                     # 31/48 = 0.646 not > 0.7


# ---- interpolation (ADC/multistep_refiner.cu:547-631) ------------------------------------------------------------------
def test_interpolation(orc):
    H, W = 9, 15
    img = np.zeros((H, W, 3), np.uint8)
    img[:, :7] = 10
    img[:, 7:] = 200
    disp = np.full((H, W), INF, np.float32)
    disp[4, 2] = 3.0       # along -x from the centre (direction s = 16 is not searched, but s=0.. covers +x only;
    disp[4, 12] = 7.0      # -x is reached through angle ~pi at s = 15 only approximately) -> use +x / +y hits
    disp[8, 7] = 5.0       # straight down (s = 8)
    occ = np.zeros((H, W), np.uint8)
    mis = np.zeros((H, W), np.uint8)
    mis[4, 7] = 1
    out = orc.interpolate(img, disp, occ, mis, 16)
    # collected: (4,12)=7.0 colour 200 (diff 0) and (8,7)=5.0 colour 200 -> first smallest diff wins: 7.0
    assert out[4, 7] == 7.0
    mis[4, 7] = 0
    occ[4, 7] = 1
    out = orc.interpolate(img, disp, occ, mis, 16)
    assert out[4, 7] == 5.0                          # occlusion: minimum collected disparity
    out = orc.interpolate(img, disp, occ, mis, 4)    # max_search_length 4: m = 1..3 reaches neither
    assert np.isinf(out[4, 7])
    assert out[4, 2] == 3.0 and np.isinf(out[0, 0])  # unmasked pixels untouched


# ---- edges + adjustment (ADC/multistep_refiner.cu:161-180,692-737) -------------------------------------------------------
def test_edge_detect_and_adjust(orc):
    H, W, D = 7, 9, 8
    disp = np.full((H, W), 2.0, np.float32)
    disp[:, 5:] = 6.0
    e = orc.edge_detect(disp, 5.0)
    assert e[3, 4] == 1 and e[3, 5] == 1 and e[3, 2] == 0 and e[0, 4] == 0 and e[3, 0] == 0
    d2 = disp.copy()
    d2[3, 3] = INF
    d2[3, 5] = INF
    e2 = orc.edge_detect(d2, 5.0)
    assert e2[3, 4] == 0 and e2[2, 2] == 1           # +inf and -inf in one sum -> NaN -> no edge; one-sided inf -> edge
    cost = np.full((H, W, D), 1.0, np.float32)
    cost[3, 4, 2] = 0.9                              # own cost
    cost[3, 5, 6] = 0.5                              # right neighbour's cost at its own disparity
    out = orc.edge_adjust(disp, cost, e, 0, D)
    assert out[3, 4] == 6.0
    cost[3, 3, 2] = 0.4
    out = orc.edge_adjust(disp, cost, e, 0, D)
    assert out[3, 4] == 2.0                          # left neighbour wins first (0.4), right (0.5) is not lower
    assert out[3, 5] == 6.0                          # snapshot semantics: neighbour 4 still reads as 2.0, cost 1.0 vs 0.5


# ---- median (ADC/multistep_refiner.cu:186-220) -----------------------------------------------------------------------------
def test_median(orc):
    rng = np.random.default_rng(31)
    d = rng.random((12, 14), dtype=np.float32) * 60
    d[rng.random((12, 14)) < 0.2] = INF
    out = orc.median3(d)
    assert np.array_equal(out[0], d[0]) and np.array_equal(out[:, -1], d[:, -1])
    for (y, x) in [(1, 1), (5, 7), (10, 12)]:
        assert out[y, x] == np.sort(d[y - 1:y + 2, x - 1:x + 2].ravel())[4]


# ---- whole pipeline against Middlebury 2003 ground truth -------------------------------------------------------------------
@pytest.mark.parametrize("name,expected", [("teddy", 0.07246222222222222), ("cones", 0.0801125925925926)])
def test_middlebury_bad2(orc, name, expected):
    L, R, gt = middlebury2003(name)
    disp, _ = orc.match(L, R, orc.default_option(max_disparity=64))
    bad, rms = orc.bad_fraction(disp, gt, 4.0, 2.0)
    assert bad < 0.12                                  # an AD-Census implementation lands well under this on 2003 scenes
    assert abs(bad - expected) < 1e-12                 # value recorded when the oracle was written


# ---- literal-defect switches (sensitivity study only; canonical = 0 is what every parity test uses) ----

def test_literal_macro_expansions(orc):
    # MAX(a, MAX(b, c)) with the unparenthesised macro (config.h:72): condition ((a > (b>c)) ? b : c) != 0
    assert orc.literal_max3(5, 10, 3) == 5          # true maximum would be 10
    assert orc.literal_max3(0, 7, 9) == 0           # (0 > (7>9)) false -> cond = c = 9, non-zero -> a
    for a in range(0, 6):
        for b in range(0, 6):
            for c in range(0, 6):
                cond = (b if a > int(b > c) else c)
                assert orc.literal_max3(a, b, c) == (a if cond else max(b, c))
    # MIN(MIN(l1,l2), MIN(l3,l4)) (config.h:71): min(l1,l2) unless the selected condition value is 0
    assert orc.literal_min4(2.0, 3.0, 0.5, 0.25) == 2.0      # l3, l4 ignored
    assert orc.literal_min4(3.0, 2.5, 0.5, 0.25) == 2.5
    assert orc.literal_min4(0.0, 3.0, 7.0, 5.0) == 5.0       # l1<l2 and l1 == 0 -> min(l3,l4)
    assert orc.literal_min4(3.0, 2.0, 0.0, 5.0) == 2.0       # l2 !< (l3<l4)=1 -> cond=l4=5 -> min(l1,l2)
    assert orc.literal_min4(3.0, 0.5, 0.0, 5.0) == 0.0       # l2 < 1 -> cond=l3=0 -> min(l3,l4)


def test_literal_zero_is_canonical_and_restores(orc):
    rng = np.random.default_rng(5)
    L = rng.integers(0, 256, (12, 20, 3), dtype=np.uint8)
    R = rng.integers(0, 256, (12, 20, 3), dtype=np.uint8)
    opt = orc.default_option(max_disparity=8)
    base, _ = orc.match(L, R, opt)
    with orc.literal(0):
        again, _ = orc.match(L, R, opt)
    assert np.array_equal(base.view(np.uint32), again.view(np.uint32))
    with orc.literal(orc.LIT_ALL):
        assert orc.lib().orc_get_literal() == orc.LIT_ALL
    assert orc.lib().orc_get_literal() == 0
    after, _ = orc.match(L, R, opt)
    assert np.array_equal(base.view(np.uint32), after.view(np.uint32))


def test_literal_horizontal_pass_only_copies_the_path_head(orc):
    rng = np.random.default_rng(6)
    H, W, D = 5, 9, 4
    L = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    R = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    src = rng.random((H, W, D), dtype=np.float32)
    held = np.full((H, W, D), -7.0, np.float32)

    def run(vertical, forward):
        import ctypes as C
        dst = held.copy()
        orc.lib().orc_scanline_pass(L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p),
                                    src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p),
                                    C.c_int(W), C.c_int(H), C.c_int(0), C.c_int(D), C.c_float(1.0), C.c_float(3.0),
                                    C.c_int(15), C.c_int(int(vertical)), C.c_int(int(forward)))
        return dst

    with orc.literal(orc.LIT_HPASS_RETURN):
        fwd, bwd, vert = run(False, True), run(False, False), run(True, True)
    assert np.array_equal(fwd[:, 0], src[:, 0]) and np.all(fwd[:, 1:] == -7.0)
    assert np.array_equal(bwd[:, -1], src[:, -1]) and np.all(bwd[:, :-1] == -7.0)
    assert np.array_equal(vert, orc.scanline_pass(L, R, src, 0, D, vertical=True, forward=True))  # vertical untouched


def test_literal_low_sentinel_changes_only_what_reads_it(orc):
    # one row, D=3, left->right: at k=0 the term Lp[-1]+P1 is P1 - 6.8e-20 = P1 instead of 99999+P1
    L = np.zeros((1, 3, 3), np.uint8)
    R = np.zeros((1, 3, 3), np.uint8)
    src = np.array([[[5.0, 6.0, 7.0], [1.0, 1.0, 1.0], [1.0, 1.0, 1.0]]], np.float32)
    canon = orc.scanline_pass(L, R, src, 0, 3)
    with orc.literal(orc.LIT_LOW_SENTINEL):
        lit = orc.scanline_pass(L, R, src, 0, 3)
    assert np.array_equal(canon[0, 0], lit[0, 0])
    # canonical x=1,k=0: min(5, 99999+1, 6+1, 5+3) = 5 -> (1+5)/2 = 3
    assert canon[0, 1, 0] == np.float32(3.0)
    # literal: initial min includes -6.8e-20 -> l4 = 3.0 (absorbed); l2 = 1.0 -> (1+1)/2 = 1
    assert lit[0, 1, 0] == np.float32(1.0)
    # k=2 does not read Lp[-1] but does read the polluted initial minimum: min(7, 6+1, 99999+1, -6.8e-20+3) = 3
    assert lit[0, 1, 2] == np.float32(2.0) and canon[0, 1, 2] == np.float32(4.0)


def test_literal_interpolation_return_abandons_the_row(orc):
    H, W = 4, 8
    inf = np.float32(np.inf)
    disp = np.full((H, W), inf, np.float32)
    disp[2, 7] = 3.0
    L = np.zeros((H, W, 3), np.uint8)
    mis = np.isinf(disp).astype(np.uint8)
    occ = np.zeros((H, W), np.uint8)
    canon = orc.interpolate(L, disp, occ, mis, 3)          # search length 3 -> m = 1, 2
    assert canon[2, 5] == 3.0 and canon[2, 6] == 3.0 and np.isinf(canon[2, :5]).all()
    with orc.literal(orc.LIT_INTERP_RETURN):
        lit = orc.interpolate(L, disp, occ, mis, 3)
    assert np.isinf(lit[2, :7]).all() and lit[2, 7] == 3.0  # (2,0) collected nothing -> the row's thread returned
    assert not np.isinf(canon[0, 6:]).any() and np.isinf(lit[0]).all()   # row 0 died at x=0 as well
