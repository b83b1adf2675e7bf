/*
 * CPU oracle for the AD-Census matcher path -- TEST INFRASTRUCTURE ONLY (see adcensus_oracle.h).
 * PARITY UNPINNED by the reference itself (it has no tests or golden outputs); pinned by the
 * known-answer tests in tests/ and Middlebury 2003 ground truth.
 *
 * ADC/ = /root/reference/final_version_project/Stereo_Matching/ADCensus_CUDA/
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fno-fast-math (see oracle/Makefile). Contraction must
 * stay off: the gray conversion and the float expressions below are evaluated exactly as written.
 */
#include "adcensus_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_LARGE 99999.0f            /* Large_Float, ADC/adcensus_types.h:54 */
#define ORC_INVALID ((float)INFINITY) /* Invalid_Float, ADC/adcensus_types.h:52 */
#define ORC_MAX_ARM 255               /* MAX_ARM_LENGTH, ADC/cross_aggregator.h:17 */

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Thread count of the calls that follow (the CPU baseline is timed at all cores and at one). */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* Defaults of ADCensusOption(), ADC/adcensus_types.h:86-93 with ADC/config.h:12-13,26-29. */
void orc_default_option(orc_option *o) {
    memset(o, 0, sizeof(*o));
    o->min_disparity = 0;
    o->max_disparity = 64;
    o->lambda_ad = 10;
    o->lambda_census = 30;
    o->cross_L1 = 34;
    o->cross_L2 = 17;
    o->cross_t1 = 20;
    o->cross_t2 = 6;
    o->so_p1 = 1.0f;
    o->so_p2 = 3.0f;
    o->so_tso = 15;
    o->irv_ts = 20;
    o->irv_th = 0.4f;
    o->lrcheck_thres = 1.0f;
    o->do_lr_check = 1;
    o->do_filling = 1;
    o->do_discontinuity_adjustment = 1;
}

static inline int iabs_(int v) { return v < 0 ? -v : v; }
static inline int imax_(int a, int b) { return a > b ? a : b; }
static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline float fmin_(float a, float b) { return a < b ? a : b; }

/* Literal-defect switches (sensitivity study only; 0 = canonical, the only mode the parity tests
 * and the CPU baseline use). */
static unsigned g_literal = 0;
void orc_set_literal(unsigned flags) { g_literal = flags; }
unsigned orc_get_literal(void) { return g_literal; }

/* What `MAX(a, MAX(b, c))` evaluates to with `#define MAX(a,b) (a>b)?(a):(b)` (ADC/config.h:72):
 * the text becomes (a>(b>c)?(b):(c))?(a):((b>c)?(b):(c)), i.e. the condition is
 * ((a > (b>c)) ? b : c) != 0 and the result is a when it holds, else max(b,c). */
int orc_literal_max3(int a, int b, int c) {
    const int cond = (a > (b > c)) ? b : c;
    return cond ? a : ((b > c) ? b : c);
}

/* What `MIN(MIN(l1,l2), MIN(l3,l4))` evaluates to with `#define MIN(a,b) (a<b)?(a):(b)`
 * (ADC/config.h:71): ((l1<l2)?(l1):(l2)<(l3<l4)?(l3):(l4)) ? min(l1,l2) : min(l3,l4); the condition
 * parses as (l1<l2) ? l1 : ((l2 < (float)(l3<l4)) ? l3 : l4), taken as a truth value. */
float orc_literal_min4(float l1, float l2, float l3, float l4) {
    const float cond = (l1 < l2) ? l1 : ((l2 < (float)(l3 < l4)) ? l3 : l4);
    return (cond != 0.0f) ? ((l1 < l2) ? l1 : l2) : ((l3 < l4) ? l3 : l4);
}

/* ColorDist, ADC/cross_aggregator.h:73-75: max over channels of |difference|. */
static inline int color_dist(const uint8_t *p, const uint8_t *q) {
    return imax_(iabs_((int)p[2] - (int)q[2]), imax_(iabs_((int)p[1] - (int)q[1]), iabs_((int)p[0] - (int)q[0])));
}

/* ComputeGrayCuda, ADC/cost_computor.cu:74-89: b=img[0], g=img[1], r=img[2];
 * gray = uint8(r*0.299 + g*0.587 + b*0.114) in double, evaluated left to right. */
void orc_gray(const uint8_t *bgr, int W, int H, uint8_t *gray) {
    const size_t n = (size_t)W * H;
#pragma omp parallel for
    for (long long i = 0; i < (long long)n; i++) {
        const uint8_t b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        const double v = ((double)r * 0.299 + (double)g * 0.587) + (double)b * 0.114;
        gray[i] = (uint8_t)v;
    }
}

/* CenterSymmetricCensusKernelSM2, ADC/census_cost.cu:9-98 with TOP=3, LEFT=4 (ADC/config.h:41-44).
 * Pixels outside the image read as 0 (:27-29). 27 bits from the 3 upper rows against their
 * point-mirrored partners, then 4 bits from the left half of the centre row. */
void orc_census(const uint8_t *gray, int W, int H, uint64_t *census) {
#define G_(yy, xx) (((yy) >= 0 && (yy) < H && (xx) >= 0 && (xx) < W) ? gray[(size_t)(yy) * W + (xx)] : (uint8_t)0)
#pragma omp parallel for
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            uint64_t c = 0;
            for (int k = 0; k < 3; k++) {
                for (int m = 0; m < 9; m++) {
                    const uint8_t e1 = G_(y - 3 + k, x - 4 + m);
                    const uint8_t e2 = G_(y + 3 - k, x + 4 - m);
                    c |= (uint64_t)(e1 >= e2) << (k * 9 + m);
                }
            }
            for (int m = 0; m < 4; m++) {
                const uint8_t e1 = G_(y, x - 4 + m);
                const uint8_t e2 = G_(y, x + 4 - m);
                c |= (uint64_t)(e1 >= e2) << (27 + m);
            }
            census[(size_t)y * W + x] = c;
        }
    }
#undef G_
}

/* ComputeCostCuda, ADC/cost_computor.cu:151-207, with the runtime disparity range.
 * cost = 1 - exp(-ad/lambda_ad) + 1 - exp(-hamming/lambda_census), evaluated left to right in
 * float; out-of-image right pixel => 1.0f (:177-180). exp() is the correctly rounded value
 * (float)exp((double)x) of the float argument, tabulated per integer SAD sum / Hamming distance. */
void orc_cost_init(const uint8_t *L, const uint8_t *R, const uint64_t *cenL, const uint64_t *cenR,
                   int W, int H, int dmin, int dmax, int lambda_ad, int lambda_census, float *cost) {
    const int D = dmax - dmin;
    float e_ad[766], e_cen[65];
    for (int s = 0; s < 766; s++) {
        const float ad = (float)s / 3.0f;
        const float arg = -ad / (float)lambda_ad;
        e_ad[s] = (float)exp((double)arg);
    }
    for (int h = 0; h < 65; h++) {
        const float arg = -(float)h / (float)lambda_census;
        e_cen[h] = (float)exp((double)arg);
    }
#pragma omp parallel for
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const uint8_t *pl = L + ((size_t)y * W + x) * 3;
            const uint64_t cl = cenL[(size_t)y * W + x];
            float *c = cost + ((size_t)y * W + x) * D;
            for (int d = dmin; d < dmax; d++) {
                const int xr = x - d;
                if (xr < 0 || xr >= W) {
                    c[d - dmin] = 1.0f;
                    continue;
                }
                const uint8_t *pr = R + ((size_t)y * W + xr) * 3;
                const int s = iabs_((int)pl[0] - (int)pr[0]) + iabs_((int)pl[1] - (int)pr[1]) +
                              iabs_((int)pl[2] - (int)pr[2]);
                const int h = __builtin_popcountll(cl ^ cenR[(size_t)y * W + xr]);
                c[d - dmin] = ((1.0f - e_ad[s]) + 1.0f) - e_cen[h];
            }
        }
    }
}

/* FindHorizontalArm / FindVerticalArm, ADC/cost_aggregator.cu:130-260. */
static uint8_t find_arm(const uint8_t *L, int W, int H, int x, int y, int dx, int dy,
                        int L1, int L2, int t1, int t2) {
    const uint8_t *p0 = L + ((size_t)y * W + x) * 3;
    const uint8_t *last = p0;
    int xn = x + dx, yn = y + dy;
    int arm = 0;
    const int cap = imin_(L1, ORC_MAX_ARM);
    for (int n = 0; n < cap; n++) {
        if (xn < 0 || xn >= W || yn < 0 || yn >= H) break;
        const uint8_t *p = L + ((size_t)yn * W + xn) * 3;
        const int d1 = color_dist(p, p0);
        if (d1 >= t1) break;
        if (n > 0) {
            if (color_dist(p, last) >= t1) break;
        }
        if (n + 1 > L2) {
            if (d1 >= t2) break;
        }
        arm++;
        last = p;
        xn += dx;
        yn += dy;
    }
    return (uint8_t)arm;
}

/* BuildArms, ADC/cost_aggregator.cu:71-82; struct order {left,right,top,bottom}
 * (ADC/cross_aggregator.h:12-15). */
void orc_build_arms(const uint8_t *L, int W, int H, int L1, int L2, int t1, int t2, uint8_t *arms) {
#pragma omp parallel for
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            uint8_t *a = arms + ((size_t)y * W + x) * 4;
            a[0] = find_arm(L, W, H, x, y, -1, 0, L1, L2, t1, t2);
            a[1] = find_arm(L, W, H, x, y, +1, 0, L1, L2, t1, t2);
            a[2] = find_arm(L, W, H, x, y, 0, -1, L1, L2, t1, t2);
            a[3] = find_arm(L, W, H, x, y, 0, +1, L1, L2, t1, t2);
        }
    }
}

/* ComputeSupPixelCount, ADC/cost_aggregator.cu:262-316. cnt_h: horizontal-first order
 * (sum over the vertical arm of the horizontal span lengths); cnt_v: vertical-first. */
void orc_sup_counts(const uint8_t *arms, int W, int H, uint16_t *cnt_h, uint16_t *cnt_v) {
#pragma omp parallel for
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const uint8_t *a = arms + ((size_t)y * W + x) * 4;
            int ch = 0, cv = 0;
            for (int t = -(int)a[2]; t <= (int)a[3]; t++) {
                const uint8_t *b = arms + ((size_t)(y + t) * W + x) * 4;
                ch += (int)b[0] + (int)b[1] + 1;
            }
            for (int t = -(int)a[0]; t <= (int)a[1]; t++) {
                const uint8_t *b = arms + ((size_t)y * W + x + t) * 4;
                cv += (int)b[2] + (int)b[3] + 1;
            }
            cnt_h[(size_t)y * W + x] = (uint16_t)ch;
            cnt_v[(size_t)y * W + x] = (uint16_t)cv;
        }
    }
}

/* Aggregate / AggregateInArms / AggregateInArms_cuda, ADC/cost_aggregator.cu:84-113,329-392,440-454.
 * In place on `cost`. Iteration i is horizontal-first when i is even. Per iteration pass 1 runs
 * over the whole image before pass 2 starts. Every sum starts at 0.0f and adds in ascending t
 * (:346-372); pass 2 divides by the support count of that order (:380-385). All disparities of a
 * pixel use the same arms, so the k loop is innermost here; each k still sees the reference's
 * exact addition order. */
void orc_aggregate(float *cost, const uint8_t *arms, const uint16_t *cnt_h, const uint16_t *cnt_v,
                   int W, int H, int D, int iters) {
    const size_t n = (size_t)W * H * D;
    float *tmp = (float *)malloc(n * sizeof(float));
    float *out = (float *)malloc(n * sizeof(float));
    if (!tmp || !out) { free(tmp); free(out); return; }
    for (int it = 0; it < iters; it++) {
        const int hfirst = (it % 2) == 0;
        /* pass 1: cost -> tmp; pass 2: tmp -> out */
        for (int pass = 0; pass < 2; pass++) {
            const float *src = pass == 0 ? cost : tmp;
            float *dst = pass == 0 ? tmp : out;
            const int horizontal = (pass == 0) ? hfirst : !hfirst;
            const uint16_t *cnt = hfirst ? cnt_h : cnt_v;
#pragma omp parallel for schedule(dynamic, 4)
            for (int y = 0; y < H; y++) {
                for (int x = 0; x < W; x++) {
                    const uint8_t *a = arms + ((size_t)y * W + x) * 4;
                    float *o = dst + ((size_t)y * W + x) * D;
                    for (int k = 0; k < D; k++) o[k] = 0.0f;
                    if (horizontal) {
                        for (int t = -(int)a[0]; t <= (int)a[1]; t++) {
                            const float *s = src + ((size_t)y * W + x + t) * D;
                            for (int k = 0; k < D; k++) o[k] += s[k];
                        }
                    } else {
                        for (int t = -(int)a[2]; t <= (int)a[3]; t++) {
                            const float *s = src + ((size_t)(y + t) * W + x) * D;
                            for (int k = 0; k < D; k++) o[k] += s[k];
                        }
                    }
                    if (pass == 1) {
                        const float c = (float)cnt[(size_t)y * W + x];
                        for (int k = 0; k < D; k++) o[k] = o[k] / c;
                    }
                }
            }
        }
        memcpy(cost, out, n * sizeof(float));
    }
    free(tmp);
    free(out);
}

/* One directional pass of ScanlineOptimizeLeftRight / ScanlineOptimizeUpDown,
 * ADC/scanline_optimizer.cu:4-290 (recurrence :68-140 and :215-287).
 *   first path pixel: dst = src (:51-52,:199-200)
 *   d1 = max-channel colour distance between consecutive LEFT pixels on the path (:73-74)
 *   d2 starts as d1 for each pixel and is overwritten, as d ascends, whenever 0 < xr < W-1 with
 *      xr = x - d - min_disparity, by the distance between RIGHT pixels xr and its predecessor on
 *      the path (:75-93,:222-242); otherwise it keeps the value from the previous d.
 *   (P1,P2) = (p1,p2) | (p1/4,p2/4) | (p1/10,p2/10) by how many of d1,d2 reach tso (:97-109)
 *   dst = (src + min(Lp[d], Lp[d-1]+P1, Lp[d+1]+P1, min(Lp)+P2)) / 2 (:112-121) with Lp[-1] = Lp[D]
 *      = Large_Float. */
static inline __attribute__((always_inline)) int path_dist(const uint8_t *p, const uint8_t *q, unsigned lit) {
    if (lit & ORC_LIT_MAX_MACRO)   /* argument order of ADC/scanline_optimizer.cu:73-74: channels 0,1,2 */
        return orc_literal_max3(iabs_((int)p[0] - (int)q[0]), iabs_((int)p[1] - (int)q[1]),
                                iabs_((int)p[2] - (int)q[2]));
    return color_dist(p, q);
}

static inline __attribute__((always_inline)) void
scanline_pass_body(const uint8_t *L, const uint8_t *R, const float *src, float *dst,
                   int W, int H, int dmin, int dmax, float p1, float p2, int tso,
                   int vertical, int forward, const unsigned lit) {
    const int D = dmax - dmin;
    const int dir = forward ? 1 : -1;
    const int nlines = vertical ? W : H;
    const int len = vertical ? H : W;
    const float P1t[3] = {p1, p1 / 4, p1 / 10};
    const float P2t[3] = {p2, p2 / 4, p2 / 10};
#pragma omp parallel
    {
        float *last = (float *)malloc(sizeof(float) * (size_t)(D + 2));
#pragma omp for schedule(dynamic, 4)
        for (int line = 0; line < nlines; line++) {
            int x = vertical ? line : (forward ? 0 : W - 1);
            int y = vertical ? (forward ? 0 : H - 1) : line;
            const float *s = src + ((size_t)y * W + x) * D;
            float *o = dst + ((size_t)y * W + x) * D;
            last[0] = ORC_LARGE;
            last[D + 1] = ORC_LARGE;
            float minlast = ORC_LARGE;
            if (lit & ORC_LIT_LOW_SENTINEL) {
                /* memset(p, Large_Float, D + 2) writes the byte 0x9F (99999 & 0xFF) over the first D+2
                 * BYTES (ADC/scanline_optimizer.cu:45,194); the memcpy that follows rewrites elements
                 * 1..D, so only element 0 keeps 0x9F9F9F9F (about -6.8e-20) for the whole path, and the
                 * initial minimum (:61-64) sees it. Element D+1 lies partly outside the allocation and
                 * is indeterminate in the shipped code; it stays Large_Float here. */
                const uint32_t bits = 0x9F9F9F9Fu;
                memcpy(&last[0], &bits, 4);
                minlast = fmin_(minlast, last[0]);
            }
            for (int k = 0; k < D; k++) {
                o[k] = s[k];
                last[k + 1] = o[k];
                minlast = fmin_(minlast, o[k]);
            }
            if ((lit & ORC_LIT_HPASS_RETURN) && !vertical)
                continue; /* `return;` as the first statement of the disparity loop, ADC/scanline_optimizer.cu:78:
                             the path head is copied and the thread exits; dst keeps what it held */
            const uint8_t *plast = L + ((size_t)y * W + x) * 3;
            for (int step = 1; step < len; step++) {
                if (vertical) y += dir; else x += dir;
                s = src + ((size_t)y * W + x) * D;
                o = dst + ((size_t)y * W + x) * D;
                const uint8_t *pc = L + ((size_t)y * W + x) * 3;
                const int d1 = path_dist(pc, plast, lit);
                int d2 = d1;
                float mn = ORC_LARGE;
                for (int k = 0; k < D; k++) {
                    const int xr = x - k - dmin;
                    if (xr > 0 && xr < W - 1) {
                        const uint8_t *rc = R + ((size_t)y * W + xr) * 3;
                        const uint8_t *rp = vertical ? R + ((size_t)(y - dir) * W + xr) * 3
                                                     : R + ((size_t)y * W + (xr - dir)) * 3;
                        d2 = path_dist(rc, rp, lit);
                    }
                    const int sel = (d1 >= tso) + (d2 >= tso);
                    const float P1 = P1t[sel], P2 = P2t[sel];
                    const float l1 = last[k + 1];
                    const float l2 = last[k] + P1;
                    const float l3 = last[k + 2] + P1;
                    const float l4 = minlast + P2;
                    float cs = s[k] + ((lit & ORC_LIT_MIN_MACRO) ? orc_literal_min4(l1, l2, l3, l4)
                                                                 : fmin_(fmin_(l1, l2), fmin_(l3, l4)));
                    cs /= 2;
                    o[k] = cs;
                    mn = fmin_(mn, cs);
                }
                minlast = mn;
                memcpy(last + 1, o, sizeof(float) * (size_t)D);
                plast = pc;
            }
        }
        free(last);
    }
}

void orc_scanline_pass(const uint8_t *L, const uint8_t *R, const float *src, float *dst,
                       int W, int H, int dmin, int dmax, float p1, float p2, int tso,
                       int vertical, int forward) {
    if (g_literal == 0)   /* the constant lets the compiler drop every literal branch from the canonical loop */
        scanline_pass_body(L, R, src, dst, W, H, dmin, dmax, p1, p2, tso, vertical, forward, 0u);
    else
        scanline_pass_body(L, R, src, dst, W, H, dmin, dmax, p1, p2, tso, vertical, forward, g_literal);
}

/* ScanlineOptimizer::Optimize, ADC/scanline_optimizer.cu:380-447: L->R aggr->init, R->L init->aggr,
 * U->D aggr->init, D->U init->aggr. Result in `aggr`. */
void orc_scanline(const uint8_t *L, const uint8_t *R, float *aggr, float *init,
                  int W, int H, int dmin, int dmax, float p1, float p2, int tso) {
    orc_scanline_pass(L, R, aggr, init, W, H, dmin, dmax, p1, p2, tso, 0, 1);
    orc_scanline_pass(L, R, init, aggr, W, H, dmin, dmax, p1, p2, tso, 0, 0);
    orc_scanline_pass(L, R, aggr, init, W, H, dmin, dmax, p1, p2, tso, 1, 1);
    orc_scanline_pass(L, R, init, aggr, W, H, dmin, dmax, p1, p2, tso, 1, 0);
}

static inline float subpixel(float best, float c1, float c2, float cmin) {
    const float denom = c1 + c2 - 2 * cmin;
    if (denom != 0.0f) return best + (c1 - c2) / (denom * 2.0f);
    return best;
}

/* ComputeDisparityCuda, ADC/ADCensusStereo.cu:200-267 (uniqueness check compiled out). */
void orc_wta_left(const float *cost, int W, int H, int dmin, int dmax, float *disp) {
    const int D = dmax - dmin;
#pragma omp parallel for
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const float *c = cost + ((size_t)y * W + x) * D;
            float mc = ORC_LARGE;
            int best = 0;
            for (int d = dmin; d < dmax; d++) {
                if (mc > c[d - dmin]) { mc = c[d - dmin]; best = d; }
            }
            float out;
            if (best == dmin || best == dmax - 1) out = ORC_INVALID;
            else out = subpixel((float)best, c[best - 1 - dmin], c[best + 1 - dmin], mc);
            disp[(size_t)y * W + x] = out;
        }
    }
}

/* ComputeDisparityRight, ADC/ADCensusStereo.cu:358-424: cost_R(x,d) = cost_L(x+d,d); candidates
 * whose left column falls outside the image hold Large_Float and never win; a best disparity at
 * either end of the range is written as that integer (:404-407). The scratch row is private to the
 * pixel here. */
void orc_wta_right(const float *cost, int W, int H, int dmin, int dmax, float *disp) {
    const int D = dmax - dmin;
#pragma omp parallel
    {
        float *loc = (float *)malloc(sizeof(float) * (size_t)D);
#pragma omp for
        for (int y = 0; y < H; y++) {
            for (int x = 0; x < W; x++) {
                float mc = ORC_LARGE;
                int best = 0;
                for (int d = dmin; d < dmax; d++) {
                    const int k = d - dmin;
                    const int xl = x + d;
                    if (xl >= 0 && xl < W) {
                        const float c = loc[k] = cost[((size_t)y * W + xl) * D + k];
                        if (mc > c) { mc = c; best = d; }
                    } else {
                        loc[k] = ORC_LARGE;
                    }
                }
                float out;
                if (best == dmin || best == dmax - 1 || best - 1 - dmin < 0 || best + 1 - dmin >= D)
                    out = (float)best; /* last two guards only bite when no candidate was in range and dmin != 0 */
                else
                    out = subpixel((float)best, loc[best - 1 - dmin], loc[best + 1 - dmin], mc);
                disp[(size_t)y * W + x] = out;
            }
        }
        free(loc);
    }
}

static inline long lround_f(float v) { return lroundf(v); }

/* OutlierDetectionCuda, ADC/multistep_refiner.cu:70-156 (MinDispMismatch = 0, ADC/config.h:4).
 * disp_l is read-only here (snapshot); results go to disp_out. */
void orc_outlier(const float *disp_l, const float *disp_r, int W, int H, float thres,
                 float *disp_out, uint8_t *occ, uint8_t *mis) {
#pragma omp parallel for
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const size_t i = (size_t)y * W + x;
            occ[i] = 0;
            mis[i] = 0;
            const float d = disp_l[i];
            disp_out[i] = d;
            if (d < 0) { disp_out[i] = ORC_INVALID; mis[i] = 1; continue; }
            if (d == ORC_INVALID) { mis[i] = 1; continue; }
            const long cr = lround_f((float)x - d);
            if (cr >= 0 && cr < W) {
                const float dr = disp_r[(size_t)y * W + cr];
                if (fabsf(d - dr) > thres) {
                    const long crl = lround_f((float)cr + dr);
                    if (crl > 0 && crl < W) {
                        const float dl = disp_l[(size_t)y * W + crl];
                        if (dl > d) occ[i] = 1; else mis[i] = 1;
                    } else {
                        mis[i] = 1;
                    }
                    disp_out[i] = ORC_INVALID;
                }
            } else {
                disp_out[i] = ORC_INVALID;
                mis[i] = 1;
            }
        }
    }
}

/* One RegionVotingCuda + DeleteFilledPixel pass, ADC/multistep_refiner.cu:376-449. Reads disp_in,
 * writes disp_out (all pixels), clears mask where the output is valid. Disparities whose rounded
 * value falls outside [dmin,dmax) are not counted (the reference would index out of bounds). */
void orc_region_voting_pass(const float *disp_in, float *disp_out, uint8_t *mask,
                            const uint8_t *arms, int W, int H, int dmin, int dmax,
                            int tau_s, float tau_h) {
    const int D = dmax - dmin;
#pragma omp parallel
    {
        int *hist = (int *)malloc(sizeof(int) * (size_t)D);
#pragma omp for schedule(dynamic, 4)
        for (int y = 0; y < H; y++) {
            for (int x = 0; x < W; x++) {
                const size_t i = (size_t)y * W + x;
                float d = disp_in[i];
                if (mask[i] && d == ORC_INVALID) {
                    memset(hist, 0, sizeof(int) * (size_t)D);
                    const uint8_t *a = arms + i * 4;
                    for (int t = -(int)a[2]; t <= (int)a[3]; t++) {
                        const int yt = y + t;
                        const uint8_t *a2 = arms + ((size_t)yt * W + x) * 4;
                        for (int s = -(int)a2[0]; s <= (int)a2[1]; s++) {
                            const float dd = disp_in[(size_t)yt * W + x + s];
                            if (dd != ORC_INVALID) {
                                const long di = lround_f(dd) - dmin;
                                if (di >= 0 && di < D) hist[di]++;
                            }
                        }
                    }
                    int best = 0, count = 0, maxh = 0;
                    for (int k = 0; k < D; k++) {
                        const int h = hist[k];
                        if (maxh < h) { maxh = h; best = k; }
                        count += h;
                    }
                    if (maxh > 0) {
                        if (count > tau_s && maxh * 1.0f / count > tau_h) d = best + (float)dmin;
                    }
                }
                disp_out[i] = d;
            }
        }
        free(hist);
    }
    const size_t n = (size_t)W * H;
    for (size_t i = 0; i < n; i++)
        if (disp_out[i] != ORC_INVALID) mask[i] = 0;
}

/* Direction table of FillDisp, ADC/multistep_refiner.cu:561-583: ang starts at 0.0 (double) and is
 * increased 15 times by PI/16 with PI = 3.1415926f (ADC/adcensus_types.h:34), i.e. by the float
 * quotient converted to double; sin/cos are taken in double. */
static void interp_dirs(double *sina, double *cosa) {
    double ang = 0.0;
    const float step = 3.1415926f / 16;
    for (int s = 0; s < 16; s++) {
        sina[s] = sin(ang);
        cosa[s] = cos(ang);
        ang += step;
    }
}

/* ProperInterpolation / FillDisp, ADC/multistep_refiner.cu:543-683. max_search_length =
 * |max_disparity| * MaxSearchLengthTimes(1) (:642); m runs 1..max_search_length-1 (:566). A pixel
 * that collects nothing is left unchanged. */
void orc_interpolate(const uint8_t *L, const float *disp_in, float *disp_out,
                     const uint8_t *occ, const uint8_t *mis, int W, int H, int dmax) {
    double sina[16], cosa[16];
    interp_dirs(sina, cosa);
    const int maxlen = iabs_(dmax) * 1;
    const int row_return = (g_literal & ORC_LIT_INTERP_RETURN) != 0;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; y++) {
        int dead = 0; /* literal mode: the row's thread has returned (ADC/multistep_refiner.cu:586-588) */
        for (int x = 0; x < W; x++) {
            const size_t i = (size_t)y * W + x;
            disp_out[i] = disp_in[i];
            if (dead) continue;
            if (!(occ[i] || mis[i])) continue;
            float cd[16];
            size_t cidx[16];
            int num = 0;
            for (int s = 0; s < 16; s++) {
                for (int m = 1; m < maxlen; m++) {
                    const long yy = lround((double)y + (double)m * sina[s]);
                    const long xx = lround((double)x + (double)m * cosa[s]);
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) break;
                    const float d = disp_in[(size_t)yy * W + xx];
                    if (d != ORC_INVALID) {
                        cd[num] = d;
                        cidx[num] = ((size_t)yy * W + xx) * 3;
                        num++;
                        break;
                    }
                }
            }
            if (!num) { dead = row_return; continue; }
            if (mis[i]) {
                int mind = 9999;
                float d = 0.0f;
                const uint8_t *p = L + i * 3;
                for (int j = 0; j < num; j++) {
                    const uint8_t *q = L + cidx[j];
                    const int diff = iabs_((int)p[0] - (int)q[0]) + iabs_((int)p[1] - (int)q[1]) +
                                     iabs_((int)p[2] - (int)q[2]);
                    if (mind > diff) { mind = diff; d = cd[j]; }
                }
                disp_out[i] = d;
            } else {
                float md = ORC_LARGE;
                for (int j = 0; j < num; j++) md = fmin_(cd[j], md);
                disp_out[i] = md;
            }
        }
    }
}

/* EdgeDetectCuda, ADC/multistep_refiner.cu:161-180, on a zero-filled mask. inf - inf = NaN compares
 * false against the threshold. */
void orc_edge_detect(const float *disp, int W, int H, float thres, uint8_t *edge) {
    memset(edge, 0, (size_t)W * H);
#pragma omp parallel for
    for (int y = 1; y < H - 1; y++) {
        for (int x = 1; x < W - 1; x++) {
            const float *r0 = disp + (size_t)(y - 1) * W + x;
            const float *r1 = disp + (size_t)y * W + x;
            const float *r2 = disp + (size_t)(y + 1) * W + x;
            const float gx = (-r0[-1] + r0[1]) + (-2 * r1[-1] + 2 * r1[1]) + (-r2[-1] + r2[1]);
            const float gy = (-r0[-1] - 2 * r0[0] - r0[1]) + (r2[-1] + 2 * r2[0] + r2[1]);
            const float g = fabsf(gx) + fabsf(gy);
            if (g > thres) edge[(size_t)y * W + x] = 1;
        }
    }
}

/* EdgeDispOptmize, ADC/multistep_refiner.cu:685-751. Neighbour disparities come from the snapshot;
 * the cost index subtracts min_disparity. */
void orc_edge_adjust(const float *disp_in, float *disp_out, const float *cost, const uint8_t *edge,
                     int W, int H, int dmin, int dmax) {
    const int D = dmax - dmin;
#pragma omp parallel for
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const size_t i = (size_t)y * W + x;
            float d = disp_in[i];
            if (x >= 1 && x < W - 1 && edge[i] == 1 && d != ORC_INVALID) {
                const float *cp = cost + i * D;
                long di = lround_f(d) - dmin;
                if (di >= 0 && di < D) {
                    float c0 = cp[di];
                    for (int k = 0; k < 2; k++) {
                        const int x2 = (k == 0) ? x - 1 : x + 1;
                        const float d2 = disp_in[(size_t)y * W + x2];
                        if (d2 != ORC_INVALID) {
                            const long d2i = lround_f(d2) - dmin;
                            if (d2i < 0 || d2i >= D) continue;
                            const float c = (k == 0) ? cp[-D + d2i] : cp[D + d2i];
                            if (c < c0) { d = d2; c0 = c; }
                        }
                    }
                }
            }
            disp_out[i] = d;
        }
    }
}

/* MedianFilter<3>, ADC/multistep_refiner.cu:186-224: interior pixels get the 5th smallest of the
 * 3x3 window by the same partial selection sort, border pixels are copied. */
void orc_median3(const float *in, float *out, int W, int H) {
#pragma omp parallel for
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const size_t i = (size_t)y * W + x;
            if (y >= 1 && x >= 1 && y < H - 1 && x < W - 1) {
                float w[9];
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < 3; b++) w[a * 3 + b] = in[(size_t)(y - 1 + a) * W + x - 1 + b];
                for (int a = 0; a < 5; a++) {
                    int mi = a;
                    for (int b = a + 1; b < 9; b++)
                        if (w[b] < w[mi]) mi = b;
                    const float t = w[a];
                    w[a] = w[mi];
                    w[mi] = t;
                }
                out[i] = w[4];
            } else {
                out[i] = in[i];
            }
        }
    }
}

#define TAP_COPY(dst, src, bytes) do { if (taps && taps->dst) memcpy(taps->dst, (src), (bytes)); } while (0)

/* ADCensusStereo::Match, ADC/ADCensusStereo.cu:64-145, and MultiStepRefiner::Refine,
 * ADC/multistep_refiner.cu:283-371 (do_filling drives both voting and interpolation,
 * ADC/ADCensusStereo.cu:195-196; IterativeNums = 5, ADC/config.h:46). */
int orc_match(const uint8_t *L, const uint8_t *R, int W, int H, const orc_option *opt,
              float *disp_left, const orc_taps *taps) {
    if (!L || !R || !disp_left || !opt) return 1;
    if (W <= 0 || H <= 0) return 2;
    const int dmin = opt->min_disparity, dmax = opt->max_disparity;
    const int D = dmax - dmin;
    if (D <= 0) return 3;
    const size_t npx = (size_t)W * H;
    const size_t nvol = npx * D;
    int rc = 0;

    uint8_t *gl = malloc(npx), *gr = malloc(npx);
    uint64_t *cl = malloc(npx * 8), *cr = malloc(npx * 8);
    float *init = malloc(nvol * sizeof(float));
    float *aggr = malloc(nvol * sizeof(float));
    uint8_t *arms = malloc(npx * 4);
    uint16_t *ch = malloc(npx * 2), *cv = malloc(npx * 2);
    float *dl = malloc(npx * 4), *dr = malloc(npx * 4), *d2 = malloc(npx * 4);
    uint8_t *occ = malloc(npx), *mis = malloc(npx), *edge = malloc(npx);
    if (!gl || !gr || !cl || !cr || !init || !aggr || !arms || !ch || !cv || !dl || !dr || !d2 ||
        !occ || !mis || !edge) { rc = 4; goto done; }

    orc_gray(L, W, H, gl);
    orc_gray(R, W, H, gr);
    TAP_COPY(gray_left, gl, npx);
    TAP_COPY(gray_right, gr, npx);
    orc_census(gl, W, H, cl);
    orc_census(gr, W, H, cr);
    TAP_COPY(census_left, cl, npx * 8);
    TAP_COPY(census_right, cr, npx * 8);
    orc_cost_init(L, R, cl, cr, W, H, dmin, dmax, opt->lambda_ad, opt->lambda_census, init);
    TAP_COPY(cost_init, init, nvol * 4);

    orc_build_arms(L, W, H, opt->cross_L1, opt->cross_L2, opt->cross_t1, opt->cross_t2, arms);
    orc_sup_counts(arms, W, H, ch, cv);
    TAP_COPY(arms, arms, npx * 4);
    TAP_COPY(sup_count_h, ch, npx * 2);
    TAP_COPY(sup_count_v, cv, npx * 2);
    memcpy(aggr, init, nvol * 4);
    orc_aggregate(aggr, arms, ch, cv, W, H, D, 4);
    TAP_COPY(cost_aggr, aggr, nvol * 4);

    orc_scanline(L, R, aggr, init, W, H, dmin, dmax, opt->so_p1, opt->so_p2, opt->so_tso);
    TAP_COPY(cost_scan, aggr, nvol * 4);
    TAP_COPY(cost_scan_aux, init, nvol * 4);

    orc_wta_left(aggr, W, H, dmin, dmax, dl);
    orc_wta_right(aggr, W, H, dmin, dmax, dr);
    TAP_COPY(disp_left_wta, dl, npx * 4);
    TAP_COPY(disp_right_wta, dr, npx * 4);

    memset(occ, 0, npx);
    memset(mis, 0, npx);
    if (opt->do_lr_check) {
        orc_outlier(dl, dr, W, H, opt->lrcheck_thres, d2, occ, mis);
        memcpy(dl, d2, npx * 4);
    }
    TAP_COPY(disp_outlier, dl, npx * 4);
    TAP_COPY(mask_occ, occ, npx);
    TAP_COPY(mask_mis, mis, npx);

    if (opt->do_filling) {
        for (int it = 0; it < 5; it++) {
            for (int k = 0; k < 2; k++) {
                uint8_t *mask = (k == 0) ? mis : occ;
                orc_region_voting_pass(dl, d2, mask, arms, W, H, dmin, dmax, opt->irv_ts, opt->irv_th);
                memcpy(dl, d2, npx * 4);
            }
        }
    }
    TAP_COPY(disp_voting, dl, npx * 4);
    TAP_COPY(mask_occ_voting, occ, npx);
    TAP_COPY(mask_mis_voting, mis, npx);

    if (opt->do_filling) {
        orc_interpolate(L, dl, d2, occ, mis, W, H, dmax);
        memcpy(dl, d2, npx * 4);
    }
    TAP_COPY(disp_interp, dl, npx * 4);

    memset(edge, 0, npx);
    if (opt->do_discontinuity_adjustment) {
        orc_edge_detect(dl, W, H, 5.0f, edge);
        orc_edge_adjust(dl, d2, aggr, edge, W, H, dmin, dmax);
        memcpy(dl, d2, npx * 4);
    }
    TAP_COPY(edge_mask, edge, npx);
    TAP_COPY(disp_edge, dl, npx * 4);

    orc_median3(dl, d2, W, H);
    memcpy(disp_left, d2, npx * 4);

done:
    free(gl); free(gr); free(cl); free(cr); free(init); free(aggr); free(arms); free(ch); free(cv);
    free(dl); free(dr); free(d2); free(occ); free(mis); free(edge);
    return rc;
}

double orc_bad_fraction(const float *disp, const uint8_t *gt, int W, int H, float gt_scale,
                        float delta, double *rms) {
    double bad = 0.0, sq = 0.0;
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const size_t i = (size_t)y * W + x;
            const float g = (float)gt[i];
            if (g == 0) continue;
            float dlt;
            if (disp[i] != ORC_INVALID) dlt = disp[i] - g / gt_scale;
            else dlt = 0 - g / gt_scale;
            sq += (double)dlt * dlt;
            bad += (fabsf(dlt) > delta);
        }
    }
    if (rms) *rms = sqrt(sq / ((double)W * H));
    return bad / ((double)W * H);
}
