// Per-pixel front end of the AD-Census path on gfx950: BGR packing + gray, 9x7 centre-symmetric
// census, colour-step flags, cross arms, support counts and the initial AD-Census cost volume.
// Reference: ADC/ = final_version_project/Stereo_Matching/ADCensus_CUDA/.
#include "adc_common.h"

namespace adc {

__device__ __forceinline__ int color_dist(uint32_t p, uint32_t q) {
    // max over the three channels of |p - q| (ColorDist, ADC/cross_aggregator.h:73-75)
    // one channel masked in place on both sides: v_sad_u8 (sum of absolute byte differences) is that channel's |p - q|
    const int d0 = (int)__builtin_amdgcn_sad_u8(p & 0xffu, q & 0xffu, 0u);
    const int d1 = (int)__builtin_amdgcn_sad_u8(p & 0xff00u, q & 0xff00u, 0u);
    const int d2 = (int)__builtin_amdgcn_sad_u8(p & 0xff0000u, q & 0xff0000u, 0u);
    return max(d0, max(d1, d2));
}

// ---- pack + gray ------------------------------------------------------------------------------
// ComputeGrayCuda (ADC/cost_computor.cu:74-89): gray = uint8(r*0.299 + g*0.587 + b*0.114) in double,
// left to right, no contraction (the library is built with -ffp-contract=off).
// Both images of the pair in one launch (blockIdx.y picks the image).
__global__ void __launch_bounds__(256) k_pack_gray(const uint8_t *__restrict__ bgr_l, const uint8_t *__restrict__ bgr_r,
                                                   uint32_t *__restrict__ pix_l, uint32_t *__restrict__ pix_r,
                                                   uint8_t *__restrict__ gray_l, uint8_t *__restrict__ gray_r, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool right = blockIdx.y != 0;
    const uint8_t *bgr = right ? bgr_r : bgr_l;
    const uint32_t b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
    (right ? pix_r : pix_l)[i] = b | (g << 8) | (r << 16);
    const double v = ((double)r * 0.299 + (double)g * 0.587) + (double)b * 0.114;
    (right ? gray_r : gray_l)[i] = (uint8_t)(uint32_t)v;
}

void launch_pack_gray(const uint8_t *bgr_l, const uint8_t *bgr_r, uint32_t *pix_l, uint32_t *pix_r, uint8_t *gray_l,
                      uint8_t *gray_r, int W, int H, hipStream_t s) {
    const size_t n = (size_t)W * H;
    k_pack_gray<<<dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, s>>>(bgr_l, bgr_r, pix_l, pix_r, gray_l, gray_r, n);
}

// ---- census -----------------------------------------------------------------------------------
// CenterSymmetricCensusKernelSM2 (ADC/census_cost.cu:9-98): 9x7 window, pixels outside the image
// read as 0, 27 bits from the three upper rows against their point-mirrored partners plus 4 bits
// from the left half of the centre row. One 64x16 output tile per 256-thread workgroup, the
// (64+8)x(16+6) gray tile staged in LDS.
#define CEN_TW 64
#define CEN_TH 16
#define CEN_LW (CEN_TW + 8)
#define CEN_LH (CEN_TH + 6)
__global__ void __launch_bounds__(256) k_census(const uint8_t *__restrict__ gray_l, const uint8_t *__restrict__ gray_r,
                                                uint32_t *__restrict__ census_l, uint32_t *__restrict__ census_r,
                                                int W, int H) {
    __shared__ uint8_t tile[CEN_LH][CEN_LW];
    const uint8_t *gray = blockIdx.z ? gray_r : gray_l;     // both images of the pair in one launch
    uint32_t *census = blockIdx.z ? census_r : census_l;
    const int x0 = blockIdx.x * CEN_TW, y0 = blockIdx.y * CEN_TH;
    for (int i = threadIdx.x; i < CEN_LH * CEN_LW; i += 256) {
        const int ly = i / CEN_LW, lx = i - ly * CEN_LW;
        const int gy = y0 + ly - 3, gx = x0 + lx - 4;
        tile[ly][lx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? gray[(size_t)gy * W + gx] : (uint8_t)0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int ly = ty + 4 * r;           // tile row of the output pixel
        const int x = x0 + tx, y = y0 + ly;
        if (x >= W || y >= H) continue;
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int m = 0; m < 9; m++) {
                const uint8_t e1 = tile[ly + k][tx + m];
                const uint8_t e2 = tile[ly + 6 - k][tx + 8 - m];
                c |= (uint32_t)(e1 >= e2) << (k * 9 + m);
            }
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const uint8_t e1 = tile[ly + 3][tx + m];
            const uint8_t e2 = tile[ly + 3][tx + 8 - m];
            c |= (uint32_t)(e1 >= e2) << (27 + m);
        }
        census[(size_t)y * W + x] = c;
    }
}

void launch_census(const uint8_t *gray_l, const uint8_t *gray_r, uint32_t *census_l, uint32_t *census_r, int W, int H,
                   hipStream_t s) {
    dim3 grid((W + CEN_TW - 1) / CEN_TW, (H + CEN_TH - 1) / CEN_TH, 2);
    k_census<<<grid, dim3(256), 0, s>>>(gray_l, gray_r, census_l, census_r, W, H);
}

__global__ void __launch_bounds__(256) k_census_to_u64(const uint32_t *__restrict__ c32, uint64_t *__restrict__ c64, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c64[i] = (uint64_t)c32[i];
}
void launch_census_to_u64(const uint32_t *c32, uint64_t *c64, size_t n, hipStream_t s) {
    k_census_to_u64<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(c32, c64, n);
}

// ---- colour-step flags ------------------------------------------------------------------------
// The scanline recurrence compares the max-channel colour distance between consecutive path pixels
// with so_tso (ADC/scanline_optimizer.cu:73-74,90-91,98-109). Both directions of a path see the same
// pixel pair, so one flag per pixel pair is enough: bit0 = dist(p[y][x], p[y][x-1]) >= tso,
// bit1 = dist(p[y][x], p[y-1][x]) >= tso.
__global__ void __launch_bounds__(256) k_step_flags(const uint32_t *__restrict__ pix_l, const uint32_t *__restrict__ pix_r,
                                                    uint8_t *__restrict__ flags_l, uint8_t *__restrict__ flags_r,
                                                    int W, int H, int tso) {
    const uint32_t *pix = blockIdx.z ? pix_r : pix_l;       // both images of the pair in one launch
    uint8_t *flags = blockIdx.z ? flags_r : flags_l;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t i = (size_t)y * W + x;
    const uint32_t p = pix[i];
    uint8_t f = 0;
    if (x >= 1 && color_dist(p, pix[i - 1]) >= tso) f |= 1;
    if (y >= 1 && color_dist(p, pix[i - W]) >= tso) f |= 2;
    flags[i] = f;
}
void launch_step_flags(const uint32_t *pix_l, const uint32_t *pix_r, uint8_t *flags_l, uint8_t *flags_r, int W, int H,
                       int tso, hipStream_t s) {
    dim3 grid((W + 63) / 64, (H + 3) / 4, 2);
    k_step_flags<<<grid, dim3(256), 0, s>>>(pix_l, pix_r, flags_l, flags_r, W, H, tso);
}

// ---- cross arms -------------------------------------------------------------------------------
// FindHorizontalArm / FindVerticalArm (ADC/cost_aggregator.cu:130-260) on the left image.
// The four arms of a pixel are walked together, one step per iteration, so that four loads are in
// flight instead of one dependent load per step. A walk stops (for good) at the image border, at a
// pixel too far from the anchor (>= t1), too far from the previous pixel of the walk (>= t1, from the
// second step on) or, beyond L2 pixels, not within t2 of the anchor.
__global__ void __launch_bounds__(256) k_arms(const uint32_t *__restrict__ pix, uint32_t *__restrict__ arms,
                                              int W, int H, int cap, int L2, int t1, int t2) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool in = x < W && y < H;
    const size_t i0 = in ? (size_t)y * W + x : 0;
    const uint32_t p0 = pix[i0];
    const int dx[4] = {-1, 1, 0, 0}, dy[4] = {0, 0, -1, 1};   // left, right, top, bottom (ADC/cross_aggregator.h:13)
    // A pixel is kept as its three channels masked in place: |difference| of one channel is then one v_sad_u8
    // (sum of absolute byte differences, the other bytes being zero), the colour distance three of them and a max3.
    struct Ch { uint32_t b, g, r; };
    auto split = [](uint32_t v) { return Ch{v & 0xffu, v & 0xff00u, v & 0xff0000u}; };
    auto dist = [](const Ch &a, const Ch &c) {
        const int d0 = (int)__builtin_amdgcn_sad_u8(a.b, c.b, 0u);
        const int d1 = (int)__builtin_amdgcn_sad_u8(a.g, c.g, 0u);
        const int d2 = (int)__builtin_amdgcn_sad_u8(a.r, c.r, 0u);
        return max(d0, max(d1, d2));
    };
    const Ch c0 = split(p0);
    uint32_t arm[4] = {0, 0, 0, 0};
    Ch last[4] = {c0, c0, c0, c0};
    bool alive[4] = {in, in, in, in};
    for (int n = 0; n < cap && __any(alive[0] | alive[1] | alive[2] | alive[3]); n++) {
        uint32_t p[4];
        bool inside[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int xn = x + (n + 1) * dx[d], yn = y + (n + 1) * dy[d];
            inside[d] = alive[d] & ((unsigned)xn < (unsigned)W) & ((unsigned)yn < (unsigned)H);
            p[d] = pix[inside[d] ? (size_t)yn * W + xn : i0];
        }
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const Ch c = split(p[d]);
            const int d1 = dist(c, c0);
            bool go = inside[d] & (d1 < t1);
            go &= (n == 0) | (dist(c, last[d]) < t1);
            go &= (n + 1 <= L2) | (d1 < t2);
            arm[d] += go;
            last[d] = c;
            alive[d] = go;
        }
    }
    if (in) arms[i0] = arm[0] | (arm[1] << 8) | (arm[2] << 16) | (arm[3] << 24);
}
// (The same walk out of an LDS tile -- a 1024-thread workgroup staging its 64 x 16 pixels plus a cap-wide frame once --
// measured 162 us against this kernel's 150 us at 1920 x 1080: the walk is bound by its vector ALU work, about twenty
// instructions per step and direction, not by its loads, which the caches serve.)
void launch_arms(const uint32_t *pix, uint32_t *arms, int W, int H, int L1, int L2, int t1, int t2, hipStream_t s) {
    const int cap = L1 < 255 ? L1 : 255;  // min(cross_L1, MAX_ARM_LENGTH), ADC/cost_aggregator.cu:145
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    k_arms<<<grid, dim3(256), 0, s>>>(pix, arms, W, H, cap, L2, t1, t2);
}

// ---- support pixel counts -----------------------------------------------------------------------
// ComputeSupPixelCount (ADC/cost_aggregator.cu:262-316).
__global__ void __launch_bounds__(256) k_sup_counts(const uint32_t *__restrict__ arms, uint16_t *__restrict__ cnt_h,
                                                    uint16_t *__restrict__ cnt_v, uint32_t *__restrict__ cnt_h32,
                                                    uint32_t *__restrict__ cnt_v32, int W, int H) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t i = (size_t)y * W + x;
    const uint32_t a = arms[i];
    const int l = a & 0xff, r = (a >> 8) & 0xff, t = (a >> 16) & 0xff, b = a >> 24;
    // four rows / columns per trip, so that four loads are in flight per lane (the sums are integers: any order)
    int ch = 0, cv = 0;
    for (int dy = -t; dy <= b; dy += 4) {
        uint32_t q[4];
#pragma unroll
        for (int u = 0; u < 4; u++) q[u] = arms[i + (ptrdiff_t)min(dy + u, b) * W];
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (dy + u <= b) ch += (int)(q[u] & 0xff) + (int)((q[u] >> 8) & 0xff) + 1;
    }
    for (int dx = -l; dx <= r; dx += 4) {
        uint32_t q[4];
#pragma unroll
        for (int u = 0; u < 4; u++) q[u] = arms[i + min(dx + u, r)];
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (dx + u <= r) cv += (int)((q[u] >> 16) & 0xff) + (int)(q[u] >> 24) + 1;
    }
    cnt_h[i] = (uint16_t)ch;
    cnt_v[i] = (uint16_t)cv;
    cnt_h32[i] = (uint32_t)(uint16_t)ch;
    cnt_v32[i] = (uint32_t)(uint16_t)cv;
}
#define SUP_TW 64
#define SUP_TH 16
// The same sums out of LDS: a 1024-thread workgroup stages the arms of its 64 x 16 pixels with `cap` rows above and
// below (for the sums over a pixel's vertical arms) and `cap` columns on either side (for the sums over its horizontal
// arms) once, instead of up to 4 cap + 2 cached loads per pixel. Integer sums: the same values.
__global__ void __launch_bounds__(1024) k_sup_counts_tile(const uint32_t *__restrict__ arms, uint16_t *__restrict__ cnt_h,
                                                          uint16_t *__restrict__ cnt_v, uint32_t *__restrict__ cnt_h32,
                                                          uint32_t *__restrict__ cnt_v32, int W, int H, int cap) {
    extern __shared__ uint32_t sarm[];                  // tall slab: (16 + 2 cap) x 64, then wide slab: 16 x (64 + 2 cap)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * SUP_TW, y0 = blockIdx.y * SUP_TH;
    const int th = SUP_TH + 2 * cap, tw = SUP_TW + 2 * cap;
    uint32_t *tall = sarm, *wide = sarm + th * SUP_TW;
    for (int ty = wave; ty < th; ty += 16) {
        const int gy = y0 - cap + ty, gx = x0 + lane;
        tall[ty * SUP_TW + lane] = ((unsigned)gy < (unsigned)H && gx < W) ? arms[(size_t)gy * W + gx] : 0u;
    }
    {
        const int gy = y0 + wave;
        for (int tx = lane; tx < tw; tx += 64) {
            const int gx = x0 - cap + tx;
            wide[wave * tw + tx] = (gy < H && (unsigned)gx < (unsigned)W) ? arms[(size_t)gy * W + gx] : 0u;
        }
    }
    __syncthreads();
    const int x = x0 + lane, y = y0 + wave;
    if (x >= W || y >= H) return;
    const uint32_t a = tall[(wave + cap) * SUP_TW + lane];
    const int l = a & 0xff, r = (a >> 8) & 0xff, t = (a >> 16) & 0xff, b = a >> 24;   // each <= cap: inside the slabs
    int ch = 0, cv = 0;
    for (int dy = -t; dy <= b; dy++) {
        const uint32_t q = tall[(wave + cap + dy) * SUP_TW + lane];
        ch += (int)(q & 0xff) + (int)((q >> 8) & 0xff) + 1;
    }
    for (int dx = -l; dx <= r; dx++) {
        const uint32_t q = wide[wave * tw + lane + cap + dx];
        cv += (int)((q >> 16) & 0xff) + (int)(q >> 24) + 1;
    }
    const size_t i = (size_t)y * W + x;
    cnt_h[i] = (uint16_t)ch;
    cnt_v[i] = (uint16_t)cv;
    cnt_h32[i] = (uint32_t)(uint16_t)ch;
    cnt_v32[i] = (uint32_t)(uint16_t)cv;
}
// cap: the arm length cap the arms were built with (every arm <= cap)
void launch_sup_counts(const uint32_t *arms, uint16_t *cnt_h, uint16_t *cnt_v, uint32_t *cnt_h32, uint32_t *cnt_v32,
                       int W, int H, int cap, hipStream_t s) {
    const size_t lds = ((size_t)(SUP_TH + 2 * cap) * SUP_TW + (size_t)SUP_TH * (SUP_TW + 2 * cap)) * sizeof(uint32_t);
    if (cap >= 0 && lds <= 64 * 1024) {
        dim3 grid((W + SUP_TW - 1) / SUP_TW, (H + SUP_TH - 1) / SUP_TH);
        k_sup_counts_tile<<<grid, dim3(1024), lds, s>>>(arms, cnt_h, cnt_v, cnt_h32, cnt_v32, W, H, cap);
        return;
    }
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    k_sup_counts<<<grid, dim3(256), 0, s>>>(arms, cnt_h, cnt_v, cnt_h32, cnt_v32, W, H);
}

// ---- initial cost volume ------------------------------------------------------------------------
// ComputeCostCuda (ADC/cost_computor.cu:151-207) with the runtime disparity range:
//   cost = ((1 - e_ad[sum|dBGR|]) + 1) - e_cen[popcount(cenL ^ cenR)], 1.0f when x-d is outside the image.
// e_ad / e_cen are the host-built tables of expf(-ad/lambda_ad) and expf(-h/lambda_census).
// One workgroup = one row segment of 64 pixels; the right-image entries {packed BGR, census} that
// any of its (pixel, disparity) pairs can touch are staged once in LDS; lane <-> disparity so every
// store instruction writes 256 contiguous bytes of the [H][W][Dp] volume.
#define COST_TX 64
__global__ void __launch_bounds__(256) k_cost_init(const uint32_t *__restrict__ pix_l, const uint32_t *__restrict__ pix_r,
                                                   const uint32_t *__restrict__ cen_l, const uint32_t *__restrict__ cen_r,
                                                   const float *__restrict__ lut_ad, const float *__restrict__ lut_cen,
                                                   float *__restrict__ cost, int W, int H, int dmin, int D, int Dp) {
    extern __shared__ __align__(16) uint8_t smem[];
    float *s_ad = reinterpret_cast<float *>(smem);        // 766 (+2 pad)
    float *s_cen = s_ad + 768;                            // 65 (+3 pad)
    uint2 *s_r = reinterpret_cast<uint2 *>(s_cen + 68);   // COST_TX + D entries
    const int y = blockIdx.y;
    const int x0 = blockIdx.x * COST_TX;
    const int dmax = dmin + D;
    const int base = x0 - dmax + 1;           // smallest right column any pair of this segment touches
    const int nr = COST_TX + D - 1;
    for (int i = threadIdx.x; i < 766; i += 256) s_ad[i] = lut_ad[i];
    if (threadIdx.x < 65) s_cen[threadIdx.x] = lut_cen[threadIdx.x];
    for (int i = threadIdx.x; i < nr; i += 256) {
        const int xr = base + i;
        uint2 v = make_uint2(0u, 0u);
        if (xr >= 0 && xr < W) {
            const size_t q = (size_t)y * W + xr;
            v = make_uint2(pix_r[q], cen_r[q]);
        }
        s_r[i] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int xi = wave; xi < COST_TX; xi += 4) {
        const int x = x0 + xi;
        if (x >= W) break;
        const size_t p = (size_t)y * W + x;
        const uint32_t pl = pix_l[p], cl = cen_l[p];
        float *out = cost + p * (size_t)Dp;
        for (int k = lane; k < D; k += 64) {
            const int xr = x - dmin - k;
            float c = 1.0f;
            if (xr >= 0 && xr < W) {
                const uint2 r = s_r[xr - base];
                const uint32_t sad = __builtin_amdgcn_sad_u8(pl, r.x, 0u);
                const uint32_t ham = __builtin_popcount(cl ^ r.y);
                c = ((1.0f - s_ad[sad]) + 1.0f) - s_cen[ham];
            }
            out[k] = c;
        }
    }
}

void launch_cost_init(const Planes &p, const Geom &g, const float *lut_ad, const float *lut_cen,
                      float *cost, hipStream_t s) {
    dim3 grid((g.W + COST_TX - 1) / COST_TX, g.H);
    const size_t lds = (768 + 68) * sizeof(float) + (size_t)(COST_TX + g.D) * sizeof(uint2);
    k_cost_init<<<grid, dim3(256), lds, s>>>(p.pix_l, p.pix_r, p.cen_l, p.cen_r, lut_ad, lut_cen, cost,
                                            g.W, g.H, g.dmin, g.D, g.Dp);
}

}  // namespace adc
