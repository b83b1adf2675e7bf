// Winner-take-all + sub-pixel fit for the left view (ComputeDisparityCuda, ADC/ADCensusStereo.cu:200-267)
// and the right view (ComputeDisparityRight, ADC/ADCensusStereo.cu:358-424) on gfx950.
// ADC/ = final_version_project/Stereo_Matching/ADCensus_CUDA/.
//
// One wavefront per pixel, lanes along the disparity axis. The reference scans d ascending and
// keeps the first strict minimum, i.e. the lexicographic minimum of (cost, d); that is what the
// cross-lane reduction below computes.
#include "adc_common.h"

namespace adc {

__device__ __forceinline__ void wave_argmin(float &v, int &k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int ok = __shfl_xor(k, off, 64);
        if (ov < v || (ov == v && ok < k)) { v = ov; k = ok; }
    }
}

__device__ __forceinline__ float subpixel(float best, float c1, float c2, float cmin) {
    const float denom = c1 + c2 - 2 * cmin;
    if (denom != 0.0f) return best + (c1 - c2) / (denom * 2.0f);
    return best;
}

__global__ void __launch_bounds__(256)
k_wta_left(const float *__restrict__ cost, float *__restrict__ disp, size_t npx, int dmin, int D, int Dp) {
    const int lane = threadIdx.x & 63;
    const size_t p = (size_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (p >= npx) return;
    const float *c = cost + p * (size_t)Dp;
    float mv = ADC_LARGE_F;
    int mk = 0x7fffffff;
    for (int k = lane; k < D; k += 64) {
        const float v = c[k];
        if (mv > v) { mv = v; mk = k; }
    }
    wave_argmin(mv, mk);
    if (lane == 0) {
        // no cost below Large_Float: the reference's best_disparity stays 0
        const int best = (mk == 0x7fffffff) ? 0 : mk + dmin;
        float out;
        if (best == dmin || best == dmin + D - 1 || best - dmin < 0 || best - dmin >= D) {
            out = __int_as_float(0x7f800000);  // Invalid_Float
        } else {
            const int kb = best - dmin;
            out = subpixel((float)best, c[kb - 1], c[kb + 1], mv);
        }
        disp[p] = out;
    }
}

// cost_R(xr, d) = cost_L(xr + d, d); candidates whose left column is outside the image hold
// Large_Float and are never selected; a best disparity at either end of the range is written as
// that integer (ADC/ADCensusStereo.cu:404-407).
__global__ void __launch_bounds__(256)
k_wta_right(const float *__restrict__ cost, float *__restrict__ disp, int W, int H, int dmin, int D, int Dp) {
    const int lane = threadIdx.x & 63;
    const size_t p = (size_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (p >= (size_t)W * H) return;
    const int y = (int)(p / W), xr = (int)(p - (size_t)y * W);
    const float *row = cost + (size_t)y * W * Dp;
    float mv = ADC_LARGE_F;
    int mk = 0x7fffffff;
    for (int k = lane; k < D; k += 64) {
        const int xl = xr + dmin + k;
        if (xl >= 0 && xl < W) {
            const float v = row[(size_t)xl * Dp + k];
            if (mv > v) { mv = v; mk = k; }
        }
    }
    wave_argmin(mv, mk);
    if (lane == 0) {
        const int best = (mk == 0x7fffffff) ? 0 : mk + dmin;
        const int kb = best - dmin;
        float out;
        if (best == dmin || best == dmin + D - 1 || kb - 1 < 0 || kb + 1 >= D) {
            out = (float)best;
        } else {
            const int x1 = xr + best - 1, x2 = xr + best + 1;
            const float c1 = (x1 >= 0 && x1 < W) ? row[(size_t)x1 * Dp + kb - 1] : ADC_LARGE_F;
            const float c2 = (x2 >= 0 && x2 < W) ? row[(size_t)x2 * Dp + kb + 1] : ADC_LARGE_F;
            out = subpixel((float)best, c1, c2, mv);
        }
        disp[p] = out;
    }
}

void launch_wta(const float *cost, float *disp_l, float *disp_r, const Geom &g, hipStream_t s) {
    const size_t npx = (size_t)g.W * g.H;
    const unsigned blocks = (unsigned)((npx + 3) / 4);
    k_wta_left<<<dim3(blocks), dim3(256), 0, s>>>(cost, disp_l, npx, g.dmin, g.D, g.Dp);
    k_wta_right<<<dim3(blocks), dim3(256), 0, s>>>(cost, disp_r, g.W, g.H, g.dmin, g.D, g.Dp);
}

}  // namespace adc
