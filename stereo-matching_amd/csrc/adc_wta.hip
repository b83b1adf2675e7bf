// Right-view winner-take-all + sub-pixel fit (ComputeDisparityRight, ADC/ADCensusStereo.cu:358-424) on
// gfx950. ADC/ = final_version_project/Stereo_Matching/ADCensus_CUDA/. (The left view is produced by the
// last scanline pass while the final costs are still in registers.)
//
// cost_R(xr, d) = cost_L(xr + d, d): a right pixel's candidates lie on a diagonal of its row's [W][D]
// cost matrix, one float per 4*(Dp+1) bytes. Read straight from HBM that is one cache line per
// candidate; instead a workgroup stages, per chunk of 64 disparities, the 64-float row pieces of the
// 192 left pixels its 128 right pixels can reach (coalesced 256-byte reads) into LDS with a padded
// row stride, and every thread walks its diagonal there in ascending d with the reference's strict
// '>' test, so the first minimum wins exactly as in the sequential loop.
#include "adc_common.h"

namespace adc {

#define WR_PX 128   // right pixels per workgroup
#define WR_KC 64    // disparities per staged chunk
#define WR_ST 65    // LDS row stride in floats: diagonal walks hit 32 different banks

__device__ __forceinline__ float subpixel(float best, float c1, float c2, float cmin) {
    const float denom = c1 + c2 - 2 * cmin;
    if (denom != 0.0f) return best + (c1 - c2) / (denom * 2.0f);
    return best;
}

__global__ void __launch_bounds__(256)
k_wta_right(const float *__restrict__ cost, float *__restrict__ disp, int W, int H, int dmin, int D, int Dp) {
    __shared__ float S[(WR_PX + WR_KC) * WR_ST];
    __shared__ float pm[2][WR_PX];
    __shared__ int pk[2][WR_PX];
    const int y = blockIdx.y, xr0 = blockIdx.x * WR_PX;
    const int t = threadIdx.x;
    const int l = t & (WR_PX - 1), q = t >> 7;   // scan role: right pixel l, half q of the chunk
    const int j = t & 63, r = t >> 6;            // staging role: column j, rows r, r+4, ...
    const float *row = cost + (size_t)y * W * Dp;
    float bv = ADC_LARGE_F;    // running minimum over the chunks done so far (threads with q == 0)
    int bk = 0x7fffffff;
    for (int kc = 0; kc < D; kc += WR_KC) {
        const int k = kc + j;
        const int xbase = xr0 + dmin + kc;
        // 48 rows per thread, 16 loads in flight at a time
#pragma unroll 1
        for (int i0 = r; i0 < WR_PX + WR_KC; i0 += 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int x = xbase + i0 + 4 * u;
                v[u] = ADC_LARGE_F;   // outside the image or past D: never selected (ADC/ADCensusStereo.cu:398-400)
                if (x >= 0 && x < W && k < D) v[u] = row[(size_t)x * Dp + k];
            }
#pragma unroll
            for (int u = 0; u < 16; u++) S[(i0 + 4 * u) * WR_ST + j] = v[u];
        }
        __syncthreads();
        float mv = ADC_LARGE_F;
        int mk = 0x7fffffff;
        const int kq = q * (WR_KC / 2);
#pragma unroll 8
        for (int kk = 0; kk < WR_KC / 2; kk++) {
            const int k2 = kq + kk;
            const float v = S[(l + k2) * WR_ST + k2];
            if (mv > v) { mv = v; mk = kc + k2; }
        }
        pm[q][l] = mv;
        pk[q][l] = mk;
        __syncthreads();
        if (q == 0) {
#pragma unroll
            for (int qq = 0; qq < 2; qq++) {
                const float v = pm[qq][l];
                if (bv > v) { bv = v; bk = pk[qq][l]; }
            }
        }
    }
    const int xr = xr0 + l;
    if (q == 0 && xr < W) {
        const int best = (bk == 0x7fffffff) ? 0 : bk + dmin;   // nothing below Large_Float: best_disparity stays 0
        const int kb = best - dmin;
        float out;
        if (best == dmin || best == dmin + D - 1 || kb - 1 < 0 || kb + 1 >= D) {
            out = (float)best;                                  // written as the integer (:404-407)
        } else {
            const int x1 = xr + best - 1, x2 = xr + best + 1;
            const float c1 = (x1 >= 0 && x1 < W) ? row[(size_t)x1 * Dp + kb - 1] : ADC_LARGE_F;
            const float c2 = (x2 >= 0 && x2 < W) ? row[(size_t)x2 * Dp + kb + 1] : ADC_LARGE_F;
            out = subpixel((float)best, c1, c2, bv);
        }
        disp[(size_t)y * W + xr] = out;
    }
}

void launch_wta_right(const float *cost, float *disp_r, const Geom &g, hipStream_t s) {
    dim3 grid((g.W + WR_PX - 1) / WR_PX, g.H);
    k_wta_right<<<grid, dim3(256), 0, s>>>(cost, disp_r, g.W, g.H, g.dmin, g.D, g.Dp);
}

}  // namespace adc
