// Right-view winner-take-all + sub-pixel fit (ComputeDisparityRight, ADC/ADCensusStereo.cu:358-424) on
// gfx950. ADC/ = final_version_project/Stereo_Matching/ADCensus_CUDA/. (The left view is produced by the
// last scanline pass while the final costs are still in registers.)
//
// cost_R(xr, d) = cost_L(xr + d, d): a right pixel's candidates lie on a diagonal of its row's [W][D]
// cost matrix, one float per 4*(Dp+1) bytes. Read straight from HBM that is one cache line per
// candidate; instead a workgroup stages, per chunk of WR_KC disparities, the WR_KC-float row pieces of the
// WR_PX + WR_KC left pixels its WR_PX right pixels can reach (coalesced reads) into LDS with a padded
// row stride, and every thread walks its diagonal there in ascending d with the reference's strict
// '>' test, so the first minimum wins exactly as in the sequential loop.
#include "adc_common.h"

#include <cstdlib>

namespace adc {

#ifndef WR_PX
#define WR_PX 128   // right pixels per workgroup (128 + 32 staged left pixels per chunk: 1.25x over-read; 256-pixel tiles read
                    // 1.125x but run 3 % slower: 40 KB of LDS per workgroup leaves fewer loads in flight per CU)
#endif
#define WR_NQ (256 / WR_PX)   // threads sharing a right pixel: each scans 1/WR_NQ of a chunk
#ifndef WR_KC
#define WR_KC 32    // disparities per staged chunk (32: 128-byte row pieces, 1.25x over-read; 64 was 10 % slower)
#endif
#define WR_ST (WR_KC + 1)   // LDS row stride in floats: odd, so diagonal walks spread over the banks

__device__ __forceinline__ float subpixel(float best, float c1, float c2, float cmin) {
    const float denom = c1 + c2 - 2 * cmin;
    if (denom != 0.0f) return best + (c1 - c2) / (denom * 2.0f);
    return best;
}

// V4: the disparity stride is a multiple of 4 floats, so a thread stages four consecutive disparities of a
// left pixel with one 16-byte load (WR_KC/4 lanes per row piece).
template <bool V4>
__global__ void __launch_bounds__(256)
k_wta_right(const float *__restrict__ cost, float *__restrict__ disp, int W, int H, int dmin, int D, int Dp) {
    __shared__ float S[(WR_PX + WR_KC) * WR_ST];
    __shared__ float pm[WR_NQ][WR_PX];
    __shared__ int pk[WR_NQ][WR_PX];
    const int y = blockIdx.y, xr0 = blockIdx.x * WR_PX;
    const int t = threadIdx.x;
    const int l = t % WR_PX, q = t / WR_PX;      // scan role: right pixel l, part q of the chunk
    constexpr int SRP = 256 / WR_KC;             // rows per staging pass of the scalar path
    const int j = t % WR_KC, r = t / WR_KC;      // staging role: column j, rows r, r+SRP, ...
    const float *row = cost + (size_t)y * W * Dp;
    float bv = ADC_LARGE_F;    // running minimum over the chunks done so far (threads with q == 0)
    int bk = 0x7fffffff;
    for (int kc = 0; kc < D; kc += WR_KC) {
        const int k = kc + j;
        const int xbase = xr0 + dmin + kc;
        if constexpr (V4) {
            // every thread's NIT loads are in flight together
            constexpr int LPR = WR_KC / 4, RPP = 256 / LPR, NIT = (WR_PX + WR_KC) / RPP;   // lanes per row piece, rows per pass
            static_assert((WR_PX + WR_KC) % RPP == 0, "staged rows must divide evenly over the threads");
            const int j4 = t % LPR, r16 = t / LPR;
            const int k4 = kc + 4 * j4;
            float4 v[NIT];
#pragma unroll
            for (int u = 0; u < NIT; u++) {
                const int x = xbase + r16 + RPP * u;
                v[u] = make_float4(ADC_LARGE_F, ADC_LARGE_F, ADC_LARGE_F, ADC_LARGE_F);
                if (x >= 0 && x < W && k4 < D) v[u] = *reinterpret_cast<const float4 *>(row + (size_t)x * Dp + k4);   // D % 4 == 0
            }
#pragma unroll
            for (int u = 0; u < NIT; u++) {
                float *sp = S + (r16 + RPP * u) * WR_ST + 4 * j4;
                sp[0] = v[u].x; sp[1] = v[u].y; sp[2] = v[u].z; sp[3] = v[u].w;
            }
        } else
        // 48 rows per thread, 16 loads in flight at a time
#pragma unroll 1
        for (int i0 = r; i0 < WR_PX + WR_KC; i0 += 16 * SRP) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int x = xbase + i0 + SRP * u;
                v[u] = ADC_LARGE_F;   // outside the image or past D: never selected (ADC/ADCensusStereo.cu:398-400)
                if (x >= 0 && x < W && k < D && i0 + SRP * u < WR_PX + WR_KC) v[u] = row[(size_t)x * Dp + k];
            }
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (i0 + SRP * u < WR_PX + WR_KC) S[(i0 + SRP * u) * WR_ST + j] = v[u];
        }
        __syncthreads();
        float mv = ADC_LARGE_F;
        int mk = 0x7fffffff;
        const int kq = q * (WR_KC / WR_NQ);
#pragma unroll 8
        for (int kk = 0; kk < WR_KC / WR_NQ; kk++) {
            const int k2 = kq + kk;
            const float v = S[(l + k2) * WR_ST + k2];
            if (mv > v) { mv = v; mk = kc + k2; }
        }
        if constexpr (WR_NQ == 1) {
            if (bv > mv) { bv = mv; bk = mk; }
            __syncthreads();             // the tile is restaged by the next chunk
        } else {
            pm[q][l] = mv;
            pk[q][l] = mk;
            __syncthreads();
            if (q == 0) {
#pragma unroll
                for (int qq = 0; qq < WR_NQ; qq++) {
                    const float v = pm[qq][l];
                    if (bv > v) { bv = v; bk = pk[qq][l]; }
                }
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

// Left-view winner-take-all + sub-pixel fit (ComputeDisparityCuda, ADC/ADCensusStereo.cu:200-267) as a kernel of
// its own: one wavefront per pixel, lanes along the disparities. Used for small frames, where reading the final
// volume once more costs less than the extra instructions the fused form puts on the serial chain of the last
// scanline pass; large frames keep the fused form (the volume read would cost more than the hidden instructions).
__global__ void __launch_bounds__(256)
k_wta_left(const float *__restrict__ cost, float *__restrict__ disp, unsigned npx, int dmin, int D, int Dp) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = gridDim.x * 4u;
    constexpr int U = 4;                             // pixels per trip: their loads are in flight together
    for (unsigned p0 = wave * U; p0 < npx; p0 += nwaves * U) {
        float bv[U];
        int bk[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            bv[u] = ADC_LARGE_F;                     // the reference's initial minimum: a cost has to be below it to be taken
            bk[u] = 0x7fffffff;
        }
        for (int k = lane; k < D; k += 64) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = (p0 + u < npx) ? cost[(size_t)(p0 + u) * Dp + k] : ADC_LARGE_F;
#pragma unroll
            for (int u = 0; u < U; u++)
                if (bv[u] > v[u]) { bv[u] = v[u]; bk[u] = k; }   // strict: the first minimum of this lane's disparities
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (p0 + u >= npx) break;                // uniform
            const float m = wave_min(bv[u]);
            const int best = wave_min_i32((bv[u] == m) ? bk[u] : 0x7fffffff);   // the first disparity at the minimum
            if (lane == 0) {
                const float *row = cost + (size_t)(p0 + u) * Dp;
                float out = __int_as_float(0x7f800000);  // Invalid_Float: minimum at either end of the range
                if (best > 0 && best < D - 1 && m < ADC_LARGE_F) out = subpixel((float)(best + dmin), row[best - 1], row[best + 1], m);
                disp[p0 + u] = out;
            }
        }
    }
}

void launch_wta_left(const float *cost, float *disp_l, const Geom &g, hipStream_t s) {
    const unsigned npx = (unsigned)((size_t)g.W * g.H);
    const unsigned wgs = (npx + 15) / 16;
    k_wta_left<<<dim3(wgs < 16384u ? wgs : 16384u), dim3(256), 0, s>>>(cost, disp_l, npx, g.dmin, g.D, g.Dp);
}

void launch_wta_right(const float *cost, float *disp_r, const Geom &g, hipStream_t s) {
    dim3 grid((g.W + WR_PX - 1) / WR_PX, g.H);
    if (g.Dp % 4 == 0 && g.D % 4 == 0 && !g.tune.wta_scalar)
        k_wta_right<true><<<grid, dim3(256), 0, s>>>(cost, disp_r, g.W, g.H, g.dmin, g.D, g.Dp);
    else
        k_wta_right<false><<<grid, dim3(256), 0, s>>>(cost, disp_r, g.W, g.H, g.dmin, g.D, g.Dp);
}

}  // namespace adc
