// 4-direction scanline optimisation on gfx950 (ScanlineOptimizeLeftRight / ScanlineOptimizeUpDown,
// ADC/scanline_optimizer.cu:4-290; ADC/ = final_version_project/Stereo_Matching/ADCensus_CUDA/).
//
// One wavefront owns one scanline (a row for the horizontal passes, a column for the vertical ones)
// and walks it sequentially. The D path costs of the current pixel live in registers, NV per lane
// with lane l holding disparities l*NV .. l*NV+NV-1, so
//   * the load/store of a pixel's disparity vector is one contiguous NV*256-byte access,
//   * Lr(p-r, d-1) / Lr(p-r, d+1) are register neighbours except at lane boundaries, where one DPP
//     wave shift per direction supplies them (lanes past D hold Large_Float = the reference's
//     sentinels Lp[-1], Lp[D]),
//   * min_d Lr(p-r, d) is a DPP row_shr / row_bcast reduction, read back with v_readlane.
// The recurrence, evaluated in the reference's operation order (so results are bit-identical to a
// sequential CPU evaluation):
//   dst[p][d] = (src[p][d] + min(Lp[d], Lp[d-1]+P1, Lp[d+1]+P1, min(Lp)+P2)) / 2
// (P1,P2) chosen by how many of the two colour steps reach so_tso (:97-109); first path pixel
// copies src (:51-52).
#include "adc_common.h"

namespace adc {

template <int CTRL>
__device__ __forceinline__ float dpp_f(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, false));
}

// min over the 64 lanes; every lane returns the result.
__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, dpp_f<0x111>(v, v));  // row_shr:1
    v = fminf(v, dpp_f<0x112>(v, v));  // row_shr:2
    v = fminf(v, dpp_f<0x114>(v, v));  // row_shr:4
    v = fminf(v, dpp_f<0x118>(v, v));  // row_shr:8   -> lane 15 of each row holds the row minimum
    v = fminf(v, dpp_f<0x142>(v, v));  // row_bcast:15 -> rows 1..3 fold in the previous row
    v = fminf(v, dpp_f<0x143>(v, v));  // row_bcast:31 -> lane 63 holds the wave minimum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int NV>
__device__ __forceinline__ void load_vec(const float *p, float (&v)[NV]) {
    if constexpr (NV == 1) {
        v[0] = p[0];
    } else if constexpr (NV == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        v[0] = t.x; v[1] = t.y;
    } else {
#pragma unroll
        for (int q = 0; q < NV / 4; q++) {
            const float4 t = reinterpret_cast<const float4 *>(p)[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    }
}
template <int NV>
__device__ __forceinline__ void store_vec(float *p, const float (&v)[NV]) {
    if constexpr (NV == 1) {
        p[0] = v[0];
    } else if constexpr (NV == 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    } else {
#pragma unroll
        for (int q = 0; q < NV / 4; q++)
            reinterpret_cast<float4 *>(p)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
}

#define SCAN_U 4  // path pixels whose loads are in flight together

// VERT = 0: wave walks row `line` along x; VERT = 1: walks column `line` along y.
template <int NV, int VERT>
__global__ void __launch_bounds__(256)
k_scanline(const float *__restrict__ src, float *__restrict__ dst, const uint8_t *__restrict__ step_l,
           const uint8_t *__restrict__ step_r, int W, int H, int dmin, int D, int Dp, int forward, ScanParams sp) {
    const int lane = threadIdx.x & 63;
    const int line = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nlines = VERT ? W : H;
    if (line >= nlines) return;
    const int LEN = VERT ? H : W;
    const int dir = forward ? 1 : -1;
    const int k0 = lane * NV;
    const bool active = k0 < Dp;              // lane owns part of the (padded) disparity vector
    const int koff = active ? k0 : 0;
    // position of the path head and per-step strides
    int x = VERT ? line : (forward ? 0 : W - 1);
    int y = VERT ? (forward ? 0 : H - 1) : line;
    const ptrdiff_t pstep = VERT ? (ptrdiff_t)dir * W : (ptrdiff_t)dir;   // in pixels
    ptrdiff_t pix = (ptrdiff_t)y * W + x;

    float Lp[NV];
    {
        float s[NV];
        if (active) load_vec<NV>(src + (size_t)pix * Dp + koff, s);
#pragma unroll
        for (int j = 0; j < NV; j++) Lp[j] = (active && k0 + j < D) ? s[j] : ADC_LARGE_F;
        if (active) {
            float o[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) o[j] = (k0 + j < D) ? s[j] : 0.0f;
            store_vec<NV>(dst + (size_t)pix * Dp + koff, o);
        }
    }
    float lmin = Lp[0];
#pragma unroll
    for (int j = 1; j < NV; j++) lmin = fminf(lmin, Lp[j]);
    float minLp = wave_min(lmin);

    // Colour-step flag of the pair (current, previous path pixel) is stored at the pixel with the
    // larger coordinate: bit0 for horizontal pairs, bit1 for vertical pairs.
    const int fbit = VERT ? 1 : 0;
    const ptrdiff_t fshift = forward ? 0 : (VERT ? (ptrdiff_t)W : (ptrdiff_t)1);  // backward: pair stored one pixel further

    for (int st0 = 1; st0 < LEN; st0 += SCAN_U) {
        float s[SCAN_U][NV];
        uint32_t f2[SCAN_U];
        uint32_t f1v = 0;
        // ---- issue every load of this group first
#pragma unroll
        for (int u = 0; u < SCAN_U; u++) {
            const bool ok = st0 + u < LEN;
            const ptrdiff_t pu = pix + (ptrdiff_t)(u + 1) * pstep;
            if (ok && active) load_vec<NV>(src + (size_t)pu * Dp + koff, s[u]);
            else {
#pragma unroll
                for (int j = 0; j < NV; j++) s[u][j] = 0.0f;
            }
            // right-image flags for this lane's disparities at path pixel pu
            uint32_t bits = 0;
            if (ok) {
                const int xu = VERT ? x : x + (u + 1) * dir;
                const int yu = VERT ? y + (u + 1) * dir : y;
                const bool none = (xu - dmin < 1) || (W < 3);   // no disparity of this pixel has 0 < xr < W-1
                const uint8_t *row = step_r + (ptrdiff_t)yu * W + fshift;
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    const int xe = xu - dmin - (k0 + j);
                    // xr >= W-1 happens only before the first usable disparity: d2 is still d1 there.
                    // xr <= 0 keeps the flag of the last usable disparity, whose xr is 1.
                    const bool use_f1 = none || xe >= W - 1;
                    const int xc = max(xe, 1);
                    uint32_t b = 2;  // 2 = "same as d1"
                    if (!use_f1) b = (row[min(xc, W - 2)] >> fbit) & 1u;
                    bits |= b << (2 * j);
                }
            }
            f2[u] = bits;
        }
        // left-image flags of the group: lane u fetches the one for step u
        {
            const int u = lane & (SCAN_U - 1);
            if (st0 + u < LEN) {
                const ptrdiff_t pu = pix + (ptrdiff_t)(u + 1) * pstep;
                f1v = (step_l[pu + fshift] >> fbit) & 1u;
            }
        }
        // ---- sequential recurrence over the group
#pragma unroll
        for (int u = 0; u < SCAN_U; u++) {
            if (st0 + u >= LEN) break;
            const uint32_t f1 = __builtin_amdgcn_readlane(f1v, u);
            const float P1a = f1 ? sp.P1[1] : sp.P1[0], P1b = f1 ? sp.P1[2] : sp.P1[1];
            const float P2a = f1 ? sp.P2[1] : sp.P2[0], P2b = f1 ? sp.P2[2] : sp.P2[1];
            const float up = dpp_f<0x138>(ADC_LARGE_F, Lp[NV - 1]);  // wave_shr:1 -> Lp[k0-1]
            const float dn = dpp_f<0x130>(ADC_LARGE_F, Lp[0]);       // wave_shl:1 -> Lp[k0+NV]
            float nw[NV];
            float lm = ADC_LARGE_F;
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const uint32_t b = (f2[u] >> (2 * j)) & 3u;
                const bool hi = (b == 2u) ? (f1 != 0) : (b != 0);
                const float P1 = hi ? P1b : P1a;
                const float P2 = hi ? P2b : P2a;
                const float l1 = Lp[j];
                const float l2 = (j == 0 ? up : Lp[j - 1]) + P1;
                const float l3 = (j == NV - 1 ? dn : Lp[j + 1]) + P1;
                const float l4 = minLp + P2;
                const float m = fminf(fminf(l1, l2), fminf(l3, l4));
                const float v = (s[u][j] + m) / 2.0f;
                nw[j] = (active && k0 + j < D) ? v : ADC_LARGE_F;
                lm = fminf(lm, nw[j]);
            }
            const ptrdiff_t pu = pix + (ptrdiff_t)(u + 1) * pstep;
            if (active) {
                float o[NV];
#pragma unroll
                for (int j = 0; j < NV; j++) o[j] = (k0 + j < D) ? nw[j] : 0.0f;
                store_vec<NV>(dst + (size_t)pu * Dp + koff, o);
            }
#pragma unroll
            for (int j = 0; j < NV; j++) Lp[j] = nw[j];
            minLp = wave_min(lm);
        }
        pix += (ptrdiff_t)SCAN_U * pstep;
        if (VERT) y += SCAN_U * dir; else x += SCAN_U * dir;
    }
}

template <int NV>
static void launch_nv(const float *src, float *dst, const Planes &p, const Geom &g, const ScanParams &sp,
                      int vertical, int forward, hipStream_t s) {
    const int nlines = vertical ? g.W : g.H;
    dim3 grid((nlines + 3) / 4);
    if (vertical)
        k_scanline<NV, 1><<<grid, dim3(256), 0, s>>>(src, dst, p.step_l, p.step_r, g.W, g.H, g.dmin, g.D, g.Dp, forward, sp);
    else
        k_scanline<NV, 0><<<grid, dim3(256), 0, s>>>(src, dst, p.step_l, p.step_r, g.W, g.H, g.dmin, g.D, g.Dp, forward, sp);
}

void launch_scanline_pass(const float *src, float *dst, const Planes &p, const Geom &g,
                          const ScanParams &sp, int vertical, int forward, hipStream_t s) {
    switch (g.NV) {
        case 1: launch_nv<1>(src, dst, p, g, sp, vertical, forward, s); break;
        case 2: launch_nv<2>(src, dst, p, g, sp, vertical, forward, s); break;
        case 4: launch_nv<4>(src, dst, p, g, sp, vertical, forward, s); break;
        case 8: launch_nv<8>(src, dst, p, g, sp, vertical, forward, s); break;
        default: launch_nv<16>(src, dst, p, g, sp, vertical, forward, s); break;
    }
}

}  // namespace adc
