// Cross-based cost aggregation on gfx950 (AggregateInArms_cuda, ADC/cost_aggregator.cu:329-392,
// ADC/ = final_version_project/Stereo_Matching/ADCensus_CUDA/).
//
// One launch = one directional pass over the whole f32[H][W][Dp] volume:
//     dst[p][k] = sum_{t=-neg(p)..pos(p)} src[p + t*step][k]      (optionally / support count of p)
// with (neg,pos,step) = (left,right,one pixel) for a horizontal pass and (top,bottom,one row) for a
// vertical pass. An aggregation iteration is two such launches, which is the full-image barrier
// between pass 1 and pass 2 that the reference's single launch lacks.
//
// Lanes run along the disparity axis: the arms of a pixel are the same for all of its disparities,
// so the span bounds are wave-uniform scalars, every global access of a wave is 256 contiguous
// bytes, and each lane adds its span in ascending t starting from 0.0f -- the reference's order
// (ADC/cost_aggregator.cu:346-372), so the sums are bit-identical to a sequential CPU evaluation.
//
// Ring kernel: a 256-thread workgroup owns one scanline (a row for horizontal passes, a column for
// vertical ones) x 64 disparities and marches along it. The last 2*cap+1 (+2 blocks) input
// positions live in an LDS ring [R][64] floats, so each input element is fetched from HBM exactly
// once per pass and re-read up to 2*cap+1 times from LDS (conflict-free: lane <-> bank).
#include "adc_common.h"

namespace adc {

#define AGG_B 16  // output positions per workgroup iteration (4 per wave)

template <bool DIV>
__global__ void __launch_bounds__(256)
k_agg_ring(const float *__restrict__ src, float *__restrict__ dst, const uint32_t *__restrict__ arms,
           const uint16_t *__restrict__ cnt, int W, int H, int Dp, int vertical, int seglen, int cap, int rmask) {
    extern __shared__ float ring[];  // (rmask+1) x 64
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    const bool kok = k < Dp;
    const int line = blockIdx.y;
    const int LEN = vertical ? H : W;
    const int s0 = blockIdx.z * seglen;
    if (s0 >= LEN) return;
    const int s1 = min(LEN, s0 + seglen);
    const size_t mstride = vertical ? (size_t)W * Dp : (size_t)Dp;   // elements between march positions
    const size_t base = (vertical ? (size_t)line * Dp : (size_t)line * W * Dp) + (size_t)(kok ? k : 0);
    const size_t abase = vertical ? (size_t)line : (size_t)line * W;
    const size_t astride = vertical ? (size_t)W : 1;
    const int in_hi = min(LEN, s1 + cap);   // exclusive upper bound of input positions this segment needs
    const int sh_neg = vertical ? 16 : 0, sh_pos = vertical ? 24 : 8;

    // preload [max(0, s0-cap), min(in_hi, s0 + B + cap))
    {
        const int lo = max(0, s0 - cap);
        const int hi = min(in_hi, s0 + AGG_B + cap);
        for (int i = lo + wave; i < hi; i += 4)
            ring[(i & rmask) * 64 + lane] = kok ? src[base + (size_t)i * mstride] : 0.0f;
    }
    __syncthreads();

    for (int o0 = s0; o0 < s1; o0 += AGG_B) {
        // (a) issue the loads of the next input block early
        float nxt[AGG_B / 4];
        const int n0 = o0 + AGG_B + cap;
#pragma unroll
        for (int j = 0; j < AGG_B / 4; j++) {
            const int i = n0 + wave + 4 * j;
            nxt[j] = (kok && i < in_hi) ? src[base + (size_t)i * mstride] : 0.0f;
        }
        // (b) outputs of this block
#pragma unroll 1
        for (int j = 0; j < AGG_B / 4; j++) {
            const int o = o0 + wave + 4 * j;
            if (o >= s1) break;
            const size_t ap = abase + (size_t)o * astride;
            const uint32_t a = __builtin_amdgcn_readfirstlane(arms[ap]);
            const int neg = (a >> sh_neg) & 0xff, pos = (a >> sh_pos) & 0xff;
            const int n = neg + pos + 1;
            const int start = (o - neg) & rmask;
            const int n1 = min(n, rmask + 1 - start);
            float sum = 0.0f;
            const float *p = ring + start * 64 + lane;
            for (int i = 0; i < n1; i++) sum += p[i * 64];
            p = ring + lane;
            for (int i = 0; i < n - n1; i++) sum += p[i * 64];
            if (DIV) {
                const uint32_t c = __builtin_amdgcn_readfirstlane((uint32_t)cnt[ap]);
                sum = sum / (float)c;
            }
            if (kok) dst[base + (size_t)o * mstride] = sum;
        }
        // (c) park the prefetched block in the ring; its slots are disjoint from everything block o0 reads
#pragma unroll
        for (int j = 0; j < AGG_B / 4; j++) {
            const int i = n0 + wave + 4 * j;
            if (i < in_hi) ring[(i & rmask) * 64 + lane] = nxt[j];
        }
        __syncthreads();
    }
}

// Fallback without LDS (any arm cap): one wave per (position, 64 disparities), spans re-read through
// the caches. Same addition order.
template <bool DIV>
__global__ void __launch_bounds__(256)
k_agg_direct(const float *__restrict__ src, float *__restrict__ dst, const uint32_t *__restrict__ arms,
             const uint16_t *__restrict__ cnt, int W, int H, int Dp, int vertical) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    const bool kok = k < Dp;
    const int line = blockIdx.y;
    const int LEN = vertical ? H : W;
    const size_t mstride = vertical ? (size_t)W * Dp : (size_t)Dp;
    const size_t base = (vertical ? (size_t)line * Dp : (size_t)line * W * Dp) + (size_t)(kok ? k : 0);
    const size_t abase = vertical ? (size_t)line : (size_t)line * W;
    const size_t astride = vertical ? (size_t)W : 1;
    const int sh_neg = vertical ? 16 : 0, sh_pos = vertical ? 24 : 8;
    for (int o = blockIdx.z * 4 + wave; o < LEN; o += 4 * gridDim.z) {
        const size_t ap = abase + (size_t)o * astride;
        const uint32_t a = __builtin_amdgcn_readfirstlane(arms[ap]);
        const int neg = (a >> sh_neg) & 0xff, pos = (a >> sh_pos) & 0xff;
        float sum = 0.0f;
        if (kok) {
            const float *p = src + base + (size_t)(o - neg) * mstride;
            for (int i = 0; i <= neg + pos; i++) sum += p[(size_t)i * mstride];
        }
        if (DIV) {
            const uint32_t c = __builtin_amdgcn_readfirstlane((uint32_t)cnt[ap]);
            sum = sum / (float)c;
        }
        if (kok) dst[base + (size_t)o * mstride] = sum;
    }
}

void launch_aggregate_pass(const float *src, float *dst, const uint32_t *arms, const uint16_t *cnt,
                           const Geom &g, int vertical, int max_arm, int variant, hipStream_t s) {
    const int chunks = (g.Dp + 63) / 64;
    const int lines = vertical ? g.W : g.H;
    const int LEN = vertical ? g.H : g.W;
    int R = 64;
    while (R < 2 * max_arm + 1 + 2 * AGG_B) R <<= 1;
    const size_t lds = (size_t)R * 64 * sizeof(float);
    if (variant == 0 && lds <= 64 * 1024) {
        // enough workgroups to fill 256 CUs several times over; segments are multiples of AGG_B
        int nseg = 1;
        const long long wgs = (long long)chunks * lines;
        if (wgs < 2048) nseg = (int)((2048 + wgs - 1) / wgs);
        int seglen = (LEN + nseg - 1) / nseg;
        seglen = ((seglen + AGG_B - 1) / AGG_B) * AGG_B;
        if (seglen < 4 * AGG_B) seglen = 4 * AGG_B;
        nseg = (LEN + seglen - 1) / seglen;
        dim3 grid(chunks, lines, nseg);
        if (cnt)
            k_agg_ring<true><<<grid, dim3(256), lds, s>>>(src, dst, arms, cnt, g.W, g.H, g.Dp, vertical, seglen, max_arm, R - 1);
        else
            k_agg_ring<false><<<grid, dim3(256), lds, s>>>(src, dst, arms, cnt, g.W, g.H, g.Dp, vertical, seglen, max_arm, R - 1);
    } else {
        int nz = 1;
        const long long wgs = (long long)chunks * lines;
        if (wgs < 4096) nz = (int)((4096 + wgs - 1) / wgs);
        if (nz > (LEN + 3) / 4) nz = (LEN + 3) / 4;
        dim3 grid(chunks, lines, nz);
        if (cnt)
            k_agg_direct<true><<<grid, dim3(256), 0, s>>>(src, dst, arms, cnt, g.W, g.H, g.Dp, vertical);
        else
            k_agg_direct<false><<<grid, dim3(256), 0, s>>>(src, dst, arms, cnt, g.W, g.H, g.Dp, vertical);
    }
}

}  // namespace adc
