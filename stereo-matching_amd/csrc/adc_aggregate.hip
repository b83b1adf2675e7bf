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

#define AGG_B 16   // output positions per workgroup iteration
#define AGG_JB 4   // consecutive output positions one wave produces together

// Ring fill: one LDS-DMA per (march position, wave): 64 lanes x 4 B land at ring[slot][0..63] with no
// VGPR or ds_write in between. The destination is wave-uniform (slot base); the hardware adds lane*4.
__device__ __forceinline__ void ring_fill(const float *gsrc, float *ring_slot) {
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)ring_slot, 4, 0, 0);
}

// One ring value feeds up to AGG_JB accumulators: consecutive outputs have heavily overlapping
// spans, so each LDS read is shared by the JB sums whose span contains that position. Over the
// positions common to all JB spans the adds are unconditional; over the ragged head and tail a
// position is added as (inside span j ? v : 0.0f), and adding +0.0f to a non-negative sum is the
// identity, so this is bit-identical to "add" / "skip". Every accumulator still receives exactly
// its own span in ascending order, starting from 0.0f.
//
// Pipeline per workgroup iteration i (block of AGG_B outputs starting at o0):
//   issue the fills of block i+2's new input positions -> compute + store block i ->
//   wait until block i+1's fills have landed (counted vmcnt) -> barrier.
// Fills and stores retire in issue order on one counter. At the wait of iteration i a wave's newest
// operations are: 4 stores of block i-1, 4 fills issued in iteration i, 4 stores of block i -- so
// vmcnt(12) returns exactly when the 4 fills issued in iteration i-1 (block i+1's data) have landed.
template <bool DIV>
__global__ void __launch_bounds__(256)
k_agg_ring(const float *__restrict__ src, float *__restrict__ dst, const uint32_t *__restrict__ arms,
           const uint32_t *__restrict__ cnt, int W, int H, int Dp, int vertical, int seglen, int cap, int rmask) {
    extern __shared__ float ring[];  // (rmask+1) x 64
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps span bounds in SGPRs
    const int k = blockIdx.x * 64 + lane;
    const bool kok = k < Dp;
    const int line = blockIdx.y;
    const int LEN = vertical ? H : W;
    const int s0 = blockIdx.z * seglen;
    if (s0 >= LEN) return;
    const int s1 = min(LEN, s0 + seglen);
    // Strides fit 32 bits (W*Dp < 2^31 is checked by the launcher); only base offsets are 64-bit. All
    // per-iteration addresses are running pointers advanced by 32-bit strides: 64-bit multiplies on
    // the scalar unit are what this kernel can least afford.
    const int mstride = vertical ? W * Dp : Dp;          // elements between march positions
    const int astride = vertical ? W : 1;                // pixels between march positions
    const size_t base = (vertical ? (size_t)line * Dp : (size_t)line * W * Dp) + (size_t)(kok ? k : 0);
    const size_t abase = vertical ? (size_t)line : (size_t)line * W;
    const int in_hi = min(LEN, s1 + cap);   // exclusive upper bound of input positions this segment needs
    const int sh_neg = vertical ? 16 : 0, sh_pos = vertical ? 24 : 8;
    const int R = rmask + 1;
    const float *rl = ring + lane;

    // prologue: everything blocks 0 and 1 read, [max(0, s0-cap), min(in_hi, s0 + 2B + cap))
    {
        const int lo = max(0, s0 - cap);
        const int hi = min(in_hi, s0 + 2 * AGG_B + cap);
        const float *g = src + base + (size_t)(lo + wave) * mstride;
        for (int i = lo + wave; i < hi; i += 4, g += 4 * (ptrdiff_t)mstride)
            if (kok) ring_fill(g, ring + (i & rmask) * 64);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // running pointers of this wave
    const float *fill_p = src + base + (size_t)(s0 + 2 * AGG_B + cap + wave) * mstride;   // first fill of iteration 0
    float *out_p = dst + base + (size_t)(s0 + wave * AGG_JB) * mstride;                    // first output of iteration 0
    const uint32_t *arm_p = arms + abase + (size_t)(s0 + wave * AGG_JB) * astride;
    const uint32_t *cnt_p = DIV ? cnt + abase + (size_t)(s0 + wave * AGG_JB) * astride : nullptr;
    const ptrdiff_t blk_m = (ptrdiff_t)AGG_B * mstride, blk_a = (ptrdiff_t)AGG_B * astride;

    // arms / support counts of a block are fetched (scalar loads) one iteration ahead of their use
    uint32_t a_cur[AGG_JB], c_cur[AGG_JB];
    {
        const int last = max(0, s1 - 1 - (s0 + wave * AGG_JB));
#pragma unroll
        for (int j = 0; j < AGG_JB; j++) {
            a_cur[j] = arm_p[min(j, last) * astride];
            c_cur[j] = DIV ? cnt_p[min(j, last) * astride] : 1u;
        }
    }

    for (int o0 = s0; o0 < s1; o0 += AGG_B, fill_p += blk_m, out_p += blk_m, arm_p += blk_a, cnt_p += (DIV ? blk_a : 0)) {
        // (a) fills for block i+2: positions [o0 + 2B + cap, o0 + 3B + cap); their slots are disjoint
        //     from everything blocks i and i+1 read (2*cap + 3*B <= R)
        const int n0 = o0 + 2 * AGG_B + cap;
#pragma unroll
        for (int j = 0; j < AGG_B / 4; j++) {
            const int i = n0 + wave + 4 * j;
            if (i < in_hi && kok) ring_fill(fill_p + (ptrdiff_t)(4 * j) * mstride, ring + (i & rmask) * 64);
        }
        // (b) this wave's JB consecutive outputs
        const int ob = o0 + wave * AGG_JB;
        uint32_t a_nxt[AGG_JB], c_nxt[AGG_JB];
#pragma unroll
        for (int j = 0; j < AGG_JB; j++) { a_nxt[j] = 0; c_nxt[j] = 1u; }
        if (ob + AGG_B < s1) {
            const int lastn = s1 - 1 - (ob + AGG_B);
#pragma unroll
            for (int j = 0; j < AGG_JB; j++) {
                a_nxt[j] = arm_p[blk_a + min(j, lastn) * astride];
                c_nxt[j] = DIV ? cnt_p[blk_a + min(j, lastn) * astride] : 1u;
            }
        }
        if (ob < s1) {
            const int last = s1 - 1 - ob;   // outputs past the segment repeat the last valid one (same value stored twice)
            int st[AGG_JB], en[AGG_JB], jo[AGG_JB];
            uint32_t cn[AGG_JB];
#pragma unroll
            for (int j = 0; j < AGG_JB; j++) {
                jo[j] = min(j, last);
                const uint32_t a = a_cur[j];
                st[j] = ob + jo[j] - (int)((a >> sh_neg) & 0xff);
                en[j] = ob + jo[j] + (int)((a >> sh_pos) & 0xff);
                cn[j] = c_cur[j];
            }
            const int lo = min(min(st[0], st[1]), min(st[2], st[3]));
            const int hi = max(max(en[0], en[1]), max(en[2], en[3]));
            int ca = max(max(st[0], st[1]), max(st[2], st[3]));   // first position inside every span
            int cb = min(min(en[0], en[1]), min(en[2], en[3]));   // last position inside every span
            if (ca > cb) { ca = hi + 1; cb = hi; }                // no common part: everything goes through the masked path
            float sum[AGG_JB] = {0.0f, 0.0f, 0.0f, 0.0f};
            // Range tests of the ragged parts run on the VALU: the position counter lives in a VGPR the
            // compiler cannot prove uniform, so "is p inside span j" costs vector compares/selects instead
            // of scalar-unit instructions (one scalar unit serves the whole CU and is the scarce resource here).
            int pv;
            asm volatile("v_mov_b32 %0, %1" : "=v"(pv) : "s"(lo));
            uint32_t len[AGG_JB];
#pragma unroll
            for (int j = 0; j < AGG_JB; j++) len[j] = (uint32_t)(en[j] - st[j]);
            // ragged head [lo, ca), four positions per trip so four LDS reads are in flight; a position at or
            // past `ca` is turned into one that fails every range test
            for (int p = lo; p < ca; p += 4, pv += 4) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = rl[((p + u) & rmask) * 64];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = (pv + u < ca) ? pv + u : (int)0x80000000;
#pragma unroll
                    for (int j = 0; j < AGG_JB; j++) {
                        const bool in = (uint32_t)(q - st[j]) <= len[j];
                        sum[j] += in ? v[u] : 0.0f;
                    }
                }
            }
            // common part, in at most two runs that do not wrap around the ring
            for (int p = ca; p <= cb;) {
                const int sl = p & rmask;
                const int run = min(cb - p + 1, R - sl);
                const float *q = rl + sl * 64;
                int i = 0;
                for (; i + 4 <= run; i += 4) {
                    const float v0 = q[(i + 0) * 64], v1 = q[(i + 1) * 64], v2 = q[(i + 2) * 64], v3 = q[(i + 3) * 64];
#pragma unroll
                    for (int j = 0; j < AGG_JB; j++) sum[j] += v0;
#pragma unroll
                    for (int j = 0; j < AGG_JB; j++) sum[j] += v1;
#pragma unroll
                    for (int j = 0; j < AGG_JB; j++) sum[j] += v2;
#pragma unroll
                    for (int j = 0; j < AGG_JB; j++) sum[j] += v3;
                }
                for (; i < run; i++) {
                    const float v = q[i * 64];
#pragma unroll
                    for (int j = 0; j < AGG_JB; j++) sum[j] += v;
                }
                p += run;
            }
            // ragged tail (cb, hi], same scheme
            asm volatile("v_mov_b32 %0, %1" : "=v"(pv) : "s"(cb + 1));
            for (int p = cb + 1; p <= hi; p += 4, pv += 4) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = rl[((p + u) & rmask) * 64];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = (pv + u <= hi) ? pv + u : (int)0x80000000;
#pragma unroll
                    for (int j = 0; j < AGG_JB; j++) {
                        const bool in = (uint32_t)(q - st[j]) <= len[j];
                        sum[j] += in ? v[u] : 0.0f;
                    }
                }
            }
            // always JB store instructions per wave, so the number of operations younger than last
            // iteration's fills is fixed
#pragma unroll
            for (int j = 0; j < AGG_JB; j++) {
                float r = sum[j];
                if (DIV) r = r / (float)cn[j];
                if (kok) out_p[(ptrdiff_t)jo[j] * mstride] = r;
            }
        }
#pragma unroll
        for (int j = 0; j < AGG_JB; j++) { a_cur[j] = a_nxt[j]; c_cur[j] = c_nxt[j]; }
        // (c) block i+1's fills (issued one iteration ago) must have landed before anyone reads them.
        //     Once this iteration issued fewer than 4 fills (end of the line / segment) the count of
        //     younger operations is no longer fixed: drain instead.
        if (n0 + AGG_B > in_hi) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// Fallback without LDS (any arm cap): one wave per (position, 64 disparities), spans re-read through
// the caches. Same addition order.
template <bool DIV>
__global__ void __launch_bounds__(256)
k_agg_direct(const float *__restrict__ src, float *__restrict__ dst, const uint32_t *__restrict__ arms,
             const uint32_t *__restrict__ cnt, int W, int H, int Dp, int vertical) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps span bounds in SGPRs
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
            const uint32_t c = cnt[ap];
            sum = sum / (float)c;
        }
        if (kok) dst[base + (size_t)o * mstride] = sum;
    }
}

void launch_aggregate_pass(const float *src, float *dst, const uint32_t *arms, const uint32_t *cnt,
                           const Geom &g, int vertical, int max_arm, int variant, hipStream_t s) {
    const int chunks = (g.Dp + 63) / 64;
    const int lines = vertical ? g.W : g.H;
    const int LEN = vertical ? g.H : g.W;
    int R = 64;
    while (R < 2 * max_arm + 1 + 3 * AGG_B) R <<= 1;
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
