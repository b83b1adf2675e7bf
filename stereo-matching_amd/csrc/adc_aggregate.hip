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
// so the span bounds are wave-uniform scalars, every global access of a wave is contiguous, and each
// lane adds its span in ascending t starting from 0.0f -- the reference's order
// (ADC/cost_aggregator.cu:346-372), so the sums are bit-identical to a sequential CPU evaluation.
//
// Ring kernels: a 1024-thread workgroup (16 waves) owns one scanline (a row for horizontal passes, a
// column for vertical ones) x 64*VEC disparities and marches along it. The input positions any
// in-flight block can touch live in an LDS ring of slots of 64*VEC floats filled by LDS-DMA, so each
// input element is fetched from HBM once per pass and re-read up to 2*cap+1 times from LDS (lane l
// reads floats l*VEC.. of a slot: conflict-free ds_read_b32/b64).
//   k_agg_pair   two adjacent positions per wave share their ring reads; used whenever its ring fits
//                twice into a CU's LDS; optionally computes the initial costs itself (first pass)
//   k_agg_ring   one position per wave (power-of-two ring), kept for arm caps k_agg_pair cannot hold
//   k_agg_direct no LDS, any arm cap
// The ring kernels are bound by instruction issue and by the ~5.2 TB/s a mixed read+write stream gets
// from HBM, not by LDS bandwidth.
#include "adc_common.h"

#include <algorithm>

namespace adc {

#ifndef AGG_WAVES
#define AGG_WAVES 16   // waves per workgroup = output positions per iteration
#endif
#ifndef AGG_BATCH
#define AGG_BATCH 16   // ring reads in flight per wave before the adds start
#endif

template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float T __attribute__((ext_vector_type(2))); };

// Ring fill: one LDS-DMA per 64 consecutive floats of a slot: 64 lanes x 4 B land at dst[0..63] with
// no VGPR or ds_write in between. The destination is wave-uniform; the hardware adds lane*4.
__device__ __forceinline__ void ring_fill(const float *gsrc, float *lds_dst) {
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_dst, 4, 0, 0);
}
// The second 64 floats of a 128-float slot: the instruction offset moves the global and the LDS address alike,
// so both halves use one address pair and one M0 value.
__device__ __forceinline__ void ring_fill_hi(const float *gsrc, float *lds_dst) {
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_dst, 4, 256, 0);
}

// Four floats per lane: 64 lanes x 16 B = 1 KiB land at dst in one instruction (global_load_lds_dwordx4). Every
// lane brings its own source address (16-byte aligned), so one instruction can fill neighbouring slots from
// different image positions.
__device__ __forceinline__ void ring_fill_x4(const float *gsrc, float *lds_dst) {
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// One ring read = one ds_read_b32 / ds_read_b64. The access is volatile so that the compiler does not
// pair two of them into ds_read2(st64)_b64: on gfx950 the paired form runs at half the LDS rate
// (8 cycles per wave-instruction against 2 for a single ds_read_b64).
template <typename T>
__device__ __forceinline__ T ring_read(const float *p) {
    typedef const volatile T __attribute__((address_space(3))) *lds_ptr;
    return *(lds_ptr)(uintptr_t)p;   // generic LDS address: its low 32 bits are the LDS offset
}

// Pipeline per workgroup iteration i (outputs o0 .. o0+7, wave w produces o0+w):
//   issue the fills of the positions block i+DEPTH adds to the window -> compute + store block i ->
//   wait until block i+1's fills have landed (counted vmcnt) -> barrier.
// Fills and stores retire in issue order on one counter. In steady state every wave issues VEC fills
// and one store per iteration, so at the wait of iteration i the fills issued in iteration
// i-(DEPTH-1) -- block i+1's data -- are older than exactly 1 + (DEPTH-1)*(nfill+1) operations
// (nfill = 1 when the second half of a VEC = 2 chunk lies entirely past Dp).
template <int VEC, bool DIV, int AGG_DEPTH>   // AGG_DEPTH: ring fills run this many iterations ahead of their consumer
__global__ void __launch_bounds__(AGG_WAVES * 64)
k_agg_ring(const float *__restrict__ src, float *__restrict__ dst, const uint32_t *__restrict__ arms,
           const uint32_t *__restrict__ cnt, int W, int H, int Dp, int vertical, int seglen, int cap, int rmask) {
    typedef typename VecT<VEC>::T vec_t;
    constexpr int SLOT = 64 * VEC;          // floats per ring slot
    constexpr int B = AGG_WAVES;
    extern __shared__ __align__(16) float ring[];  // (rmask+1) x SLOT
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps span bounds in SGPRs
    const int kc = blockIdx.x * SLOT;       // first disparity of this workgroup's chunk
    const int line = blockIdx.y;
    const int LEN = vertical ? H : W;
    const int s0 = blockIdx.z * seglen;
    if (s0 >= LEN) return;
    const int s1 = min(LEN, s0 + seglen);
    // Strides fit 32 bits (checked by the launcher); only base offsets are 64-bit. Per-iteration
    // addresses are running pointers advanced by 32-bit strides: the scalar unit is shared by the
    // whole CU and 64-bit multiplies on it are what this kernel can least afford.
    const int mstride = vertical ? W * Dp : Dp;          // elements between march positions
    const int astride = vertical ? W : 1;                // pixels between march positions
    const size_t lbase = (vertical ? (size_t)line * Dp : (size_t)line * W * Dp) + (size_t)kc;
    const size_t abase = vertical ? (size_t)line : (size_t)line * W;
    const int in_hi = min(LEN, s1 + cap);   // exclusive upper bound of input positions this segment needs
    const int sh_neg = vertical ? 16 : 0, sh_pos = vertical ? 24 : 8;
    const int R = rmask + 1;
    // fill addressing: lane <-> float `lane` (+64 for the second half) of the slot
    bool fok[VEC];
#pragma unroll
    for (int h = 0; h < VEC; h++) fok[h] = kc + h * 64 + lane < Dp;
    const int nfill = (VEC == 2 && kc + 64 >= Dp) ? 1 : VEC;   // fill instructions this workgroup's waves issue per position
    const float *fsrc = src + lbase + lane;
    // compute addressing: lane <-> floats lane*VEC .. lane*VEC+VEC-1 of the slot
    const bool kok = kc + lane * VEC < Dp;  // Dp is a multiple of VEC
    const float *rl = ring + lane * VEC;

    // prologue: everything blocks 0 .. DEPTH-1 read: [max(0, s0-cap), min(in_hi, s0 + DEPTH*B + cap))
    {
        const int lo = max(0, s0 - cap);
        const int hi = min(in_hi, s0 + AGG_DEPTH * B + cap);
        const float *g = fsrc + (size_t)(lo + wave) * mstride;
        for (int i = lo + wave; i < hi; i += B, g += (ptrdiff_t)B * mstride) {
#pragma unroll
            for (int h = 0; h < VEC; h++)
                if (fok[h]) ring_fill(g + h * 64, ring + (i & rmask) * SLOT + h * 64);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // running pointers of this wave
    const float *fill_p = fsrc + (size_t)(s0 + AGG_DEPTH * B + cap + wave) * mstride;   // this wave's fill of iteration 0
    float *out_p = dst + lbase + lane * VEC + (size_t)(s0 + wave) * mstride;            // this wave's output of iteration 0
    const uint32_t *arm_p = arms + abase + (size_t)(s0 + wave) * astride;
    const uint32_t *cnt_p = DIV ? cnt + abase + (size_t)(s0 + wave) * astride : nullptr;
    const ptrdiff_t blk_m = (ptrdiff_t)B * mstride, blk_a = (ptrdiff_t)B * astride;

    // arms / support count of an output are fetched (scalar loads) one iteration ahead of their use
    uint32_t a_cur = 0, c_cur = 1;
    if (s0 + wave < s1) {
        a_cur = arm_p[0];
        if (DIV) c_cur = cnt_p[0];
    }

    for (int o0 = s0; o0 < s1; o0 += B, fill_p += blk_m, out_p += blk_m, arm_p += blk_a, cnt_p += (DIV ? blk_a : 0)) {
        // (a) fill of the position block i+DEPTH adds to the window; its slot is not read by blocks i .. i+DEPTH-1
        const int n0 = o0 + AGG_DEPTH * B + cap;
        if (n0 + wave < in_hi) {
#pragma unroll
            for (int h = 0; h < VEC; h++)
                if (fok[h]) ring_fill(fill_p + h * 64, ring + ((n0 + wave) & rmask) * SLOT + h * 64);
        }
        const int o = o0 + wave;
        uint32_t a_nxt = 0, c_nxt = 1;
        if (o + B < s1) {
            a_nxt = arm_p[blk_a];
            if (DIV) c_nxt = cnt_p[blk_a];
        }
        // (b) this wave's output
        if (o < s1) {
            const int neg = (a_cur >> sh_neg) & 0xff, pos = (a_cur >> sh_pos) & 0xff;
#if defined(AGG_DIAG) && AGG_DIAG == 1
            const int n = 1;                        // diagnostic build: one term per output
#else
            const int n = neg + pos + 1;
#endif
            const int sl = (o - neg) & rmask;
            const int n1 = min(n, R - sl);          // terms before the ring wraps
            vec_t acc = {};
            // Reads are issued in batches of AGG_BATCH before the (ordered, dependent) adds consume them:
            // the span is a serial chain, so what matters is how few LDS round trips it is cut into.
            const float *q = rl + sl * SLOT;
            int i = 0;
            for (; i + AGG_BATCH <= n1; i += AGG_BATCH) {
                vec_t v[AGG_BATCH];
#pragma unroll
                for (int u = 0; u < AGG_BATCH; u++) v[u] = ring_read<vec_t>(q + (i + u) * SLOT);
#pragma unroll
                for (int u = 0; u < AGG_BATCH; u++) acc += v[u];
            }
            for (; i + 4 <= n1; i += 4) {
                vec_t v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = ring_read<vec_t>(q + (i + u) * SLOT);
#pragma unroll
                for (int u = 0; u < 4; u++) acc += v[u];
            }
            for (; i < n1; i++) acc += ring_read<vec_t>(q + i * SLOT);
            const int n2 = n - n1;                  // terms after the wrap, from slot 0
            i = 0;
            for (; i + AGG_BATCH <= n2; i += AGG_BATCH) {
                vec_t v[AGG_BATCH];
#pragma unroll
                for (int u = 0; u < AGG_BATCH; u++) v[u] = ring_read<vec_t>(rl + (i + u) * SLOT);
#pragma unroll
                for (int u = 0; u < AGG_BATCH; u++) acc += v[u];
            }
            for (; i + 4 <= n2; i += 4) {
                vec_t v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = ring_read<vec_t>(rl + (i + u) * SLOT);
#pragma unroll
                for (int u = 0; u < 4; u++) acc += v[u];
            }
            for (; i < n2; i++) acc += ring_read<vec_t>(rl + i * SLOT);
            if (DIV) acc = acc / (float)c_cur;
            if (kok) *reinterpret_cast<vec_t *>(out_p) = acc;
        }
        a_cur = a_nxt;
        c_cur = c_nxt;
        // (c) block i+1's fills must have landed before anyone reads them. Once a wave skipped a fill
        //     (end of the line / segment) the count of younger operations is no longer fixed: drain.
        if (n0 + B > in_hi) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nfill == VEC) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + (AGG_DEPTH - 1) * (VEC + 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + (AGG_DEPTH - 1) * 2) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// ---- two adjacent positions per wave ------------------------------------------------------------------
// The windows of two neighbouring positions of a scanline overlap almost entirely. A wave that owns
// both reads every ring slot of the overlap once and adds it to both running sums (two independent
// dependency chains that interleave), then finishes the few terms only one of them has. Each sum
// still receives exactly its own terms in ascending position order. Per output this halves the LDS
// reads, the barriers and the per-iteration bookkeeping -- the kernel is bound by instruction issue.
//
// sum of n consecutive ring slots starting at slot s (no wrap inside) added to one / two chains
// G consecutive slots from q added to one chain (CH == 1) or to both (CH == 2): all reads, then the adds
template <int VEC, int CH, int G>
__device__ __forceinline__ void group_add(typename VecT<VEC>::T &a0, typename VecT<VEC>::T &a1, const float *q) {
    typedef typename VecT<VEC>::T vec_t;
    constexpr int SLOT = 64 * VEC;
    vec_t v[G];
#pragma unroll
    for (int u = 0; u < G; u++) v[u] = ring_read<vec_t>(q + u * SLOT);
#pragma unroll
    for (int u = 0; u < G; u++) {
        a0 += v[u];
        if (CH == 2) a1 += v[u];
    }
}
// n consecutive ring slots starting at slot s (no wrap inside): batches of 16, then the remainder as
// one group each of 8, 4, 2, 1 terms as the bits of n say (four tests instead of a counted loop)
template <int VEC, int CH>
__device__ __forceinline__ void span(typename VecT<VEC>::T &a0, typename VecT<VEC>::T &a1, const float *rl, int s, int n) {
    constexpr int SLOT = 64 * VEC;
    const float *q = rl + s * SLOT;
    for (int i = 16; i <= n; i += 16, q += 16 * SLOT) group_add<VEC, CH, 16>(a0, a1, q);
    if (n & 8) { group_add<VEC, CH, 8>(a0, a1, q); q += 8 * SLOT; }
    if (n & 4) { group_add<VEC, CH, 4>(a0, a1, q); q += 4 * SLOT; }
    if (n & 2) { group_add<VEC, CH, 2>(a0, a1, q); q += 2 * SLOT; }
    if (n & 1) group_add<VEC, CH, 1>(a0, a1, q);
}
template <int VEC>
__device__ __forceinline__ void span1(typename VecT<VEC>::T &acc, const float *rl, int s, int n) {
    span<VEC, 1>(acc, acc, rl, s, n);
}
template <int VEC>
__device__ __forceinline__ void span2(typename VecT<VEC>::T &a0, typename VecT<VEC>::T &a1, const float *rl, int s, int n) {
    span<VEC, 2>(a0, a1, rl, s, n);
}
// the same over virtual slots vs .. vs+n-1 of an R-slot ring (vs in (-R, 2R)): split at the wrap
template <int VEC>
__device__ __forceinline__ void range1(typename VecT<VEC>::T &acc, const float *rl, int vs, int n, int R) {
    if (n <= 0) return;
    int s = vs < 0 ? vs + R : (vs >= R ? vs - R : vs);
    const int n1 = min(n, R - s);
    span1<VEC>(acc, rl, s, n1);
    if (n > n1) span1<VEC>(acc, rl, 0, n - n1);
}
template <int VEC>
__device__ __forceinline__ void range2(typename VecT<VEC>::T &a0, typename VecT<VEC>::T &a1, const float *rl, int vs, int n, int R) {
    if (n <= 0) return;
    int s = vs < 0 ? vs + R : (vs >= R ? vs - R : vs);
    const int n1 = min(n, R - s);
    span2<VEC>(a0, a1, rl, s, n1);
    if (n > n1) span2<VEC>(a0, a1, rl, 0, n - n1);
}

// a / c, correctly rounded (the same bits as the IEEE division the reference does), for a sum a >= 0 that
// is zero or a normal number and a support count c that is an integer in [1, 2^14]:
//     q0 = a * rcp(c);   e = fma(-c, q0, a);   q = fma(e, rcp(c), q0)
// q0 is within two ulps, so the remainder e = a - c*q0 is a small multiple of ulp(q0) and exact; e*rcp(c)
// misses e/c by less than 2^-22 ulp, while a/c keeps at least ulp/(2c) >= 2^-15 ulp away from every
// rounding boundary (a/c cannot be a midpoint between two floats for an integer c). Three instructions
// per value instead of the scale / refine / fix-up sequence of a general division.
template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::T div_by_count(typename VecT<VEC>::T a, float c) {
    const float r = __builtin_amdgcn_rcpf(c);
    if constexpr (VEC == 1) {
        const float q0 = a * r;
        const float e = __builtin_fmaf(-c, q0, a);
        return __builtin_fmaf(e, r, q0);
    } else {
        typedef typename VecT<VEC>::T vec_t;
        const vec_t rv = {r, r}, nc = {-c, -c};
        const vec_t q0 = a * rv;
        const vec_t e = __builtin_elementwise_fma(nc, q0, a);
        return __builtin_elementwise_fma(e, rv, q0);
    }
}

// Workgroup iteration i: wave w produces positions o0 + 2w and o0 + 2w + 1 (B = 32 per iteration);
// the fills of the block iteration i+1 adds to the window are issued first and must have landed by the
// end of the iteration: they are then older than exactly the two stores that follow them.
// Ring: R >= 2*cap + 2*B slots (any number; slots are tracked as running counters, not by masking).
#define PAIR_B (2 * AGG_WAVES)
// COST: the pass is the first one (horizontal, no division) and its input is not read from a volume but
// computed on the spot: ComputeCostCuda's expression (ADC/cost_computor.cu:151-207)
//     ((1 - e_ad[sum|dBGR|]) + 1) - e_cen[popcount(cenL ^ cenR)],   1.0f where x-d is outside the image,
// evaluated per ring slot from the right row's {pixel, census} entries and two tables staged in LDS:
// t_ad[s] = (1 - e_ad[s]) + 1 (the same two fp32 operations, done once per table entry on the host instead of
// once per volume element here) and e_cen. The initial cost volume then never exists in HBM (one volume write
// and one volume read less).
struct CostSrc {
    const uint32_t *pix_l, *pix_r, *cen_l, *cen_r;
    const float *lut_ad1, *lut_cen;   // t_ad[766], e_cen[65]
    int dmin, D;
};
// (defined with the fused kernel below) the sums of the two windows of a pair described by a pair descriptor
template <int VEC>
__device__ __forceinline__ void desc_windows(typename VecT<VEC>::T &acc0, typename VecT<VEC>::T &acc1, uint32_t base, int R,
                                             int q, uint32_t d);
// DESC: the window geometry of a wave's pair comes from the line's pair descriptors (see k_pair_desc below: worked out
// once per frame, with the support counts as floats) instead of from the arms and counts of the two pixels -- less
// than half the scalar instructions and branches per pair. Horizontal passes of frames that build the descriptors.
// FULLD: Dp is a multiple of the chunk (64*VEC disparities), so every lane of every chunk is live and the
// fills / stores need no lane predicate (each predicate costs an execution-mask save, a branch and a restore).
template <int VEC, bool DIV, bool COST, bool FULLD, bool DESC>
__global__ void __launch_bounds__(AGG_WAVES * 64)
k_agg_pair(const float *__restrict__ src, float *__restrict__ dst, const uint32_t *__restrict__ arms,
           const uint32_t *__restrict__ cnt, int W, int H, int Dp, int vertical, int seglen, int cap, int R,
           CostSrc cs, const uint4 *__restrict__ desc, int dpad) {
    typedef typename VecT<VEC>::T vec_t;
    constexpr int SLOT = 64 * VEC;
    constexpr int B = PAIR_B;
    extern __shared__ __align__(16) float ring[];  // R x SLOT
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kc = blockIdx.x * SLOT;
    const int line = blockIdx.y;
    const int LEN = vertical ? H : W;
    const int s0 = blockIdx.z * seglen;
    if (s0 >= LEN) return;
    const int s1 = min(LEN, s0 + seglen);
    const int in_lo = max(0, s0 - cap), in_hi = min(LEN, s1 + cap);
    const int mstride = vertical ? W * Dp : Dp;
    const int astride = vertical ? W : 1;
    const size_t lbase = (vertical ? (size_t)line * Dp : (size_t)line * W * Dp) + (size_t)kc;
    const size_t abase = vertical ? (size_t)line : (size_t)line * W;
    const int sh_neg = vertical ? 16 : 0, sh_pos = vertical ? 24 : 8;
    bool fok[VEC];
#pragma unroll
    for (int h = 0; h < VEC; h++) fok[h] = FULLD || kc + h * 64 + lane < Dp;
    const float *fsrc = src + lbase + lane;
    const bool kok = FULLD || kc + lane * VEC < Dp;
    const float *rl = ring + lane * VEC;

    // COST: right-row entries any (position, disparity) pair of this segment and chunk can touch, i.e.
    // columns xr_base .. xr_base + nr - 1 with xr = p - dmin - k, and the two exp tables
    uint2 *s_r = reinterpret_cast<uint2 *>(ring + (size_t)R * SLOT);
    const int nr = COST ? (in_hi - in_lo) + SLOT - 1 : 0;
    float *s_ad = reinterpret_cast<float *>(s_r + nr);     // 766 (+2)
    float *s_cen = s_ad + 768;                             // 65 (+3)
    const int xr_base = in_lo - cs.dmin - kc - (SLOT - 1);
    if (COST) {
        for (int i = threadIdx.x; i < nr; i += AGG_WAVES * 64) {
            const int xr = xr_base + i;
            uint2 v = make_uint2(0u, 0u);
            if (xr >= 0 && xr < W) {
                const size_t q = (size_t)line * W + xr;
                v = make_uint2(cs.pix_r[q], cs.cen_r[q]);
            }
            s_r[i] = v;
        }
        for (int i = threadIdx.x; i < 766; i += AGG_WAVES * 64) s_ad[i] = cs.lut_ad1[i];
        if (threadIdx.x < 65) s_cen[threadIdx.x] = cs.lut_cen[threadIdx.x];
        __syncthreads();
    }
    // costs of position p (a column of row `line`) for this chunk's disparities into ring slot q
    const bool has_pad = COST && cs.D != Dp;               // (uniform: most sizes have no padding disparities at all)
    auto cost_fill = [&](int p, int q) {
        const size_t pi = (size_t)line * W + p;
        const uint32_t pl = cs.pix_l[pi], cl = cs.cen_l[pi];
#pragma unroll
        for (int h = 0; h < VEC; h++) {
            const int jj = h * 64 + lane;                  // disparity index inside the chunk = float inside the slot
            const int xr = p - cs.dmin - kc - jj;
            float c = 1.0f;
            if ((unsigned)xr < (unsigned)W) {
                const uint2 r = s_r[xr - xr_base];
                const uint32_t sad = __builtin_amdgcn_sad_u8(pl, r.x, 0u);
                const uint32_t ham = __builtin_popcount(cl ^ r.y);
                c = s_ad[sad] - s_cen[ham];
            }
            if (has_pad && kc + jj >= cs.D) c = 0.0f;      // padding past D holds zeros, as the volume does
            ring[q * SLOT + jj] = c;
        }
    };

    // prologue: input position in_lo + j lands in slot j; everything block 0 reads
    {
        const int pro_n = min(in_hi, s0 + B + cap) - in_lo;     // <= R - B
        if (COST) {
            for (int j = wave; j < pro_n; j += AGG_WAVES) cost_fill(in_lo + j, j);
        } else {
            const float *g = fsrc + (size_t)(in_lo + wave) * mstride;
            for (int j = wave; j < pro_n; j += AGG_WAVES, g += (ptrdiff_t)AGG_WAVES * mstride) {
#pragma unroll
                for (int h = 0; h < VEC; h++)
                    if (fok[h]) ring_fill(g + h * 64, ring + j * SLOT + h * 64);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int o = s0 + 2 * wave;                        // first of this wave's two positions
    int qo = s0 - in_lo + 2 * wave;               // its ring slot (< R)
    int qf = s0 - in_lo + B + cap + 2 * wave;     // ring slot of this wave's first fill of the iteration (< R)
    const ptrdiff_t blk_m = (ptrdiff_t)B * mstride, blk_a = (ptrdiff_t)B * astride;
    const float *fill_p = fsrc + (size_t)(s0 + B + cap + 2 * wave) * mstride;
    float *out_p = dst + lbase + lane * VEC + (size_t)o * mstride;
    const uint32_t *arm_p = arms + abase + (size_t)o * astride;
    const uint32_t *cnt_p = DIV ? cnt + abase + (size_t)o * astride : nullptr;

    // arms / support counts -- or the pair's descriptor -- are fetched (scalar loads) one iteration ahead of their use
    uint32_t a_cur[2] = {0u, 0u}, c_cur[2] = {1u, 1u};
    const uint4 *dsc_p = DESC ? desc + (size_t)line * (size_t)(((LEN + 1) >> 1) + 2 * dpad) + dpad + (o >> 1) : nullptr;
    uint4 d_cur = make_uint4(0u, 0u, 0u, 0u);
    const uint32_t ro = (uint32_t)(uintptr_t)rl;   // this lane's floats of slot 0 as an LDS byte offset
    if (DESC) {
        if (o < s1) d_cur = dsc_p[0];
    } else {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (o + j < s1) {
                a_cur[j] = arm_p[j * astride];
                if (DIV) c_cur[j] = cnt_p[j * astride];
            }
        }
    }

    bool fills_live = true;
    for (int o0 = s0; o0 < s1; o0 += B) {
        // (a) fills of the block the next iteration adds to the window
        const int n0 = o0 + B + cap;
        __builtin_amdgcn_s_setprio(3);   // get the fills out ahead of the other waves' arithmetic (about 0.5 % per pass in an alternating A/B)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (n0 + 2 * wave + j < in_hi) {
                int q = qf + j;
                if (q >= R) q -= R;
                if (COST) {
                    cost_fill(n0 + 2 * wave + j, q);
                } else {
                    if constexpr (FULLD && VEC == 2) {
                        ring_fill(fill_p + (ptrdiff_t)j * mstride, ring + q * SLOT);
                        ring_fill_hi(fill_p + (ptrdiff_t)j * mstride, ring + q * SLOT);
                    } else {
#pragma unroll
                        for (int h = 0; h < VEC; h++)
                            if (fok[h]) ring_fill(fill_p + (ptrdiff_t)j * mstride + h * 64, ring + q * SLOT + h * 64);
                    }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        uint32_t a_nxt[2] = {0u, 0u}, c_nxt[2] = {1u, 1u};
        uint4 d_nxt = make_uint4(0u, 0u, 0u, 0u);
        if (DESC) {
            if (o + B < s1) d_nxt = dsc_p[B / 2];
        } else {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (o + B + j < s1) {
                    a_nxt[j] = arm_p[blk_a + j * astride];
                    if (DIV) c_nxt[j] = cnt_p[blk_a + j * astride];
                }
            }
        }
        // (b) this wave's two outputs. Windows in virtual ring slots (slot of position o is qo):
        //     chain 0 = [qo - neg0, qo + pos0], chain 1 = [qo + 1 - neg1, qo + 1 + pos1]
        if (DESC) {
            if (d_cur.x) {               // (zero: no position here -- behind the end of the segment or of the line)
                vec_t acc0, acc1;
                desc_windows<VEC>(acc0, acc1, ro, R, qo, d_cur.x);
                if (DIV) {
                    acc0 = div_by_count<VEC>(acc0, __uint_as_float(d_cur.y));
                    acc1 = div_by_count<VEC>(acc1, __uint_as_float(d_cur.z));
                }
                if (kok) {
                    *reinterpret_cast<vec_t *>(out_p) = acc0;
                    if (d_cur.x >> 31) *reinterpret_cast<vec_t *>(out_p + mstride) = acc1;
                }
            }
        } else if (o < s1) {
            const bool two = o + 1 < s1;
            const int b0 = qo - (int)((a_cur[0] >> sh_neg) & 0xff), e0 = qo + (int)((a_cur[0] >> sh_pos) & 0xff);
            // a missing second position gets an empty window placed after the first one's
            const int b1 = two ? qo + 1 - (int)((a_cur[1] >> sh_neg) & 0xff) : e0 + 1;
            const int e1 = two ? qo + 1 + (int)((a_cur[1] >> sh_pos) & 0xff) : e0;
            vec_t acc0 = {}, acc1 = {};
            // head: the terms in front of the other chain's first term
            if (b0 < b1) range1<VEC>(acc0, rl, b0, min(e0, b1 - 1) - b0 + 1, R);
            else if (b1 < b0) range1<VEC>(acc1, rl, b1, min(e1, b0 - 1) - b1 + 1, R);
            // middle: the terms both windows hold
            const int m0 = max(b0, b1), m1 = min(e0, e1);
            if (b0 == b1) {   // a common start: one chain of additions serves both windows up to the end of the shorter one
                range1<VEC>(acc0, rl, m0, m1 - m0 + 1, R);
                acc1 = acc0;
            } else {
                range2<VEC>(acc0, acc1, rl, m0, m1 - m0 + 1, R);
            }
            // tail: the terms behind the other chain's last term
            if (e0 > e1) { const int t = max(b0, e1 + 1); range1<VEC>(acc0, rl, t, e0 - t + 1, R); }
            else if (e1 > e0) { const int t = max(b1, e0 + 1); range1<VEC>(acc1, rl, t, e1 - t + 1, R); }
            if (DIV) {
                acc0 = div_by_count<VEC>(acc0, (float)c_cur[0]);
                acc1 = div_by_count<VEC>(acc1, (float)c_cur[1]);
            }
            if (kok) {
                *reinterpret_cast<vec_t *>(out_p) = acc0;
                if (two) *reinterpret_cast<vec_t *>(out_p + mstride) = acc1;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; j++) { a_cur[j] = a_nxt[j]; c_cur[j] = c_nxt[j]; }
        d_cur = d_nxt;
        if (DESC) dsc_p += B / 2;
        // (c) the next block's fills must have landed before anyone reads them. Once a fill has been
        //     skipped (end of the line / segment) no further fill follows: drain once, then stop waiting.
        if (!COST && fills_live) {            // (COST: the ring is written by ds_write, ordered by the wait below)
            if (n0 + B > in_hi) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                fills_live = false;
            } else {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        o += B;
        qo += B; if (qo >= R) qo -= R;
        qf += B; if (qf >= R) qf -= R;
        fill_p += blk_m;
        out_p += blk_m;
        arm_p += blk_a;
        if (DIV) cnt_p += blk_a;
    }
}

// ---- two consecutive passes along the same axis in one launch -------------------------------------------
// The eight passes of an aggregation run H V | V H | H V | V H: passes 1+2, 3+4 and 5+6 are a dividing pass
// followed by a plain pass along the SAME axis, and along one scanline the second only needs the first one's
// results of that scanline. k_agg_fused runs such a pair in one sweep:
//     stage 1   A[p] = ( sum over p's arm span of src ) / count[p]        (the dividing pass)
//     stage 2   dst[o] = sum over o's arm span of A                        (the pass that follows it)
// with A kept in a second LDS ring instead of travelling to HBM and back (one volume read and one volume
// write for two passes). Stage 1 runs cap + one block ahead of stage 2, so that every A a stage-2 window
// touches was written before the previous barrier. Each sum is the same ascending sequence of additions as
// in the unfused passes, the division is the same exact one, and A is a float either way: identical bits.
// Both rings hold 2*cap + 2*B slots; with 128 disparities per slot that is one workgroup per CU, so the
// kernel lives on how few instructions a wave issues per window. The window geometry of a pair of positions
// depends on the arms only: it is worked out once per frame and axis (k_pair_desc) instead of by every
// wave of every pass and disparity chunk.
//
// Pair descriptor, for the positions p (even) and p + 1 of a line. The two windows [p-neg0, p+pos0] and
// [p+1-neg1, p+1+pos1] always form one contiguous run of ring slots: head (terms only one window has, in front
// of the other's first term), middle (terms of both), tail (terms only one has, behind the other's last).
//   bits  0..6   first slot of the run, as a distance below p          (0 .. 127)
//   bits  7..13  head length      bits 14..21  middle length      bits 22..28  tail length
//   bit   29     the head belongs to the second window      bit 30   the tail belongs to the second window
//   bit   31     the pair has a second position (only the last pair of a line of odd length has none)
// A lone last position has an empty second window: everything is head of the first.
// {descriptor, (float)count of p, (float)count of p+1, 0} per pair; the pairs of a line are contiguous and every
// line is padded with pair_desc_pad() all-zero records in front and behind (zeroed when the handle is initialised,
// never written): a zero descriptor is "no position here", so the fused kernel needs no range checks of its own
// for the blocks it processes before the start and behind the end of a line.
// (the kernel starts cap + B positions, i.e. up to cap + B pairs' worth of blocks, before a line and its look-ahead
// ends up to cap/2 + 3B/2 pairs behind it)
int pair_desc_pad(int max_arm) { return ((max_arm + 1) & ~1) + 2 * PAIR_B + 2; }
size_t pair_desc_count(int lines, int len, int max_arm) { return (size_t)lines * ((len + 1) / 2 + 2 * pair_desc_pad(max_arm)); }

__global__ void __launch_bounds__(256)
k_pair_desc(const uint32_t *__restrict__ arms, const uint32_t *__restrict__ cnt_for_h, const uint32_t *__restrict__ cnt_for_v,
            uint4 *__restrict__ desc_h, uint4 *__restrict__ desc_v, int W, int H, int pad) {
    const int vertical = blockIdx.z;
    const int LEN = vertical ? H : W, lines = vertical ? W : H;
    const int npairs = (LEN + 1) >> 1;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)lines * npairs) return;
    // consecutive threads walk along image rows in both cases (coalesced arm reads)
    int line, pair;
    if (vertical) { line = (int)(gid % W); pair = (int)(gid / W); }
    else { line = (int)(gid / npairs); pair = (int)(gid - (size_t)line * npairs); }
    const int p = 2 * pair;
    const bool two = p + 1 < LEN;
    const size_t i0 = vertical ? (size_t)p * W + line : (size_t)line * W + p;
    const size_t i1 = two ? (vertical ? i0 + W : i0 + 1) : i0;
    const uint32_t a0 = arms[i0], a1 = arms[i1];
    const int sh_neg = vertical ? 16 : 0, sh_pos = vertical ? 24 : 8;
    const int b0 = -(int)((a0 >> sh_neg) & 0xff), e0 = (int)((a0 >> sh_pos) & 0xff);
    const int b1 = two ? 1 - (int)((a1 >> sh_neg) & 0xff) : e0 + 1;
    const int e1 = two ? 1 + (int)((a1 >> sh_pos) & 0xff) : e0;
    const bool h0 = b0 < b1, h1 = b1 < b0;
    const int hn = h0 ? min(e0, b1 - 1) - b0 + 1 : (h1 ? min(e1, b0 - 1) - b1 + 1 : 0);
    const int m0 = max(b0, b1), m1 = min(e0, e1);
    const int mn = max(m1 - m0 + 1, 0);
    const bool t0 = e0 > e1, t1 = e1 > e0;
    const int tn = t0 ? e0 - max(b0, e1 + 1) + 1 : (t1 ? e1 - max(b1, e0 + 1) + 1 : 0);
    const uint32_t d = (uint32_t)(-min(b0, b1)) | (uint32_t)hn << 7 | (uint32_t)mn << 14 | (uint32_t)tn << 22 |
                       (h1 ? 1u << 29 : 0u) | (t1 ? 1u << 30 : 0u) | (two ? 1u << 31 : 0u);
    const uint32_t *cnt = vertical ? cnt_for_v : cnt_for_h;
    const uint4 rec = make_uint4(d, __float_as_uint((float)cnt[i0]), __float_as_uint((float)cnt[i1]), 0u);
    (vertical ? desc_v : desc_h)[(size_t)line * (npairs + 2 * pad) + pad + pair] = rec;
}

// cnt_for_h: the counts horizontal dividing passes divide by (the vertical-first order's), cnt_for_v: the others
void launch_pair_descriptors(const uint32_t *arms, const uint32_t *cnt_for_h, const uint32_t *cnt_for_v, uint4 *desc_h,
                             uint4 *desc_v, const Geom &g, int max_arm, hipStream_t s) {
    const size_t n = std::max((size_t)g.H * ((g.W + 1) / 2), (size_t)g.W * ((g.H + 1) / 2));
    k_pair_desc<<<dim3((unsigned)((n + 255) / 256), 1, 2), dim3(256), 0, s>>>(arms, cnt_for_h, cnt_for_v, desc_h, desc_v, g.W, g.H,
                                                                              pair_desc_pad(max_arm));
}

// ---- window sums of the fused kernel: ring positions as 32-bit LDS byte offsets ---------------------------------
template <typename T>
__device__ __forceinline__ T ring_read_o(uint32_t off) {
    typedef const volatile T __attribute__((address_space(3))) *lds_ptr;
    return *(lds_ptr)(uintptr_t)off;
}
// G consecutive slots from byte offset o added to one chain (CH == 1) or to both (CH == 2): all reads, then the adds
template <int VEC, int CH, int G>
__device__ __forceinline__ void group_add_o(typename VecT<VEC>::T &a0, typename VecT<VEC>::T &a1, uint32_t o) {
    typedef typename VecT<VEC>::T vec_t;
    constexpr uint32_t SB = 256 * VEC;          // bytes per slot
    vec_t v[G];
#pragma unroll
    for (int u = 0; u < G; u++) v[u] = ring_read_o<vec_t>(o + u * SB);
#pragma unroll
    for (int u = 0; u < G; u++) {
        a0 += v[u];
        if (CH == 2) a1 += v[u];
    }
}
// n consecutive slots (no wrap inside): batches of 16, then one group each of 8, 4, 2, 1 terms as the bits of n say
template <int VEC, int CH>
__device__ __forceinline__ void span_o(typename VecT<VEC>::T &a0, typename VecT<VEC>::T &a1, uint32_t o, int n) {
    constexpr uint32_t SB = 256 * VEC;
    for (int i = 16; i <= n; i += 16, o += 16 * SB) group_add_o<VEC, CH, 16>(a0, a1, o);
    if (n & 8) { group_add_o<VEC, CH, 8>(a0, a1, o); o += 8 * SB; }
    if (n & 4) { group_add_o<VEC, CH, 4>(a0, a1, o); o += 4 * SB; }
    if (n & 2) { group_add_o<VEC, CH, 2>(a0, a1, o); o += 2 * SB; }
    if (n & 1) group_add_o<VEC, CH, 1>(a0, a1, o);
}
// Head and tail terms belong to one of the two windows only. Instead of summing them into a copy of that window's
// accumulator and selecting the results back (four conditional moves per part), every such term goes to BOTH
// accumulators through a fused multiply-add with a wave-uniform factor: 1.0f for the window it belongs to, 0.0f for
// the other. fma(v, 1, a) rounds v + a exactly as the addition does, fma(v, 0, a) is a + 0 = a (v is a finite cost,
// a >= +0): the same bits as before. Heads and tails are one term long most of the time.
template <int VEC>
__device__ __forceinline__ void masked_term(typename VecT<VEC>::T &a0, typename VecT<VEC>::T &a1, typename VecT<VEC>::T v, float m0, float m1) {
    if constexpr (VEC == 1) {
        a0 = __builtin_fmaf(v, m0, a0);
        a1 = __builtin_fmaf(v, m1, a1);
    } else {
        typedef typename VecT<VEC>::T vec_t;
        const vec_t w0 = {m0, m0}, w1 = {m1, m1};
        a0 = __builtin_elementwise_fma(v, w0, a0);
        a1 = __builtin_elementwise_fma(v, w1, a1);
    }
}
template <int VEC>
__device__ __forceinline__ void masked_span_o(typename VecT<VEC>::T &a0, typename VecT<VEC>::T &a1, uint32_t o, int n, float m0, float m1) {
    typedef typename VecT<VEC>::T vec_t;
    constexpr uint32_t SB = 256 * VEC;
    if (n == 1) {
        masked_term<VEC>(a0, a1, ring_read_o<vec_t>(o), m0, m1);
        return;
    }
    for (; n >= 4; n -= 4, o += 4 * SB) {
        vec_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = ring_read_o<vec_t>(o + u * SB);
#pragma unroll
        for (int u = 0; u < 4; u++) masked_term<VEC>(a0, a1, v[u], m0, m1);
    }
    for (; n > 0; n--, o += SB) masked_term<VEC>(a0, a1, ring_read_o<vec_t>(o), m0, m1);
}

// The sums of the two windows of a pair from the ring at LDS byte offset `base` (this lane's floats; R slots);
// q = slot of the pair's first position.
template <int VEC>
__device__ __forceinline__ void desc_windows(typename VecT<VEC>::T &acc0, typename VecT<VEC>::T &acc1, uint32_t base, int R,
                                             int q, uint32_t d) {
    typedef typename VecT<VEC>::T vec_t;
    constexpr uint32_t SB = 256 * VEC;
#ifdef AGG_DIAG_ONE_TERM   // diagnostic: every window is one term (what is left is the kernel's memory skeleton)
    d = (d & 0x80000000u) | (1u << 14);
#endif
#ifdef AGG_DIAG_UNIFORM    // diagnostic: every pair has the same windows (1 + 28 + 1 terms, the frame's mean): no imbalance between waves
    d = 0x80000000u | 14u | (1u << 7) | (28u << 14) | (1u << 22) | (1u << 30);
#endif
    const int hn = (d >> 7) & 127, mn = (d >> 14) & 255, tn = (d >> 22) & 127;
    // factors of the head's and the tail's terms for window 0 / window 1
    // (1.0f or 0.0f made from the descriptor's bits by integer arithmetic: they stay in scalar registers)
    const uint32_t one = 0x3f800000u;
    const uint32_t h1b = ((d >> 29) & 1u) * one, t1b = ((d >> 30) & 1u) * one;
    const float h1 = __uint_as_float(h1b), h0 = __uint_as_float(one - h1b);
    const float t1 = __uint_as_float(t1b), t0 = __uint_as_float(one - t1b);
    int u = q - (int)(d & 127);
    if (u < 0) u += R;
    const vec_t zero = {};
    if (u + hn + mn + tn <= R) {
        // the run does not cross the end of the ring (three times out of four)
        uint32_t o = base + (uint32_t)u * SB;
        if (hn == 0) {
            // both windows start at the same place: up to the end of the shorter one they are one and the same chain
            // of additions (a third to a half of all pairs: both arms stop at the same edge)
            vec_t c = zero;
            span_o<VEC, 1>(c, c, o, mn);
            acc0 = c;
            acc1 = c;
        } else {
            acc0 = zero;
            acc1 = zero;
            masked_span_o<VEC>(acc0, acc1, o, hn, h0, h1);
            o += (uint32_t)hn * SB;
            span_o<VEC, 2>(acc0, acc1, o, mn);
        }
        o += (uint32_t)mn * SB;
        if (tn) masked_span_o<VEC>(acc0, acc1, o, tn, t0, t1);
    } else {
        // the run crosses the end of the ring somewhere: every part is split at the wrap
        auto part = [&](int vs, int n, int kind) {   // kind 0: head, 1: single chain, 2: both chains, 3: tail
            if (n <= 0) return;
            const int n1 = min(n, R - vs);
            for (int half = 0; half < 2; half++) {
                const int cnt = half ? n - n1 : n1;
                if (cnt <= 0) continue;
                const uint32_t o = base + (half ? 0u : (uint32_t)vs * SB);
                if (kind == 0) masked_span_o<VEC>(acc0, acc1, o, cnt, h0, h1);
                else if (kind == 1) span_o<VEC, 1>(acc0, acc0, o, cnt);
                else if (kind == 2) span_o<VEC, 2>(acc0, acc1, o, cnt);
                else masked_span_o<VEC>(acc0, acc1, o, cnt, t0, t1);
            }
        };
        acc0 = zero;
        acc1 = zero;
        int v = u + hn;
        if (v >= R) v -= R;
        if (hn == 0) {
            part(v, mn, 1);
            acc1 = acc0;
        } else {
            part(u, hn, 0);
            part(v, mn, 2);
        }
        v += mn;
        if (v >= R) v -= R;
        part(v, tn, 3);
    }
}

#ifdef AGG_STAMP
// diagnostic build: cycles the waves spend in the parts of an iteration, summed over all waves of all launches
__device__ unsigned long long g_agg_stamp[8];
#define STAMP(i) { const unsigned long long t_ = __builtin_readcyclecounter(); st[i] += t_ - tprev; tprev = t_; }
#else
#define STAMP(i)
#endif
// MODE: how ring 1 is filled when a slot is a full 128 floats (FULLD, VEC = 2; other shapes always use mode 0).
//   0  every wave brings its own pair of positions by one 16-byte-per-lane LDS-DMA per iteration
//   1  a loader wave: wave 15 issues the fills of all pairs and the descriptor DMA, waves 0..14 only compute
//      (30 positions per iteration) and never touch the vector-memory wait counter
// (A third form -- every wave loads its pair into registers with global_load_dwordx4 and writes it to the ring behind
// stage 2 -- measured the same as mode 0 and was dropped.)
template <int VEC, bool FULLD, int MODE> struct FusedShape {
    static constexpr bool X4 = FULLD && VEC == 2;
    static constexpr bool LOADER = X4 && MODE == 1;
    static constexpr int CW = LOADER ? AGG_WAVES - 1 : AGG_WAVES;   // computing waves
    static constexpr int B = 2 * CW;                                 // positions per iteration
};
template <int VEC, bool FULLD, int MODE>
__global__ void __launch_bounds__(AGG_WAVES * 64)
k_agg_fused(const float *__restrict__ src, float *__restrict__ dst, const uint4 *__restrict__ desc,
            int W, int H, int Dp, int vertical, int cap, int R1, int R2, int dpad, int nwhole, int npieces, int piece_len) {
    // One workgroup = one line x one chunk of disparities. cap: the arm cap rounded up to an even number.
    // Blocks start at multiples of B, so every pair of positions starts at an even position of the line and at
    // an even ring slot (R1, R2 are even): a pair never straddles the end of a ring.
    // Work items (blockIdx.y): the first nwhole lines whole; every later line as npieces pieces of piece_len positions
    // (a multiple of B), so that the workgroups left over after the last full round over the CUs are short ones
    // (see launch_aggregate_fused_pair). A piece [org, end) runs the march from org - B*nlead: stage 1 recomputes
    // the cap positions in front of it, stage 2 starts at org.
    typedef typename VecT<VEC>::T vec_t;
    typedef volatile vec_t __attribute__((address_space(3))) *lds_vec_ptr;
    constexpr int SLOT = 64 * VEC;
    typedef FusedShape<VEC, FULLD, MODE> Shape;
    constexpr int B = Shape::B, CW = Shape::CW;
    constexpr bool LOADER = Shape::LOADER;
    extern __shared__ __align__(16) float ring[];   // ring 1 (src): R1 x SLOT, then ring 2 (A): R2 x SLOT
    float *const ring2 = ring + (size_t)R1 * SLOT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kc = blockIdx.x * SLOT;
    const int LEN = vertical ? H : W;
    int line = blockIdx.y, org = 0, end = LEN;
    if (line >= nwhole) {
        const int j = line - nwhole, li = j / npieces;
        line = nwhole + li;
        org = (j - li * npieces) * piece_len;
        end = min(LEN, org + piece_len);
    }
    const int mstride = vertical ? W * Dp : Dp;
    const size_t lbase = (vertical ? (size_t)line * Dp : (size_t)line * W * Dp) + (size_t)kc;
    bool fok[VEC];
#pragma unroll
    for (int h = 0; h < VEC; h++) fok[h] = FULLD || kc + h * 64 + lane < Dp;
    const float *fsrc = src + lbase + lane;
    const bool kok = FULLD || kc + lane * VEC < Dp;
    const float *rl2 = ring2 + lane * VEC;
    // this lane's floats of slot 0 of either ring, as LDS byte offsets (a generic LDS address: its low 32 bits)
    const uint32_t ro1 = (uint32_t)(uintptr_t)(ring + lane * VEC), ro2 = (uint32_t)(uintptr_t)rl2;

    // Iteration j = -nlead, ..., 0, 1, ...: stage 1 on positions pa .. pa+B-1, pa = cap + B*(j+1); stage 2 on
    // o0 .. o0+B-1, o0 = B*j; fills of the inputs stage 1 of iteration j+2 adds to its windows (two blocks ahead:
    // one workgroup per CU has to keep that much of HBM's latency covered by itself). Positions before the line
    // (the lead-in: stage 1 has to run cap + B ahead) and behind it have zero descriptors and are skipped.
    const int nlead = (2 * cap + B - 1) / B + 1;
    int pa = org + cap + B * (1 - nlead);         // <= org - cap
    // ring 1 (R1 = 2*cap + 3*B slots): slot 0 <-> position pa - cap of the first iteration;
    // ring 2 (R2 = 2*cap + 2*B slots): slot 0 <-> the first pa
    {
        const int base1 = pa - cap;
        for (int i = wave; i < 2 * cap + 2 * B; i += AGG_WAVES) {
            const int x = base1 + i;
            if (x >= 0 && x < LEN) {
                const float *g = fsrc + (size_t)x * mstride;
#pragma unroll
                for (int h = 0; h < VEC; h++)
                    if (fok[h]) ring_fill(g + h * 64, ring + i * SLOT + h * 64);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int q1 = cap + 2 * wave;                       // ring-1 slot of this wave's first stage-1 position
    int qf = 2 * cap + 2 * B + 2 * wave;           // ring-1 slot of this wave's first fill of the iteration (< R1)
    int qw = 2 * wave;                             // ring-2 slot of this wave's first stage-1 position
    int q2 = R2 - cap - B + 2 * wave;              // ring-2 slot of this wave's first stage-2 position, -B*nlead + 2*wave
    if (q2 >= R2) q2 -= R2;
    const ptrdiff_t blk_m = (ptrdiff_t)B * mstride;
    // Fills: this wave's pair of positions f, f + 1 of every iteration. Two (VEC = 2: four) fill instructions are
    // issued in EVERY iteration, so that the count of operations in flight behind them is fixed: once the pair
    // lies behind the end of the line the pointer stops at the last pair it reached and the fills re-read that
    // into slots no window ever reads (fp = the position fill_p points to).
    const int f0 = pa + 2 * B + cap + 2 * wave;    // > 0
    int fp = min(f0, LEN - 1);
    const float *fill_p = fsrc + (ptrdiff_t)fp * mstride;
    // (full 128-float slots: ONE 16-byte-per-lane DMA fills both slots of the pair, lanes 0..31 the first position,
    //  lanes 32..63 the second -- issuing a fill instruction costs a wave hundreds of cycles, whatever its width)
    constexpr bool FILL_X4 = Shape::X4;
    constexpr int NFILL = FILL_X4 ? 1 : 2 * VEC;   // fill instructions per wave and iteration
    const float *fill_p4 = src + lbase + (lane & 31) * 4 + (ptrdiff_t)fp * mstride;
    const ptrdiff_t second = (lane >> 5) ? (ptrdiff_t)mstride : (ptrdiff_t)0;
    float *out_p = dst + lbase + lane * VEC + (ptrdiff_t)(org + 2 * wave - B * nlead) * mstride;   // (stored to inside the piece only)
    // The workgroup's pair descriptors travel like the fills: by LDS-DMA, two iterations ahead of their use, into a
    // small staging area behind the rings (three 512-byte buffers used in turn: one being read, one landed, one in
    // flight). ONE instruction of one wave (wave 0, or the loader wave) moves
    // them all: lanes 0..15 bring the stage-1 records of the 16 waves (consecutive pairs: 256 contiguous bytes),
    // lanes 16..31 the stage-2 records. It is issued in front of that wave's fill, so the wait for the fill
    // covers it, and the barrier that publishes the fills publishes the descriptors; behind that barrier every wave
    // reads its own records (every lane the same address) and makes the values uniform registers with v_readfirstlane. Scalar
    // loads would be the natural form, but they share the LDS's wait counter and return out of order: the first wait
    // for a ring read would sit out the whole scalar load. Ordinary vector loads would make the compiler wait for
    // every fill in flight in front of their first use. (An LDS-DMA instruction costs the CU some tens of cycles
    // whatever it moves, and all waves of the CU queue for that: one per wave and iteration doubled the queue.)
    constexpr int DW = LOADER ? CW : 0;            // the wave that brings the descriptors
    const size_t drow = (size_t)line * (((LEN + 1) >> 1) + 2 * dpad) + dpad;
    const uint4 *d1p = desc + drow + (pa >> 1);                    // stage-1 pairs of the first iteration
    const uint4 *d2p = desc + drow + (org - B * nlead) / 2;        // stage-2 pairs of the first iteration (org, B even)
    const uint4 *dsrc = (lane < 16 ? d1p : d2p) + (lane & 15);
    float *const stage = ring2 + (size_t)R2 * SLOT;                // three buffers of 128 dwords
    const uint32_t stage_off = (uint32_t)(uintptr_t)stage + (uint32_t)wave * 16u;
    auto desc_dma = [&](int it_ahead, int buf) {   // the descriptors `it_ahead` iterations ahead of dsrc into buffer `buf`
        if (wave == DW && lane < 32)
            __builtin_amdgcn_global_load_lds(dsrc + (ptrdiff_t)it_ahead * (B / 2),
                                             (__attribute__((address_space(3))) void *)(stage + buf * 128), 16, 0, 0);
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    // (the buffer is read through an address rebuilt from an integer, like the rings: ordering against the DMA is
    // what the counted waits and the barrier provide)
    auto read_slot = [&](int buf, u32x4 &a, uint32_t &b) {
        typedef const volatile u32x4 __attribute__((address_space(3))) *p4;
        typedef const volatile uint32_t __attribute__((address_space(3))) *p1;
        a = *(p4)(uintptr_t)(stage_off + buf * 512);
        b = *(p1)(uintptr_t)(stage_off + buf * 512 + 256);
    };
    desc_dma(0, 0);
    desc_dma(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if constexpr (LOADER) {
        if (wave == CW) {
            // The loader wave: per iteration the descriptor DMA and the fills of all CW pairs (positions fl + 2w,
            // fl + 2w + 1 into slots ql + 2w, ql + 2w + 1), then the wait for the PREVIOUS iteration's (everything but
            // the 1 + CW operations just issued), then the barrier that publishes them. Pairs behind the end of the
            // line are skipped; from then on the wait is for everything.
            int fl = pa + 2 * B + cap;             // > 0
            int ql = 2 * cap + 2 * B;              // < R1
            int sbl = 2;
            const float *lsrc = src + lbase + (lane & 31) * 4;
            const int hi = lane >> 5;
#ifdef AGG_STAMP
            unsigned long long lt[3] = {0, 0, 0};
#endif
            for (int o0 = org - B * nlead; o0 < end; o0 += B) {
#ifdef AGG_STAMP
                const unsigned long long t0_ = __builtin_readcyclecounter();
#endif
                desc_dma(2, sbl);
                for (int w = 0; w < CW; w++) {
                    const int p = fl + 2 * w;
                    if (p < LEN) {
                        int q = ql + 2 * w;
                        if (q >= R1) q -= R1;
                        const int pl = min(p + hi, LEN - 1);   // a lone last position: its second half re-reads it
                        ring_fill_x4(lsrc + (ptrdiff_t)pl * mstride, ring + q * SLOT);
                    }
                }
#ifdef AGG_STAMP
                const unsigned long long t1_ = __builtin_readcyclecounter();
#endif
                if (fl + B <= LEN) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + CW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef AGG_STAMP
                const unsigned long long t2_ = __builtin_readcyclecounter();
#endif
                __builtin_amdgcn_s_barrier();
#ifdef AGG_STAMP
                lt[0] += t1_ - t0_; lt[1] += t2_ - t1_; lt[2] += __builtin_readcyclecounter() - t2_;
                if (o0 + B >= end && lane == 0) {   // (diagnostic build: the loader's issue / fill wait / barrier wait, in 1024 cycles)
                    atomicAdd(&g_agg_stamp[6], (lt[0] >> 10) | ((lt[1] >> 10) << 20) | ((lt[2] >> 10) << 40));
                }
#endif
                fl += B;
                ql += B; if (ql >= R1) ql -= R1;
                sbl = sbl == 2 ? 0 : sbl + 1;      // (the order the computing waves read them in)
                dsrc += B / 2;
            }
            return;
        }
    }

    u32x4 d1v;
    uint32_t d2v;
    read_slot(0, d1v, d2v);
    uint4 d1c = make_uint4(uni(d1v.x), uni(d1v.y), uni(d1v.z), 0u);
    uint32_t d2c = uni(d2v);
    int nstore_prev = 0, sb = 0, sb2 = 2;         // staging buffers: sb holds this iteration's descriptors, sb2 takes the DMA

#ifdef AGG_STAMP
    unsigned long long st[8] = {}, tprev = __builtin_readcyclecounter();
#endif
    for (int o0 = org - B * nlead; o0 < end; o0 += B) {
        // (a) the descriptors of the iteration after the next into the staging buffer that is free, then the fill.
        //     (Issuing the fills of half the waves between the stages instead, so that their queueing for the CU's
        //     DMA path overlaps the other half's sums, moved time between the phases of an iteration and left its
        //     length as it was.)
        if constexpr (!LOADER) {
            desc_dma(2, sb2);
            const float *p1 = fill_p + (fp + 1 < LEN ? (ptrdiff_t)mstride : (ptrdiff_t)0);
            __builtin_amdgcn_s_setprio(3);
#ifdef AGG_DIAG_NO_MEM   // diagnostic: the fill reads one and the same line (cache hits), the stores are dropped
            if constexpr (FILL_X4) {
                ring_fill_x4(src + lbase + (lane & 31) * 4, ring + qf * SLOT);
            } else
#endif
            if constexpr (FILL_X4) {
                ring_fill_x4(fill_p4 + (fp + 1 < LEN ? second : (ptrdiff_t)0), ring + qf * SLOT);
            } else {
#pragma unroll
                for (int h = 0; h < VEC; h++) {
                    if (fok[h]) ring_fill(fill_p + h * 64, ring + qf * SLOT + h * 64);
                    if (fok[h]) ring_fill(p1 + h * 64, ring + (qf + 1) * SLOT + h * 64);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        STAMP(0)
        // (b) stage 1: A of positions p, p + 1 into ring 2
        if (d1c.x) {
            vec_t acc0, acc1;
            desc_windows<VEC>(acc0, acc1, ro1, R1, q1, d1c.x);
            acc0 = div_by_count<VEC>(acc0, __uint_as_float(d1c.y));
            *(lds_vec_ptr)(uintptr_t)(rl2 + qw * SLOT) = acc0;
            if (d1c.x >> 31) {
                acc1 = div_by_count<VEC>(acc1, __uint_as_float(d1c.z));
                *(lds_vec_ptr)(uintptr_t)(rl2 + (qw + 1) * SLOT) = acc1;
            }
        }
        STAMP(1)
        // (c) stage 2: outputs o, o + 1 from ring 2 (written before the previous barrier)
        int nstore = 0;
        vec_t out0 = {}, out1 = {};
        if (d2c && o0 >= org) {                   // (in front of a piece the positions are real, but they are another piece's)
            desc_windows<VEC>(out0, out1, ro2, R2, q2, d2c);
            nstore = (d2c >> 31) ? 2 : 1;
        }
#ifdef AGG_DIAG_NO_MEM
        if (nstore && reinterpret_cast<float &>(out0) == -1.f) *reinterpret_cast<vec_t *>(out_p) = out0 + out1;   // (never true)
        nstore = 0;
#else
        if (nstore && kok) {
            *reinterpret_cast<vec_t *>(out_p) = out0;
            if (nstore == 2) *reinterpret_cast<vec_t *>(out_p + mstride) = out1;
        }
#endif
        STAMP(2)
        // (d) the fills of the PREVIOUS iteration must have landed before the next iteration reads them (and with
        //     them the descriptors issued in front of them, which are the next iteration's). Issued since then, in
        //     order: that iteration's stores, this one's descriptor DMA, its fills and its stores -- 2 + 1 + NFILL + 2
        //     operations in the steady state, and never fewer than 1 + NFILL (wave 0; the others issue no descriptor DMA).
        if constexpr (LOADER) {
            // (nothing to wait for: the loader wave waits for its fills)
        } else if constexpr (FULLD) {
            if (wave == 0) {
                if (nstore == 2 && nstore_prev == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 + NFILL) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + NFILL) : "memory");
            } else {
                if (nstore == 2 && nstore_prev == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NFILL) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NFILL) : "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (lane-predicated fills: their number is not fixed)
        }
        nstore_prev = nstore;
        STAMP(3)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's ring-2 slots are written
        __builtin_amdgcn_s_barrier();
        STAMP(4)
        sb2 = sb;
        sb = sb == 2 ? 0 : sb + 1;
        read_slot(sb, d1v, d2v);                             // the next iteration's descriptors (wave DW brought them)
        q1 += B; if (q1 >= R1) q1 -= R1;
        qf += B; if (qf >= R1) qf -= R1;
        qw += B; if (qw >= R2) qw -= R2;
        q2 += B; if (q2 >= R2) q2 -= R2;
        if (fp + B < LEN) { fp += B; fill_p += blk_m; fill_p4 += blk_m; }
        out_p += blk_m;
        dsrc += B / 2;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d1v), "+v"(d2v) : : "memory");
        d1c = make_uint4(uni(d1v.x), uni(d1v.y), uni(d1v.z), 0u);
        d2c = uni(d2v);
        STAMP(5)
    }
#ifdef AGG_STAMP
    if (lane == 0) {
        for (int i = 0; i < 6; i++) atomicAdd(&g_agg_stamp[i], st[i]);
        atomicAdd(&g_agg_stamp[7], 1ull);
    }
#endif
}
#ifdef AGG_STAMP
extern "C" int adc_debug_agg_stamps(unsigned long long *out) {
    unsigned long long z[8] = {};
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_agg_stamp), sizeof(z));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_agg_stamp), z, sizeof(z));
    return (int)e;
}
#endif

// ---- the same-axis pair of passes on the matrix pipe ---------------------------------------------------------
// v_mfma_f32_4x4x1_16b_f32 computes, for sixteen independent 4x4 blocks, D[i][j] = C[i][j] + a[i] * b[j]: with K = 1
// every element of D receives exactly ONE product and ONE rounding per instruction. With b[j] in {0.0, 1.0} that is
// "add the term a[i] to the window sum j if the term belongs to window j" -- the masked addition of the vector kernels
// above (fma(v, 1, acc) rounds as v + acc does, fma(v, 0, acc) is acc), 256 of them per instruction:
//     block k, column j  <->  disparity 4k + j of a 64-disparity term       (B: lane 4k + j holds the term's value)
//     row i              <->  one of FOUR neighbouring positions of the line  (A: lane 4k + i holds window i's factor)
//     D                  <->  lane 4k + j, register i: the running sum of window i at disparity 4k + j
// so a lane keeps "its" disparity throughout (ring reads, results, stores are all lane <-> disparity) and the four
// windows are the four accumulator registers.
// Each sum still receives its own terms in ascending position order from 0.0f, one IEEE addition per term (checked
// against sequential fp32 sums on a million accumulators by tools/microbench/mfma_add_probe.hip): identical bits.
// What the matrix pipe offers: a term is read from the ring once for four windows, the window geometry is decoded once
// per four outputs and no per-window branching is left (a term outside a window costs a factor, not a branch); alone, a
// SIMD retires ~27 such additions per clock this way (tools/microbench/mfma_add_probe.hip) against the ~9 the vector
// kernel sustains. Inside this kernel it does not get there: a 4x4x1 instruction holds the SIMD's vector issue for its
// whole 8 cycles, the factors of head and tail terms are vector instructions of their own, and with one workgroup per CU
// the waves pass through their matrix phases together and leave the pipe idle through the rest of an iteration
// (divide, ring and memory writes, barrier): 1.72 ms per launch at 1920x1080x256 against 1.39 ms of k_agg_fused. It is
// kept as Tuning::agg_mode = 2, bit-identical and tested, not as the default.
//
// Quad descriptor (8 dwords) of the positions p .. p+3 of a line, p a multiple of 4. The union of the four windows is
// one run of ring slots; "head" = its terms in front of the last window's first term, "core" = the terms all four
// windows hold, "tail" = the rest (all of the run is head when there is no term common to all four):
//   w0  bits 0..7 distance of the run's first term below p | 8..15 head terms | 16..23 core terms | 24..31 tail terms
//   w1  bits 0..3: position p + j exists (a line's last quad may be short); 0 = no position here at all
//   w2  byte j: first term of window j, counted from the run's first term (255 for a missing position)
//   w3  byte j: last term of window j, likewise (255 for a missing position: a run has fewer than 255 terms)
//   w4..w7  support count of p + j as float (1.0f for a missing position)
// Rows are padded with quad_desc_pad() all-zero records in front and behind, like the pair descriptors.
#ifndef QUAD_NQ
#define QUAD_NQ 8                       // quads per iteration = computing waves (a wave owns a quad x all 128 disparities)
#endif
#ifndef QUAD_NL
#define QUAD_NL 2                       // loader waves
#endif
#define QUAD_B (4 * QUAD_NQ)            // positions per iteration
#define QUAD_WAVES (QUAD_NQ + QUAD_NL)
int quad_cap(int max_arm) { return (max_arm + 3) & ~3; }
int quad_desc_pad(int max_arm) { return (quad_cap(max_arm) + 3 * QUAD_B) / 4 + 8; }
size_t quad_desc_count(int lines, int len, int max_arm) { return (size_t)lines * ((len + 3) / 4 + 2 * quad_desc_pad(max_arm)) * 2; }   // in uint4

__global__ void __launch_bounds__(256)
k_quad_desc(const uint32_t *__restrict__ arms, const uint32_t *__restrict__ cnt_for_h, const uint32_t *__restrict__ cnt_for_v,
            uint4 *__restrict__ desc_h, uint4 *__restrict__ desc_v, int W, int H, int pad) {
    const int vertical = blockIdx.z;
    const int LEN = vertical ? H : W, lines = vertical ? W : H;
    const int nquads = (LEN + 3) >> 2;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)lines * nquads) return;
    int line, quad;
    if (vertical) { line = (int)(gid % W); quad = (int)(gid / W); }
    else { line = (int)(gid / nquads); quad = (int)(gid - (size_t)line * nquads); }
    const int p = 4 * quad;
    const uint32_t *cnt = vertical ? cnt_for_v : cnt_for_h;
    const int sh_neg = vertical ? 16 : 0, sh_pos = vertical ? 24 : 8;
    int b[4], e[4];
    float c[4];
    uint32_t valid = 0;
    int umin = 1 << 20, umax = -(1 << 20), bmax = -(1 << 20), emin = 1 << 20;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        c[j] = 1.0f;
        b[j] = 0; e[j] = -1;
        if (p + j < LEN) {
            const size_t i = vertical ? (size_t)(p + j) * W + line : (size_t)line * W + p + j;
            const uint32_t a = arms[i];
            b[j] = j - (int)((a >> sh_neg) & 0xff);
            e[j] = j + (int)((a >> sh_pos) & 0xff);
            c[j] = (float)cnt[i];
            valid |= 1u << j;
            umin = min(umin, b[j]); umax = max(umax, e[j]);
            bmax = max(bmax, b[j]); emin = min(emin, e[j]);
        }
    }
    int nh, nc, nt;
    if (bmax <= emin) { nh = bmax - umin; nc = emin - bmax + 1; nt = umax - emin; }
    else { nh = umax - umin + 1; nc = 0; nt = 0; }
    uint32_t rb = 0, re = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool ok = (valid >> j) & 1u;
        rb |= (uint32_t)(ok ? b[j] - umin : 255) << (8 * j);     // (a missing position: a window no term index reaches)
        re |= (uint32_t)(ok ? e[j] - umin : 255) << (8 * j);
    }
    const uint32_t w0 = (uint32_t)(-umin) | (uint32_t)nh << 8 | (uint32_t)nc << 16 | (uint32_t)nt << 24;
    uint4 *out = (vertical ? desc_v : desc_h) + ((size_t)line * (nquads + 2 * pad) + pad + quad) * 2;
    out[0] = make_uint4(w0, valid, rb, re);
    out[1] = make_uint4(__float_as_uint(c[0]), __float_as_uint(c[1]), __float_as_uint(c[2]), __float_as_uint(c[3]));
}
void launch_quad_descriptors(const uint32_t *arms, const uint32_t *cnt_for_h, const uint32_t *cnt_for_v, uint4 *desc_h,
                             uint4 *desc_v, const Geom &g, int max_arm, hipStream_t s) {
    const size_t n = std::max((size_t)g.H * ((g.W + 3) / 4), (size_t)g.W * ((g.H + 3) / 4));
    k_quad_desc<<<dim3((unsigned)((n + 255) / 256), 1, 2), dim3(256), 0, s>>>(arms, cnt_for_h, cnt_for_v, desc_h, desc_v, g.W, g.H,
                                                                              quad_desc_pad(max_arm));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// A lane holds the disparities 2*lane and 2*lane + 1 of the workgroup's 128 (one ds_read_b64 per term, as in the vector
// kernels): two independent accumulator sets, whose matrix instructions alternate -- a dependent chain of its own would
// leave the pipe idle between its instructions.
// G consecutive ring slots from LDS byte offset o (this lane's two floats of slot 0 included): the terms i0 .. i0+G-1 of
// a quad's run added to its four window sums. MASKED: a term goes to window j (the lane's lane & 3) only if
// rb <= i <= rb + span; otherwise (core terms) to all four.
template <int G, bool MASKED>
__device__ __forceinline__ void quad_group(f32x4 &a0, f32x4 &a1, uint32_t o, int i0, uint32_t rb, uint32_t span) {
    f32x2 v[G];
#pragma unroll
    for (int u = 0; u < G; u++) v[u] = ring_read_o<f32x2>(o + u * 512u);
#pragma unroll
    for (int u = 0; u < G; u++) {
        float f = 1.0f;
        if (MASKED) f = ((uint32_t)(i0 + u) - rb <= span) ? 1.0f : 0.0f;
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(f, v[u].x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(f, v[u].y, a1, 0, 0, 0);
    }
}
// n terms from byte offset o (no ring wrap inside), the first being term i0 of the run; [c0, c1) are the core terms
__device__ __forceinline__ void quad_span(f32x4 &a0, f32x4 &a1, uint32_t o, int i0, int n, int c0, int c1, uint32_t rb, uint32_t span) {
    for (; n >= 8; n -= 8, i0 += 8, o += 8 * 512u) {
        if (i0 >= c0 && i0 + 8 <= c1) quad_group<8, false>(a0, a1, o, i0, rb, span);
        else quad_group<8, true>(a0, a1, o, i0, rb, span);
    }
    if (n & 4) { quad_group<4, true>(a0, a1, o, i0, rb, span); i0 += 4; o += 4 * 512u; }
    if (n & 2) { quad_group<2, true>(a0, a1, o, i0, rb, span); i0 += 2; o += 2 * 512u; }
    if (n & 1) quad_group<1, true>(a0, a1, o, i0, rb, span);
}
// The four window sums of a quad from the ring at LDS byte offset `base` (this lane's floats of slot 0; R slots of
// 512 bytes); q = slot of the quad's first position; w0: the descriptor's first word; rb / re: this lane's window.
__device__ __forceinline__ void quad_windows(f32x4 &a0, f32x4 &a1, uint32_t base, int R, int q, uint32_t w0, uint32_t rb, uint32_t re) {
    const int nh = (w0 >> 8) & 255, nc = (w0 >> 16) & 255, nt = w0 >> 24;
    const int n = nh + nc + nt;
    int u = q - (int)(w0 & 255u);
    if (u < 0) u += R;
    const uint32_t span = re - rb;          // (a missing position: rb = re = 255, never reached)
    a0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    a1 = a0;
    const int n1 = min(n, R - u);
    quad_span(a0, a1, base + (uint32_t)u * 512u, 0, n1, nh, nh + nc, rb, span);
    if (n > n1) quad_span(a0, a1, base, n1, n - n1, nh, nh + nc, rb, span);
}

__global__ void __launch_bounds__(QUAD_WAVES * 64)
k_agg_fused_q(const float *__restrict__ src, float *__restrict__ dst, const uint4 *__restrict__ desc,
              int W, int H, int Dp, int vertical, int cap, int R1, int R2, int dpad, int nwhole, int npieces, int piece_len) {
    // One workgroup = one line (or piece of a line) x 128 disparities: NQ computing waves, one quad of positions each,
    // and NL loader waves. cap: the arm cap rounded up to a multiple of 4; blocks start at multiples of B = 4 NQ and
    // R1, R2 are multiples of 4: a quad's four ring slots never wrap.
    constexpr int SLOT = 128, B = QUAD_B, NQ = QUAD_NQ, NL = QUAD_NL;
    static_assert(NQ == 8, "the descriptor DMA below moves sixteen 16-byte chunks per stage");
    extern __shared__ __align__(16) float ring[];   // ring 1 (src): R1 x SLOT, ring 2 (A): R2 x SLOT, descriptor staging
    float *const ring2 = ring + (size_t)R1 * SLOT;
    float *const stage = ring2 + (size_t)R2 * SLOT; // three buffers of 128 dwords
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kc = blockIdx.x * SLOT;
    const int LEN = vertical ? H : W;
    int line = blockIdx.y, org = 0, end = LEN;
    if (line >= nwhole) {
        const int j = line - nwhole, li = j / npieces;
        line = nwhole + li;
        org = (j - li * npieces) * piece_len;
        end = min(LEN, org + piece_len);
    }
    const int mstride = vertical ? W * Dp : Dp;
    const size_t lbase = (vertical ? (size_t)line * Dp : (size_t)line * W * Dp) + (size_t)kc;
    const int nlead = (2 * cap + B - 1) / B + 1;
    const int pa = org + cap + B * (1 - nlead);   // stage-1 positions of the first iteration start here (<= org - cap)
    // prologue: ring-1 slot i <-> position pa - cap + i, i < 2*cap + 2*B
    {
        const float *fsrc = src + lbase + lane;
        for (int i = wave; i < 2 * cap + 2 * B; i += QUAD_WAVES) {
            const int x = pa - cap + i;
            if (x >= 0 && x < LEN) {
                ring_fill(fsrc + (size_t)x * mstride, ring + i * SLOT);
                ring_fill_hi(fsrc + (size_t)x * mstride, ring + i * SLOT);
            }
        }
    }
    // descriptors: 8 quads x 32 bytes per stage and iteration, brought by ONE 16-byte-per-lane LDS-DMA of the first
    // loader wave two iterations ahead (lanes 0..15: the stage-1 records, lanes 16..31: the stage-2 records)
    const size_t drow = ((size_t)line * (size_t)(((LEN + 3) >> 2) + 2 * dpad) + dpad) * 2;
    const uint4 *d1p = desc + drow + (ptrdiff_t)(pa / 4) * 2;                      // (pa is a multiple of 4, maybe negative)
    const uint4 *d2p = desc + drow + (ptrdiff_t)((org - B * nlead) / 4) * 2;
    const uint4 *dsrc = (lane < 16 ? d1p : d2p) + (lane & 15);
    auto desc_dma = [&](int it_ahead, int buf) {
        if (wave == NQ && lane < 32)
            __builtin_amdgcn_global_load_lds(dsrc + (ptrdiff_t)it_ahead * (2 * NQ),
                                             (__attribute__((address_space(3))) void *)(stage + buf * 128), 16, 0, 0);
    };
    desc_dma(0, 0);
    desc_dma(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if (wave >= NQ) {
        // A loader wave: per iteration (the first one: the descriptor DMA and) its share of the fills of the block two
        // iterations ahead, one 16-byte-per-lane DMA per pair of positions (pairs li, li + NL, ...); it waits for the
        // PREVIOUS iteration's, then joins the barrier that publishes them.
        const int li = wave - NQ;
        constexpr int MINE = (2 * NQ + NL - 1) / NL;     // pairs per loader wave and iteration (16 pairs in all)
        static_assert((2 * NQ) % NL == 0, "every loader wave issues the same number of fills");
        int fl = pa + 2 * B + cap;             // > 0
        int ql = 2 * cap + 2 * B;              // < R1
        int sbl = 2;
        const float *lsrc = src + lbase + (lane & 31) * 4;
        const int hi = lane >> 5;
        for (int o0 = org - B * nlead; o0 < end; o0 += B) {
            desc_dma(2, sbl);
#pragma unroll
            for (int k = 0; k < MINE; k++) {
                const int w = li + k * NL;
                const int p = fl + 2 * w;
                if (p < LEN) {
                    int q = ql + 2 * w;
                    if (q >= R1) q -= R1;
                    const int pl = min(p + hi, LEN - 1);
                    ring_fill_x4(lsrc + (ptrdiff_t)pl * mstride, ring + q * SLOT);
                }
            }
            if (fl + B <= LEN) {
                if (li == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + MINE) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MINE) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            fl += B;
            ql += B; if (ql >= R1) ql -= R1;
            sbl = sbl == 2 ? 0 : sbl + 1;
            dsrc += 2 * NQ;
        }
        return;
    }

    const int quad = wave;
    const int col = lane & 3;                              // the window this lane's factor belongs to
    int q1 = cap + 4 * quad;                              // ring-1 slot of the quad's first stage-1 position
    int qw = 4 * quad;                                    // ring-2 slot of it
    int q2 = R2 - ((cap + B) % R2) + 4 * quad;            // ring-2 slot of the quad's first stage-2 position
    if (q2 >= R2) q2 -= R2;
    const uint32_t ro1 = (uint32_t)(uintptr_t)ring + (uint32_t)lane * 8u;
    const uint32_t ro2 = (uint32_t)(uintptr_t)ring2 + (uint32_t)lane * 8u;
    // register i of an accumulator = window i (position 4*quad + i of the block); the two accumulators = this lane's two disparities
    float *out_p = dst + lbase + 2 * lane + (ptrdiff_t)(org + 4 * quad - B * nlead) * mstride;
    const ptrdiff_t blk_m = (ptrdiff_t)B * mstride;
    const uint32_t stage_off = (uint32_t)(uintptr_t)stage + (uint32_t)quad * 32u;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4q __attribute__((ext_vector_type(4)));
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    int sb = 0;
#ifdef AGG_STAMP
    unsigned long long st[8] = {}, tprev = __builtin_readcyclecounter();
#endif
    for (int o0 = org - B * nlead; o0 < end; o0 += B) {
        // this iteration's records (published by the previous barrier): uniform words, then this lane's window bounds
        const uint32_t rec1 = stage_off + (uint32_t)sb * 512u, rec2 = rec1 + 256u;
        const u32x2 h1 = ring_read_o<u32x2>(rec1), h2 = ring_read_o<u32x2>(rec2);
        const uint32_t rb1 = ring_read_o<uint8_t>(rec1 + 8u + (uint32_t)col), re1 = ring_read_o<uint8_t>(rec1 + 12u + (uint32_t)col);
        const uint32_t rb2 = ring_read_o<uint8_t>(rec2 + 8u + (uint32_t)col), re2 = ring_read_o<uint8_t>(rec2 + 12u + (uint32_t)col);
        const u32x4q cv = ring_read_o<u32x4q>(rec1 + 16u);          // the four support counts of the stage-1 quad
        const uint32_t w0a = uni(h1.x), va = uni(h1.y), w0b = uni(h2.x), vb = uni(h2.y);
        STAMP(0)
        // stage 1: A of the quad's positions into ring 2
        if (va) {
            f32x4 a0, a1;
            quad_windows(a0, a1, ro1, R1, q1, w0a, rb1, re1);
            asm volatile("" : "+v"(a0), "+v"(a1));
            STAMP(1)
            const float cq[4] = {__uint_as_float(uni(cv.x)), __uint_as_float(uni(cv.y)), __uint_as_float(uni(cv.z)), __uint_as_float(uni(cv.w))};
            typedef volatile f32x2 __attribute__((address_space(3))) *wp;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if ((va >> i) & 1u) {
                    const f32x2 q = div_by_count<2>(f32x2{a0[i], a1[i]}, cq[i]);   // the window's count: uniform
                    *(wp)(uintptr_t)(ro2 + (uint32_t)(qw + i) * 512u) = q;
                }
            }
        }
        // stage 2: outputs of the quad B*nlead positions behind, from ring 2 (written before the previous barrier)
        if (vb && o0 >= org) {
            STAMP(2)
            f32x4 a0, a1;
            quad_windows(a0, a1, ro2, R2, q2, w0b, rb2, re2);
            asm volatile("" : "+v"(a0), "+v"(a1));
            STAMP(3)
#pragma unroll
            for (int i = 0; i < 4; i++)
                if ((vb >> i) & 1u) *reinterpret_cast<f32x2 *>(out_p + (ptrdiff_t)i * mstride) = f32x2{a0[i], a1[i]};
        }
        STAMP(4)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's ring-2 slots are written
        __builtin_amdgcn_s_barrier();
        STAMP(5)
        sb = sb == 2 ? 0 : sb + 1;
        q1 += B; if (q1 >= R1) q1 -= R1;
        qw += B; if (qw >= R2) qw -= R2;
        q2 += B; if (q2 >= R2) q2 -= R2;
        out_p += blk_m;
    }
#ifdef AGG_STAMP
    if (lane == 0) {
        for (int i = 0; i < 6; i++) atomicAdd(&g_agg_stamp[i], st[i]);
        atomicAdd(&g_agg_stamp[7], 1ull);
    }
#endif
}

// Fallback without LDS (any arm cap): one wave per (position, 64 disparities), spans re-read through
// the caches. Same addition order.
template <bool DIV>
__global__ void __launch_bounds__(256)
k_agg_direct(const float *__restrict__ src, float *__restrict__ dst, const uint32_t *__restrict__ arms,
             const uint32_t *__restrict__ cnt, int W, int H, int Dp, int vertical) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
        const uint32_t a = arms[ap];
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

template <int VEC, int DEPTH>
static void launch_ring(const float *src, float *dst, const uint32_t *arms, const uint32_t *cnt, const Geom &g,
                        int vertical, int max_arm, int R, hipStream_t s) {
    const int SLOT = 64 * VEC;
    const int chunks = (g.Dp + SLOT - 1) / SLOT;
    const int lines = vertical ? g.W : g.H;
    const int LEN = vertical ? g.H : g.W;
    const size_t lds = (size_t)R * SLOT * sizeof(float);
    // enough workgroups for several full rounds over 256 CUs; segments are multiples of the block size
    int nseg = 1;
    const long long wgs = (long long)chunks * lines;
    if (wgs < 4096) nseg = (int)((4096 + wgs - 1) / wgs);
    int seglen = (LEN + nseg - 1) / nseg;
    seglen = ((seglen + AGG_WAVES - 1) / AGG_WAVES) * AGG_WAVES;
    if (seglen < 128) seglen = 128;
    nseg = (LEN + seglen - 1) / seglen;
    dim3 grid(chunks, lines, nseg), block(AGG_WAVES * 64);
    if (cnt)
        k_agg_ring<VEC, true, DEPTH><<<grid, block, lds, s>>>(src, dst, arms, cnt, g.W, g.H, g.Dp, vertical, seglen, max_arm, R - 1);
    else
        k_agg_ring<VEC, false, DEPTH><<<grid, block, lds, s>>>(src, dst, arms, cnt, g.W, g.H, g.Dp, vertical, seglen, max_arm, R - 1);
}

template <int VEC>
static void launch_pair(const float *src, float *dst, const uint32_t *arms, const uint32_t *cnt, const Geom &g,
                        int vertical, int max_arm, const CostSrc *cost, const uint4 *desc, hipStream_t s) {
    const int SLOT = 64 * VEC;
    const int chunks = (g.Dp + SLOT - 1) / SLOT;
    const int lines = vertical ? g.W : g.H;
    const int LEN = vertical ? g.H : g.W;
    const int R = 2 * max_arm + 2 * PAIR_B;
    size_t lds = (size_t)R * SLOT * sizeof(float);
    int nseg = 1;
    const long long wgs = (long long)chunks * lines;
    if (wgs < 4096) nseg = (int)((4096 + wgs - 1) / wgs);
    int seglen = (LEN + nseg - 1) / nseg;
    seglen = ((seglen + PAIR_B - 1) / PAIR_B) * PAIR_B;
    if (seglen < 128) seglen = 128;
    if (cost) {
        // the staged right-row entries (seglen + 2*cap + SLOT - 1 of 8 B) and the exp tables share the 80 KB
        const long long room = (80 * 1024 - (long long)lds - 836 * 4) / 8 - 2 * max_arm - (SLOT - 1);
        const int max_seg = (int)(room / PAIR_B) * PAIR_B;
        if (seglen > max_seg) seglen = max_seg;
        lds += (size_t)(seglen + 2 * max_arm + SLOT - 1) * 8 + 836 * 4;
    }
    nseg = (LEN + seglen - 1) / seglen;
    dim3 grid(chunks, lines, nseg), block(AGG_WAVES * 64);
#ifdef AGG_NO_FULLD
    const bool fulld = false;
#else
    const bool fulld = g.Dp % SLOT == 0;
#endif
    const CostSrc none = {};
    const int dpad = pair_desc_pad(max_arm);
#define ADC_PAIR_ARGS(CS_) src, dst, arms, cnt, g.W, g.H, g.Dp, vertical, seglen, max_arm, R, CS_, desc, dpad
#define ADC_PAIR_GO(D_, C_, CS_)                                                                                  \
    do {                                                                                                          \
        if (desc) {                                                                                               \
            if (fulld) k_agg_pair<VEC, D_, C_, true, true><<<grid, block, lds, s>>>(ADC_PAIR_ARGS(CS_));          \
            else k_agg_pair<VEC, D_, C_, false, true><<<grid, block, lds, s>>>(ADC_PAIR_ARGS(CS_));               \
        } else {                                                                                                  \
            if (fulld) k_agg_pair<VEC, D_, C_, true, false><<<grid, block, lds, s>>>(ADC_PAIR_ARGS(CS_));         \
            else k_agg_pair<VEC, D_, C_, false, false><<<grid, block, lds, s>>>(ADC_PAIR_ARGS(CS_));              \
        }                                                                                                         \
    } while (0)
    if (cost) ADC_PAIR_GO(false, true, *cost);
    else if (cnt) ADC_PAIR_GO(true, false, none);
    else ADC_PAIR_GO(false, false, none);
#undef ADC_PAIR_GO
#undef ADC_PAIR_ARGS
}

// The dynamic-LDS opt-in of every instantiation a launcher above can pick (the ring kernels use up to 64 KB,
// the pair kernel up to 80 KB), on the current device. A kernel attribute belongs to the device it was set
// on, so every handle does this for its own device when it is initialised; nothing is cached per process.
template <int VEC>
static hipError_t setup_vec() {
    hipError_t e = hipSuccess;
    auto set = [&e](const void *f, int bytes) {
        if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    };
#define ADC_RING_ATTR(D_) \
    set(reinterpret_cast<const void *>(&k_agg_ring<VEC, true, D_>), 64 * 1024); \
    set(reinterpret_cast<const void *>(&k_agg_ring<VEC, false, D_>), 64 * 1024)
    ADC_RING_ATTR(2); ADC_RING_ATTR(4); ADC_RING_ATTR(6);
#undef ADC_RING_ATTR
#define ADC_PAIR_ATTR(D_, C_, F_)                                                               \
    set(reinterpret_cast<const void *>(&k_agg_pair<VEC, D_, C_, F_, false>), 80 * 1024);        \
    set(reinterpret_cast<const void *>(&k_agg_pair<VEC, D_, C_, F_, true>), 80 * 1024)
    ADC_PAIR_ATTR(true, false, true); ADC_PAIR_ATTR(false, false, true); ADC_PAIR_ATTR(false, true, true);
    ADC_PAIR_ATTR(true, false, false); ADC_PAIR_ATTR(false, false, false); ADC_PAIR_ATTR(false, true, false);
#undef ADC_PAIR_ATTR
    set(reinterpret_cast<const void *>(&k_agg_fused<VEC, true, 0>), 160 * 1024);
    set(reinterpret_cast<const void *>(&k_agg_fused<VEC, false, 0>), 160 * 1024);
    if (VEC == 2) {
        set(reinterpret_cast<const void *>(&k_agg_fused<2, true, 1>), 160 * 1024);
        set(reinterpret_cast<const void *>(&k_agg_fused_q), 160 * 1024);
    }
    return e;
}
hipError_t aggregate_setup_device() {
    const hipError_t e = setup_vec<1>();
    return e != hipSuccess ? e : setup_vec<2>();
}

// forward declaration (defined below)
static bool pair_kernel_ok(const Geom &g, int max_arm);

// Both rings of k_agg_fused must fit one CU's LDS, and the frame must be large enough for whole-line
// workgroups to fill the chip (a segment's lead-in recomputes 2*cap + B stage-1 positions).
bool aggregate_pair_fusion_supported(const Geom &g, int max_arm) {
    // Measured from 320 x 240 x 64 to 3840 x 2160 x 512: the fused launches win at every size (3 % of the
    // aggregation stage at the small end, 12-20 % from 640 x 480 up). Tuning::agg_fuse = 0 (test hook) turns them off.
    const int env = g.tune.agg_fuse;
    const int vec = (g.Dp >= 128 && g.Dp % 2 == 0) ? 2 : 1;
    const size_t lds = (size_t)(4 * ((max_arm + 1) & ~1) + 5 * PAIR_B) * 64 * vec * 4 + 3 * 512;
    return env && pair_kernel_ok(g, max_arm) && lds <= 160 * 1024;
}

// src -> (dividing pass along the axis) -> (plain pass along the same axis) -> dst; desc: that axis's pair descriptors
// The matrix-pipe form (k_agg_fused_q) runs full 128-float ring slots whose two rings fit the LDS with quads.
bool aggregate_quad_fusion_supported(const Geom &g, int max_arm) {
    const int cap = quad_cap(max_arm);
    const size_t lds = (size_t)(4 * cap + 5 * QUAD_B) * 512 + 3 * 512;
    return g.tune.agg_mode == 2 && aggregate_pair_fusion_supported(g, max_arm) && g.Dp % 128 == 0 && lds <= 160 * 1024;
}

void launch_aggregate_fused_pair(const float *src, float *dst, const uint4 *desc, const uint4 *qdesc, const Geom &g, int vertical,
                                 int max_arm, hipStream_t s) {
    const int vec = (g.Dp >= 128 && g.Dp % 2 == 0) ? 2 : 1;
    const int SLOT = 64 * vec;
    const int chunks = (g.Dp + SLOT - 1) / SLOT;
    const int lines = vertical ? g.W : g.H;
    const bool fulld = g.Dp % SLOT == 0;
    const bool quads = qdesc && aggregate_quad_fusion_supported(g, max_arm);
    const int cap = quads ? quad_cap(max_arm) : (max_arm + 1) & ~1;   // a multiple of 4 / even (see the kernels)
    const int mode = (vec == 2 && fulld) ? (g.tune.agg_mode == 2 ? 1 : g.tune.agg_mode) : 0;
    const int B = quads ? QUAD_B : (mode == 1 ? FusedShape<2, true, 1>::B : PAIR_B);
    const int R1 = 2 * cap + 3 * B, R2 = 2 * cap + 2 * B;
    const size_t lds = (size_t)(R1 + R2) * SLOT * sizeof(float) + 3 * 512;   // + the descriptor staging buffers
    // One workgroup per CU at a time: chunks * lines workgroups run as so many rounds over the CUs, and when the last
    // round is far from full (1080 rows x 2 chunks on 256 CUs: 8.44 rounds) most of the chip idles through it. The
    // lines of that last round are then cut into pieces, as many as still fit one round, so that it is a short one.
    const int LEN = vertical ? g.H : g.W;
    const int slots = g.simds / 4;
    int nwhole = lines, npieces = 1, piece_len = LEN;
    {
        const long long items = (long long)chunks * lines;
        const int rem = (int)(items % slots);
        const int k = rem ? slots / rem : 1;                     // pieces per line that keep the last round a single one
        const int rem_lines = (rem + chunks - 1) / chunks;
        const int plen = ((LEN + k - 1) / k + B - 1) / B * B;
        if (g.tune.agg_split >= 2) {                             // test hook: every line in that many pieces
            const int kk = g.tune.agg_split, pl = ((LEN + kk - 1) / kk + B - 1) / B * B;
            if ((long long)pl * (kk - 1) < LEN) { nwhole = 0; npieces = kk; piece_len = pl; }
        } else if (g.tune.agg_split && items > slots && k >= 2 && plen >= 4 * B && (long long)plen * (k - 1) < LEN) {
            nwhole = lines - rem_lines;
            npieces = k;
            piece_len = plen;
        }
    }
    dim3 grid(chunks, nwhole + (lines - nwhole) * npieces), block(AGG_WAVES * 64);
    if (quads) {
        k_agg_fused_q<<<grid, dim3(QUAD_WAVES * 64), lds, s>>>(src, dst, qdesc, g.W, g.H, g.Dp, vertical, cap, R1, R2,
                                                                quad_desc_pad(max_arm), nwhole, npieces, piece_len);
        return;
    }
#define ADC_FUSED_GO(V_, F_, M_) k_agg_fused<V_, F_, M_><<<grid, block, lds, s>>>(src, dst, desc, g.W, g.H, g.Dp, vertical, cap, R1, R2, pair_desc_pad(max_arm), nwhole, npieces, piece_len)
    if (vec == 2) {
        if (!fulld) ADC_FUSED_GO(2, false, 0);
        else if (mode == 1) ADC_FUSED_GO(2, true, 1);
        else ADC_FUSED_GO(2, true, 0);
    } else { if (fulld) ADC_FUSED_GO(1, true, 0); else ADC_FUSED_GO(1, false, 0); }
#undef ADC_FUSED_GO
}

static bool pair_kernel_ok(const Geom &g, int max_arm) {
    const int pair_env = g.tune.agg_pair;
    const int vec = (g.Dp >= 128 && g.Dp % 2 == 0) ? 2 : 1;
    // div_by_count needs a count in [1, 2^14]: the 16-bit support counts (wrap-around, as the reference's
    // vec_sup_count_) stay below 2^14 while (2*cap+1)^2 does, i.e. for arm caps up to 63; longer arms take the
    // ring / direct kernels, whose plain IEEE division also covers a count that wrapped to 0.
    return pair_env && (2 * max_arm + 1) * (2 * max_arm + 1) <= (1 << 14) &&
           (long long)g.W * g.Dp * (PAIR_B + 2) < (1ll << 31) &&
           (size_t)(2 * max_arm + 2 * PAIR_B) * 64 * vec * 4 <= 80 * 1024;
}

// The first aggregation pass (horizontal, no division) with the initial costs computed inside it.
bool aggregate_cost_fusion_supported(const Geom &g, int max_arm) {
    const int env = g.tune.cost_fuse;
    const int vec = (g.Dp >= 128 && g.Dp % 2 == 0) ? 2 : 1;
    // room for at least a 128-position segment of right-row entries next to the ring and the tables
    const long long room = (80 * 1024 - (long long)(2 * max_arm + 2 * PAIR_B) * 64 * vec * 4 - 836 * 4) / 8 - 2 * max_arm - (64 * vec - 1);
    return env && pair_kernel_ok(g, max_arm) && room >= 128;
}
void launch_aggregate_first_pass_with_cost(const Planes &p, const float *lut_ad1, const float *lut_cen, float *dst,
                                           const Geom &g, int max_arm, const uint4 *desc_h, hipStream_t s) {
    const CostSrc cs = {p.pix_l, p.pix_r, p.cen_l, p.cen_r, lut_ad1, lut_cen, g.dmin, g.D};
    const int vec = (g.Dp >= 128 && g.Dp % 2 == 0) ? 2 : 1;
    if (!g.tune.agg_desc) desc_h = nullptr;
    if (vec == 2) launch_pair<2>(nullptr, dst, p.arms, nullptr, g, 0, max_arm, &cs, desc_h, s);
    else launch_pair<1>(nullptr, dst, p.arms, nullptr, g, 0, max_arm, &cs, desc_h, s);
}

// desc: the pair descriptors of the pass's axis whose counts are the ones this pass divides by (or does not divide), or
// null: then the windows come from the arms (and cnt)
void launch_aggregate_pass(const float *src, float *dst, const uint32_t *arms, const uint32_t *cnt,
                           const Geom &g, int vertical, int max_arm, int variant, const uint4 *desc, hipStream_t s) {
    if (!g.tune.agg_desc) desc = nullptr;
    // deepest fill look-ahead (6, 4 or 2 iterations) whose ring still fits 64 KB of LDS
    const int depth_env = g.tune.agg_depth;
    const bool strides_ok = (long long)g.W * g.Dp * (AGG_WAVES + 1) < (1ll << 31);
    const int vec = (g.Dp >= 128 && g.Dp % 2 == 0) ? 2 : 1;
    int depth = 0, R = 0;
    for (int d : {6, 4, 2}) {
        if (d > depth_env) continue;
        int r = 64;
        while (r < 2 * max_arm + (d + 1) * AGG_WAVES) r <<= 1;
        if ((size_t)r * 64 * vec * 4 <= 64 * 1024) { depth = d; R = r; break; }
    }
    // two positions per wave when its ring (2*cap + 64 slots) leaves room for two workgroups per CU
    if (variant == 0 && pair_kernel_ok(g, max_arm)) {
        if (vec == 2) launch_pair<2>(src, dst, arms, cnt, g, vertical, max_arm, nullptr, desc, s);
        else launch_pair<1>(src, dst, arms, cnt, g, vertical, max_arm, nullptr, desc, s);
    } else if (variant == 0 && strides_ok && depth) {
#define ADC_RING(V, D) launch_ring<V, D>(src, dst, arms, cnt, g, vertical, max_arm, R, s)
        if (vec == 2) { if (depth == 6) ADC_RING(2, 6); else if (depth == 4) ADC_RING(2, 4); else ADC_RING(2, 2); }
        else { if (depth == 6) ADC_RING(1, 6); else if (depth == 4) ADC_RING(1, 4); else ADC_RING(1, 2); }
#undef ADC_RING
    } else {
        const int chunks = (g.Dp + 63) / 64;
        const int lines = vertical ? g.W : g.H;
        const int LEN = vertical ? g.H : g.W;
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
