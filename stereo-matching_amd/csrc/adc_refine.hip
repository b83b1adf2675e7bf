// Multi-step disparity refinement on gfx950 (MultiStepRefiner, ADC/multistep_refiner.cu;
// ADC/ = final_version_project/Stereo_Matching/ADCensus_CUDA/): left-right outlier detection,
// iterative region voting, 16-direction interpolation, sobel edges + discontinuity adjustment and
// the 3x3 median. Every step reads the disparity map as it was when the step started and writes its
// results afterwards (the reference updates the map in place while other threads read it).
//
// Outlier pixels are compacted into two lists (mismatch / occlusion) when they are detected; the
// voting and interpolation steps then run over the lists only instead of over the whole image.
#include "adc_common.h"

namespace adc {

#define ADC_INF __int_as_float(0x7f800000)

// lround() of the reference: round half away from zero.
__device__ __forceinline__ int lround_f(float v) { return (int)roundf(v); }

// Disparity histogram bin of a map entry as the voting step sees it: lround(d) - dmin when the value is
// valid and inside [0, D), else -1 (RegionVotingCuda's guards, ADC/multistep_refiner.cu:405-409).
__device__ __forceinline__ short vote_bin(float d, int dmin, int D) {
    if (d == ADC_INF) return -1;
    const int b = lround_f(d) - dmin;
    return (b >= 0 && b < D) ? (short)b : (short)-1;
}

// ---- outlier detection (OutlierDetectionCuda, ADC/multistep_refiner.cu:70-156) --------------------
// 64 x 16 pixel tile per 1024-thread workgroup. Flagged pixels are appended to the mismatch / occlusion
// lists with one atomic per workgroup and list (all appends hit the same two counters, and same-address
// atomics are served one at a time): wave ballots -> per-wave counts in LDS -> one reservation -> ranks.
#define OUT_WAVES 16
__global__ void __launch_bounds__(64 * OUT_WAVES)
k_outlier(const float *__restrict__ dl, const float *__restrict__ dr, float *__restrict__ out,
          uint8_t *__restrict__ occ, uint8_t *__restrict__ mis, short *__restrict__ binmap,
          uint32_t *__restrict__ list_mis, uint32_t *__restrict__ list_occ, uint32_t *__restrict__ counters,
          int W, int H, int dmin, int D, float thres) {
    __shared__ uint32_t s_off[2][OUT_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * OUT_WAVES + wave;
    const bool in = x < W && y < H;
    const size_t i = in ? (size_t)y * W + x : 0;
    const float d = dl[i];
    float o = d;
    bool is_occ = false, is_mis = false;
    if (!in) {
    } else if (d < 0) {               // MinDispMismatch = 0 (ADC/config.h:4)
        o = ADC_INF; is_mis = true;
    } else if (d == ADC_INF) {
        is_mis = true;
    } else {
        const int cr = lround_f((float)x - d);
        if (cr >= 0 && cr < W) {
            const float r = dr[(size_t)y * W + cr];
            if (fabsf(d - r) > thres) {
                const int crl = lround_f((float)cr + r);
                if (crl > 0 && crl < W) {
                    const float l2 = dl[(size_t)y * W + crl];
                    if (l2 > d) is_occ = true; else is_mis = true;
                } else {
                    is_mis = true;
                }
                o = ADC_INF;
            }
        } else {
            o = ADC_INF; is_mis = true;
        }
    }
    if (in) {
        out[i] = o;
        occ[i] = is_occ;
        mis[i] = is_mis;
        binmap[i] = vote_bin(o, dmin, D);
    }
    const unsigned long long m_mis = __ballot(is_mis), m_occ = __ballot(is_occ);
    if (lane == 0) {
        s_off[0][wave] = (uint32_t)__popcll(m_mis);
        s_off[1][wave] = (uint32_t)__popcll(m_occ);
    }
    __syncthreads();
    if (threadIdx.x < 2) {            // one thread per list: reserve, then turn the counts into offsets
        uint32_t tot = 0;
        for (int w = 0; w < OUT_WAVES; w++) tot += s_off[threadIdx.x][w];
        uint32_t base = tot ? atomicAdd(&counters[threadIdx.x], tot) : 0u;
        for (int w = 0; w < OUT_WAVES; w++) {
            const uint32_t c = s_off[threadIdx.x][w];
            s_off[threadIdx.x][w] = base;
            base += c;
        }
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    if (is_mis) list_mis[s_off[0][wave] + (uint32_t)__popcll(m_mis & below)] = (uint32_t)i;
    if (is_occ) list_occ[s_off[1][wave] + (uint32_t)__popcll(m_occ & below)] = (uint32_t)i;
}

hipError_t launch_outlier(const float *dl, const float *dr, float *out, uint8_t *occ, uint8_t *mis, short *binmap,
                          uint32_t *list_mis, uint32_t *list_occ, uint32_t *counters,
                          const Geom &g, float thres, hipStream_t s) {
    const hipError_t e = hipMemsetAsync(counters, 0, ADC_NUM_COUNTERS * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    dim3 grid((g.W + 63) / 64, (g.H + OUT_WAVES - 1) / OUT_WAVES);
    k_outlier<<<grid, dim3(64 * OUT_WAVES), 0, s>>>(dl, dr, out, occ, mis, binmap, list_mis, list_occ, counters,
                                                    g.W, g.H, g.dmin, g.D, thres);
    return hipSuccess;
}

hipError_t launch_clear_masks(uint8_t *occ, uint8_t *mis, uint32_t *counters, size_t n, hipStream_t s) {
    hipError_t e = hipMemsetAsync(occ, 0, n, s);
    if (e == hipSuccess) e = hipMemsetAsync(mis, 0, n, s);
    if (e == hipSuccess) e = hipMemsetAsync(counters, 0, ADC_NUM_COUNTERS * sizeof(uint32_t), s);
    return e;
}

// ---- iterative region voting (RegionVotingCuda + DeleteFilledPixel, ADC/multistep_refiner.cu:376-449)
// One wavefront (= one 64-thread workgroup) per listed pixel: the disparity histogram of the cross
// support region is built in LDS. The arms of up to 64 region rows are fetched with one load (lane <->
// row). A region row is at most 2*cap + 1 pixels long, far fewer than a wave has lanes, so one wave
// instruction serves SEVERAL rows: a lane reads VOTE_PX neighbouring pixels of the histogram-bin map with
// one (unaligned) 16-byte load, ceil((2*cap + 1) / VOTE_PX) lanes cover a row, and as many rows as fit
// into 64 lanes share every load and every histogram update (seven rows for the default cap of 34). The
// kernel is bound by the instructions it issues per region row, not by bytes. Counts are integers: the
// order in which a region's pixels reach the histogram does not matter.
//
// A pixel that failed the vote fails it again for as long as no disparity inside its support region has
// changed, so from a list's second pass on the histogram is rebuilt only for pixels that had a fill
// land near them since they were last looked at. Fills are tracked per 8x8 tile: k_apply stamps the
// tile with the pass number, and a pixel is re-voted when any tile touching the bounding box of its
// region (its own vertical arms, the arm cap horizontally) carries a stamp from the two passes since
// its previous turn. Stamps grow monotonically over frames (stamp0 = this frame's pass 0), so the tile
// array is never cleared.
#define VOTE_PX 8
#ifndef VOTE_GROUPS
#define VOTE_GROUPS 2   // loads in flight per lane
#endif
// Most pixels of a support region carry the same one or two disparities, so the lanes of a histogram
// update mostly hit one LDS word and are served one after the other. Every bin therefore has VOTE_COPIES
// counters, one per lane & (VOTE_COPIES - 1), next to each other (16 bytes per bin: four different banks,
// one ds_read_b128 when the histogram is read out); they are summed at the read-out. (Eight or sixteen
// counters per bin measured slower: fewer workgroups fit a CU.)
// LDS layout in bytes: [0, VOTE_BINB) the words "bin -1" (no valid disparity, or past the end of the row's
// span) maps to -- never updated, only addressed --, then VOTE_BINB bytes per bin, then nine 16-byte masks.
#ifndef VOTE_COPIES
#define VOTE_COPIES 4
#endif
#define VOTE_BINB (4 * VOTE_COPIES)            // bytes per bin
#define VOTE_LUT(D) (VOTE_BINB + VOTE_BINB * (D))
#ifndef VOTE_TILE_SHIFT
#define VOTE_TILE_SHIFT 3
#endif
#ifndef VOTE_SKIP_FROM
#define VOTE_SKIP_FROM 2
#endif
static size_t vote_lds_bytes(int D) { return (size_t)VOTE_LUT(D) + 9 * 16; }
// The loads of a list entry depend on each other: list -> the pixel's arms, mask and disparity -> the arms of its
// region rows and the stamps of its tiles -> the bin map. Taken one after the other they cost four or five trips
// to memory per pixel, and a wave does little else. The loop is therefore a software pipeline over the entries of
// a workgroup, each level issued one iteration ahead of the next:
//   A  entry i+3: its pixel index                     B  entry i+2: arms, mask and disparity of the pixel
//   C  entry i+1: arms of its (first 64) region rows and, from the third pass on, its tile stamps
//   P  entry i:   the vote
// The loads of A, B and C are issued unconditionally (addresses clamped to valid ones) between the first bin-map
// loads of P and the histogram updates that wait for them, so that a wait counts loads and never drains them.
// A workgroup is one wave: LDS operations of one wave execute in order, no barrier is needed.
__global__ void __launch_bounds__(64)
k_vote(const float *__restrict__ disp, const short *__restrict__ binmap, const uint8_t *__restrict__ mask,
       const uint32_t *__restrict__ list, const uint32_t *__restrict__ count, float *__restrict__ newval,
       const uint32_t *__restrict__ arms, const uint32_t *__restrict__ tile_stamp, int W, int dmin, int D,
       int tau_s, float tau_h, int cap, int tiles_w, int tiles_n, uint32_t stamp0, int pass) {
    extern __shared__ __align__(16) unsigned char vote_lds[];
    const int lane = threadIdx.x;
    int4 *const hist4 = reinterpret_cast<int4 *>(vote_lds + VOTE_BINB);    // VOTE_COPIES counters per bin
    // byte offset (an LDS address) of this lane's counter of bin 0
    const uint32_t hbase = (uint32_t)(uintptr_t)(vote_lds + VOTE_BINB) + 4u * (uint32_t)(lane & (VOTE_COPIES - 1));
    const uint32_t lut_off = (uint32_t)(uintptr_t)(vote_lds + VOTE_LUT(D));
    // mask c (c = 0 .. 8): all ones in the 16-bit fields of the pixels c .. 7 of a load (ORed in, they become bin -1)
    if (lane < 9) {
        uint32_t m[4];
#pragma unroll
        for (int j = 0; j < 4; j++) m[j] = (2 * j >= lane ? 0xffffu : 0u) | (2 * j + 1 >= lane ? 0xffff0000u : 0u);
        reinterpret_cast<uint4 *>(vote_lds + VOTE_LUT(D))[lane] = make_uint4(m[0], m[1], m[2], m[3]);
    }
    // lane <-> (row slot, pixel group) of a wave instruction
    const int lpr = (2 * cap + 1 + VOTE_PX - 1) / VOTE_PX;     // lanes per region row (<= 64: cap <= 255)
    const int rpi = 64 / lpr;                                  // region rows per wave instruction
    const int rslot = lane / lpr, grp = lane - rslot * lpr;
    const bool lane_on = rslot < rpi;
    const uint32_t n = *count, G = gridDim.x, base = blockIdx.x;
    if (base >= n) return;
    const uint32_t mine = (n - base + G - 1) / G;              // entries of this workgroup
    const bool skip = pass >= VOTE_SKIP_FROM;
    const uint32_t fresh_from = stamp0 + (uint32_t)(pass - 2); // (> 0: stamps start above 2)

    // pipeline registers (B: pB; C: pC, aC, mC, dC; P: pP, aP, mP, dP, arow, st0, st1, ntP)
    uint32_t pB = 0, pC = 0, aC = 0, pP = 0, aP = 0, arow = 0, st0 = 0, st1 = 0;
    uint32_t mC = 0, mP = 0;
    float dC = 0.0f, dP = 0.0f;
    int ntP = 0;
    for (uint32_t it = 0; it < mine + 3; it++) {
        // ---- P, first half: the decision, the cleared histogram and the first bin-map loads
        const bool vP = it >= 3;                               // (it - 3 < mine by the loop bound)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(arow), "+v"(st0), "+v"(st1));   // settled here, not in front of every cross-lane read
        bool todo = vP && mP != 0u && dP == ADC_INF;           // uniform across the wave
        if (todo && skip) {
            bool fresh = st0 >= fresh_from || st1 >= fresh_from;
            if (ntP > 128) {                                   // (arm caps beyond 34: more tiles than two loads per lane cover)
                const int y = (int)(pP / (uint32_t)W), x = (int)(pP - (uint32_t)y * (uint32_t)W);
                const int tx0 = max(x - cap, 0) >> VOTE_TILE_SHIFT, tx1 = min(x + cap, W - 1) >> VOTE_TILE_SHIFT;
                const int ty0 = (y - (int)((aP >> 16) & 0xff)) >> VOTE_TILE_SHIFT;
                const int ntx = tx1 - tx0 + 1;
                for (int t = lane + 128; t < ntP; t += 64) {
                    const int ty = ty0 + t / ntx, tx = tx0 + t % ntx;
                    fresh |= tile_stamp[ty * tiles_w + tx] >= fresh_from;
                }
            }
            todo = __ballot(fresh) != 0ull;
        }
        const int top = (aP >> 16) & 0xff, bot = aP >> 24;
        const int nrows = top + bot + 1;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        uint4 bw[VOTE_GROUPS];
        u32x4 bm[VOTE_GROUPS];
        int last[VOTE_GROUPS];              // this lane's pixels 0 .. last of the load are inside the row's span
        // the loads of region rows t0 .. t0 + rpi*VOTE_GROUPS - 1 of a chunk of nr rows starting at pixel qrow
        auto issue = [&](uint32_t ar, size_t qrow, int t0, int nr) {
            uint32_t a2[VOTE_GROUPS];
#pragma unroll
            for (int u = 0; u < VOTE_GROUPS; u++)              // (rows past the region read a zero or a stale arm: unused)
                a2[u] = (uint32_t)__shfl((int)ar, (t0 + u * rpi + rslot) & 63, 64);
#pragma unroll
            for (int u = 0; u < VOTE_GROUPS; u++) {
                last[u] = -1;
                bw[u] = make_uint4(0u, 0u, 0u, 0u);
                const int r = t0 + u * rpi + rslot;
                const int l2 = a2[u] & 0xff, r2 = (a2[u] >> 8) & 0xff;
                if (lane_on && r < nr) last[u] = l2 + r2 - VOTE_PX * grp;
                if (last[u] >= 0) {
                    const short *src = binmap + qrow + (size_t)r * W - l2 + VOTE_PX * grp;
                    __builtin_memcpy(&bw[u], src, 16);         // (2-byte aligned)
                }
                typedef const u32x4 __attribute__((address_space(3))) *lut_ptr;
                bm[u] = *(lut_ptr)(uintptr_t)(lut_off + 16u * (uint32_t)min(max(last[u] + 1, 0), VOTE_PX));
            }
        };
        // One histogram update per pixel and nothing else: the LDS address of a pixel's counter is one v_mad_i32_i16
        // (16 * bin + this lane's base, the bin taken from either half of the loaded word), and pixels that do not
        // count carry bin -1, whose counters are the four spare words in front of bin 0. Lanes with no pixel of the
        // row at all are switched off for the whole load.
        auto update = [&]() {
            typedef int __attribute__((address_space(3))) *cnt_ptr;
#pragma unroll
            for (int u = 0; u < VOTE_GROUPS; u++) {
                if (last[u] >= 0) {
                    const uint32_t w[4] = {bw[u].x | bm[u].x, bw[u].y | bm[u].y, bw[u].z | bm[u].z, bw[u].w | bm[u].w};
#pragma unroll
                    for (int k = 0; k < VOTE_PX; k++) {
                        uint32_t addr;
                        if (k & 1) asm("v_mad_i32_i16 %0, %1, %3, %2 op_sel:[1,0,0,0]" : "=v"(addr) : "v"(w[k >> 1]), "v"(hbase), "n"(VOTE_BINB));
                        else asm("v_mad_i32_i16 %0, %1, %3, %2" : "=v"(addr) : "v"(w[k >> 1]), "v"(hbase), "n"(VOTE_BINB));
                        // (pixels that do not count are switched off; counting them on the spare words instead is cheaper
                        //  to issue but slower: every such lane then queues at the same four words)
                        if ((int)addr >= (int)hbase)
                        __hip_atomic_fetch_add((cnt_ptr)(uintptr_t)addr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        };
        const size_t qrow0 = (size_t)pP - (size_t)top * W;     // first row of the region
        const int nr0 = min(64, nrows);
        if (todo) {
            for (int k = lane; k < D * (VOTE_COPIES / 4); k += 64) hist4[k] = make_int4(0, 0, 0, 0);
            issue(arow, qrow0, 0, nr0);
        }

        // ---- A, B, C of the entries behind P (addresses of entries past the end are clamped: loaded, never used)
        const uint32_t nA = list[min(base + it * G, n - 1)];
        const uint32_t naB = arms[pB];
        // (the mask byte through the aligned word around it: a uniform dword load goes to the scalar unit like its
        //  neighbours; every buffer has slack behind its last element)
        const uint32_t nmB = (reinterpret_cast<const uint32_t *>(mask)[pB >> 2] >> (8u * (pB & 3u))) & 0xffu;
        const float ndB = disp[pB];
        uint32_t narow, nst0 = 0, nst1 = 0;
        int nntC = 0;
        {
            const int topC = (aC >> 16) & 0xff, botC = aC >> 24;
            const int nrC = min(64, topC + botC + 1);
            narow = arms[(size_t)pC + (ptrdiff_t)(min(lane, nrC - 1) - topC) * W];
            if (skip) {
                const int y = (int)(pC / (uint32_t)W), x = (int)(pC - (uint32_t)y * (uint32_t)W);
                const int tx0 = max(x - cap, 0) >> VOTE_TILE_SHIFT, tx1 = min(x + cap, W - 1) >> VOTE_TILE_SHIFT;
                const int ty0 = (y - topC) >> VOTE_TILE_SHIFT, ty1 = (y + botC) >> VOTE_TILE_SHIFT;
                const int ntx = tx1 - tx0 + 1;
                nntC = ntx * (ty1 - ty0 + 1);
                // lanes past the last tile read the zero word behind the stamps (tiles_n): never fresh
                const int t0 = lane, t1 = lane + 64;
                nst0 = tile_stamp[t0 < nntC ? (ty0 + t0 / ntx) * tiles_w + tx0 + t0 % ntx : tiles_n];
                nst1 = tile_stamp[t1 < nntC ? (ty0 + t1 / ntx) * tiles_w + tx0 + t1 % ntx : tiles_n];
            }
        }

        // ---- P, second half
        float result = ADC_INF;
        if (todo) {
            update();
            for (int t0 = rpi * VOTE_GROUPS; t0 < nr0; t0 += rpi * VOTE_GROUPS) {
                issue(arow, qrow0, t0, nr0);
                update();
            }
            for (int r0 = 64; r0 < nrows; r0 += 64) {          // regions taller than 64 rows
                const int nr = min(64, nrows - r0);
                const size_t qrow = qrow0 + (size_t)r0 * W;
                uint32_t ar = 0;
                if (lane < nr) ar = arms[qrow + (size_t)lane * W];
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(ar));
                for (int t0 = 0; t0 < nr; t0 += rpi * VOTE_GROUPS) {
                    issue(ar, qrow, t0, nr);
                    update();
                }
            }
            // first bin holding the maximum count, and the total: the maximum of (count << 9 | 511 - bin) over the
            // bins with a count (counts stay below 2^18, bins below 512)
            uint32_t key = 0;
            int cnt = 0;
            for (int k = lane; k < D; k += 64) {
                int h = 0;
#pragma unroll
                for (int c = 0; c < VOTE_COPIES / 4; c++) {
                    const int4 h4 = hist4[k * (VOTE_COPIES / 4) + c];
                    h += (h4.x + h4.y) + (h4.z + h4.w);
                }
                if (h > 0) key = max(key, (uint32_t)h << 9 | (uint32_t)(511 - k));
                cnt += h;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                key = max(key, (uint32_t)__shfl_xor((int)key, off, 64));
                cnt += __shfl_xor(cnt, off, 64);
            }
            const int mh = (int)(key >> 9), mk = 511 - (int)(key & 511u);
            if (mh > 0) {
                if (cnt > tau_s && mh * 1.0f / cnt > tau_h) result = mk + (float)dmin;
            }
        }
        if (vP && lane == 0) newval[base + (it - 3) * G] = result;

        // ---- every entry moves up one level
        pP = pC; aP = aC; mP = mC; dP = dC; arow = narow; st0 = nst0; st1 = nst1; ntP = nntC;
        pC = pB; aC = naB; mC = nmB; dC = ndB;
        pB = nA;
    }
}

// Writes the pass's fills, and hands the entries that are still open to the list's next pass (out_list /
// out_count, skipped when null): a filled pixel never comes back, so later passes walk ever shorter lists
// instead of the whole outlier list. Appends are aggregated per workgroup (ballot -> per-wave counts ->
// one reservation) as in k_outlier. The order of a list does not matter: every entry is voted on its own.
__global__ void __launch_bounds__(256)
k_apply(float *__restrict__ disp, short *__restrict__ binmap, uint8_t *__restrict__ mask,
        const uint32_t *__restrict__ list, const uint32_t *__restrict__ count, const float *__restrict__ newval,
        uint32_t *__restrict__ out_list, uint32_t *__restrict__ out_count,
        uint32_t *__restrict__ tile_stamp, int W, int dmin, int D, int tiles_w, uint32_t stamp) {
    __shared__ uint32_t s_off[4];
    const uint32_t n = *count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t e0 = blockIdx.x * 256u; e0 < n; e0 += gridDim.x * 256u) {   // uniform per workgroup
        const uint32_t e = e0 + threadIdx.x;
        bool keep = false;
        uint32_t p = 0;
        if (e < n) {
            const float v = newval[e];
            p = list[e];
            if (v != ADC_INF) {
                disp[p] = v;
                binmap[p] = vote_bin(v, dmin, D);
                mask[p] = 0;
                const uint32_t y = p / (uint32_t)W, x = p - y * (uint32_t)W;
                tile_stamp[(y >> VOTE_TILE_SHIFT) * (uint32_t)tiles_w + (x >> VOTE_TILE_SHIFT)] = stamp;   // every writer stores the same value
            } else {
                keep = mask[p] && disp[p] == ADC_INF;   // only this entry's own fill could change either
            }
        }
        if (out_list) {
            const unsigned long long m = __ballot(keep);
            if (lane == 0) s_off[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t c0 = s_off[0], c1 = s_off[1], c2 = s_off[2], c3 = s_off[3];
                const uint32_t tot = c0 + c1 + c2 + c3;
                const uint32_t base = tot ? atomicAdd(out_count, tot) : 0u;
                s_off[0] = base; s_off[1] = base + c0; s_off[2] = base + c0 + c1; s_off[3] = base + c0 + c1 + c2;
            }
            __syncthreads();
            if (keep) out_list[s_off[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = p;
            __syncthreads();
        }
    }
}

int vote_tiles_w(int W) { return (W + (1 << VOTE_TILE_SHIFT) - 1) >> VOTE_TILE_SHIFT; }
int vote_tiles_h(int H) { return (H + (1 << VOTE_TILE_SHIFT) - 1) >> VOTE_TILE_SHIFT; }

void launch_region_voting_pass(float *disp, short *binmap, uint8_t *mask, const uint32_t *list, const uint32_t *count,
                               uint32_t *out_list, uint32_t *out_count,
                               float *newval, const uint32_t *arms, uint32_t *tile_stamp, uint32_t stamp0, int pass,
                               int cap, const Geom &g, int tau_s, float tau_h, hipStream_t s) {
    const int tw = vote_tiles_w(g.W);
    k_vote<<<dim3(8192), dim3(64), vote_lds_bytes(g.D), s>>>(disp, binmap, mask, list, count, newval, arms,
                                                                  tile_stamp, g.W, g.dmin, g.D, tau_s, tau_h, cap, tw,
                                                                  tw * vote_tiles_h(g.H), stamp0, pass);
    k_apply<<<dim3(512), dim3(256), 0, s>>>(disp, binmap, mask, list, count, newval, out_list, out_count, tile_stamp,
                                            g.W, g.dmin, g.D, tw, stamp0 + (uint32_t)pass);
}

// ---- list compaction ---------------------------------------------------------------------------------
// After the voting passes most listed pixels are filled. The interpolation walks long rays for the
// ones that are not, 4 pixels per wavefront, so it runs on lists holding only those.
// counters: [0] mismatch, [1] occlusion entries in; [2], [3] entries out.
__global__ void __launch_bounds__(1024)
k_compact(const uint32_t *__restrict__ list_mis, const uint32_t *__restrict__ list_occ,
          const uint8_t *__restrict__ mis, const uint8_t *__restrict__ occ, uint32_t *__restrict__ out_mis,
          uint32_t *__restrict__ out_occ, uint32_t *__restrict__ counters) {
    __shared__ uint32_t s_off[16];
    const uint32_t n_mis = counters[0], n_occ = counters[1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t e0 = blockIdx.x * 1024u; e0 < ((n_mis + 1023u) & ~1023u) + n_occ; e0 += gridDim.x * 1024u) {
        // a block handles entries of one list only: the occlusion entries start at the next multiple of 1024
        const uint32_t e = e0 + threadIdx.x;
        const uint32_t mis_end = (n_mis + 1023u) & ~1023u;
        bool keep = false, from_mis = e0 < mis_end;
        uint32_t p = 0;
        if (from_mis) {
            if (e < n_mis) { p = list_mis[e]; keep = mis[p] != 0; }
        } else if (e - mis_end < n_occ) {
            p = list_occ[e - mis_end];
            keep = occ[p] != 0;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_off[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int w = 0; w < 16; w++) tot += s_off[w];
            uint32_t base = tot ? atomicAdd(&counters[from_mis ? 2 : 3], tot) : 0u;
            for (int w = 0; w < 16; w++) { const uint32_t c = s_off[w]; s_off[w] = base; base += c; }
        }
        __syncthreads();
        if (keep) (from_mis ? out_mis : out_occ)[s_off[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = p;
        __syncthreads();
    }
}

// ---- proper interpolation (FillDisp, ADC/multistep_refiner.cu:543-633) ------------------------------
// ray_dy / ray_dx hold the host-computed products m*sin(ang_s), m*cos(ang_s) (double) for the 16
// directions, so the device only adds and rounds. 16 lanes share one listed pixel, one ray each, and
// walk their rays INTERP_STEPS samples at a time (independent loads); the reference's sequential
// scan over the directions (first smallest colour difference wins, ADC/multistep_refiner.cu:596-622)
// becomes a lexicographic (difference, direction) minimum over the 16 lanes.
// When every table entry is far from a rounding tie, round(y + m*sin) == y + round(m*sin) for every
// integer y, and the host hands over the rounded offsets instead (ray_off: int16 dy << 16 | int16 dx,
// staged in LDS when LDS_TAB); otherwise the double tables are used as they are.
#define INTERP_STEPS 4
template <bool INT_TAB, bool LDS_TAB>
__global__ void __launch_bounds__(256)
k_interp(const float *__restrict__ disp, const uint8_t *__restrict__ occ, const uint8_t *__restrict__ mis,
         const uint32_t *__restrict__ list_mis, const uint32_t *__restrict__ list_occ,
         const uint32_t *__restrict__ counters, float *__restrict__ newval, const uint32_t *__restrict__ pix,
         const double *__restrict__ ray_dy, const double *__restrict__ ray_dx, const uint32_t *__restrict__ ray_off,
         int W, int H, int maxlen, size_t npx) {
    extern __shared__ uint32_t s_off[];              // LDS_TAB: 16 x maxlen offsets
    const uint32_t n_mis = counters[0], n_occ = counters[1];
    const uint32_t n = n_mis + n_occ;
    if (blockIdx.x * (256u / 16u) >= n) return;      // nothing listed for this workgroup: skip the table set-up too
    const int s = threadIdx.x & 15;                  // this lane's direction
    const double *dy = ray_dy + (size_t)s * maxlen;
    const double *dx = ray_dx + (size_t)s * maxlen;
    const uint32_t *off = ray_off + (size_t)s * maxlen;
    if (LDS_TAB) {
        // row stride maxlen | 1: the 16 lanes of a pixel read the same column of 16 different rows, and with
        // a stride that is a multiple of 64 dwords (maxlen = 256) all of those sit in one LDS bank
        const int lstride = maxlen | 1;
        for (int i = threadIdx.x; i < 16 * maxlen; i += 256) s_off[(i / maxlen) * lstride + (i % maxlen)] = ray_off[i];
        __syncthreads();
        off = s_off + s * lstride;
    }
    const uint32_t per_block = 256 / 16;
    for (uint32_t e0 = blockIdx.x * per_block; e0 < n; e0 += gridDim.x * per_block) {
        const uint32_t e = e0 + (threadIdx.x >> 4);
        const bool live = e < n;
        const bool from_mis = e < n_mis;
        uint32_t p = 0;
        bool m = false, o = false;
        if (live) {
            p = from_mis ? list_mis[e] : list_occ[e - n_mis];
            m = mis[p] != 0;
            o = occ[p] != 0;
        }
        const int y = (int)(p / (uint32_t)W), x = (int)(p - (uint32_t)y * W);
        bool found = false;
        float fd = 0.0f;
        uint32_t fq = 0;
        bool walking = live && (m || o);
        // Branch-free body: every lane loads INTERP_STEPS samples (from its own pixel when the sample is
        // outside the image or the lane has finished), then the flags advance by selects.
        // the table entries of a batch are read one batch ahead: table read -> sample load is a dependent chain
        uint32_t onx[INTERP_STEPS];
        if (INT_TAB) {
#pragma unroll
            for (int u = 0; u < INTERP_STEPS; u++) onx[u] = off[min(1 + u, maxlen - 1)];
        }
        for (int mm = 1; mm < maxlen && __any(walking); mm += INTERP_STEPS) {
            float d[INTERP_STEPS];
            uint32_t q[INTERP_STEPS];
            bool in[INTERP_STEPS];
            uint32_t ocur[INTERP_STEPS];
            if (INT_TAB) {
#pragma unroll
                for (int u = 0; u < INTERP_STEPS; u++) {
                    ocur[u] = onx[u];
                    onx[u] = off[min(mm + INTERP_STEPS + u, maxlen - 1)];
                }
            }
#pragma unroll
            for (int u = 0; u < INTERP_STEPS; u++) {
                const int mi = min(mm + u, maxlen - 1);
                int yy, xx;
                if (INT_TAB) {
                    const uint32_t o2 = ocur[u];
                    yy = y + ((int)o2 >> 16);
                    xx = x + (int)(short)(o2 & 0xffffu);
                } else {
                    yy = (int)round((double)y + dy[mi]);
                    xx = (int)round((double)x + dx[mi]);
                }
                in[u] = (mm + u < maxlen) & ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
                q[u] = in[u] ? (uint32_t)yy * (uint32_t)W + (uint32_t)xx : p;
            }
            // lanes whose ray has ended issue no loads: the samples are scattered 4-byte reads, one cache line
            // per lane, and the address path of the vector cache is what this kernel saturates
            if (walking) {
#pragma unroll
                for (int u = 0; u < INTERP_STEPS; u++) d[u] = disp[q[u]];
            } else {
#pragma unroll
                for (int u = 0; u < INTERP_STEPS; u++) d[u] = ADC_INF;
            }
#pragma unroll
            for (int u = 0; u < INTERP_STEPS; u++) {
                const bool hit = walking & in[u] & (d[u] != ADC_INF);   // first valid sample of the ray
                found |= hit;
                fd = hit ? d[u] : fd;
                fq = hit ? q[u] : fq;
                walking = walking & in[u] & !hit;                       // the ray ends at the image border too
            }
        }
        // combine the 16 directions of a pixel
        int key = 0x7fffffff;                        // (colour difference << 4) | direction; lower wins
        float mind = ADC_LARGE_F;
        if (found) {
            const int diff = (int)__builtin_amdgcn_sad_u8(pix[p], pix[fq], 0u);
            key = (diff << 4) | s;
            mind = fd;
        }
        int bkey = key;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            bkey = min(bkey, __shfl_xor(bkey, off, 16));
            mind = fminf(mind, __shfl_xor(mind, off, 16));
        }
        // disparity collected by the winning direction
        const float best_d = __shfl(fd, (int)((threadIdx.x & 48) | (bkey & 15)), 64);
        if (live && s == 0) {
            float result = ADC_INF;
            if ((m || o) && bkey != 0x7fffffff) result = m ? best_d : mind;
            newval[from_mis ? (size_t)e : npx + (e - n_mis)] = result;
        }
    }
}

__global__ void __launch_bounds__(256)
k_apply_interp(float *__restrict__ disp, const uint32_t *__restrict__ list_mis, const uint32_t *__restrict__ list_occ,
               const uint32_t *__restrict__ counters, const float *__restrict__ newval, size_t npx) {
    const uint32_t n_mis = counters[0], n_occ = counters[1];
    const uint32_t n = n_mis + n_occ;
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const bool from_mis = e < n_mis;
        const uint32_t p = from_mis ? list_mis[e] : list_occ[e - n_mis];
        const float v = newval[from_mis ? (size_t)e : npx + (e - n_mis)];
        if (v != ADC_INF) disp[p] = v;
    }
}

hipError_t launch_interpolate(float *disp, const uint8_t *occ, const uint8_t *mis,
                        const uint32_t *list_mis, const uint32_t *list_occ, uint32_t *list2_mis, uint32_t *list2_occ,
                        uint32_t *counters, float *newval, const uint32_t *pix_l, const double *ray_dy,
                        const double *ray_dx, const uint32_t *ray_off, const Geom &g, int maxlen, hipStream_t s) {
    const size_t npx = (size_t)g.W * g.H;
    const hipError_t e = hipMemsetAsync(counters + 2, 0, 2 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    k_compact<<<dim3(1024), dim3(1024), 0, s>>>(list_mis, list_occ, mis, occ, list2_mis, list2_occ, counters);
    const size_t tab = (size_t)16 * (maxlen | 1) * sizeof(uint32_t);
#define ADC_INTERP(I, L, LDS) k_interp<I, L><<<dim3(8192), dim3(256), LDS, s>>>( \
        disp, occ, mis, list2_mis, list2_occ, counters + 2, newval, pix_l, ray_dy, ray_dx, ray_off, g.W, g.H, maxlen, npx)
    if (!ray_off) ADC_INTERP(false, false, 0);
    else if (tab <= 48 * 1024) ADC_INTERP(true, true, tab);
    else ADC_INTERP(true, false, 0);
#undef ADC_INTERP
    k_apply_interp<<<dim3(512), dim3(256), 0, s>>>(disp, list2_mis, list2_occ, counters + 2, newval, npx);
    return hipSuccess;
}

// ---- sobel edges (EdgeDetectCuda, ADC/multistep_refiner.cu:161-180) ---------------------------------
__global__ void __launch_bounds__(256)
k_edge_detect(const float *__restrict__ disp, uint8_t *__restrict__ edge, int W, int H, float thres) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    uint8_t e = 0;
    if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) {
        const float *r0 = disp + (size_t)(y - 1) * W + x;
        const float *r1 = disp + (size_t)y * W + x;
        const float *r2 = disp + (size_t)(y + 1) * W + x;
        const float gx = (-r0[-1] + r0[1]) + (-2 * r1[-1] + 2 * r1[1]) + (-r2[-1] + r2[1]);
        const float gy = (-r0[-1] - 2 * r0[0] - r0[1]) + (r2[-1] + 2 * r2[0] + r2[1]);
        const float gsum = fabsf(gx) + fabsf(gy);
        if (gsum > thres) e = 1;   // NaN (inf - inf) compares false
    }
    edge[(size_t)y * W + x] = e;
}
void launch_edge_detect(const float *disp, uint8_t *edge, int W, int H, float thres, hipStream_t s) {
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    k_edge_detect<<<grid, dim3(256), 0, s>>>(disp, edge, W, H, thres);
}

// ---- discontinuity adjustment (EdgeDispOptmize, ADC/multistep_refiner.cu:685-751) -------------------
__global__ void __launch_bounds__(256)
k_edge_adjust(const float *__restrict__ din, float *__restrict__ dout, const float *__restrict__ cost,
              const uint8_t *__restrict__ edge, int W, int H, int dmin, int D, int Dp) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t i = (size_t)y * W + x;
    float d = din[i];
    if (x >= 1 && x < W - 1 && edge[i] == 1 && d != ADC_INF) {
        const float *cp = cost + i * (size_t)Dp;
        const int di = lround_f(d) - dmin;
        if (di >= 0 && di < D) {
            float c0 = cp[di];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const float d2 = din[k == 0 ? i - 1 : i + 1];
                if (d2 != ADC_INF) {
                    const int d2i = lround_f(d2) - dmin;
                    if (d2i >= 0 && d2i < D) {
                        const float c = (k == 0) ? cp[-(ptrdiff_t)Dp + d2i] : cp[Dp + d2i];
                        if (c < c0) { d = d2; c0 = c; }
                    }
                }
            }
        }
    }
    dout[i] = d;
}
void launch_edge_adjust(const float *disp_in, float *disp_out, const float *cost, const uint8_t *edge,
                        const Geom &g, hipStream_t s) {
    dim3 grid((g.W + 63) / 64, (g.H + 3) / 4);
    k_edge_adjust<<<grid, dim3(256), 0, s>>>(disp_in, disp_out, cost, edge, g.W, g.H, g.dmin, g.D, g.Dp);
}

// ---- 3x3 median (MedianFilter<3>, ADC/multistep_refiner.cu:186-224) ----------------------------------
__global__ void __launch_bounds__(256)
k_median3(const float *__restrict__ in, float *__restrict__ out, int W, int H) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t i = (size_t)y * W + x;
    if (y >= 1 && x >= 1 && y < H - 1 && x < W - 1) {
        float w[9];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) w[a * 3 + b] = in[(size_t)(y - 1 + a) * W + x - 1 + b];
        // the reference's partial selection sort, as a fixed compare-exchange network over (a, b > a)
#pragma unroll
        for (int a = 0; a < 5; a++) {
#pragma unroll
            for (int b = a + 1; b < 9; b++) {
                const float lo = w[b] < w[a] ? w[b] : w[a];
                const float hi = w[b] < w[a] ? w[a] : w[b];
                w[a] = lo;
                w[b] = hi;
            }
        }
        out[i] = w[4];
    } else {
        out[i] = in[i];
    }
}
void launch_median3(const float *in, float *out, int W, int H, hipStream_t s) {
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    k_median3<<<grid, dim3(256), 0, s>>>(in, out, W, H);
}

}  // namespace adc
