// 4-direction scanline optimisation on gfx950 (ScanlineOptimizeLeftRight / ScanlineOptimizeUpDown,
// ADC/scanline_optimizer.cu:4-290; ADC/ = final_version_project/Stereo_Matching/ADCensus_CUDA/).
//
// One wavefront owns one scanline (a row for the horizontal passes, a column for the vertical ones)
// and walks it sequentially. The D path costs of the current pixel live in registers, NV per lane
// with lane l holding disparities l*NV .. l*NV+NV-1, so
//   * a pixel's disparity vector moves as one contiguous NV*256-byte access,
//   * Lr(p-r, d-1) / Lr(p-r, d+1) are register neighbours except at lane boundaries, where one DPP
//     wave shift per direction supplies them (lanes past D hold Large_Float = the reference's
//     sentinels Lp[-1], Lp[D]),
//   * min_d Lr(p-r, d) is a DPP row_shr / row_bcast reduction, read back with v_readlane.
// The recurrence, evaluated in the reference's operation order (so results are bit-identical to a
// sequential CPU evaluation):
//   dst[p][d] = (src[p][d] + min(Lp[d], Lp[d-1]+P1, Lp[d+1]+P1, min(Lp)+P2)) / 2
// (P1,P2) chosen by how many of the two colour steps reach so_tso (:97-109); first path pixel
// copies src (:51-52).
//
// A scanline is a serial chain with only ~1-2 waves per SIMD to run it, so HBM latency has to be
// hidden inside the wave: every path pixel's cost vector and its colour-step record are brought in
// by LDS-DMA into a per-wave ring RS pixels ahead of their use, with counted vmcnt waits.
#include "adc_common.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace adc {

// ---- colour-step flags of a path pixel ---------------------------------------------------------------
// For every pixel p and every disparity k the recurrence needs one bit: does d2 reach so_tso
// (ADC/scanline_optimizer.cu:75-109)? d2 is the colour step of the RIGHT image at xr = x - k - dmin
// along the path when 0 < xr < W-1, keeps the value of the previous disparity otherwise, and starts
// as d1 (the LEFT image step). As k grows xr only decreases, so "previous value" is: d1 while
// xr >= W-1 (nothing usable yet), and the step at xr = 1 once xr <= 0 -- or d1 for every k when not
// even k = 0 reaches a usable column.
//
// Per image row and path family (horizontal forward, horizontal backward, vertical) the right-image flags
// are packed into a bit string E indexed by j = (W-2) - xr, clamped to the usable columns
// (E[j] = flag at column clamp(W-2-j, 1, W-2)): along a pixel's disparities j = (W-2-x+dmin) + k ascends
// with k, so the D flags of pixel x are D consecutive bits of its row's string, starting at j0 = W-2-x+dmin.
// A second string F per row and family holds the LEFT image's d1 flag of pixel x at the same index j0.
// The kernels do not read the strings themselves but what is cut from them once per frame, in one of two forms
// chosen by the disparity count (a window = 32 consecutive bits of a string, i.e. the string funnel-shifted):
//   records per pixel (NV <= 2)  for every pixel the 2*NV windows of E that hold its D flags, with the pixel's own d1
//                      in place of the first disparities where the string has nothing to offer (j < 0, or every
//                      disparity when no column is usable), and a dword whose bit 0 is d1. One contiguous fetch per
//                      path pixel whose address steps by a constant along any path: the least work per pixel, which
//                      is what the short serial chains of small frames are made of. (3 families x W x H x (2*NV+1)
//                      dwords: 6 MB at 450 x 375.)
//   windows by shift (NV >= 4)   per row and family 32 copies of the two strings, copy c shifted by c bits: window j
//                      is dword j >> 5 of copy j & 31, so the windows of a pixel are consecutive dwords of one copy,
//                      fetched by LDS-DMA straight from the copies -- no per-pixel table (it would take 224 MB at
//                      1920 x 1080 x 256; the copies take 57 MB). The address cycles through the copies along a row,
//                      and the few disparities at the ends of a row whose flag is the pixel's own d1 are patched in
//                      the kernels with a mask (rec_nfill / rec_apply_fill). Large frames are bound by memory
//                      traffic, not by the chain, and do not feel either.
//
// Layout: [family][row][x][2*NV + 1] (records) or [family][row][shift 0..31][E | F][ne] (windows by shift).
// String index 0 <-> j = -joff (joff a multiple of 32 that makes every pixel's start index non-negative).
struct RecSrc {
    const uint32_t *fam;   // this pass's family
    int ne;                // dwords per string (32 * ne windows)
    int jbase;             // W - 2 + dmin + joff: pixel x starts at window jbase - x
    int dmin, D;
    int row;               // dwords per image row of the family's table
};
// the form of the table follows from the floats per lane of a pixel's disparities (Geom::NV)
__host__ __device__ constexpr bool rec_per_pixel(int nv) { return nv <= 2; }
__host__ __device__ constexpr int rec_row_dwords(int nv, int ne, int W) { return rec_per_pixel(nv) ? W * (2 * nv + 1) : 64 * ne; }
template <int NVR>   // NVR: Geom::NV
__device__ __forceinline__ uint32_t rec_index(const RecSrc &r, int x, int y) {   // index of pixel (x, y)'s first E window
    if (rec_per_pixel(NVR)) return (uint32_t)y * (uint32_t)r.row + (uint32_t)(x * (2 * NVR + 1));
    const int j = r.jbase - x;
    return (uint32_t)(y * 32 + (j & 31)) * (uint32_t)(2 * r.ne) + (uint32_t)(j >> 5);
}
// what DMA lane `lane` of a pixel's 2*NVR + 1 adds to that index: E window number `lane`, or the F window
template <int NVR>
__device__ __forceinline__ uint32_t rec_lane_offset(const RecSrc &r, int lane) {
    if (rec_per_pixel(NVR)) return (uint32_t)lane;
    return lane < 2 * NVR ? (uint32_t)lane : (uint32_t)r.ne;
}
// number of leading disparities of a pixel in column x whose flag is the pixel's own d1 flag
__device__ __forceinline__ int rec_nfill(const RecSrc &r, int x, int W) {
    const int xhi = W - 2 + r.dmin;              // columns up to here have every disparity in the string
    if ((unsigned)(x - r.dmin - 1) < (unsigned)max(W - 2, 0)) return 0;   // dmin < x <= xhi: all but the ends of a row
    if (x <= r.dmin || W < 3) return r.D;
    return min(x - xhi, r.D);
}
// kbits: the flags of disparities k0 .. k0+nv-1 (bit 0 = k0); the first nf disparities of the pixel take f1
__device__ __forceinline__ uint32_t rec_apply_fill(uint32_t kbits, int nf, int k0, int nv, bool f1) {
    const int c = min(max(nf - k0, 0), nv);
    const uint32_t fm = (1u << c) - 1u;      // nv <= 16
    return f1 ? (kbits | fm) : (kbits & ~fm);
}

// host side of the layout
static int scan_joff(const Geom &g) { return 32 * ((std::max(0, 1 - g.dmin) + 31) / 32); }
int scan_row_dwords(const Geom &g) {   // ne
    return ((g.W - 2 + std::max(g.dmin, 0) + scan_joff(g)) >> 5) + 2 * g.NV + 2;
}
size_t scan_rows_total_dwords(const Geom &g) { return (size_t)3 * g.H * rec_row_dwords(g.NV, scan_row_dwords(g), g.W); }
size_t scan_base_total_dwords(const Geom &g) { return (size_t)3 * g.H * 2 * (scan_row_dwords(g) + 1); }

// unshifted strings: base[family][row][E | F][nb], nb = ne + 1. One lane per bit (64 bits = two dwords per wave),
// gathered by ballot.
__global__ void __launch_bounds__(256)
k_scan_rows(const uint8_t *__restrict__ step_l, const uint8_t *__restrict__ step_r, uint32_t *__restrict__ base,
            int W, int H, int dmin, int joff, int nb) {
    const int lane = threadIdx.x & 63;
    const int pairs = (nb + 1) >> 1;                       // 64-bit pieces per string
    const size_t wid = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wid >= (size_t)H * pairs) return;                  // (whole waves leave together)
    const int y = (int)(wid / pairs), iw = 2 * (int)(wid - (size_t)y * pairs);
    const uint8_t *rrow = step_r + (size_t)y * W, *lrow = step_l + (size_t)y * W;
    const int j = iw * 32 + lane - joff;
    bool e_hf = false, e_hb = false, e_v = false, f_hf = false, f_hb = false, f_v = false;
    if (j >= 0 && W >= 3) {
        const int xc = min(max(W - 2 - j, 1), W - 2);
        const uint32_t s0 = rrow[xc], s1 = rrow[xc + 1];
        e_hf = s0 & 1u;            // step between right pixels xc and xc-1
        e_hb = s1 & 1u;            // step between right pixels xc and xc+1
        e_v = (s0 >> 1) & 1u;      // step between rows y and y-1 at column xc
    }
    const int x = W - 2 + dmin - j;                        // the left pixel whose disparities start at j
    if (x >= 0 && x < W) {
        const uint32_t sl = lrow[x];
        f_hf = sl & 1u;                                    // step between left pixels x and x-1
        f_hb = (x + 1 < W) ? (lrow[x + 1] & 1u) : false;   // ... x and x+1
        f_v = (sl >> 1) & 1u;                              // ... rows y and y-1
    }
    const unsigned long long m0 = __ballot(e_hf), m1 = __ballot(f_hf), m2 = __ballot(e_hb), m3 = __ballot(f_hb),
                             m4 = __ballot(e_v), m5 = __ballot(f_v);
    // lanes 0..5 store the low dwords of the six strings, lanes 8..13 the high dwords
    const int k = lane & 7, hi = lane >> 3;
    if (k < 6 && hi < 2 && iw + hi < nb) {
        const unsigned long long m = k == 0 ? m0 : k == 1 ? m1 : k == 2 ? m2 : k == 3 ? m3 : k == 4 ? m4 : m5;
        const size_t fam = (size_t)H * 2 * nb, row = (size_t)y * 2 * nb;
        base[(size_t)(k >> 1) * fam + row + (size_t)(k & 1) * nb + iw + hi] = (uint32_t)(m >> (32 * hi));
    }
}

// the window tables: by shift, out[family * H + row][c][E | F][i] = bits 32 i + c ... of the string
__global__ void __launch_bounds__(256)
k_scan_copies(const uint32_t *__restrict__ base, uint32_t *__restrict__ out, int H, int ne) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)3 * H * 64 * ne;
    if (gid >= total) return;
    const int i = (int)(gid % ne);
    const size_t t = gid / ne;               // ((family * H + row) * 32 + c) * 2 + part
    const int c = (int)((t >> 1) & 31);
    const size_t str = (t >> 6) * 2 + (t & 1);
    const uint32_t *b = base + str * (size_t)(ne + 1) + i;
    out[gid] = __funnelshift_r(b[0], b[1], c);
}
// ... and records per pixel, out[family][row][x][w]: E windows j0 + 32 w for w < 2 NVR with the pixel's own d1
// flag in front, then a dword holding d1. One thread per (pixel, w) writes the dword of all three families.
template <int NVR>
__global__ void __launch_bounds__(256)
k_scan_records(const uint32_t *__restrict__ base, uint32_t *__restrict__ out, int W, int H, int ne, int jbase, int dmin, int D) {
    constexpr int RECW = 2 * NVR + 1;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)W * RECW;              // dwords per image row of a family
    if (gid >= (size_t)H * per) return;
    const int y = (int)(gid / per);
    const int k = (int)(gid - (size_t)y * per);
    const int x = k / RECW, w = k - x * RECW;
    const int j = jbase - x, i = j >> 5, c = j & 31;
    const size_t nb = (size_t)ne + 1, fam_in = (size_t)H * 2 * nb, fam_out = (size_t)H * per;
    // the leading disparities that take d1 (as rec_nfill counts them); dword w covers disparities 32 w ...
    int nf = 0;
    if (x - dmin < 1 || W < 3) nf = D;
    else nf = min(max(x - (W - 2 + dmin), 0), D);
    const int m = min(max(nf - 32 * w, 0), 32);
    const uint32_t fm = m >= 32 ? 0xffffffffu : ((1u << m) - 1u);
#pragma unroll
    for (int f = 0; f < 3; f++) {
        const uint32_t *b = base + (size_t)f * fam_in + (size_t)y * 2 * nb;
        const uint32_t f1 = (b[nb + i] >> c) & 1u;
        uint32_t v = f1;
        if (w < RECW - 1) {
            v = __funnelshift_r(b[i + w], b[i + w + 1], c);
            v = f1 ? (v | fm) : (v & ~fm);
        }
        out[(size_t)f * fam_out + gid] = v;
    }
}

void launch_scan_records(const Planes &p, const Geom &g, hipStream_t s) {
    const int ne = scan_row_dwords(g), nb = ne + 1;
    const size_t nrow = (size_t)g.H * ((nb + 1) / 2) * 64;   // a wave per 64 bits of a row's strings
    k_scan_rows<<<dim3((unsigned)((nrow + 255) / 256)), dim3(256), 0, s>>>(p.step_l, p.step_r, p.rec_base, g.W, g.H, g.dmin,
                                                                          scan_joff(g), nb);
    const size_t total = scan_rows_total_dwords(g);
    if (rec_per_pixel(g.NV)) {
        const int jbase = g.W - 2 + g.dmin + scan_joff(g);
        const dim3 grid((unsigned)((total / 3 + 255) / 256));
        if (g.NV == 1) k_scan_records<1><<<grid, dim3(256), 0, s>>>(p.rec_base, p.rec_rows, g.W, g.H, ne, jbase, g.dmin, g.D);
        else k_scan_records<2><<<grid, dim3(256), 0, s>>>(p.rec_base, p.rec_rows, g.W, g.H, ne, jbase, g.dmin, g.D);
    } else
        k_scan_copies<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(p.rec_base, p.rec_rows, g.H, ne);
}

// the record source of a pass
static RecSrc rec_source(const Planes &p, const Geom &g, int vertical, int forward) {
    const int ne = scan_row_dwords(g);
    const int family = vertical ? 2 : (forward ? 0 : 1);
    const int row = rec_row_dwords(g.NV, ne, g.W);
    return RecSrc{p.rec_rows + (size_t)family * g.H * row, ne, g.W - 2 + g.dmin + scan_joff(g), g.dmin, g.D, row};
}

// ---- the pass ----------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, false));
}

// v_min without the canonicalising v_max the compiler puts in front of fminf() operands it cannot prove to be
// quiet: no value in these kernels is ever a NaN (finite costs and the Large_Float sentinel only).
__device__ __forceinline__ float min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
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
        typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q = 0; q < NV / 4; q++) {
            f4 t = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            reinterpret_cast<f4 *>(p)[q] = t;
        }
    }
}

typedef __attribute__((address_space(3))) void lds_void;

template <typename T>
__device__ __forceinline__ T lds_read(uint32_t byte_off) {
    typedef const volatile T __attribute__((address_space(3))) *lds_ptr;
    return *(lds_ptr)(uintptr_t)byte_off;
}
template <int NV>
__device__ __forceinline__ void lds_read_vec(uint32_t byte_off, float (&v)[NV]) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    if constexpr (NV == 1) {
        v[0] = lds_read<float>(byte_off);
    } else if constexpr (NV == 2) {
        const f2 t = lds_read<f2>(byte_off);
        v[0] = t.x; v[1] = t.y;
    } else {
#pragma unroll
        for (int q = 0; q < NV / 4; q++) {
            const f4 t = lds_read<f4>(byte_off + 16 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    }
}

template <int NV, bool WTA, int RSO = 0> struct ScanCfg {
    static constexpr int DATA_B = 256 * NV;                         // one pixel's cost vector
    static constexpr int FLAG_B = ((8 * NV + 4 + 15) / 16) * 16;    // its colour-step record, padded to 16 B
    static constexpr int SLOT_B = DATA_B + FLAG_B;
    static constexpr int NDMA = NV >= 4 ? NV / 4 : NV;              // LDS-DMA instructions per cost vector
    static constexpr int NST = (NV >= 4 ? NV / 4 : 1) + (WTA ? 1 : 0);  // store instructions per path pixel
    static constexpr int OPS = NDMA + 1 + NST;                      // vector-memory operations per path pixel
    static constexpr int RS_MEM = NV <= 4 ? 16 : (NV == 8 ? 8 : 4); // ring slots per wave that fit ~17 KB
    static constexpr int RS_CNT = 63 / OPS + 2;                     // ... and that keep the counted wait within 6 bits
    static constexpr int RS = RSO ? RSO : (RS_MEM < RS_CNT ? RS_MEM : RS_CNT);   // RSO: ring size chosen by the caller
    // operations younger than the fill of step s+1 at the point step s waits for it (see the loop)
    static constexpr int WAITN = (RS - 2) * OPS;
    static_assert(WAITN <= 63 && RS >= 3, "vmcnt is a 6-bit counter");
};

// VERT = 0: wave walks row `line` along x; VERT = 1: walks column `line` along y.
// FULL: Dp == D == 64*NV, every lane owns NV real disparities (no masking anywhere).
// WTA: this is the last pass; while a pixel's final costs are still in registers, also write its
//      left-view winner-take-all disparity with the sub-pixel fit (ComputeDisparityCuda,
//      ADC/ADCensusStereo.cu:200-267): first strict minimum over d, invalid (+inf) at either end of the range.
// One wavefront, one scanline: `line` is the row / column, `ring` this wave's RS x SLOT_B bytes of LDS.
template <int NV, int VERT, bool FULL, bool WTA, int RSO = 0>
__device__ __forceinline__ void
scan_wave(const float *__restrict__ src, float *__restrict__ dst, const RecSrc rec,
          float *__restrict__ disp, int W, int H, int dmin, int D, int Dp, int forward, const ScanParams &sp,
          int line, uint8_t *ring) {
    typedef ScanCfg<NV, WTA, RSO> C;
    const int lane = threadIdx.x & 63;
    const int LEN = VERT ? H : W;
    const int dir = forward ? 1 : -1;
    const int k0 = lane * NV;
    const bool active = FULL || k0 < Dp;       // lane owns part of the (padded) disparity vector
    const int koff = active ? k0 : 0;
    // path head, per-step strides (in pixels); the backward vertical pass takes the flags of the row below
    // (the step between rows y and y+1 is stored with row y+1)
    const int x0 = VERT ? line : (forward ? 0 : W - 1);
    const int y0 = VERT ? (forward ? 0 : H - 1) : line;
    const ptrdiff_t pstep = VERT ? (ptrdiff_t)dir * W : (ptrdiff_t)dir;
    const ptrdiff_t pix0 = (ptrdiff_t)y0 * W + x0;
    const int yshift = (VERT && !forward) ? 1 : 0;

    // ---- fill addressing (lane <-> DMA element, which differs from the compute mapping for NV < 4)
    // data: NV >= 4: instruction q moves floats q*256 + 4*lane .. +3 (16 B per lane);
    //       NV <  4: instruction h moves float h*64 + lane (4 B per lane)
    const float *gfill = src + (NV >= 4 ? 4 * lane : lane);
    bool fok[C::NDMA];
#pragma unroll
    for (int q = 0; q < C::NDMA; q++) fok[q] = FULL || ((NV >= 4 ? q * 256 + 4 * lane : q * 64 + lane) < Dp);
    // the counted wait below assumes every one of the NDMA fill instructions has at least one live lane
    const bool counted = FULL || (Dp > (C::NDMA - 1) * (NV >= 4 ? 256 : 64));
    // flags: lanes 0 .. 2*NV-1 fetch the E windows of the pixel, lane 2*NV its F window
    const uint32_t *gflag = rec.fam + rec_lane_offset<NV>(rec, lane);
    const bool flag_lane = lane <= 2 * NV;
    const int nf_col = VERT ? rec_nfill(rec, line, W) : 0;   // a column's pixels all patch the same disparities

    auto issue_fill = [&](int s) {     // path pixel s -> ring slot s % RS
        const ptrdiff_t p = pix0 + (ptrdiff_t)s * pstep;
        uint8_t *slot = ring + (s % C::RS) * C::SLOT_B;
        const float *g = gfill + (size_t)p * Dp;
#pragma unroll
        for (int q = 0; q < C::NDMA; q++) {
            if (fok[q]) {
                if constexpr (NV >= 4)
                    __builtin_amdgcn_global_load_lds(g + q * 256, (lds_void *)(slot + q * 1024), 16, 0, 0);
                else
                    __builtin_amdgcn_global_load_lds(g + q * 64, (lds_void *)(slot + q * 256), 4, 0, 0);
            }
        }
        if (flag_lane) {
            const int fx = VERT ? line : x0 + dir * s, fy = VERT ? y0 + dir * s + yshift : line;
            __builtin_amdgcn_global_load_lds(gflag + rec_index<NV>(rec, fx, fy), (lds_void *)(slot + C::DATA_B), 4, 0, 0);
        }
    };
    // compute-side view of a slot. The reads go through an LDS address rebuilt from an integer: the
    // compiler then has no basis to tie them to the in-flight LDS-DMA fills (it would otherwise put a
    // full vmcnt(0) in front of every ring read); ordering is what the counted waits in the loop provide.
    const uint32_t ring_off = (uint32_t)(uintptr_t)ring;
    auto read_slot = [&](int s, float (&v)[NV], uint32_t &bits, uint32_t &f1) {
        const uint32_t slot = ring_off + (uint32_t)(s % C::RS) * C::SLOT_B;
        lds_read_vec<NV>(slot + (uint32_t)k0 * 4, v);
        // raw record words: they are decoded (shift / mask) where they are used, one path pixel later,
        // so that nothing waits for these reads before the current pixel has been computed
        if constexpr (NV == 16) bits = lds_read<uint16_t>(slot + C::DATA_B + 2 * lane);
        else bits = (uint32_t)lds_read<uint8_t>(slot + C::DATA_B + ((lane * NV) >> 3));
        f1 = lds_read<uint32_t>(slot + C::DATA_B + 8 * NV);
    };
    const int bshift = NV == 16 ? 0 : (lane * NV) & 7;

    // ---- path head: dst = src, Lp = src
    float Lp[NV];
    {
        float s[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) s[j] = 0.0f;
        if (active) load_vec<NV>(src + (size_t)pix0 * Dp + koff, s);
#pragma unroll
        for (int j = 0; j < NV; j++) Lp[j] = (active && (FULL || k0 + j < D)) ? s[j] : ADC_LARGE_F;
        if (active) {
            float o[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) o[j] = (FULL || k0 + j < D) ? s[j] : 0.0f;
            store_vec<NV>(dst + (size_t)pix0 * Dp + koff, o);
        }
    }
    float lmin = Lp[0];
#pragma unroll
    for (int j = 1; j < NV; j++) lmin = fminf(lmin, Lp[j]);
    float minLp = wave_min(lmin);
    // c[] = this pixel's final costs (Large_Float past D), cmin = their minimum
    auto write_wta = [&](const float (&c)[NV], float cmin, ptrdiff_t pixel) {
        int kk = 0x7fffffff;
#pragma unroll
        for (int j = NV - 1; j >= 0; j--) kk = (c[j] == cmin) ? k0 + j : kk;   // first disparity of this lane at the minimum
        const int best = wave_min_i32(kk);                                      // first over the wave
        float out = __int_as_float(0x7f800000);                                 // Invalid_Float
        if (best > 0 && best < D - 1 && cmin < ADC_LARGE_F) {
            const int l1 = (best - 1) / NV, e1 = (best - 1) - l1 * NV;
            const int l2 = (best + 1) / NV, e2 = (best + 1) - l2 * NV;
            float s1 = c[0], s2 = c[0];
#pragma unroll
            for (int j = 1; j < NV; j++) { s1 = (e1 == j) ? c[j] : s1; s2 = (e2 == j) ? c[j] : s2; }
            const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s1), l1));
            const float c2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s2), l2));
            const float fb = (float)(best + dmin);
            const float denom = c1 + c2 - 2 * cmin;
            out = (denom != 0.0f) ? fb + (c1 - c2) / (denom * 2.0f) : fb;
        }
        if (lane == 0) disp[pixel] = out;
    };
    if (WTA) write_wta(Lp, minLp, pix0);
    if (LEN < 2) return;

    // ---- prologue: fills of path pixels 1 .. RS-1, all landed, pixel 1 in registers
    for (int t = 1; t < C::RS && t < LEN; t++) issue_fill(t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float cur[NV];
    uint32_t cbits, cf1;
    read_slot(1, cur, cbits, cf1);

    float *outp = dst + (size_t)(pix0 + pstep) * Dp + koff;   // output of the next pixel to store
    const ptrdiff_t ostep = pstep * (ptrdiff_t)Dp;

    for (int s = 1; s < LEN; s++) {
        // (A) fill of pixel s+RS-1 into the slot pixel s-1 occupied (its reads completed an iteration ago).
        //     (Issuing the fills of four consecutive pixels together every fourth step, and likewise the
        //     stores, was measured: no gain.)
        const int t = s + C::RS - 1;
        if (t < LEN) issue_fill(t);
        // (B) pixel s+1 must have landed. In steady state the operations issued after its fill are
        //     (RS-2) x (fill + store) groups -- exactly WAITN; near the end of the path that count
        //     shrinks, so drain instead.
        if (t < LEN && counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::WAITN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (C) start reading pixel s+1 while pixel s is computed
        float nxt[NV];
        uint32_t nbits = 0, nf1 = 0;
#pragma unroll
        for (int j = 0; j < NV; j++) nxt[j] = 0.0f;
        if (s + 1 < LEN) read_slot(s + 1, nxt, nbits, nf1);
        // (D) the recurrence for pixel s
        const bool d1hi = cf1 & 1u;
        uint32_t kbits = cbits >> bshift;
        if constexpr (!rec_per_pixel(NV)) {
            const int nf = VERT ? nf_col : rec_nfill(rec, x0 + dir * s, W);
            if (nf) kbits = rec_apply_fill(kbits, nf, k0, NV, d1hi);
        }
        const float P1a = d1hi ? sp.P1[1] : sp.P1[0], P1b = d1hi ? sp.P1[2] : sp.P1[1];
        const float P2a = d1hi ? sp.P2[1] : sp.P2[0], P2b = d1hi ? sp.P2[2] : sp.P2[1];
        const float up = dpp_f<0x138>(ADC_LARGE_F, Lp[NV - 1]);  // wave_shr:1 -> Lp[k0-1]
        const float dn = dpp_f<0x130>(ADC_LARGE_F, Lp[0]);       // wave_shl:1 -> Lp[k0+NV]
        float l4a = minLp + P2a, l4b = minLp + P2b;
        asm volatile("" : "+v"(l4a), "+v"(l4b));   // keep the two sums: selecting the penalty first costs an addition per disparity
        float nw[NV];
        float lm = ADC_LARGE_F;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const bool hi = (kbits >> j) & 1u;
            const float P1 = hi ? P1b : P1a;
            // min(a + P1, b + P1) == min(a, b) + P1 bit for bit (rounding is monotonic), one addition less
            const float l1 = Lp[j];
            const float l23 = min_raw(j == 0 ? up : Lp[j - 1], j == NV - 1 ? dn : Lp[j + 1]) + P1;
            const float l4 = hi ? l4b : l4a;
            const float m = min3_raw(l1, l23, l4);
            const float v = (cur[j] + m) / 2.0f;
            nw[j] = (FULL || (active && k0 + j < D)) ? v : ADC_LARGE_F;
            lm = fminf(lm, nw[j]);
        }
        {
            float o[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) o[j] = (FULL || k0 + j < D) ? nw[j] : 0.0f;
            if (active) store_vec<NV>(outp, o);
            outp += ostep;
        }
#pragma unroll
        for (int j = 0; j < NV; j++) Lp[j] = nw[j];
        minLp = wave_min(lm);
        if (WTA) write_wta(Lp, minLp, pix0 + (ptrdiff_t)s * pstep);
        // (E)
#pragma unroll
        for (int j = 0; j < NV; j++) cur[j] = nxt[j];
        cbits = nbits;
        cf1 = nf1;
    }
}

template <int NV, int VERT, bool FULL, bool WTA>
__global__ void __launch_bounds__(256)
k_scanline(const float *__restrict__ src, float *__restrict__ dst, const RecSrc rec,
           float *__restrict__ disp, int W, int H, int dmin, int D, int Dp, int forward, ScanParams sp) {
    typedef ScanCfg<NV, WTA> C;
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int line = blockIdx.x * 4 + wave;
    if (line >= (VERT ? W : H)) return;
    scan_wave<NV, VERT, FULL, WTA>(src, dst, rec, disp, W, H, dmin, D, Dp, forward, sp, line,
                                   lds_raw + (size_t)wave * C::RS * C::SLOT_B);
}

// f(integral_constant<int, U>) for U = FROM .. TO-1, unrolled at compile time
template <int FROM, int TO, class F>
__device__ __forceinline__ void unroll_pixels(F &&f) {
    if constexpr (FROM < TO) {
        f(std::integral_constant<int, FROM>{});
        unroll_pixels<FROM + 1, TO>(f);
    }
}

// ---- several wavefronts per scanline --------------------------------------------------------------------
// A pass is a serial chain per scanline. One wave per scanline is the cheapest way to run it, but the wave
// count then has to fit the chip: a SIMD that gets a second full-width wave needs twice as long per path
// pixel, and the pass lasts as long as its slowest scanline (1024 rows: 0.83 ms, 1080 rows: 1.19 ms on 1024
// SIMDs). Here a scanline is shared by the NPARTS = 2 or 4 waves of a workgroup, each holding a contiguous
// share of the disparities (NVW per lane). Per path pixel they exchange, through LDS and one barrier, what
// crosses the seams: their partial minima of Lr(p-r, .) and the costs next to a neighbour's share. min() is
// exact and commutative, so the results are the same bits as the one-wave kernel's.
// Only for Dp == D == NPARTS*64*NVW, passes without the fused WTA.
// Scanlines of up to this many floats per lane (D <= 64) get a ring of 8 slots instead of 16: the main loop is unrolled
// over one round of the ring and keeps an address per slot and stream in scalar registers; 16 slots spill them
// (450 x 375 x 64, four passes: 0.30 ms rolled, 0.33 ms unrolled with 16 slots, 0.28 ms with 8).
#ifndef SCAN_RS8_UPTO
#define SCAN_RS8_UPTO 1
#endif
template <int NVW, int NPARTS = 2, bool WTA = false> struct ScanCfgSplit {
    static constexpr int NVF = NPARTS * NVW;                          // disparities per lane over all the scanline's waves = the record's NV
    static constexpr int DATA_B = 256 * NVW;                          // this wave's half of a pixel's cost vector
    static constexpr int FLAG_B = ((8 * NVF + 4 + 15) / 16) * 16;     // the pixel's colour-step record (both halves)
    static constexpr int SLOT_B = DATA_B + FLAG_B;
    static constexpr int NDMA = NVW >= 4 ? NVW / 4 : NVW;
    static constexpr int NST = (NVW >= 4 ? NVW / 4 : 1) + (WTA ? 1 : 0);
    static constexpr int OPS = NDMA + 1 + NST;
    static constexpr int RS_MEM = (NVW <= 4 && NPARTS * NVW > SCAN_RS8_UPTO) ? 16 : 8;
    static constexpr int RS_CNT = 63 / OPS + 2;
    static constexpr int RS = (RS_MEM <= RS_CNT) ? RS_MEM : (RS_CNT >= 8 ? 8 : 4);   // a power of two (slots are masked)
    static constexpr int WAITN = (RS - 2) * OPS;
    static_assert(WAITN <= 63 && RS >= 3, "vmcnt is a 6-bit counter");
};

// The per-pixel instruction count is kept down (a wave issues at most one instruction per 4-cycle turn,
// and with ~2 waves per SIMD nothing else fills the turns it wastes):
//   * the path pointers live in scalar registers and advance by scalar adds; fills, the record fetch and
//     the store address memory as scalar base + 32-bit lane offset,
//   * the LDS-DMA pieces of one pixel share one M0 value (the instruction offset moves both addresses),
//   * the loop is peeled (no end-of-path tests in the main part) and runs two pixels per trip, so the
//     exchange parity is static and the "next pixel" registers swap roles instead of being copied,
//   * every lane writes {wave minimum, its own boundary cost} to its own exchange slot; the other wave
//     reads the slot of the lane next to the seam -- no execution-mask juggling around the LDS write.
// NPARTS = 2 or 4 waves share the scanline `line`; this is wave `part` of them (its disparities:
// part*64*NVW ...). lds0: LDS byte offset of the scanline's area = NPARTS rings of RS x SLOT_B bytes, then
// the exchange area [parity][part][lane] x XB bytes. With four parts a wave needs the minima of the three
// others and one boundary cost from each neighbour: every lane writes {wave minimum, its lowest cost, its
// highest cost}, and the readers pick lane 63 of the wave below and lane 0 of the wave above.
// WTA (one wave per scanline only): the pass is the last one and also writes the left-view winner-take-all
// disparity of every pixel, as scan_wave does.
template <int NVW, int NPARTS, int VERT, bool WTA = false>
__device__ __forceinline__ void
scan_split_wave(const float *__restrict__ src, float *__restrict__ dst, const RecSrc rec,
                int W, int H, int Dp, int forward, const ScanParams &sp, int line, int part, uint32_t lds0,
                float *__restrict__ disp = nullptr, int dmin = 0) {
    typedef ScanCfgSplit<NVW, NPARTS, WTA> C;
    static_assert(!WTA || NPARTS == 1, "the fused WTA needs a pixel's whole disparity range in one wave");
    constexpr int NV = NVW;
    constexpr int RS = C::RS;
    static_assert((RS & (RS - 1)) == 0, "ring size must be a power of two");
    static_assert(NPARTS == 1 || NPARTS == 2 || NPARTS == 4, "one, two or four waves per scanline");
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr uint32_t XB = NPARTS == 4 ? 16u : 8u;              // bytes per lane slot of the exchange area
    constexpr uint32_t XPAR = NPARTS * 64u * XB;                 // bytes per parity
    const int lane = threadIdx.x & 63;
    const int half = part;                                       // (two parts: 0 = low disparities, 1 = high)
    const uint32_t ring_off = lds0 + (uint32_t)part * RS * C::SLOT_B;
    const uint32_t xch = lds0 + (uint32_t)NPARTS * RS * C::SLOT_B;
    const uint32_t xw = xch + (uint32_t)part * 64u * XB + (uint32_t)lane * XB;
    // two parts: the other wave's lane at the seam. Four parts: lane 63 of the wave below, lane 0 of the
    // wave above, lane 0 of the remaining one (indices wrap; what the end parts read there is discarded).
    const uint32_t xr = NPARTS == 2 ? xch + (uint32_t)(1 - part) * 64u * XB + (part ? 63u : 0u) * XB
                                    : xch + (uint32_t)((part + NPARTS - 1) % NPARTS) * 64u * XB + 63u * XB;
    const uint32_t xr_up = xch + (uint32_t)((part + 1) % NPARTS) * 64u * XB;
    const uint32_t xr_far = xch + (uint32_t)((part + 2) % NPARTS) * 64u * XB;
    const bool has_below = part > 0, has_above = part < NPARTS - 1;
    const int LEN = VERT ? H : W;
    const int dir = forward ? 1 : -1;
    const int kbase = part * 64 * NV;             // first disparity of this wave
    const int k0 = kbase + lane * NV;
    const int x0 = VERT ? line : (forward ? 0 : W - 1);
    const int y0 = VERT ? (forward ? 0 : H - 1) : line;
    const ptrdiff_t pstep = VERT ? (ptrdiff_t)dir * W : (ptrdiff_t)dir;
    const ptrdiff_t pix0 = (ptrdiff_t)y0 * W + x0;
    const int yshift = (VERT && !forward) ? 1 : 0;   // the backward vertical pass takes the flags of the row below

    // uniform byte addresses of path pixel 0 and the byte steps along the path
    const char *const src_b = reinterpret_cast<const char *>(src + (size_t)pix0 * Dp + kbase);
    char *const dst_b = reinterpret_cast<char *>(dst + (size_t)pix0 * Dp + kbase);
    const ptrdiff_t vstep = pstep * (ptrdiff_t)Dp * 4;
    // the flags of path pixel t: uniform address of its first E window; a constant step along the path when the
    // table is linear or the path a column
    constexpr bool PER_PIXEL = rec_per_pixel(C::NVF);
    const char *const rec_b = reinterpret_cast<const char *>(rec.fam + rec_index<C::NVF>(rec, x0, y0 + yshift));
    const ptrdiff_t rstep = VERT ? (ptrdiff_t)dir * rec.row * 4 : (ptrdiff_t)dir * (2 * C::NVF + 1) * 4;
    auto rec_ptr = [&](int t) {
        if constexpr (VERT || PER_PIXEL) return rec_b + (ptrdiff_t)t * rstep;
        else return reinterpret_cast<const char *>(rec.fam + rec_index<C::NVF>(rec, x0 + dir * t, line));
    };
    const int nf_col = VERT ? rec_nfill(rec, line, W) : 0;
    const uint32_t vo_fill = (uint32_t)lane * (NV >= 4 ? 16u : 4u);
    const uint32_t vo_rec = rec_lane_offset<C::NVF>(rec, lane) * 4u;
    const uint32_t vo_out = (uint32_t)lane * (uint32_t)(NV * 4);
    const bool flag_lane = lane <= 2 * C::NVF;

    auto issue_fill = [&](const char *sp_, const char *rp_, uint32_t slot) {
        lds_void *l = (lds_void *)(uintptr_t)(ring_off + slot);
        const char *g = sp_ + vo_fill;
        static_assert(C::NDMA <= 2, "one or two LDS-DMA pieces per pixel and wave");
        if constexpr (NV >= 4) {
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(g), l, 16, 0, 0);
            if constexpr (C::NDMA == 2) __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(g), l, 16, 1024, 0);
        } else {
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(g), l, 4, 0, 0);
            if constexpr (C::NDMA == 2) __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(g), l, 4, 256, 0);
        }
        if (flag_lane)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(rp_ + vo_rec),
                                             (lds_void *)(uintptr_t)(ring_off + slot + C::DATA_B), 4, 0, 0);
    };
    const uint32_t rd_data = ring_off + (uint32_t)(lane * NV) * 4u;
    const uint32_t rd_bits = ring_off + C::DATA_B + (uint32_t)(k0 >> 3);
    const uint32_t rd_f1 = ring_off + C::DATA_B + 8u * C::NVF;
    auto read_slot = [&](uint32_t slot, float (&v)[NV], uint32_t &bits, uint32_t &f1) {
        lds_read_vec<NV>(rd_data + slot, v);
        if constexpr (NV == 16) bits = lds_read<uint16_t>(rd_bits + slot);
        else bits = (uint32_t)lds_read<uint8_t>(rd_bits + slot);
        f1 = lds_read<uint32_t>(rd_f1 + slot);
    };
    const int bshift = NV == 16 ? 0 : k0 & 7;
    // in: this wave's minimum and its lanes' lowest / highest cost. out: the minimum over the other waves,
    // and the costs just below this wave's first / just above its last disparity (Large_Float at the ends of
    // the range; with two parts only the one at the seam is meaningful, the other is not used).
    auto exchange = [&](uint32_t par, float own_min, float lo, float hi, float &omin, float &e_below, float &e_above) {
        if constexpr (NPARTS == 1) {          // the wave holds the whole disparity range: nothing to exchange
            omin = own_min;
            e_below = ADC_LARGE_F;
            e_above = ADC_LARGE_F;
        } else if constexpr (NPARTS == 2) {
            typedef volatile f2 __attribute__((address_space(3))) *xp;
            f2 v = {own_min, half ? lo : hi};
            *(xp)(uintptr_t)(xw + par * XPAR) = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const f2 o = lds_read<f2>(xr + par * XPAR);
            omin = o.x;
            e_below = o.y;
            e_above = o.y;
        } else {
            typedef volatile f4 __attribute__((address_space(3))) *xp;
            f4 v = {own_min, lo, hi, 0.0f};
            *(xp)(uintptr_t)(xw + par * XPAR) = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const f4 a = lds_read<f4>(xr + par * XPAR);        // wave below: {min, ., highest}
            const f4 b = lds_read<f4>(xr_up + par * XPAR);     // wave above: {min, lowest, .}
            const float c = lds_read<float>(xr_far + par * XPAR);
            omin = fminf(fminf(a.x, b.x), c);
            e_below = has_below ? a.z : ADC_LARGE_F;
            e_above = has_above ? b.y : ADC_LARGE_F;
        }
    };

    // c[] = a pixel's final costs, cmin = their minimum (ComputeDisparityCuda, ADC/ADCensusStereo.cu:200-267)
    const int D = Dp;
    auto write_wta = [&](const float (&c)[NV], float cmin, ptrdiff_t pixel) {
        int kk = 0x7fffffff;
#pragma unroll
        for (int j = NV - 1; j >= 0; j--) kk = (c[j] == cmin) ? k0 + j : kk;   // first disparity of this lane at the minimum
        const int best = wave_min_i32(kk);                                      // first over the wave
        float out = __int_as_float(0x7f800000);                                 // Invalid_Float
        if (best > 0 && best < D - 1 && cmin < ADC_LARGE_F) {
            const int l1 = (best - 1) / NV, e1 = (best - 1) - l1 * NV;
            const int l2 = (best + 1) / NV, e2 = (best + 1) - l2 * NV;
            float s1 = c[0], s2 = c[0];
#pragma unroll
            for (int j = 1; j < NV; j++) { s1 = (e1 == j) ? c[j] : s1; s2 = (e2 == j) ? c[j] : s2; }
            const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s1), l1));
            const float c2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s2), l2));
            const float fb = (float)(best + dmin);
            const float denom = c1 + c2 - 2 * cmin;
            out = (denom != 0.0f) ? fb + (c1 - c2) / (denom * 2.0f) : fb;
        }
        if (lane == 0) disp[pixel] = out;
    };

    // ---- path head: dst = src, Lp = src
    float Lp[NV];
    load_vec<NV>(reinterpret_cast<const float *>(src_b + vo_out), Lp);
    store_vec<NV>(reinterpret_cast<float *>(dst_b + vo_out), Lp);
    float lmin = Lp[0];
#pragma unroll
    for (int j = 1; j < NV; j++) lmin = fminf(lmin, Lp[j]);
    float minLp, e_below, e_above;
    {
        const float own = wave_min(lmin);
        float omin;
        exchange(0u, own, Lp[0], Lp[NV - 1], omin, e_below, e_above);
        minLp = fminf(own, omin);
    }
    if constexpr (WTA) write_wta(Lp, minLp, pix0);
    if (LEN < 2) return;

    for (int t = 1; t < RS && t < LEN; t++)
        issue_fill(src_b + (ptrdiff_t)t * vstep, rec_ptr(t), (uint32_t)(t & (RS - 1)) * C::SLOT_B);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float ca[NV], cb[NV];
    uint32_t ba, fa, bb = 0, fb = 0;
    read_slot(1u * C::SLOT_B, ca, ba, fa);

    // one path pixel: costs/record of this pixel in (cur, cbits, cf1); the next pixel's are fetched into
    // (nxt, nbits, nf1). FILL / NEXT: the fill RS-1 pixels ahead / the next pixel exist.
    // PATCH: the pixel may be one whose first disparities take its own d1 flag (the ends of a row).
    // FS / NS: the ring slots of the fill and of the next pixel when the caller knows them (the unrolled loop: they
    // become immediates of the instructions), -1 to have them computed from s.
    auto pixel = [&](int s, uint32_t par, bool FILL, bool NEXT, bool PATCH, int FS, int NS, const float (&cur)[NV], uint32_t cbits,
                     uint32_t cf1, float (&nxt)[NV], uint32_t &nbits, uint32_t &nf1) {
        if (FILL) {
            const int t = s + RS - 1;
            issue_fill(src_b + (ptrdiff_t)t * vstep, rec_ptr(t), (uint32_t)(FS >= 0 ? FS : (t & (RS - 1))) * C::SLOT_B);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::WAITN) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (NEXT) {
            read_slot((uint32_t)(NS >= 0 ? NS : ((s + 1) & (RS - 1))) * C::SLOT_B, nxt, nbits, nf1);
        } else {
#pragma unroll
            for (int j = 0; j < NV; j++) nxt[j] = 0.0f;
            nbits = 0; nf1 = 0;
        }
        const bool d1hi = cf1 & 1u;
        uint32_t kbits = cbits >> bshift;
        if (PATCH && !PER_PIXEL) {
            const int nf = VERT ? nf_col : rec_nfill(rec, x0 + dir * s, W);
            if (nf) kbits = rec_apply_fill(kbits, nf, k0, NV, d1hi);
        }
        const float P1a = d1hi ? sp.P1[1] : sp.P1[0], P1b = d1hi ? sp.P1[2] : sp.P1[1];
        const float P2a = d1hi ? sp.P2[1] : sp.P2[0], P2b = d1hi ? sp.P2[2] : sp.P2[1];
        float up = dpp_f<0x138>(ADC_LARGE_F, Lp[NV - 1]);   // wave_shr:1 -> Lp[k0-1]
        float dn = dpp_f<0x130>(ADC_LARGE_F, Lp[0]);        // wave_shl:1 -> Lp[k0+NV]
        if constexpr (NPARTS == 1) {
            // the DPP shifts already put Large_Float at both ends of the range
        } else if constexpr (NPARTS == 2) {
            if (half) up = lane == 0 ? e_below : up;
            else dn = lane == 63 ? e_above : dn;
        } else {
            up = lane == 0 ? e_below : up;
            dn = lane == 63 ? e_above : dn;
        }
        float l4a = minLp + P2a, l4b = minLp + P2b;
        asm volatile("" : "+v"(l4a), "+v"(l4b));   // keep the two sums: selecting the penalty first costs an addition per disparity
        float nw[NV];
        float lm = ADC_LARGE_F;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const bool hi = (kbits >> j) & 1u;
            const float P1 = hi ? P1b : P1a;
            // min(a + P1, b + P1) == min(a, b) + P1 bit for bit (rounding is monotonic), one addition less
            const float l1 = Lp[j];
            const float l23 = min_raw(j == 0 ? up : Lp[j - 1], j == NV - 1 ? dn : Lp[j + 1]) + P1;
            const float l4 = hi ? l4b : l4a;
            const float m = min3_raw(l1, l23, l4);
            nw[j] = (cur[j] + m) / 2.0f;
            lm = fminf(lm, nw[j]);
        }
        store_vec<NV>(reinterpret_cast<float *>(dst_b + (ptrdiff_t)s * vstep + vo_out), nw);
#pragma unroll
        for (int j = 0; j < NV; j++) Lp[j] = nw[j];
        const float own = wave_min(lm);
        float omin;
        exchange(par, own, Lp[0], Lp[NV - 1], omin, e_below, e_above);
        minLp = fminf(own, omin);
        if constexpr (WTA) write_wta(Lp, minLp, pix0 + (ptrdiff_t)s * pstep);
    };

    int s = 1;
    // two pixels per trip: every fill and every next pixel exists (s + 1 + RS - 1 < LEN for the second pixel)
    auto two = [&](bool patch) {
        pixel(s, 1u, true, true, patch, -1, -1, ca, ba, fa, cb, bb, fb);
        pixel(s + 1, 0u, true, true, patch, -1, -1, cb, bb, fb, ca, ba, fa);
        s += 2;
    };
    // RS unpatched pixels per trip, for s = 1 (mod RS): every ring offset is then known at compile time (a wave alone on
    // its SIMD pays an issue slot for every scalar instruction of the slot arithmetic)
    auto ring_round = [&]() {
        unroll_pixels<0, RS>([&](auto u) {
            constexpr int U = decltype(u)::value;
            if constexpr (U % 2 == 0) pixel(s + U, 1u, true, true, false, U, (U + 2) & (RS - 1), ca, ba, fa, cb, bb, fb);
            else pixel(s + U, 0u, true, true, false, U, (U + 2) & (RS - 1), cb, bb, fb, ca, ba, fa);
        });
        s += RS;
    };
    if constexpr (PER_PIXEL) {
        while (s + 2 * RS - 2 < LEN) ring_round();
        while (s + RS < LEN) two(false);
    } else {
        // path pixels s_lo <= s < s_hi have every flag in the strings (rec_nfill is 0 for them): all of a column, or a
        // row without its ends
        int s_lo = LEN, s_hi = LEN;
        if (VERT) s_lo = nf_col ? LEN : 0;
        else if (W >= 3 && forward) { s_lo = max(rec.dmin + 1, 0); s_hi = min(W - 1 + rec.dmin, W); }
        else if (W >= 3) { s_lo = max(1 - rec.dmin, 0); s_hi = min(W - 1 - rec.dmin, W); }
        while (s + RS < LEN && s < s_lo) two(true);
        while (s + RS < LEN && s + 1 < s_hi && (s & (RS - 1)) != 1) two(false);
        while (s + 2 * RS - 2 < LEN && s + RS - 1 < s_hi) ring_round();
        while (s + RS < LEN && s + 1 < s_hi) two(false);
        while (s + RS < LEN) two(true);
    }
    for (; s < LEN; s++) {
        pixel(s, (uint32_t)(s & 1), s + RS - 1 < LEN, s + 1 < LEN, true, -1, -1, ca, ba, fa, cb, bb, fb);
#pragma unroll
        for (int j = 0; j < NV; j++) ca[j] = cb[j];
        ba = bb; fa = fb;
    }
}

template <int NVW, int NPARTS> constexpr size_t scan_split_lds() {
    typedef ScanCfgSplit<NVW, NPARTS> C;
    return (size_t)NPARTS * C::RS * C::SLOT_B + (NPARTS == 1 ? 0u : 2u * NPARTS * 64u * (NPARTS == 2 ? 8u : 16u));
}

template <int NVW, int VERT>
__global__ void __launch_bounds__(128)
k_scanline_split2(const float *__restrict__ src, float *__restrict__ dst, const RecSrc rec,
                  int W, int H, int Dp, int forward, ScanParams sp) {
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    scan_split_wave<NVW, 2, VERT>(src, dst, rec, W, H, Dp, forward, sp, blockIdx.x, part, (uint32_t)(uintptr_t)lds_raw);
}

// One wave per scanline with the lean loop of scan_split_wave (NPARTS = 1): every pass without the fused WTA whose
// disparity count fills the lanes exactly (Dp == D == 64*NV, NV <= 8). Four scanlines per workgroup.
template <int NV, int VERT, bool WTA>
__global__ void __launch_bounds__(256)
k_scanline_lean(const float *__restrict__ src, float *__restrict__ dst, const RecSrc rec,
                float *__restrict__ disp, int W, int H, int dmin, int Dp, int forward, ScanParams sp) {
    typedef ScanCfgSplit<NV, 1, WTA> C;
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int line = blockIdx.x * 4 + wave;
    if (line >= (VERT ? W : H)) return;
    scan_split_wave<NV, 1, VERT, WTA>(src, dst, rec, W, H, Dp, forward, sp, line, 0,
                                      (uint32_t)(uintptr_t)lds_raw + (uint32_t)wave * C::RS * C::SLOT_B, disp, dmin);
}

// Horizontal pass over slightly more rows than the chip has SIMDs (1080 rows, 1024 SIMDs): a second
// full-width wave on a SIMD doubles that SIMD's per-pixel time and with it the whole pass. The first
// `rows_main` rows (a multiple of 4, at most one per SIMD) run one wave each; every remaining row is
// shared by the four waves of its own workgroup, 64 disparities each, which adds only a light wave to
// the SIMDs it lands on. Dp == D == 256 only (NV = 4 for the main rows, one disparity per lane for the rest).
#ifndef HYB_RS
#define HYB_RS 8
#endif
// The one-wave rows run the lean loop (scan_split_wave with one part: 4 passes 3.43 -> 3.40 ms at 1080 rows) when
// its deeper ring leaves room for the launch's workgroups (69 KB of LDS each: two per CU), else scan_wave with the
// ring of HYB_RS slots (1440 rows: 672 workgroups on 256 CUs, 7.7 ms with the lean loop against 6.5 ms).
__global__ void __launch_bounds__(256)
k_scanline_hybrid(const float *__restrict__ src, float *__restrict__ dst, const RecSrc rec,
                  int W, int H, int Dp, int forward, ScanParams sp, int rows_main, int lean_main) {
    typedef ScanCfg<4, false, HYB_RS> C;
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int main_wgs = rows_main >> 2;
    if ((int)blockIdx.x < main_wgs) {
        typedef ScanCfgSplit<4, 1, false> CL;
        if (lean_main)
            scan_split_wave<4, 1, 0, false>(src, dst, rec, W, H, Dp, forward, sp, blockIdx.x * 4 + wave, 0,
                                            (uint32_t)(uintptr_t)lds_raw + (uint32_t)wave * CL::RS * CL::SLOT_B, nullptr, 0);
        else
            scan_wave<4, 0, true, false, HYB_RS>(src, dst, rec, nullptr, W, H, 0, Dp, Dp, forward, sp, blockIdx.x * 4 + wave,
                                                 lds_raw + (size_t)wave * C::RS * C::SLOT_B);
    } else {
        __builtin_amdgcn_s_setprio(2);   // the row's four waves wait for each other every pixel: let them go first
        scan_split_wave<1, 4, 0>(src, dst, rec, W, H, Dp, forward, sp, rows_main + ((int)blockIdx.x - main_wgs), wave,
                                 (uint32_t)(uintptr_t)lds_raw);
    }
}

template <int NVW>
static void launch_split(const float *src, float *dst, const Planes &p, const Geom &g, const ScanParams &sp,
                         int vertical, int forward, hipStream_t s) {
    const int nlines = vertical ? g.W : g.H;
    const RecSrc rec = rec_source(p, g, vertical, forward);
    const size_t lds = scan_split_lds<NVW, 2>();
    if (vertical)
        k_scanline_split2<NVW, 1><<<dim3(nlines), dim3(128), lds, s>>>(src, dst, rec, g.W, g.H, g.Dp, forward, sp);
    else
        k_scanline_split2<NVW, 0><<<dim3(nlines), dim3(128), lds, s>>>(src, dst, rec, g.W, g.H, g.Dp, forward, sp);
}

template <int NV, bool WTA>
static void launch_nv(const float *src, float *dst, float *disp, const Planes &p, const Geom &g, const ScanParams &sp,
                      int vertical, int forward, hipStream_t s) {
    typedef ScanCfg<NV, WTA> C;
    const int nlines = vertical ? g.W : g.H;
    dim3 grid((nlines + 3) / 4), block(256);
    const size_t lds = (size_t)4 * C::RS * C::SLOT_B;
    const bool full = (g.D == 64 * NV) && (g.Dp == g.D);
    const RecSrc rec = rec_source(p, g, vertical, forward);
#define ADC_SCAN(V, F) k_scanline<NV, V, F, WTA><<<grid, block, lds, s>>>(src, dst, rec, disp, g.W, g.H, g.dmin, g.D, g.Dp, forward, sp)
    if (vertical) { if (full) ADC_SCAN(1, true); else ADC_SCAN(1, false); }
    else { if (full) ADC_SCAN(0, true); else ADC_SCAN(0, false); }
#undef ADC_SCAN
}

// The dynamic-LDS opt-in (up to 96 KB: four per-wave rings of a workgroup) of every instantiation the launchers
// can pick, on the current device; done by every handle for its own device when it is initialised.
template <int NV>
static hipError_t setup_nv() {
    hipError_t e = hipSuccess;
    auto set = [&e](const void *f) {
        if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    };
#define ADC_ATTR4(K, ...) \
    set(reinterpret_cast<const void *>(&K<NV, 0, __VA_ARGS__>)); set(reinterpret_cast<const void *>(&K<NV, 1, __VA_ARGS__>))
    ADC_ATTR4(k_scanline, true, true); ADC_ATTR4(k_scanline, true, false);
    ADC_ATTR4(k_scanline, false, true); ADC_ATTR4(k_scanline, false, false);
    if constexpr (NV <= 8) { ADC_ATTR4(k_scanline_lean, true); ADC_ATTR4(k_scanline_lean, false); }
#undef ADC_ATTR4
    return e;
}
hipError_t scanline_setup_device() {
    hipError_t e = setup_nv<1>();
    if (e == hipSuccess) e = setup_nv<2>();
    if (e == hipSuccess) e = setup_nv<4>();
    if (e == hipSuccess) e = setup_nv<8>();
    return e;
}

void launch_scanline_pass(const float *src, float *dst, const Planes &p, const Geom &g,
                          const ScanParams &sp, int vertical, int forward, float *disp_wta, hipStream_t s) {
    // How a pass is spread over the chip (Tuning::scan_split: 0 one wave per scanline always; 1 automatic; 2 two
    // waves per scanline wherever the kernel allows; Tuning::scan_hybrid = 0 takes the mixed launch out of 1).
    // Automatic, horizontal passes only -- a vertical pass has a scanline per column, plenty for the SIMDs:
    //   rows <= SIMDs                 one wave per row (the pass then runs at the HBM ceiling);
    //   SIMDs < rows <= 1.5 x SIMDs   mixed launch: one wave per row for `rows_main` rows, four light waves
    //                                 for each remaining row (1080 rows on 1024 SIMDs: 3.77 -> 3.45 ms for
    //                                 the four passes); rows_main = max(3/4 SIMDs, 2 rows - 3/2 SIMDs), the
    //                                 best of the values measured at 1080, 1200 and 1440 rows;
    //   more rows, or D = 512         two waves per row.
    const int split_env = g.tune.scan_split;
    const bool split_ok = !disp_wta && g.Dp == g.D && (g.NV == 4 || g.NV == 8) && g.D == 64 * g.NV;
    bool split = split_ok && split_env >= 2;
    if (split_ok && split_env == 1 && !vertical) {
        const int simds = g.simds;   // of the handle's own device
        split = g.H > simds;
        if (split && g.NV == 4 && simds >= 16 && g.H <= simds + simds / 2 && g.tune.scan_hybrid) {
            typedef ScanCfg<4, false, HYB_RS> C;
            int rows_main = std::max(simds / 4 * 3, 2 * g.H - simds / 2 * 3);
            if (g.tune.hyb_main >= 0) rows_main = g.tune.hyb_main;
            rows_main = std::min(std::min(rows_main, simds), g.H - 1) & ~3;
            typedef ScanCfgSplit<4, 1, false> CL;
            const int wgs = rows_main / 4 + (g.H - rows_main);
            const int lean_main = wgs <= simds / 2 && g.tune.hyb_lean;   // two workgroups per CU
            const size_t lds_main = lean_main ? (size_t)4 * CL::RS * CL::SLOT_B : (size_t)4 * C::RS * C::SLOT_B;
            const size_t lds_rest = scan_split_lds<1, 4>();
            k_scanline_hybrid<<<dim3(rows_main / 4 + (g.H - rows_main)), dim3(256), std::max(lds_main, lds_rest), s>>>(
                src, dst, rec_source(p, g, 0, forward), g.W, g.H, g.Dp, forward, sp, rows_main, lean_main);
            return;
        }
    }
    if (split) {
        if (g.NV == 4) launch_split<2>(src, dst, p, g, sp, vertical, forward, s);
        else launch_split<4>(src, dst, p, g, sp, vertical, forward, s);
        return;
    }
    if (g.Dp == g.D && g.D == 64 * g.NV && g.NV <= 8 && g.tune.scan_lean) {
        const int nlines = vertical ? g.W : g.H;
        const RecSrc rec = rec_source(p, g, vertical, forward);
        dim3 grid((nlines + 3) / 4), block(256);
#define ADC_LEAN_K(N, V, T)                                                                                                 \
    do {                                                                                                                    \
        typedef ScanCfgSplit<N, 1, T> CL;                                                                                   \
        const size_t lds = (size_t)4 * CL::RS * CL::SLOT_B;                                                                 \
        k_scanline_lean<N, V, T><<<grid, block, lds, s>>>(src, dst, rec, disp_wta, g.W, g.H, g.dmin, g.Dp, forward, sp);    \
    } while (0)
#define ADC_LEAN(N)                                                                                                         \
    do {                                                                                                                    \
        if (vertical) { if (disp_wta) ADC_LEAN_K(N, 1, true); else ADC_LEAN_K(N, 1, false); }                               \
        else { if (disp_wta) ADC_LEAN_K(N, 0, true); else ADC_LEAN_K(N, 0, false); }                                        \
    } while (0)
        switch (g.NV) {
            case 1: ADC_LEAN(1); break;
            case 2: ADC_LEAN(2); break;
            case 4: ADC_LEAN(4); break;
            default: ADC_LEAN(8); break;
        }
#undef ADC_LEAN_K
#undef ADC_LEAN
        return;
    }
#define ADC_NV(N)                                                                    \
    if (disp_wta) launch_nv<N, true>(src, dst, disp_wta, p, g, sp, vertical, forward, s); \
    else launch_nv<N, false>(src, dst, nullptr, p, g, sp, vertical, forward, s)
    switch (g.NV) {
        case 1: ADC_NV(1); break;
        case 2: ADC_NV(2); break;
        case 4: ADC_NV(4); break;
        default: ADC_NV(8); break;   // (adc_initialize admits at most 512 disparities: NV <= 8)
    }
#undef ADC_NV
}

}  // namespace adc
