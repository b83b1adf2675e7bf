// Internal declarations shared by the HIP translation units of libadcensus_hip.so.
// gfx950 only: wave = 64 lanes everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define ADC_LARGE_F 99999.0f  // Large_Float of the reference (adcensus_types.h:54)
#define ADC_WAVE 64

namespace adc {

// Development switches (kernel variants the tests and the A/B tools force). A handle takes them from the
// environment ONCE, in adc_create, and only when ADC_DEBUG_ENV is set to a non-zero value; otherwise these defaults
// hold and the launch path reads no environment variable.
struct Tuning {
    int agg_variant = 0;    // ADC_AGG_VARIANT   1: direct-global aggregation kernel
    int agg_fuse = 1;       // ADC_AGG_FUSE      0: eight single aggregation passes
    int agg_pair = 1;       // ADC_AGG_PAIR      0: one position per wave (ring kernel)
    int cost_fuse = 1;      // ADC_COST_FUSE     0: the initial costs as a kernel of their own
    int agg_depth = 6;      // ADC_AGG_DEPTH     fill look-ahead of the ring kernel
    int agg_mode = 1;       // ADC_AGG_MODE      0 / 1: how the vector form of the fused aggregation launch fills its first ring (see k_agg_fused); 2: the matrix-pipe form where it applies (k_agg_fused_q; bit-identical, measured slower), else 1
    int agg_desc = 1;       // ADC_AGG_DESC      0: the single-pass pair kernel works its windows out from the arms even where pair descriptors exist
    int agg_split = 1;      // ADC_AGG_SPLIT     0: never cut the lines of a fused aggregation launch's last round over the CUs into pieces; n >= 2: every line in n pieces
    int wta_fused = -1;     // ADC_WTA_FUSED     0 / 1: left WTA as its own kernel / inside the last scanline pass
    int wta_scalar = 0;     // ADC_WTA_SCALAR    1: right WTA without 16-byte loads
    int interp_double = 0;  // ADC_INTERP_DOUBLE 1: interpolation rays from the double tables
    int scan_split = 1;     // ADC_SCAN_SPLIT    0 / 1 / 2: waves per scanline (never two / automatic / two where possible)
    int scan_hybrid = 1;    // ADC_SCAN_HYBRID   0: no mixed launch
    int hyb_main = -1;      // ADC_HYB_MAIN      rows of the mixed launch that get a wave of their own
    int hyb_lean = 1;       // ADC_HYB_LEAN
    int scan_lean = 1;      // ADC_SCAN_LEAN
};

// Frame geometry handed to every launcher. Volumes are f32[H][W][Dp] with Dp >= D the internal
// disparity stride (D rounded up so each lane's disparity group is whole); entries k >= D are padding.
struct Geom {
    int W, H;
    int dmin, dmax;
    int D;   // dmax - dmin
    int Dp;  // internal stride of the disparity axis
    int NV;  // disparities per lane in the wave-per-pixel kernels: 64*NV >= D
    int simds;  // SIMDs of the handle's device (4 per compute unit): how many one-wave scanlines run side by side
    Tuning tune;
};

// Per-pixel 2-D planes living in HBM for one handle.
struct Planes {
    const uint8_t *bgr_l, *bgr_r;   // caller images, u8[H][W][3]
    uint32_t *pix_l, *pix_r;        // packed b | g<<8 | r<<16
    uint8_t *gray_l, *gray_r;
    uint32_t *cen_l, *cen_r;        // 31-bit centre-symmetric census
    uint8_t *step_l, *step_r;       // colour-step flags: bit0 = vs left neighbour, bit1 = vs upper neighbour
    uint32_t *arms;                 // left | right<<8 | top<<16 | bottom<<24
    uint16_t *cnt_h, *cnt_v;
    uint32_t *cnt_h32, *cnt_v32;    // same counts widened to 32 bits (dword scalar loads in the aggregation kernel)
    uint32_t *rec_rows;                 // packed colour-step flag strings of the scanline passes: 3 families x rows x 32 shifted copies
    uint32_t *rec_base;                 // the unshifted strings the copies are made from
    uint4 *pair_h, *pair_v;             // window descriptors of the position pairs of every row / column (fused aggregation)
    uint4 *quad_h, *quad_v;             // the same for groups of four positions (two uint4 each; the matrix-pipe form of the fused launch)
};

// --- adc_pre.hip -------------------------------------------------------------------------------
// (one launch each for both images of the pair)
void launch_pack_gray(const uint8_t *bgr_l, const uint8_t *bgr_r, uint32_t *pix_l, uint32_t *pix_r, uint8_t *gray_l,
                      uint8_t *gray_r, int W, int H, hipStream_t s);
void launch_census(const uint8_t *gray_l, const uint8_t *gray_r, uint32_t *census_l, uint32_t *census_r, int W, int H,
                   hipStream_t s);
void launch_step_flags(const uint32_t *pix_l, const uint32_t *pix_r, uint8_t *flags_l, uint8_t *flags_r, int W, int H,
                       int tso, hipStream_t s);
void launch_arms(const uint32_t *pix, uint32_t *arms, int W, int H, int L1, int L2, int t1, int t2, hipStream_t s);
// cap: the arm length cap the arms were built with (min(cross_L1, 255), not negative)
void launch_sup_counts(const uint32_t *arms, uint16_t *cnt_h, uint16_t *cnt_v, uint32_t *cnt_h32, uint32_t *cnt_v32,
                       int W, int H, int cap, hipStream_t s);
void launch_cost_init(const Planes &p, const Geom &g, const float *lut_ad, const float *lut_cen,
                      float *cost, hipStream_t s);
void launch_census_to_u64(const uint32_t *c32, uint64_t *c64, size_t n, hipStream_t s);

// --- adc_aggregate.hip -------------------------------------------------------------------------
// Allows every aggregation kernel its dynamic LDS on the CURRENT device (more than the default 64 KB for some
// instantiations). Called once per handle from adc_initialize, after hipSetDevice.
hipError_t aggregate_setup_device();
// One directional pass: dst[p][k] = sum over the arm span of src (optionally / cnt[p]).
// vertical = 0: span along x (left/right arms); 1: along y (top/bottom arms). cnt == nullptr: no division.
// variant: 0 = ring kernel (LDS), 1 = direct-global fallback.
void launch_aggregate_pass(const float *src, float *dst, const uint32_t *arms, const uint32_t *cnt,
                           const Geom &g, int vertical, int max_arm, int variant, const uint4 *desc, hipStream_t s);
// The first pass (horizontal, no division) with the initial cost volume computed inside the kernel
// instead of being read: only when aggregate_cost_fusion_supported().
bool aggregate_cost_fusion_supported(const Geom &g, int max_arm);
// lut_ad1[s] = (1.0f - e_ad[s]) + 1.0f, evaluated in fp32 on the host
void launch_aggregate_first_pass_with_cost(const Planes &p, const float *lut_ad1, const float *lut_cen, float *dst,
                                           const Geom &g, int max_arm, const uint4 *desc_h, hipStream_t s);

// A dividing pass and the plain pass that follows it along the same axis (passes 1+2, 3+4, 5+6 of the eight)
// as one launch, the intermediate volume staying in LDS: only when aggregate_pair_fusion_supported().
bool aggregate_pair_fusion_supported(const Geom &g, int max_arm);
// per frame: the window geometry of every pair of neighbouring positions, per axis (Planes::pair_h / pair_v)
size_t pair_desc_count(int lines, int len, int max_arm);   // records of one axis's descriptor plane (rows are padded)
void launch_pair_descriptors(const uint32_t *arms, const uint32_t *cnt_for_h, const uint32_t *cnt_for_v, uint4 *desc_h,
                             uint4 *desc_v, const Geom &g, int max_arm, hipStream_t s);
void launch_aggregate_fused_pair(const float *src, float *dst, const uint4 *desc, const uint4 *qdesc, const Geom &g, int vertical,
                                 int max_arm, hipStream_t s);
// the matrix-pipe form of that launch (groups of four positions): its descriptor planes and when it applies
bool aggregate_quad_fusion_supported(const Geom &g, int max_arm);
size_t quad_desc_count(int lines, int len, int max_arm);   // uint4 elements of one axis's quad descriptor plane
void launch_quad_descriptors(const uint32_t *arms, const uint32_t *cnt_for_h, const uint32_t *cnt_for_v, uint4 *desc_h,
                             uint4 *desc_v, const Geom &g, int max_arm, hipStream_t s);

// --- adc_scanline.hip --------------------------------------------------------------------------
hipError_t scanline_setup_device();   // as aggregate_setup_device, for the scanline kernels
// One directional scanline pass (src -> dst). Ptab = {P1[0..2], P2[0..2]} indexed by how many of
// (d1,d2) reach tso.
struct ScanParams { float P1[3]; float P2[3]; };
int scan_row_dwords(const Geom &g);
size_t scan_rows_total_dwords(const Geom &g);   // Planes::rec_rows
size_t scan_base_total_dwords(const Geom &g);   // Planes::rec_base
void launch_scan_records(const Planes &p, const Geom &g, hipStream_t s);
// disp_wta != nullptr: also write the left-view WTA disparity of every pixel (last pass only).
void launch_scanline_pass(const float *src, float *dst, const Planes &p, const Geom &g,
                          const ScanParams &sp, int vertical, int forward, float *disp_wta, hipStream_t s);

// --- adc_wta.hip -------------------------------------------------------------------------------
void launch_wta_right(const float *cost, float *disp_r, const Geom &g, hipStream_t s);
void launch_wta_left(const float *cost, float *disp_l, const Geom &g, hipStream_t s);   // when not fused into the last scanline pass

// --- adc_refine.hip ----------------------------------------------------------------------------
struct RefineBufs {
    float *disp_a, *disp_b;     // ping-pong disparity maps
    const float *disp_r;
    uint8_t *occ, *mis, *edge;
    uint32_t *list_mis, *list_occ;   // compacted outlier pixel lists
    uint32_t *counters;              // [0] = #mis, [1] = #occ
    float *newval;                   // one pending value per list entry (mis list first, then occ)
    const double *ray_dy, *ray_dx;   // [16][maxlen] products m*sin, m*cos
};
// (launchers that also enqueue memsets return the first failing runtime call's status)
hipError_t launch_outlier(const float *dl, const float *dr, float *out, uint8_t *occ, uint8_t *mis, short *binmap,
                          uint32_t *list_mis, uint32_t *list_occ, uint32_t *counters,
                          const Geom &g, float thres, hipStream_t s);
hipError_t launch_clear_masks(uint8_t *occ, uint8_t *mis, uint32_t *counters, size_t n, hipStream_t s);
// counters: [0] mismatch, [1] occlusion entries found by the outlier check; [2], [3] entries handed to the
// interpolation; [4 + p] entries still open after voting pass p (0..9)
#define ADC_NUM_COUNTERS 16
int vote_tiles_w(int W);
int vote_tiles_h(int H);
// pass: 0..9 within the frame (even = mismatch list, odd = occlusion list); stamp0: the frame's first stamp,
// larger than every stamp of earlier frames; cap: the arm length cap (bounds a region's horizontal extent)
// out_list / out_count (may be null): receive the entries still open after this pass, for the list's next pass
void launch_region_voting_pass(float *disp, short *binmap, uint8_t *mask, const uint32_t *list, const uint32_t *count,
                               uint32_t *out_list, uint32_t *out_count, float *newval, const uint32_t *arms, uint32_t *tile_stamp, uint32_t stamp0, int pass,
                               int cap, const Geom &g, int tau_s, float tau_h, hipStream_t s);
hipError_t launch_interpolate(float *disp, const uint8_t *occ, const uint8_t *mis,
                        const uint32_t *list_mis, const uint32_t *list_occ, uint32_t *list2_mis, uint32_t *list2_occ,
                        uint32_t *counters, float *newval, const uint32_t *pix_l, const double *ray_dy,
                        const double *ray_dx, const uint32_t *ray_off /* may be null */, const Geom &g, int maxlen,
                        hipStream_t s);
void launch_edge_detect(const float *disp, uint8_t *edge, int W, int H, float thres, hipStream_t s);
void launch_edge_adjust(const float *disp_in, float *disp_out, const float *cost, const uint8_t *edge,
                        const Geom &g, hipStream_t s);
void launch_median3(const float *in, float *out, int W, int H, hipStream_t s);


#ifdef __HIPCC__
// ---- wave-wide reductions shared by the scanline and WTA kernels ----------------------------------------
// min over the 64 lanes; every lane returns the result. Six DPP steps with the lane shuffle folded into
// the v_min itself (a lane without a source keeps its value); written out because the compiler emits a
// copy, a move-DPP, a canonicalising max and the min per step -- on the serial critical path of every
// path pixel. A VGPR written by a VALU op needs two wait states before a DPP op reads it.
__device__ __forceinline__ float wave_min(float v) {
    asm volatile(
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"   // lane 15 of each row: row minimum
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xf bank_mask:0xf\n\t" // rows 1..3 fold in the previous row
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xf bank_mask:0xf\n\t" // lane 63: wave minimum
        "s_nop 0"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ int wave_min_i32(int v) {
    asm volatile(
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

#endif

}  // namespace adc
