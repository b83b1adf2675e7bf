// C-ABI of libadcensus_hip.so: handle life cycle, the match pipeline (all device-resident, one
// stream, no allocation after adc_initialize), taps, profiling and the frame-sharded batch call.
// Mirrors ADCensusStereo (ADC/ADCensusStereo.h:9-90, ADC/ADCensusStereo.cu:16-157;
// ADC/ = final_version_project/Stereo_Matching/ADCensus_CUDA/).
#include "adc_common.h"
#include "../../include/adcensus.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace adc;

namespace {
thread_local std::string g_thread_error;
enum { N_AGG_PASSES = 8, N_EVENTS = ADC_STAGE_COUNT_ };  // stage boundaries: PRE|COST|AGGR|SCAN|WTA|REFINE|end
enum { N_KEV = ADC_KERNEL_COUNT_ + 1 };                  // kernel k runs between kernel events k and k + 1
}

// Device memory of the entry point that allocates nothing (adc_match_device_static): one array per GPU the
// library's code object is loaded on, held until the process ends. Host code never uses this name as an
// address; it asks for the current device's copy with hipGetSymbolAddress (static_arena below).
static __device__ __align__(256) unsigned char scratch[ADC_STATIC_SCRATCH_BYTES];

struct adc_handle {
    int device = 0;
    bool initialized = false;
    bool debug_taps = false;
    bool profiling = false;
    bool have_times = false;
    int agg_launches = 8;  // aggregation launches of the last match: 5 with fused same-axis pairs, else 8
    Tuning tune{};                     // development switches, read in adc_create (see Tuning)
    adc_option opt{};
    Geom g{};
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void *> dev_allocs;
    std::vector<void *> host_allocs;
    std::vector<hipEvent_t> events;
    std::vector<hipEvent_t> kevents;   // around every volume kernel (profiling only)
    hipEvent_t xfer_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // upload start/end, download start/end of adc_match
    bool have_xfer = false;
    float host_ms[2] = {0, 0};         // adc_match: staging copies in / out on the host (0 for pinned callers)
    enum { N_PIN_CACHE = 6 };
    const void *pin_ptr[N_PIN_CACHE] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // is_pinned_host's memory
    bool pin_yes[N_PIN_CACHE] = {false, false, false, false, false, false};
    int pin_next = 0;
    // arena mode: every device buffer is carved from [arena, arena + arena_size) instead of hipMalloc; no stream,
    // event or pinned host buffer is created (the caller's stream is used, profiling and adc_match are unavailable)
    unsigned char *arena = nullptr;
    size_t arena_size = 0, arena_used = 0;

    // images
    uint8_t *d_bgr_l = nullptr, *d_bgr_r = nullptr;
    uint8_t *h_bgr_l = nullptr, *h_bgr_r = nullptr;  // pinned staging
    float *h_disp = nullptr;
    Planes pl{};
    // volumes: B holds the initial cost, then every even result; A is the other ping-pong buffer
    float *vol_a = nullptr, *vol_b = nullptr;
    float *vol_final = nullptr, *vol_aux = nullptr;   // which of the two holds the final scanline volume / the third pass's output
    float *snap_init = nullptr, *snap_aggr = nullptr;
    // tables
    float *lut_ad = nullptr, *lut_cen = nullptr, *lut_ad1 = nullptr;   // e_ad, e_cen, (1 - e_ad) + 1
    double *ray_dy = nullptr, *ray_dx = nullptr;
    uint32_t *ray_off = nullptr;   // rounded (dy, dx) offsets, or null when some entry sits too close to a rounding tie
    int ray_len = 0;
    ScanParams sp{};
    // disparity maps and masks
    float *disp_a = nullptr, *disp_b = nullptr, *disp_r = nullptr, *disp_final = nullptr;
    float *cur_disp = nullptr;        // which of disp_a/disp_b holds the map after edge adjustment
    uint8_t *occ = nullptr, *mis = nullptr, *edge = nullptr;
    uint32_t *list_mis = nullptr, *list_occ = nullptr, *counters = nullptr;
    uint32_t *list2_mis = nullptr, *list2_occ = nullptr;   // the entries still unfilled after voting
    float *newval = nullptr;
    uint32_t *tile_stamp = nullptr; // per 8x8 tile: stamp of the last voting pass that filled a pixel in it
    uint32_t vote_stamp = 16;       // first stamp of the next frame (stamps only grow, the array is never cleared)
    // second set of staging buffers of the pipelined batch path (allocated on first use)
    uint8_t *pipe_h_l = nullptr, *pipe_h_r = nullptr, *pipe_d_l = nullptr, *pipe_d_r = nullptr;
    float *pipe_h_o = nullptr, *pipe_d_o[2] = {nullptr, nullptr};
    uint32_t *list3_mis = nullptr, *list3_occ = nullptr;   // with list2_*: ping-pong lists of the entries still open between voting passes
    short *binmap = nullptr;       // histogram bin of every pixel of the current disparity map (voting), -1 = none
    // debug snapshots
    float *snap_wta_l = nullptr, *snap_outlier = nullptr, *snap_voting = nullptr, *snap_interp = nullptr;
    uint8_t *snap_occ = nullptr, *snap_mis = nullptr;
    uint64_t *tap_u64 = nullptr;      // census widening scratch
    float times[ADC_STAGE_COUNT_] = {0};

    int fail(int code, const std::string &msg) {
        err = msg;
        return code;
    }
    int fail_hip(const char *what, hipError_t e) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return ADC_ERR_HIP;
    }
};

#define HIPCHK(h, call)                                       \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return (h)->fail_hip(#call, e_); \
    } while (0)

namespace {

template <typename T>
int dev_alloc(adc_handle *h, T **p, size_t count) {
    void *q = nullptr;
    if (h->arena) {
        const size_t bytes = (count * sizeof(T) + 256 + 255) & ~(size_t)255;   // same slack as below, 256-byte aligned
        if (bytes > h->arena_size - h->arena_used) {
            h->arena_used = h->arena_size + 1;       // remember the overflow; the caller reports the size needed
            h->err = "frame does not fit the static scratch";
            return ADC_ERR_UNSUPPORTED;
        }
        *p = reinterpret_cast<T *>(h->arena + h->arena_used);
        h->arena_used += bytes;
        return ADC_OK;
    }
    hipError_t e = hipMalloc(&q, count * sizeof(T) + 256);
    if (e != hipSuccess) {
        h->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? ADC_ERR_NOMEM : ADC_ERR_HIP;
    }
    h->dev_allocs.push_back(q);
    *p = static_cast<T *>(q);
    return ADC_OK;
}
template <typename T>
int host_alloc(adc_handle *h, T **p, size_t count) {
    void *q = nullptr;
    hipError_t e = hipHostMalloc(&q, count * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) {
        h->err = std::string("hipHostMalloc: ") + hipGetErrorString(e);
        return ADC_ERR_NOMEM;
    }
    h->host_allocs.push_back(q);
    *p = static_cast<T *>(q);
    return ADC_OK;
}

void release(adc_handle *h) {
    if (h->stream) hipStreamSynchronize(h->stream);
    for (void *p : h->dev_allocs) hipFree(p);
    for (void *p : h->host_allocs) hipHostFree(p);
    for (hipEvent_t e : h->events) hipEventDestroy(e);
    for (hipEvent_t e : h->kevents) hipEventDestroy(e);
    for (hipEvent_t &e : h->xfer_ev) { if (e) hipEventDestroy(e); e = nullptr; }
    h->dev_allocs.clear();
    h->host_allocs.clear();
    h->events.clear();
    h->kevents.clear();
    h->have_xfer = false;
    if (h->stream) { hipStreamDestroy(h->stream); h->stream = nullptr; }
    h->snap_init = h->snap_aggr = nullptr;
    h->pipe_d_o[0] = h->pipe_d_o[1] = nullptr;
    h->initialized = false;
    h->have_times = false;
}

int alloc_debug(adc_handle *h) {
    if (h->snap_init) return ADC_OK;
    const size_t npx = (size_t)h->g.W * h->g.H, nvol = npx * h->g.Dp;
    int rc;
    if ((rc = dev_alloc(h, &h->snap_init, nvol))) return rc;
    if ((rc = dev_alloc(h, &h->snap_aggr, nvol))) return rc;
    if ((rc = dev_alloc(h, &h->snap_wta_l, npx))) return rc;
    if ((rc = dev_alloc(h, &h->snap_outlier, npx))) return rc;
    if ((rc = dev_alloc(h, &h->snap_voting, npx))) return rc;
    if ((rc = dev_alloc(h, &h->snap_interp, npx))) return rc;
    if ((rc = dev_alloc(h, &h->snap_occ, npx))) return rc;
    if ((rc = dev_alloc(h, &h->snap_mis, npx))) return rc;
    return ADC_OK;
}

// s0: the stream the set-up work (memsets, table uploads) runs on; the handle's own stream unless the handle
// lives in the static scratch, where it is the caller's.
int do_initialize(adc_handle *h, int32_t W, int32_t H, const adc_option *opt, hipStream_t s0 = nullptr) {
    if (!opt) return h->fail(ADC_ERR_ARG, "adc_initialize: option is null");
    // ADCensusStereo::Initialize rejects w,h <= 0 and an empty disparity range (ADC/ADCensusStereo.cu:26-35)
    if (W <= 0 || H <= 0) return h->fail(ADC_ERR_ARG, "adc_initialize: width and height must be positive");
    const long long D64 = (long long)opt->max_disparity - (long long)opt->min_disparity;
    if (D64 <= 0) return h->fail(ADC_ERR_ARG, "adc_initialize: max_disparity must exceed min_disparity");
    // (the widest layout the scanline kernels are built and tested for: 8 disparities per lane)
    if (D64 > 512) return h->fail(ADC_ERR_UNSUPPORTED, "adc_initialize: more than 512 disparities not supported");
    if (opt->lambda_ad <= 0 || opt->lambda_census <= 0)
        return h->fail(ADC_ERR_ARG, "adc_initialize: lambda_ad and lambda_census must be positive");
    if ((unsigned long long)W * (unsigned long long)H >= (1ull << 31))
        return h->fail(ADC_ERR_UNSUPPORTED, "adc_initialize: image larger than 2^31 pixels");
    if (H > 65535 || W > 65535)
        return h->fail(ADC_ERR_UNSUPPORTED, "adc_initialize: image side larger than 65535");

    const bool in_arena = h->arena != nullptr;
    HIPCHK(h, hipSetDevice(h->device));
    if (!in_arena) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        s0 = h->stream;
    }
    // kernel attributes belong to the device they were set on: every handle opts its own device in
    HIPCHK(h, aggregate_setup_device());
    HIPCHK(h, scanline_setup_device());
    h->opt = *opt;
    Geom &g = h->g;
    g.tune = h->tune;
    g.W = W; g.H = H; g.dmin = opt->min_disparity; g.dmax = opt->max_disparity; g.D = (int)D64;
    g.NV = 1;
    while (64 * g.NV < g.D) g.NV *= 2;
    g.Dp = ((g.D + g.NV - 1) / g.NV) * g.NV;
    {
        int cus = 0;
        HIPCHK(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
        g.simds = 4 * cus;
    }

    const size_t npx = (size_t)W * H, nvol = npx * g.Dp;
    int rc;
#define A(p, n) if ((rc = dev_alloc(h, &(p), (n)))) return rc
    if (!in_arena) { A(h->d_bgr_l, npx * 3); A(h->d_bgr_r, npx * 3); }   // adc_match's upload targets
    A(h->pl.pix_l, npx); A(h->pl.pix_r, npx);
    A(h->pl.gray_l, npx); A(h->pl.gray_r, npx);
    A(h->pl.cen_l, npx); A(h->pl.cen_r, npx);
    A(h->pl.step_l, npx); A(h->pl.step_r, npx);
    A(h->pl.arms, npx); A(h->pl.cnt_h, npx); A(h->pl.cnt_v, npx); A(h->pl.cnt_h32, npx); A(h->pl.cnt_v32, npx);
    A(h->pl.rec_rows, scan_rows_total_dwords(g)); A(h->pl.rec_base, scan_base_total_dwords(g));
    const int arm_cap = opt->cross_L1 < 0 ? 0 : (opt->cross_L1 < 255 ? opt->cross_L1 : 255);
    const size_t n_pair_h = pair_desc_count(H, W, arm_cap), n_pair_v = pair_desc_count(W, H, arm_cap);
    A(h->pl.pair_h, n_pair_h); A(h->pl.pair_v, n_pair_v);
    // (the quad descriptor planes of the matrix-pipe form only where that form is selected)
    const bool want_quads = h->tune.agg_mode == 2;
    const size_t n_quad_h = want_quads ? quad_desc_count(H, W, arm_cap) : 0, n_quad_v = want_quads ? quad_desc_count(W, H, arm_cap) : 0;
    if (want_quads) { A(h->pl.quad_h, n_quad_h); A(h->pl.quad_v, n_quad_v); }
    A(h->vol_a, nvol); A(h->vol_b, nvol);
    A(h->lut_ad, 768); A(h->lut_cen, 68); A(h->lut_ad1, 768);
    A(h->disp_a, npx); A(h->disp_b, npx); A(h->disp_r, npx); A(h->disp_final, npx);
    A(h->occ, npx); A(h->mis, npx); A(h->edge, npx);
    A(h->list_mis, npx); A(h->list_occ, npx); A(h->counters, ADC_NUM_COUNTERS);
    A(h->list2_mis, npx); A(h->list2_occ, npx);
    A(h->list3_mis, npx); A(h->list3_occ, npx);
    A(h->newval, 2 * npx);
    A(h->binmap, npx);
    A(h->tile_stamp, (size_t)vote_tiles_w(W) * vote_tiles_h(H) + 1);   // + a word that stays zero (read by lanes with no tile)
    if (!in_arena) A(h->tap_u64, npx);
    // ray tables of the 16-direction search: m*sin, m*cos (double) and, when usable, their rounded offsets
    const int maxlen = std::abs(opt->max_disparity) * 1;  // MaxSearchLengthTimes = 1 (ADC/config.h:24)
    h->ray_len = maxlen > 1 ? maxlen : 1;
    const size_t nray = (size_t)16 * h->ray_len;
    A(h->ray_dy, nray); A(h->ray_dx, nray); A(h->ray_off, nray);
#undef A
    if (!in_arena) {
        if ((rc = host_alloc(h, &h->h_bgr_l, npx * 3))) return rc;
        if ((rc = host_alloc(h, &h->h_bgr_r, npx * 3))) return rc;
        if ((rc = host_alloc(h, &h->h_disp, npx))) return rc;
    }
    // padding columns of the volumes are never produced by the cost kernel; give them a defined value once
    HIPCHK(h, hipMemsetAsync(h->vol_a, 0, nvol * sizeof(float), s0));
    HIPCHK(h, hipMemsetAsync(h->vol_b, 0, nvol * sizeof(float), s0));
    HIPCHK(h, hipMemsetAsync(h->tile_stamp, 0, ((size_t)vote_tiles_w(W) * vote_tiles_h(H) + 1) * sizeof(uint32_t), s0));
    h->vote_stamp = 16;
    // the padding records in front of and behind every line of the pair descriptors stay zero for good
    HIPCHK(h, hipMemsetAsync(h->pl.pair_h, 0, n_pair_h * sizeof(uint4), s0));
    HIPCHK(h, hipMemsetAsync(h->pl.pair_v, 0, n_pair_v * sizeof(uint4), s0));
    if (want_quads) {
        HIPCHK(h, hipMemsetAsync(h->pl.quad_h, 0, n_quad_h * sizeof(uint4), s0));
        HIPCHK(h, hipMemsetAsync(h->pl.quad_v, 0, n_quad_v * sizeof(uint4), s0));
    }

    // exp tables of the initial cost (ADC/cost_computor.cu:186,200-203): entry = expf of the float
    // argument, taken as the correctly rounded (float)exp((double)arg).
    std::vector<float> ad(768, 0.0f), cen(68, 0.0f), ad1(768, 0.0f);
    for (int s = 0; s < 766; s++) {
        const float a = (float)s / 3.0f;
        const float arg = -a / (float)opt->lambda_ad;
        ad[s] = (float)std::exp((double)arg);
        // the first two operations of ((1 - e1) + 1) - e2 (ADC/cost_computor.cu:203), in fp32 as the kernels did them
        // per element (this file is compiled without FMA contraction; volatile keeps each rounding step)
        volatile float t = 1.0f - ad[s];
        t = t + 1.0f;
        ad1[s] = t;
    }
    for (int k = 0; k < 65; k++) {
        const float arg = -(float)k / (float)opt->lambda_census;
        cen[k] = (float)std::exp((double)arg);
    }
    HIPCHK(h, hipMemcpyAsync(h->lut_ad, ad.data(), 768 * sizeof(float), hipMemcpyHostToDevice, s0));
    HIPCHK(h, hipMemcpyAsync(h->lut_cen, cen.data(), 68 * sizeof(float), hipMemcpyHostToDevice, s0));
    HIPCHK(h, hipMemcpyAsync(h->lut_ad1, ad1.data(), 768 * sizeof(float), hipMemcpyHostToDevice, s0));
    // ray tables (ADC/multistep_refiner.cu:561-583, :642): ang += PI/16 with PI = 3.1415926f, accumulated in
    // double; entries m*sin(ang), m*cos(ang).
    std::vector<double> dy(nray, 0.0), dx(nray, 0.0);
    {
        double ang = 0.0;
        const float step = 3.1415926f / 16;
        for (int s = 0; s < 16; s++) {
            const double sa = std::sin(ang), ca = std::cos(ang);
            for (int m = 0; m < h->ray_len; m++) {
                dy[(size_t)s * h->ray_len + m] = (double)m * sa;
                dx[(size_t)s * h->ray_len + m] = (double)m * ca;
            }
            ang += step;
        }
    }
    HIPCHK(h, hipMemcpyAsync(h->ray_dy, dy.data(), nray * sizeof(double), hipMemcpyHostToDevice, s0));
    HIPCHK(h, hipMemcpyAsync(h->ray_dx, dx.data(), nray * sizeof(double), hipMemcpyHostToDevice, s0));
    // round((double)y + v) == y + round(v) for every integer |y| < 2^16 unless v is within a few ulps
    // (of 2^17) of a half-integer; with a margin of 1e-6 the integer offsets can replace the doubles.
    bool int_ok = h->ray_len <= 32000 && !h->tune.interp_double;
    std::vector<uint32_t> off(nray);
    for (size_t i = 0; int_ok && i < nray; i++) {
        const double ry = std::round(dy[i]), rx = std::round(dx[i]);
        if (std::fabs(std::fabs(dy[i] - ry) - 0.5) < 1e-6 || std::fabs(std::fabs(dx[i] - rx) - 0.5) < 1e-6) int_ok = false;
        off[i] = (uint32_t)(uint16_t)(int16_t)ry << 16 | (uint32_t)(uint16_t)(int16_t)rx;
    }
    if (int_ok) HIPCHK(h, hipMemcpyAsync(h->ray_off, off.data(), nray * sizeof(uint32_t), hipMemcpyHostToDevice, s0));
    else h->ray_off = nullptr;   // the kernels then use the double tables
    // the host tables above go out of scope: wait for the uploads
    HIPCHK(h, hipStreamSynchronize(s0));
    // penalty table (ADC/scanline_optimizer.cu:97-109)
    h->sp.P1[0] = opt->so_p1; h->sp.P1[1] = opt->so_p1 / 4; h->sp.P1[2] = opt->so_p1 / 10;
    h->sp.P2[0] = opt->so_p2; h->sp.P2[1] = opt->so_p2 / 4; h->sp.P2[2] = opt->so_p2 / 10;

    if (!in_arena) {
        h->events.resize(N_EVENTS);
        for (auto &e : h->events) HIPCHK(h, hipEventCreate(&e));
        h->kevents.resize(N_KEV);
        for (auto &e : h->kevents) HIPCHK(h, hipEventCreate(&e));
        for (auto &e : h->xfer_ev) HIPCHK(h, hipEventCreate(&e));
        if (h->debug_taps && (rc = alloc_debug(h))) return rc;
    }
    h->pl.bgr_l = h->d_bgr_l;
    h->pl.bgr_r = h->d_bgr_r;
    h->cur_disp = h->disp_a;
    h->vol_final = h->vol_b;
    h->vol_aux = h->vol_a;
    h->initialized = true;
    h->err.clear();
    return ADC_OK;
}

// The development switches of a new handle: the defaults, unless ADC_DEBUG_ENV is set to a non-zero value -- then
// every ADC_* variable named in Tuning overrides its field. Nothing else in the library reads the environment.
Tuning tuning_from_env() {
    Tuning t;
    const char *dbg = std::getenv("ADC_DEBUG_ENV");
    if (!dbg || std::atoi(dbg) == 0) return t;
    auto get = [](const char *name, int &field) {
        const char *v = std::getenv(name);
        if (v) field = std::atoi(v);
    };
    get("ADC_AGG_VARIANT", t.agg_variant); get("ADC_AGG_FUSE", t.agg_fuse); get("ADC_AGG_PAIR", t.agg_pair);
    get("ADC_COST_FUSE", t.cost_fuse); get("ADC_AGG_DEPTH", t.agg_depth); get("ADC_AGG_MODE", t.agg_mode);
    get("ADC_AGG_SPLIT", t.agg_split); get("ADC_AGG_DESC", t.agg_desc);
    get("ADC_WTA_FUSED", t.wta_fused); get("ADC_WTA_SCALAR", t.wta_scalar); get("ADC_INTERP_DOUBLE", t.interp_double);
    get("ADC_SCAN_SPLIT", t.scan_split); get("ADC_SCAN_HYBRID", t.scan_hybrid); get("ADC_HYB_MAIN", t.hyb_main);
    get("ADC_HYB_LEAN", t.hyb_lean); get("ADC_SCAN_LEAN", t.scan_lean);
    return t;
}

inline int max_arm_of(const adc_option &o) {
    int cap = o.cross_L1 < 255 ? o.cross_L1 : 255;
    return cap < 0 ? 0 : cap;
}

// Enqueue the whole pipeline on stream s. d_out receives the final map (device pointer).
// Every runtime call is checked; kernel launch failures are collected by the hipGetLastError at the end.
int enqueue_match(adc_handle *h, const uint8_t *d_l, const uint8_t *d_r, float *d_out, hipStream_t s) {
    const Geom &g = h->g;
    const adc_option &o = h->opt;
    const size_t npx = (size_t)g.W * g.H, nvol = npx * g.Dp;
    Planes pl = h->pl;
    pl.bgr_l = d_l;
    pl.bgr_r = d_r;
    const bool prof = h->profiling && !h->events.empty();
#define EQ(call)                                              \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return h->fail_hip(#call, e_);  \
    } while (0)
#define MARK(i) do { if (prof) EQ(hipEventRecord(h->events[i], s)); } while (0)
#define KMARK(i) do { if (prof) EQ(hipEventRecord(h->kevents[i], s)); } while (0)
#define SNAP(dst, src, bytes) do { if (h->debug_taps) EQ(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s)); } while (0)

    MARK(ADC_STAGE_PRE);
    launch_pack_gray(d_l, d_r, pl.pix_l, pl.pix_r, pl.gray_l, pl.gray_r, g.W, g.H, s);
    launch_census(pl.gray_l, pl.gray_r, pl.cen_l, pl.cen_r, g.W, g.H, s);
    launch_step_flags(pl.pix_l, pl.pix_r, pl.step_l, pl.step_r, g.W, g.H, o.so_tso, s);
    launch_arms(pl.pix_l, pl.arms, g.W, g.H, o.cross_L1, o.cross_L2, o.cross_t1, o.cross_t2, s);
    const int cap = max_arm_of(o);
    launch_sup_counts(pl.arms, pl.cnt_h, pl.cnt_v, pl.cnt_h32, pl.cnt_v32, g.W, g.H, cap, s);
    launch_scan_records(pl, g, s);
    const bool fuse_pairs = h->tune.agg_variant == 0 && aggregate_pair_fusion_supported(g, cap);
    // horizontal dividing passes divide by the vertical-first counts, vertical ones by the horizontal-first counts
    if (fuse_pairs) launch_pair_descriptors(pl.arms, pl.cnt_v32, pl.cnt_h32, pl.pair_h, pl.pair_v, g, cap, s);
    const bool fuse_quads = fuse_pairs && pl.quad_h && aggregate_quad_fusion_supported(g, cap);
    if (fuse_quads) launch_quad_descriptors(pl.arms, pl.cnt_v32, pl.cnt_h32, pl.quad_h, pl.quad_v, g, cap, s);

    // The initial cost volume is computed inside the first aggregation pass when that is possible; it is
    // materialised (ComputeCostCuda as a kernel of its own) when its tap is wanted or the pass cannot hold it.
    const bool fuse_cost = !h->debug_taps && h->tune.agg_variant == 0 && aggregate_cost_fusion_supported(g, cap);
    MARK(ADC_STAGE_COST);
    KMARK(ADC_KERNEL_COST);
    if (!fuse_cost) launch_cost_init(pl, g, h->lut_ad, h->lut_cen, h->vol_b, s);
    SNAP(h->snap_init, h->vol_b, nvol * sizeof(float));

    // CrossAggregator::Aggregate: 4 iterations, horizontal-first on even ones (ADC/cost_aggregator.cu:112,440-454):
    // passes H, V/cntH | V, H/cntV | H, V/cntH | V, H/cntV. `cur` is the volume holding the latest result.
    MARK(ADC_STAGE_AGGR);
    float *cur_v, *oth_v;
    h->agg_launches = fuse_pairs ? 5 : N_AGG_PASSES;
    if (fuse_pairs) {
        // five launches: pass 0 | passes 1+2 | 3+4 | 5+6 | pass 7 -- a dividing pass and the same-axis pass behind
        // it share a launch and the volume between them never leaves the CU
        KMARK(ADC_KERNEL_AGGR0);
        if (fuse_cost) {
            cur_v = h->vol_b; oth_v = h->vol_a;    // nothing lives in either volume yet
            launch_aggregate_first_pass_with_cost(pl, h->lut_ad1, h->lut_cen, cur_v, g, cap, pl.pair_h, s);
        } else {
            cur_v = h->vol_a; oth_v = h->vol_b;
            launch_aggregate_pass(h->vol_b, cur_v, pl.arms, nullptr, g, 0, cap, 0, pl.pair_h, s);
        }
        for (int k = 0; k < 3; k++) {
            const int vertical = (k % 2) == 0;
            KMARK(ADC_KERNEL_AGGR0 + 1 + 2 * k);
            launch_aggregate_fused_pair(cur_v, oth_v, vertical ? pl.pair_v : pl.pair_h,
                                        fuse_quads ? (vertical ? pl.quad_v : pl.quad_h) : nullptr, g, vertical, cap, s);
            KMARK(ADC_KERNEL_AGGR0 + 2 + 2 * k);   // (the pair's second pass has no launch of its own: 0 ms)
            std::swap(cur_v, oth_v);
        }
        KMARK(ADC_KERNEL_AGGR0 + 7);
        launch_aggregate_pass(cur_v, oth_v, pl.arms, pl.cnt_v32, g, 0, cap, 0, pl.pair_h, s);
        std::swap(cur_v, oth_v);
    } else {
        for (int it = 0; it < 4; it++) {
            const bool hfirst = (it % 2) == 0;
            KMARK(ADC_KERNEL_AGGR0 + 2 * it);
            if (it == 0 && fuse_cost)
                launch_aggregate_first_pass_with_cost(pl, h->lut_ad1, h->lut_cen, h->vol_a, g, cap, nullptr, s);
            else
                launch_aggregate_pass(h->vol_b, h->vol_a, pl.arms, nullptr, g, hfirst ? 0 : 1, cap, h->tune.agg_variant, nullptr, s);
            KMARK(ADC_KERNEL_AGGR0 + 2 * it + 1);
            launch_aggregate_pass(h->vol_a, h->vol_b, pl.arms, hfirst ? pl.cnt_h32 : pl.cnt_v32, g, hfirst ? 1 : 0, cap,
                                  h->tune.agg_variant, nullptr, s);
        }
        cur_v = h->vol_b; oth_v = h->vol_a;
    }
    KMARK(ADC_KERNEL_SCAN0);
    SNAP(h->snap_aggr, cur_v, nvol * sizeof(float));

    // ScanlineOptimizer::Optimize pass order and ping-pong (ADC/scanline_optimizer.cu:380-447): the result ends
    // in the volume the aggregation left its result in, the other one holds the third pass's output
    MARK(ADC_STAGE_SCAN);
    launch_scanline_pass(cur_v, oth_v, pl, g, h->sp, 0, 1, nullptr, s);
    KMARK(ADC_KERNEL_SCAN0 + 1);
    launch_scanline_pass(oth_v, cur_v, pl, g, h->sp, 0, 0, nullptr, s);
    KMARK(ADC_KERNEL_SCAN0 + 2);
    launch_scanline_pass(cur_v, oth_v, pl, g, h->sp, 1, 1, nullptr, s);
    KMARK(ADC_KERNEL_SCAN0 + 3);
    // The last pass also produces the left-view WTA map while each pixel's final costs are in registers -- on
    // frames where the pass is bound by HBM traffic and the extra instructions hide (1080p: +0.03 ms against a
    // 0.4 ms volume read). On small frames the pass is a bare serial chain per column and those instructions
    // more than double it (450x375x64: 154 against 66 us), while reading the small volume once more is cheap.
    const int wta_env = h->tune.wta_fused;   // (test hook: 0 / 1 force the separate / fused form)
    const bool wta_fused = wta_env >= 0 ? wta_env != 0 : (long long)g.W * g.Dp >= 150000;
    launch_scanline_pass(oth_v, cur_v, pl, g, h->sp, 1, 0, wta_fused ? h->disp_a : nullptr, s);
    h->vol_final = cur_v;
    h->vol_aux = oth_v;

    MARK(ADC_STAGE_WTA);
    KMARK(ADC_KERNEL_WTA_RIGHT);   // (in front of a separate left WTA: the last scanline pass's interval ends here)
    if (!wta_fused) launch_wta_left(cur_v, h->disp_a, g, s);
    launch_wta_right(cur_v, h->disp_r, g, s);
    KMARK(ADC_KERNEL_COUNT_);
    SNAP(h->snap_wta_l, h->disp_a, npx * sizeof(float));

    // MultiStepRefiner::Refine (ADC/multistep_refiner.cu:283-371)
    MARK(ADC_STAGE_REFINE);
    float *cur = h->disp_a, *oth = h->disp_b;
    if (o.do_lr_check) {
        EQ(launch_outlier(cur, h->disp_r, oth, h->occ, h->mis, h->binmap, h->list_mis, h->list_occ, h->counters, g,
                          o.lrcheck_thres, s));
        std::swap(cur, oth);
    } else {
        EQ(launch_clear_masks(h->occ, h->mis, h->counters, npx, s));
    }
    SNAP(h->snap_outlier, cur, npx * sizeof(float));
    SNAP(h->snap_occ, h->occ, npx);
    SNAP(h->snap_mis, h->mis, npx);
    if (o.do_filling) {  // drives both voting and interpolation (ADC/ADCensusStereo.cu:195-196)
        if (h->vote_stamp > 0xffffff00u) {   // stamps would wrap: start over
            EQ(hipMemsetAsync(h->tile_stamp, 0, ((size_t)vote_tiles_w(g.W) * vote_tiles_h(g.H) + 1) * sizeof(uint32_t), s));
            h->vote_stamp = 16;
        }
        const uint32_t stamp0 = h->vote_stamp;
        h->vote_stamp += 16;
        // Each pass hands its still-open entries to the list's next pass: full list -> list2 -> list3 -> list2 -> list3.
        const uint32_t *in_mis = h->list_mis, *in_occ = h->list_occ;
        uint32_t *n_mis = h->counters + 0, *n_occ = h->counters + 1;
        for (int it = 0; it < 5; it++) {  // IterativeNums (ADC/config.h:46): mismatch pass, then occlusion pass
            const bool last = it == 4;
            uint32_t *out_mis = last ? nullptr : (it % 2 == 0 ? h->list2_mis : h->list3_mis);
            uint32_t *out_occ = last ? nullptr : (it % 2 == 0 ? h->list2_occ : h->list3_occ);
            launch_region_voting_pass(cur, h->binmap, h->mis, in_mis, n_mis, out_mis, h->counters + 4 + 2 * it, h->newval,
                                      pl.arms, h->tile_stamp, stamp0, 2 * it, cap, g, o.irv_ts, o.irv_th, s);
            launch_region_voting_pass(cur, h->binmap, h->occ, in_occ, n_occ, out_occ, h->counters + 5 + 2 * it, h->newval,
                                      pl.arms, h->tile_stamp, stamp0, 2 * it + 1, cap, g, o.irv_ts, o.irv_th, s);
            in_mis = out_mis; in_occ = out_occ;
            n_mis = h->counters + 4 + 2 * it; n_occ = h->counters + 5 + 2 * it;
        }
    }
    SNAP(h->snap_voting, cur, npx * sizeof(float));
    if (o.do_filling) {
        EQ(launch_interpolate(cur, h->occ, h->mis, h->list_mis, h->list_occ, h->list2_mis, h->list2_occ, h->counters, h->newval, pl.pix_l,
                              h->ray_dy, h->ray_dx, h->ray_off, g, h->ray_len, s));
    }
    SNAP(h->snap_interp, cur, npx * sizeof(float));
    if (o.do_discontinuity_adjustment) {
        launch_edge_detect(cur, h->edge, g.W, g.H, 5.0f, s);
        launch_edge_adjust(cur, oth, cur_v, h->edge, g, s);
        std::swap(cur, oth);
    } else {
        EQ(hipMemsetAsync(h->edge, 0, npx, s));
    }
    h->cur_disp = cur;
    launch_median3(cur, h->disp_final, g.W, g.H, s);
    if (d_out && d_out != h->disp_final)
        EQ(hipMemcpyAsync(d_out, h->disp_final, npx * sizeof(float), hipMemcpyDeviceToDevice, s));
    MARK(ADC_STAGE_TOTAL);
    h->have_times = prof;
    EQ(hipGetLastError());
#undef EQ
#undef MARK
#undef KMARK
#undef SNAP
    return ADC_OK;
}

}  // namespace

extern "C" {

void adc_default_option(adc_option *o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->min_disparity = 0;
    o->max_disparity = 64;
    o->lambda_ad = 10;
    o->lambda_census = 30;
    o->cross_L1 = 34;
    o->cross_L2 = 17;
    o->cross_t1 = 20;
    o->cross_t2 = 6;
    o->so_p1 = 1.0f;
    o->so_p2 = 3.0f;
    o->so_tso = 15;
    o->irv_ts = 20;
    o->irv_th = 0.4f;
    o->lrcheck_thres = 1.0f;
    o->do_lr_check = 1;
    o->do_filling = 1;
    o->do_discontinuity_adjustment = 1;
}

adc_handle *adc_create(int device) {
    adc_handle *h = new (std::nothrow) adc_handle();
    if (!h) { g_thread_error = "adc_create: out of host memory"; return nullptr; }
    h->device = device;
    h->tune = tuning_from_env();
    return h;
}

void adc_destroy(adc_handle *h) {
    if (!h) return;
    hipSetDevice(h->device);
    release(h);
    delete h;
}

int adc_initialize(adc_handle *h, int32_t width, int32_t height, const adc_option *opt) {
    if (!h) return ADC_ERR_ARG;
    if (h->initialized || !h->dev_allocs.empty()) { hipSetDevice(h->device); release(h); }
    int rc = do_initialize(h, width, height, opt);
    if (rc != ADC_OK) { std::string keep = h->err; hipSetDevice(h->device); release(h); h->err = keep; }
    return rc;
}

int adc_reset(adc_handle *h, int32_t width, int32_t height, const adc_option *opt) {
    return adc_initialize(h, width, height, opt);
}

int adc_match_device(adc_handle *h, const uint8_t *d_left, const uint8_t *d_right, float *d_disp, void *stream) {
    if (!h) return ADC_ERR_ARG;
    if (!h->initialized) return h->fail(ADC_ERR_NOT_INIT, "adc_match_device: handle not initialised");
    if (!d_left || !d_right || !d_disp) return h->fail(ADC_ERR_ARG, "adc_match_device: null pointer");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    return enqueue_match(h, d_left, d_right, d_disp, s);
}

// true when p is host memory the GPU can copy from / to directly (hipHostMalloc or hipHostRegister). The runtime's
// query takes a noticeable fraction of a millisecond, and callers pass the same few buffers frame after frame: the
// handle remembers the answers for the last pointers it saw. (A remembered "pinned" for memory that has meanwhile
// been freed and handed out again as pageable is harmless: hipMemcpyAsync stages such a copy itself.)
static bool is_pinned_host(adc_handle *h, const void *p) {
    for (int i = 0; i < adc_handle::N_PIN_CACHE; i++)
        if (h->pin_ptr[i] == p) return h->pin_yes[i];
    hipPointerAttribute_t a;
    bool yes = false;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) (void)hipGetLastError();   // an ordinary malloc pointer: not an error of ours
    else yes = a.type == hipMemoryTypeHost;
    h->pin_ptr[h->pin_next] = p;
    h->pin_yes[h->pin_next] = yes;
    h->pin_next = (h->pin_next + 1) % adc_handle::N_PIN_CACHE;
    return yes;
}

int adc_match(adc_handle *h, const uint8_t *left, const uint8_t *right, float *disp) {
    if (!h) return ADC_ERR_ARG;
    if (!h->initialized) return h->fail(ADC_ERR_NOT_INIT, "adc_match: handle not initialised");
    if (h->arena) return h->fail(ADC_ERR_ARG, "adc_match: not available on the static-scratch context");
    if (!left || !right || !disp) return h->fail(ADC_ERR_ARG, "adc_match: null pointer");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t npx = (size_t)h->g.W * h->g.H;
    const bool prof = h->profiling;
    typedef std::chrono::steady_clock clk;
    // Pageable caller memory goes through the handle's pinned staging buffers; pinned caller memory is the
    // DMA source / target itself (no host copy on either side).
    const bool pin_in = is_pinned_host(h, left) && is_pinned_host(h, right), pin_out = is_pinned_host(h, disp);
    const uint8_t *src_l = left, *src_r = right;
    const auto t0 = clk::now();
    if (!pin_in) {
        std::memcpy(h->h_bgr_l, left, npx * 3);
        std::memcpy(h->h_bgr_r, right, npx * 3);
        src_l = h->h_bgr_l; src_r = h->h_bgr_r;
    }
    const auto t1 = clk::now();
    if (prof) HIPCHK(h, hipEventRecord(h->xfer_ev[0], h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_bgr_l, src_l, npx * 3, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_bgr_r, src_r, npx * 3, hipMemcpyHostToDevice, h->stream));
    if (prof) HIPCHK(h, hipEventRecord(h->xfer_ev[1], h->stream));
    int rc = enqueue_match(h, h->d_bgr_l, h->d_bgr_r, h->disp_final, h->stream);
    if (rc != ADC_OK) { hipStreamSynchronize(h->stream); return rc; }
    if (prof) HIPCHK(h, hipEventRecord(h->xfer_ev[2], h->stream));
    HIPCHK(h, hipMemcpyAsync(pin_out ? disp : h->h_disp, h->disp_final, npx * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (prof) HIPCHK(h, hipEventRecord(h->xfer_ev[3], h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const auto t2 = clk::now();
    if (!pin_out) std::memcpy(disp, h->h_disp, npx * sizeof(float));
    const auto t3 = clk::now();
    h->host_ms[0] = pin_in ? 0.0f : std::chrono::duration<float, std::milli>(t1 - t0).count();
    h->host_ms[1] = pin_out ? 0.0f : std::chrono::duration<float, std::milli>(t3 - t2).count();
    h->have_xfer = prof;
    return ADC_OK;
}

int adc_synchronize(adc_handle *h) {
    if (!h) return ADC_ERR_ARG;
    if (!h->initialized) return h->fail(ADC_ERR_NOT_INIT, "adc_synchronize: handle not initialised");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ADC_OK;
}

int adc_set_debug_taps(adc_handle *h, int enable) {
    if (!h) return ADC_ERR_ARG;
    h->debug_taps = enable != 0;
    if (h->debug_taps && h->initialized) {
        HIPCHK(h, hipSetDevice(h->device));
        return alloc_debug(h);
    }
    return ADC_OK;
}

int adc_set_profiling(adc_handle *h, int enable) {
    if (!h) return ADC_ERR_ARG;
    h->profiling = enable != 0;
    return ADC_OK;
}

int adc_stage_times(adc_handle *h, float *ms, int n) {
    if (!h || !ms || n < 0) return ADC_ERR_ARG;
    if (!h->initialized) return h->fail(ADC_ERR_NOT_INIT, "adc_stage_times: handle not initialised");
    if (!h->have_times) return h->fail(ADC_ERR_ARG, "adc_stage_times: no profiled match has run");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventSynchronize(h->events[ADC_STAGE_TOTAL]));
    for (int i = 0; i < ADC_STAGE_TOTAL; i++)
        HIPCHK(h, hipEventElapsedTime(&h->times[i], h->events[i], h->events[i + 1]));
    HIPCHK(h, hipEventElapsedTime(&h->times[ADC_STAGE_TOTAL], h->events[0], h->events[ADC_STAGE_TOTAL]));
    for (int i = 0; i < n && i < ADC_STAGE_COUNT_; i++) ms[i] = h->times[i];
    return ADC_OK;
}

int adc_kernel_times(adc_handle *h, float *ms, int n) {
    if (!h || !ms || n < 0) return ADC_ERR_ARG;
    if (!h->initialized) return h->fail(ADC_ERR_NOT_INIT, "adc_kernel_times: handle not initialised");
    if (!h->have_times) return h->fail(ADC_ERR_ARG, "adc_kernel_times: no profiled match has run");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventSynchronize(h->events[ADC_STAGE_TOTAL]));
    for (int i = 0; i < n && i < ADC_KERNEL_COUNT_; i++)
        HIPCHK(h, hipEventElapsedTime(&ms[i], h->kevents[i], h->kevents[i + 1]));
    return ADC_OK;
}

int adc_aggr_kernel_stats(adc_handle *h, int *launches, float *total_ms) {
    if (!h) return ADC_ERR_ARG;
    float t[ADC_KERNEL_COUNT_];
    int rc = adc_kernel_times(h, t, ADC_KERNEL_COUNT_);
    if (rc != ADC_OK) return rc;
    float sum = 0.0f;
    for (int i = 0; i < N_AGG_PASSES; i++) sum += t[ADC_KERNEL_AGGR0 + i];
    if (launches) *launches = h->agg_launches;
    if (total_ms) *total_ms = sum;
    return ADC_OK;
}

int adc_transfer_times(adc_handle *h, float *ms, int n) {
    if (!h || !ms || n < 0) return ADC_ERR_ARG;
    if (!h->initialized) return h->fail(ADC_ERR_NOT_INIT, "adc_transfer_times: handle not initialised");
    if (!h->have_xfer) return h->fail(ADC_ERR_ARG, "adc_transfer_times: no profiled adc_match has run");
    HIPCHK(h, hipSetDevice(h->device));
    float v[4] = {h->host_ms[0], 0.0f, 0.0f, h->host_ms[1]};
    HIPCHK(h, hipEventElapsedTime(&v[1], h->xfer_ev[0], h->xfer_ev[1]));
    HIPCHK(h, hipEventElapsedTime(&v[2], h->xfer_ev[2], h->xfer_ev[3]));
    for (int i = 0; i < n && i < 4; i++) ms[i] = v[i];
    return ADC_OK;
}

size_t adc_buffer_bytes(adc_handle *h, adc_buffer_id id) {
    if (!h || !h->initialized) return 0;
    const size_t npx = (size_t)h->g.W * h->g.H;
    switch (id) {
        case ADC_BUF_GRAY_LEFT: case ADC_BUF_GRAY_RIGHT:
        case ADC_BUF_MASK_OCC: case ADC_BUF_MASK_MIS: case ADC_BUF_MASK_OCC_VOTING: case ADC_BUF_MASK_MIS_VOTING:
        case ADC_BUF_EDGE_MASK:
            return npx;
        case ADC_BUF_CENSUS_LEFT: case ADC_BUF_CENSUS_RIGHT: return npx * 8;
        case ADC_BUF_ARMS: return npx * 4;
        case ADC_BUF_SUP_COUNT_H: case ADC_BUF_SUP_COUNT_V: return npx * 2;
        case ADC_BUF_COST_INIT: case ADC_BUF_COST_AGGR: case ADC_BUF_COST_SCAN: case ADC_BUF_COST_SCAN_AUX:
            return npx * (size_t)h->g.D * 4;
        case ADC_BUF_DISP_LEFT_WTA: case ADC_BUF_DISP_RIGHT_WTA: case ADC_BUF_DISP_OUTLIER:
        case ADC_BUF_DISP_VOTING: case ADC_BUF_DISP_INTERP: case ADC_BUF_DISP_EDGE: case ADC_BUF_DISP_FINAL:
            return npx * 4;
        default: return 0;
    }
}

int adc_get_buffer(adc_handle *h, adc_buffer_id id, void *dst, size_t bytes) {
    if (!h) return ADC_ERR_ARG;
    if (!h->initialized) return h->fail(ADC_ERR_NOT_INIT, "adc_get_buffer: handle not initialised");
    const size_t need = adc_buffer_bytes(h, id);
    if (!dst || need == 0 || bytes < need) return h->fail(ADC_ERR_ARG, "adc_get_buffer: bad id, null or short destination");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());  // a match may have been enqueued on a caller's stream
    const size_t npx = (size_t)h->g.W * h->g.H;
    const void *src = nullptr;
    const float *vol = nullptr;
    bool debug_only = false;
    switch (id) {
        case ADC_BUF_GRAY_LEFT: src = h->pl.gray_l; break;
        case ADC_BUF_GRAY_RIGHT: src = h->pl.gray_r; break;
        case ADC_BUF_CENSUS_LEFT:
        case ADC_BUF_CENSUS_RIGHT:
            launch_census_to_u64(id == ADC_BUF_CENSUS_LEFT ? h->pl.cen_l : h->pl.cen_r, h->tap_u64, npx, h->stream);
            HIPCHK(h, hipStreamSynchronize(h->stream));
            src = h->tap_u64;
            break;
        case ADC_BUF_ARMS: src = h->pl.arms; break;  // little-endian u32 = bytes {left,right,top,bottom}
        case ADC_BUF_SUP_COUNT_H: src = h->pl.cnt_h; break;
        case ADC_BUF_SUP_COUNT_V: src = h->pl.cnt_v; break;
        case ADC_BUF_COST_INIT: vol = h->snap_init; debug_only = true; break;
        case ADC_BUF_COST_AGGR: vol = h->snap_aggr; debug_only = true; break;
        case ADC_BUF_COST_SCAN: vol = h->vol_final; break;
        case ADC_BUF_COST_SCAN_AUX: vol = h->vol_aux; break;
        case ADC_BUF_DISP_LEFT_WTA: src = h->snap_wta_l; debug_only = true; break;
        case ADC_BUF_DISP_RIGHT_WTA: src = h->disp_r; break;
        case ADC_BUF_DISP_OUTLIER: src = h->snap_outlier; debug_only = true; break;
        case ADC_BUF_MASK_OCC: src = h->snap_occ; debug_only = true; break;
        case ADC_BUF_MASK_MIS: src = h->snap_mis; debug_only = true; break;
        case ADC_BUF_DISP_VOTING: src = h->snap_voting; debug_only = true; break;
        case ADC_BUF_MASK_OCC_VOTING: src = h->occ; break;
        case ADC_BUF_MASK_MIS_VOTING: src = h->mis; break;
        case ADC_BUF_DISP_INTERP: src = h->snap_interp; debug_only = true; break;
        case ADC_BUF_EDGE_MASK: src = h->edge; break;
        case ADC_BUF_DISP_EDGE: src = h->cur_disp; break;
        case ADC_BUF_DISP_FINAL: src = h->disp_final; break;
        default: break;
    }
    if (debug_only && !h->snap_init)
        return h->fail(ADC_ERR_ARG, "adc_get_buffer: this tap needs adc_set_debug_taps(h, 1) before the match");
    if (vol) {
        // internal volumes have row stride Dp >= D; hand out the reference's dense [H][W][D]
        HIPCHK(h, hipMemcpy2D(dst, (size_t)h->g.D * 4, vol, (size_t)h->g.Dp * 4, (size_t)h->g.D * 4, npx,
                              hipMemcpyDeviceToHost));
        return ADC_OK;
    }
    if (!src) return h->fail(ADC_ERR_ARG, "adc_get_buffer: unknown buffer id");
    HIPCHK(h, hipMemcpy(dst, src, need, hipMemcpyDeviceToHost));
    return ADC_OK;
}

}  // extern "C" (reopened below)

namespace {

// One device's share of a batch, pipelined: while the GPU matches pair k the host thread copies pair k+1
// into pinned memory and queues its upload, and pair k-1's result travels back and is copied out. Two slots
// of staging buffers; three streams (upload, the handle's compute stream, download) ordered by events. Per
// frame the GPU then sees only its own match time, not the 2 x 6.2 MB in and 8.3 MB out that a synchronous
// adc_match adds in front of and behind it. Pairs are drawn from the counter `next`, which all the devices
// of the call share: a device that finishes early takes the next pair instead of waiting for its turn.
int run_pipelined(adc_handle *h, const adc_batch_desc *d, std::atomic<int> *next) {
    const size_t npx = (size_t)h->g.W * h->g.H;
    HIPCHK(h, hipSetDevice(h->device));
    struct Slot {
        uint8_t *h_l = nullptr, *h_r = nullptr, *d_l = nullptr, *d_r = nullptr;
        float *h_o = nullptr, *d_o = nullptr;
        hipEvent_t in = nullptr, done = nullptr, out = nullptr;
        int pair = -1;
    } slot[2];
    hipStream_t s_in = nullptr, s_out = nullptr;
    int rc = ADC_OK;
    // nothing may still be running on the staging buffers or the events when they go away
    auto cleanup = [&]() {
        if (s_in) hipStreamSynchronize(s_in);
        hipStreamSynchronize(h->stream);
        if (s_out) hipStreamSynchronize(s_out);
        if (s_in) hipStreamDestroy(s_in);
        if (s_out) hipStreamDestroy(s_out);
        for (Slot &q : slot) {
            if (q.in) hipEventDestroy(q.in);
            if (q.done) hipEventDestroy(q.done);
            if (q.out) hipEventDestroy(q.out);
        }
    };
    auto finish = [&](Slot &q) -> int {   // pair q.pair: wait for its download, hand the map to the caller
        if (q.pair < 0) return ADC_OK;
        hipError_t e = hipEventSynchronize(q.out);
        if (e != hipSuccess) { q.pair = -1; return h->fail_hip("hipEventSynchronize", e); }
        std::memcpy(d->disp_left[q.pair], q.h_o, npx * sizeof(float));
        q.pair = -1;
        return ADC_OK;
    };
    // on a failure: pairs whose download was already queued are still handed over, then everything is drained
    auto bail = [&](int code) {
        std::string keep = h->err;
        for (Slot &q : slot) (void)finish(q);
        cleanup();
        h->err = keep;
        return code;
    };
#define PIPE(call)                                             \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return bail(h->fail_hip(#call, e_)); \
    } while (0)
    PIPE(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
    PIPE(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    // slot 0 reuses the handle's own staging buffers; slot 1 gets a second set (kept by the handle)
    slot[0].h_l = h->h_bgr_l; slot[0].h_r = h->h_bgr_r; slot[0].h_o = h->h_disp;
    slot[0].d_l = h->d_bgr_l; slot[0].d_r = h->d_bgr_r;
    if (!h->pipe_d_o[0]) {
        if ((rc = dev_alloc(h, &h->pipe_d_o[0], npx)) || (rc = dev_alloc(h, &h->pipe_d_o[1], npx)) ||
            (rc = dev_alloc(h, &h->pipe_d_l, npx * 3)) || (rc = dev_alloc(h, &h->pipe_d_r, npx * 3)) ||
            (rc = host_alloc(h, &h->pipe_h_l, npx * 3)) || (rc = host_alloc(h, &h->pipe_h_r, npx * 3)) ||
            (rc = host_alloc(h, &h->pipe_h_o, npx))) return bail(rc);
    }
    slot[0].d_o = h->pipe_d_o[0];
    slot[1].h_l = h->pipe_h_l; slot[1].h_r = h->pipe_h_r; slot[1].h_o = h->pipe_h_o;
    slot[1].d_l = h->pipe_d_l; slot[1].d_r = h->pipe_d_r; slot[1].d_o = h->pipe_d_o[1];
    for (Slot &q : slot) {
        PIPE(hipEventCreateWithFlags(&q.in, hipEventDisableTiming));
        PIPE(hipEventCreateWithFlags(&q.done, hipEventDisableTiming));
        PIPE(hipEventCreateWithFlags(&q.out, hipEventDisableTiming));
    }
    int k = 0;
    for (int i; (i = next->fetch_add(1)) < d->n_pairs; k++) {
        Slot &q = slot[k & 1];
        // the slot's previous pair (k-2) was finished at the end of iteration k-1, so every buffer of it is free
        if (!d->left_bgr[i] || !d->right_bgr[i] || !d->disp_left[i])
            return bail(h->fail(ADC_ERR_ARG, "adc_match_batch: null pointer in pair " + std::to_string(i)));
        std::memcpy(q.h_l, d->left_bgr[i], npx * 3);
        std::memcpy(q.h_r, d->right_bgr[i], npx * 3);
        PIPE(hipMemcpyAsync(q.d_l, q.h_l, npx * 3, hipMemcpyHostToDevice, s_in));
        PIPE(hipMemcpyAsync(q.d_r, q.h_r, npx * 3, hipMemcpyHostToDevice, s_in));
        PIPE(hipEventRecord(q.in, s_in));
        PIPE(hipStreamWaitEvent(h->stream, q.in, 0));
        if ((rc = enqueue_match(h, q.d_l, q.d_r, q.d_o, h->stream)) != ADC_OK) return bail(rc);
        PIPE(hipEventRecord(q.done, h->stream));
        PIPE(hipStreamWaitEvent(s_out, q.done, 0));
        PIPE(hipMemcpyAsync(q.h_o, q.d_o, npx * sizeof(float), hipMemcpyDeviceToHost, s_out));
        PIPE(hipEventRecord(q.out, s_out));
        q.pair = i;
        // while the GPU works on pair k, hand back pair k-1
        if ((rc = finish(slot[(k + 1) & 1])) != ADC_OK) return bail(rc);
    }
    if ((rc = finish(slot[(k + 1) & 1])) != ADC_OK) return bail(rc);
    if ((rc = finish(slot[k & 1])) != ADC_OK) return bail(rc);
#undef PIPE
    cleanup();
    return ADC_OK;
}

// ---- the context that lives in the static scratch ---------------------------------------------------------
enum { MAX_STATIC_DEVICES = 64 };
struct StaticCtx {
    std::mutex mu;
    unsigned char *ptr = nullptr;     // this device's copy of `scratch`, from hipGetSymbolAddress
    adc_handle h;                     // buffers carved from ptr; re-laid out when the frame size or the option changes
    int32_t W = 0, H = 0;
    adc_option opt{};
    bool valid = false;
};
StaticCtx g_static[MAX_STATIC_DEVICES];

}  // namespace

extern "C" {

int adc_match_batch(const adc_batch_desc *d) {
    if (!d || d->n_pairs < 0 || d->n_devices <= 0 || !d->devices || (d->n_pairs > 0 && (!d->left_bgr || !d->right_bgr || !d->disp_left))) {
        g_thread_error = "adc_match_batch: bad descriptor";
        return ADC_ERR_ARG;
    }
    const int nd = d->n_devices;
    if (d->handles) {   // the caller's handles: each must sit on its device, initialised for this frame size and option
        for (int di = 0; di < nd; di++) {
            const adc_handle *h = d->handles[di];
            if (!h || !h->initialized || h->arena || h->device != d->devices[di] || h->g.W != d->width || h->g.H != d->height ||
                std::memcmp(&h->opt, &d->option, sizeof(adc_option)) != 0) {
                g_thread_error = "adc_match_batch: handles[" + std::to_string(di) + "] is not initialised for this device, size and option";
                return ADC_ERR_ARG;
            }
            for (int dj = 0; dj < di; dj++)
                if (d->handles[dj] == h) { g_thread_error = "adc_match_batch: the same handle twice"; return ADC_ERR_ARG; }
        }
    }
    std::vector<int> rcs(nd, ADC_OK);
    std::vector<std::string> errs(nd);
    std::vector<std::thread> threads;
    std::atomic<int> next(0);
    for (int di = 0; di < nd; di++) {
        if (di >= d->n_pairs) break;   // fewer pairs than devices: the rest would only allocate and free
        threads.emplace_back([&, di]() {
            adc_handle *h = d->handles ? d->handles[di] : adc_create(d->devices[di]);
            if (!h) { rcs[di] = ADC_ERR_NOMEM; errs[di] = "adc_create failed"; return; }
            int rc = d->handles ? ADC_OK : adc_initialize(h, d->width, d->height, &d->option);
            if (rc == ADC_OK) rc = run_pipelined(h, d, &next);
            if (rc != ADC_OK) { rcs[di] = rc; errs[di] = h->err; }
            if (!d->handles) adc_destroy(h);
        });
    }
    for (auto &t : threads) t.join();
    for (int di = 0; di < nd; di++)
        if (rcs[di] != ADC_OK) {
            g_thread_error = "device " + std::to_string(d->devices[di]) + ": " + errs[di];
            return rcs[di];
        }
    return ADC_OK;
}

int adc_match_device_static(int device, int32_t width, int32_t height, const adc_option *opt,
                            const uint8_t *d_left, const uint8_t *d_right, float *d_disp, void *stream) {
    if (device < 0 || device >= MAX_STATIC_DEVICES) { g_thread_error = "adc_match_device_static: bad device ordinal"; return ADC_ERR_ARG; }
    if (!opt || !d_left || !d_right || !d_disp) { g_thread_error = "adc_match_device_static: null pointer"; return ADC_ERR_ARG; }
    StaticCtx &c = g_static[device];
    std::lock_guard<std::mutex> lock(c.mu);
    auto fail = [&](int code) { g_thread_error = c.h.err; return code; };
    c.h.device = device;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail(c.h.fail_hip("hipSetDevice", e));
    if (!c.ptr) {
        // once per device, with that device current: where this GPU keeps its copy of the array
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return fail(c.h.fail_hip("hipGetSymbolAddress(scratch)", e != hipSuccess ? e : hipErrorInvalidSymbol));
        c.ptr = static_cast<unsigned char *>(p);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    {
        // Not under stream capture: a re-layout sets kernel attributes, uploads tables from host memory and waits for
        // the stream, which would invalidate the capture, and a replayed graph of a steady-state call (memset and copy
        // nodes on the static array, the voting stamps of the captured frame) is not something this entry point
        // supports. Refused before anything is enqueued, so the caller's capture stays intact.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        e = hipStreamIsCapturing(s, &cap);
        if (e != hipSuccess) return fail(c.h.fail_hip("hipStreamIsCapturing", e));
        if (cap != hipStreamCaptureStatusNone)
            return fail(c.h.fail(ADC_ERR_UNSUPPORTED, "adc_match_device_static: the stream is being captured into a graph; "
                                                      "this entry point only runs on a stream that executes its work"));
    }
    if (!c.valid || c.W != width || c.H != height || std::memcmp(&c.opt, opt, sizeof(adc_option)) != 0) {
        // lay the buffers out anew in the scratch (it keeps the previous call's contents: do_initialize clears
        // what must start at zero and uploads the tables for this option)
        c.valid = false;
        adc_handle &h = c.h;
        h.initialized = false;
        h.arena = c.ptr;
        h.arena_size = ADC_STATIC_SCRATCH_BYTES;
        h.arena_used = 0;
        h.d_bgr_l = h.d_bgr_r = nullptr;
        h.tap_u64 = nullptr;
        h.profiling = h.debug_taps = false;
        const int rc = do_initialize(&h, width, height, opt, s);
        if (rc != ADC_OK) {
            if (h.arena_used > h.arena_size)
                h.err = "adc_match_device_static: a " + std::to_string(width) + "x" + std::to_string(height) + " frame with " +
                        std::to_string((long long)opt->max_disparity - opt->min_disparity) + " disparities needs more than the " +
                        std::to_string((size_t)ADC_STATIC_SCRATCH_BYTES) + " bytes of static scratch; use adc_initialize + adc_match_device";
            return fail(rc);
        }
        c.W = width; c.H = height; c.opt = *opt;
        c.valid = true;
    }
    const int rc = enqueue_match(&c.h, d_left, d_right, d_disp, s);
    if (rc != ADC_OK) { c.valid = false; return fail(rc); }
    return ADC_OK;
}

const char *adc_last_error(adc_handle *h) {
    if (!h) return g_thread_error.c_str();
    return h->err.c_str();
}

}  // extern "C"
