/*
 * adcensus.h -- C-ABI of the MI355X (gfx950) AD-Census stereo matcher.
 *
 * The reference has no FFI layer: its AD-Census path sits behind one C++ class,
 *   ADCensusStereo::Initialize / Match / Reset
 *   (final_version_project/Stereo_Matching/ADCensus_CUDA/ADCensusStereo.h:20,28,36, "ADC/" below),
 * called from ADC/main.cpp:153-180. The entry points below are what a binding for that class
 * would bind; include/ADCensusStereo.hpp is the header-only C++ class with the reference's own
 * method signatures on top of them.
 *
 * Conventions (all taken from the reference):
 *   images      BGR-interleaved uint8[H][W][3], row-major, no padding        (ADC/main.cpp:120-131)
 *   disparity   float32[H][W], left view, sub-pixel, invalid = +infinity      (ADC/adcensus_types.h:52)
 *   range       half-open [min_disparity, max_disparity), D = max - min       (ADC/ADCensusStereo.cu:32)
 * Every function returns ADC_OK (0) or a non-zero adc_status; nothing throws or exits.
 * A handle is bound to one GPU, is not thread-safe and runs one match at a time.
 * There is no CPU fallback: every entry point fails with ADC_ERR_HIP if no GPU is usable.
 */
#ifndef ADCENSUS_H_
#define ADCENSUS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum adc_status {
    ADC_OK = 0,
    ADC_ERR_ARG = 1,        /* null pointer / bad size / bad option (Match/Initialize returning false) */
    ADC_ERR_NOT_INIT = 2,   /* Match before a successful Initialize (ADC/ADCensusStereo.cu:66-68)     */
    ADC_ERR_HIP = 3,        /* a HIP runtime call failed; see adc_last_error                          */
    ADC_ERR_UNSUPPORTED = 4,/* option outside what the kernels implement; see adc_last_error          */
    ADC_ERR_NOMEM = 5
} adc_status;

/* Field-for-field ADCensusOption (ADC/adcensus_types.h:64-96) with fixed-width types.
 * adc_default_option fills the reference defaults (ADC/adcensus_types.h:86-93, ADC/config.h:12-13,26-29). */
typedef struct adc_option {
    int32_t min_disparity;
    int32_t max_disparity;
    int32_t lambda_ad;
    int32_t lambda_census;
    int32_t cross_L1;
    int32_t cross_L2;
    int32_t cross_t1;
    int32_t cross_t2;
    float   so_p1;
    float   so_p2;
    int32_t so_tso;
    int32_t irv_ts;
    float   irv_th;
    float   lrcheck_thres;
    uint8_t do_lr_check;
    uint8_t do_filling;
    uint8_t do_discontinuity_adjustment;
    uint8_t reserved_;
} adc_option;

typedef struct adc_handle adc_handle;

/* Intermediate buffers readable after a match, always in the reference's layouts.
 * Buffers marked (debug) are only kept when adc_set_debug_taps(h, 1) was called before the match,
 * because later stages overwrite them in place, exactly as the reference's ping-pong does
 * (ADC/scanline_optimizer.cu:339-342). */
typedef enum adc_buffer_id {
    ADC_BUF_GRAY_LEFT = 0,      /* u8[H][W]      ADC/cost_computor.cu:74-89                     */
    ADC_BUF_GRAY_RIGHT,
    ADC_BUF_CENSUS_LEFT,        /* u64[H][W]     ADC/census_cost.cu:9-98                        */
    ADC_BUF_CENSUS_RIGHT,
    ADC_BUF_COST_INIT,          /* f32[H][W][D]  (debug) ADC/cost_computor.cu:151-207           */
    ADC_BUF_ARMS,               /* u8[H][W][4] {left,right,top,bottom} ADC/cross_aggregator.h:12-15 */
    ADC_BUF_SUP_COUNT_H,        /* u16[H][W]     ADC/cost_aggregator.cu:262-316                 */
    ADC_BUF_SUP_COUNT_V,
    ADC_BUF_COST_AGGR,          /* f32[H][W][D]  (debug) after the 4 aggregation iterations     */
    ADC_BUF_COST_SCAN,          /* f32[H][W][D]  final scanline volume                          */
    ADC_BUF_COST_SCAN_AUX,      /* f32[H][W][D]  the other ping-pong volume (output of pass 3)  */
    ADC_BUF_DISP_LEFT_WTA,      /* f32[H][W]     (debug) ADC/ADCensusStereo.cu:200-267          */
    ADC_BUF_DISP_RIGHT_WTA,     /* f32[H][W]     ADC/ADCensusStereo.cu:358-424                  */
    ADC_BUF_DISP_OUTLIER,       /* f32[H][W]     (debug) after outlier detection                */
    ADC_BUF_MASK_OCC,           /* u8[H][W]      (debug) after outlier detection                */
    ADC_BUF_MASK_MIS,           /* u8[H][W]      (debug)                                        */
    ADC_BUF_DISP_VOTING,        /* f32[H][W]     (debug) after iterative region voting          */
    ADC_BUF_MASK_OCC_VOTING,    /* u8[H][W]      masks after voting                             */
    ADC_BUF_MASK_MIS_VOTING,
    ADC_BUF_DISP_INTERP,        /* f32[H][W]     (debug) after proper interpolation             */
    ADC_BUF_EDGE_MASK,          /* u8[H][W]      sobel mask                                     */
    ADC_BUF_DISP_EDGE,          /* f32[H][W]     after depth-discontinuity adjustment           */
    ADC_BUF_DISP_FINAL,         /* f32[H][W]     what Match returns                             */
    ADC_BUF_COUNT_
} adc_buffer_id;

/* Stage slots reported by adc_stage_times (milliseconds of device time of the last match that ran
 * with profiling on). The reference prints the same split with steady_clock under LOG
 * (ADC/ADCensusStereo.cu:76-143). */
typedef enum adc_stage_id {
    ADC_STAGE_PRE = 0,       /* pack, gray, census, colour-step flags, arms, support counts */
    ADC_STAGE_COST,          /* initial AD-Census cost volume                               */
    ADC_STAGE_AGGR,          /* 4 iterations x 2 passes                                     */
    ADC_STAGE_SCAN,          /* 4 directional passes                                        */
    ADC_STAGE_WTA,           /* left + right winner-take-all                                */
    ADC_STAGE_REFINE,        /* outlier, voting, interpolation, edges, median               */
    ADC_STAGE_TOTAL,         /* first kernel to last kernel                                 */
    ADC_STAGE_COUNT_
} adc_stage_id;

/* The volume kernels of one match, in launch order, as timed one by one by adc_kernel_times (HIP events on
 * the launch stream around every launch). The reference times whole stages only (ADC/ADCensusStereo.cu:76-143). */
typedef enum adc_kernel_id {
    ADC_KERNEL_COST = 0,     /* initial cost volume as a kernel of its own (0 ms when it is fused into AGGR0) */
    ADC_KERNEL_AGGR0 = 1,    /* aggregation passes 0..7: ids 1..8 (H,V | V,H | H,V | V,H; odd passes divide). A launch
                                that runs two passes (1+2, 3+4, 5+6) reports its time under the first of them and
                                leaves the second at 0 ms */
    ADC_KERNEL_SCAN0 = 9,    /* scanline passes 0..3: ids 9..12 (left-right, right-left, up-down, down-up)     */
    ADC_KERNEL_WTA_RIGHT = 13, /* right-view WTA; on small frames, where the left-view WTA is a kernel of its own instead
                                of a part of the last scanline pass, that kernel too. (With the debug taps on, the
                                copy of the aggregated volume falls into SCAN0's interval: time with taps off.) */
    ADC_KERNEL_COUNT_ = 14
} adc_kernel_id;

/* Fills *opt with ADCensusOption's defaults (ADC/adcensus_types.h:86-93). */
void adc_default_option(adc_option *opt);

/* ADCensusStereo::ADCensusStereo() (ADC/ADCensusStereo.cu:6-8) bound to GPU `device`.
 * Returns NULL only when out of host memory; GPU problems surface at adc_initialize. */
adc_handle *adc_create(int device);

/* ADCensusStereo::~ADCensusStereo() / Release() (ADC/ADCensusStereo.cu:10-14,426-430). */
void adc_destroy(adc_handle *h);

/* ADCensusStereo::Initialize (ADC/ADCensusStereo.cu:16-62): validates w,h > 0 and D > 0 and
 * allocates every device buffer the pipeline needs; no allocation happens in adc_match*. */
int adc_initialize(adc_handle *h, int32_t width, int32_t height, const adc_option *opt);

/* ADCensusStereo::Reset (ADC/ADCensusStereo.cu:147-157): release + initialize. */
int adc_reset(adc_handle *h, int32_t width, int32_t height, const adc_option *opt);

/* ADCensusStereo::Match (ADC/ADCensusStereo.cu:64-145): host pointers in, host pointer out,
 * synchronous. Null pointers or an uninitialised handle fail like the reference's `return false`. */
int adc_match(adc_handle *h, const uint8_t *left_bgr, const uint8_t *right_bgr, float *disp_left);

/* The same pipeline on buffers already resident in this handle's GPU memory (what a caller that
 * keeps frames on the device binds instead of Match). Enqueues on `stream` (a hipStream_t; NULL =
 * the handle's own stream) and returns without waiting; the caller synchronises the stream. */
int adc_match_device(adc_handle *h, const uint8_t *d_left_bgr, const uint8_t *d_right_bgr,
                     float *d_disp_left, void *stream);

/* Waits for everything the handle has enqueued on its own stream. */
int adc_synchronize(adc_handle *h);

/* Keep copies of the intermediates marked (debug) above. Allocates two extra volumes. */
int adc_set_debug_taps(adc_handle *h, int enable);

/* Record per-stage device times with HIP events during adc_match / adc_match_device. */
int adc_set_profiling(adc_handle *h, int enable);

/* Copies up to n (<= ADC_STAGE_COUNT_) stage times in ms into ms[]. Synchronises the stream.
 * Replaces the reference's printf timing (ADC/ADCensusStereo.cu:76-143). */
int adc_stage_times(adc_handle *h, float *ms, int n);

/* Copies up to n (<= ADC_KERNEL_COUNT_) per-launch device times in ms of the last profiled match into ms[],
 * indexed by adc_kernel_id. For per-kernel roofline accounting. */
int adc_kernel_times(adc_handle *h, float *ms, int n);

/* Number of aggregation launches of the last match (5 when same-axis passes share a launch, else 8) and the
 * device time of the last profiled match spent in them (the sum of their adc_kernel_times entries). */
int adc_aggr_kernel_stats(adc_handle *h, int *launches, float *total_ms);

/* What the last profiled adc_match spent around the pipeline, in ms: ms[0] host copy of the two images into
 * pinned staging (0 when the caller's images are pinned host memory: they are then the DMA source), ms[1]
 * upload of the two images (device time), ms[2] download of the map (device time), ms[3] host copy of the map
 * out of pinned staging (0 for a pinned destination). The reference's Match does the same transfers around
 * every stage (ADC/cost_computor.cu:91-120,209-269). */
int adc_transfer_times(adc_handle *h, float *ms, int n);

/* Size in bytes of a tap in the reference layout (0 if unknown id or not initialised). */
size_t adc_buffer_bytes(adc_handle *h, adc_buffer_id id);

/* Copies a tap to host memory `dst` (capacity `bytes` >= adc_buffer_bytes). Replaces the
 * reference's get_*_ptr() accessors (ADC/cost_computor.cu:287-295, ADC/cost_aggregator.cu:115-128). */
int adc_get_buffer(adc_handle *h, adc_buffer_id id, void *dst, size_t bytes);

/* Frame-sharded batch: n_pairs independent stereo pairs (host pointers) over n_devices GPUs, one host
 * thread per device, no inter-GPU traffic. The devices draw pair indices from one shared counter, so a
 * device that finishes early takes the next pair. handles == NULL: a handle per device is created and
 * initialised for the call and destroyed at its end; otherwise handles[i] is the caller's handle on
 * devices[i], already initialised with this width, height and option, and is reused (nothing is allocated
 * from the second call on). The reference processes its dataset list one pair after another on one GPU
 * (ADC/main.cpp:62-255); this is that loop spread over the GPUs of a node. */
typedef struct adc_batch_desc {
    int32_t width, height;
    adc_option option;
    int32_t n_pairs;
    const uint8_t *const *left_bgr;   /* n_pairs host pointers */
    const uint8_t *const *right_bgr;  /* n_pairs host pointers */
    float *const *disp_left;          /* n_pairs host pointers */
    int32_t n_devices;
    const int32_t *devices;           /* HIP device ordinals */
    adc_handle *const *handles;       /* NULL, or n_devices initialised handles (see above) */
} adc_batch_desc;
int adc_match_batch(const adc_batch_desc *desc);

/* adc_match_device for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc). It has no counterpart in the reference. It takes no handle and calls no
 * hipMalloc, hipFree or stream/event creation: every intermediate buffer of the
 * pipeline (both cost volumes, the 2-D planes, lists, tables) is laid out in one static device array of
 * ADC_STATIC_SCRATCH_BYTES that each GPU the library runs on holds until the process ends. The caller passes the
 * frame size and option with every call plus the device-resident inputs and output, exactly as to
 * adc_initialize + adc_match_device, and gets bit-identical results. Frames whose buffers need more than the
 * array (about 700 bytes per pixel at 64 disparities; every 450x375x64 Middlebury 2003 pair fits) are rejected
 * with ADC_ERR_UNSUPPORTED and a message naming the limit. The work is enqueued on `stream` (a hipStream_t;
 * NULL = the null stream); a call whose size or option differs from the previous call on that GPU first
 * re-lays the array out, sets kernel attributes, uploads the option's tables from pageable host memory (the
 * runtime may stage that copy in memory of its own) and waits for the stream once; later calls only enqueue
 * kernels, device-to-device copies and memsets. It is not for streams under graph capture: on a capturing stream
 * the call enqueues nothing, leaves the capture intact and returns ADC_ERR_UNSUPPORTED (a replayed graph would
 * carry the captured frame's set-up and voting state).
 * All calls on one GPU share that GPU's array, which keeps the previous call's contents (what must start at
 * zero is cleared by the call): calls that overlap on one GPU, from any threads or streams, race.
 * Errors: as adc_match_device; the message is read with adc_last_error(NULL) in the calling thread. */
#define ADC_STATIC_SCRATCH_BYTES (128u * 1024u * 1024u)
int adc_match_device_static(int device, int32_t width, int32_t height, const adc_option *opt,
                            const uint8_t *d_left_bgr, const uint8_t *d_right_bgr, float *d_disp_left,
                            void *stream);

/* Message of the last failure on this handle ("" if none). h == NULL returns the message of the
 * last failed adc_match_batch / adc_create in this thread. Replaces CUDA_CHECK_RETURN's
 * print-and-exit (ADC/adcensus_types.h:21-32). */
const char *adc_last_error(adc_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* ADCENSUS_H_ */
