// Header-only C++ mirror of the reference's matcher class on top of the C-ABI (adcensus.h).
//
// Same type names, method signatures, argument meaning and error behaviour as
//   final_version_project/Stereo_Matching/ADCensus_CUDA/ADCensusStereo.h:9-90   (class ADCensusStereo)
//   final_version_project/Stereo_Matching/ADCensus_CUDA/adcensus_types.h:38-96  (typedefs, ADCensusOption)
// so the reference's caller (main.cpp:136-180) compiles against this header unchanged:
//
//     ADCensusOption opt;  opt.min_disparity = 0;  opt.max_disparity = 256;
//     ADCensusStereo m;
//     if (!m.Initialize(w, h, opt)) ...
//     if (!m.Match(left_bgr, right_bgr, disp)) ...
//
// Link with -ladcensus_hip. Every method returns false where the reference returns false; GPU
// failures also return false (the reference exits the process) and LastError() says why.
#ifndef ADCENSUS_STEREO_HPP_
#define ADCENSUS_STEREO_HPP_

#include <cstdint>
#include <limits>

#include "adcensus.h"

typedef int8_t sint8;
typedef uint8_t uint8;
typedef int16_t sint16;
typedef uint16_t uint16;
typedef int32_t sint32;
typedef uint32_t uint32;
typedef int64_t sint64;
typedef uint64_t uint64;
typedef float float32;
typedef double float64;

constexpr auto Invalid_Float = std::numeric_limits<float32>::infinity();  // adcensus_types.h:52
constexpr auto Large_Float = 99999.0f;                                    // adcensus_types.h:54

/** Field-for-field ADCensusOption (adcensus_types.h:64-96) with its default constructor (:86-93). */
struct ADCensusOption {
    sint32 min_disparity;
    sint32 max_disparity;
    sint32 lambda_ad;
    sint32 lambda_census;
    sint32 cross_L1;
    sint32 cross_L2;
    sint32 cross_t1;
    sint32 cross_t2;
    float32 so_p1;
    float32 so_p2;
    sint32 so_tso;
    sint32 irv_ts;
    float32 irv_th;
    float32 lrcheck_thres;
    bool do_lr_check;
    bool do_filling;
    bool do_discontinuity_adjustment;

    ADCensusOption()
        : min_disparity(0), max_disparity(64), lambda_ad(10), lambda_census(30), cross_L1(34), cross_L2(17),
          cross_t1(20), cross_t2(6), so_p1(1.0f), so_p2(3.0f), so_tso(15), irv_ts(20), irv_th(0.4f),
          lrcheck_thres(1.0f), do_lr_check(true), do_filling(true), do_discontinuity_adjustment(true) {}
};

class ADCensusStereo {
public:
    explicit ADCensusStereo(int device = 0) : h_(adc_create(device)) {}
    ~ADCensusStereo() { adc_destroy(h_); }
    ADCensusStereo(const ADCensusStereo &) = delete;
    ADCensusStereo &operator=(const ADCensusStereo &) = delete;

    /** ADCensusStereo.h:20 */
    bool Initialize(const sint32 &width, const sint32 &height, const ADCensusOption &option) {
        const adc_option o = convert(option);
        return h_ && adc_initialize(h_, width, height, &o) == ADC_OK;
    }
    /** ADCensusStereo.h:28 -- host pointers, BGR u8[H][W][3] in, f32[H][W] out (caller-allocated). */
    bool Match(const uint8 *img_left, const uint8 *img_right, float32 *disp_left) {
        return h_ && adc_match(h_, img_left, img_right, disp_left) == ADC_OK;
    }
    /** ADCensusStereo.h:36 */
    bool Reset(const uint32 &width, const uint32 &height, const ADCensusOption &option) {
        const adc_option o = convert(option);
        return h_ && adc_reset(h_, (int32_t)width, (int32_t)height, &o) == ADC_OK;
    }
    const char *LastError() const { return adc_last_error(h_); }
    adc_handle *handle() const { return h_; }

private:
    static adc_option convert(const ADCensusOption &s) {
        adc_option o;
        adc_default_option(&o);
        o.min_disparity = s.min_disparity;
        o.max_disparity = s.max_disparity;
        o.lambda_ad = s.lambda_ad;
        o.lambda_census = s.lambda_census;
        o.cross_L1 = s.cross_L1;
        o.cross_L2 = s.cross_L2;
        o.cross_t1 = s.cross_t1;
        o.cross_t2 = s.cross_t2;
        o.so_p1 = s.so_p1;
        o.so_p2 = s.so_p2;
        o.so_tso = s.so_tso;
        o.irv_ts = s.irv_ts;
        o.irv_th = s.irv_th;
        o.lrcheck_thres = s.lrcheck_thres;
        o.do_lr_check = s.do_lr_check;
        o.do_filling = s.do_filling;
        o.do_discontinuity_adjustment = s.do_discontinuity_adjustment;
        return o;
    }
    adc_handle *h_;
};

#endif  // ADCENSUS_STEREO_HPP_
