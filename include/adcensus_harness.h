/* Host-side harness of the AD-Census path: what the reference's caller does around Match().
 * Plain C ABI, exported by libadcensus_hip.so next to adcensus.h. ADC/ = final_version_project/
 * Stereo_Matching/ADCensus_CUDA/ in the reference. Nothing here touches the GPU. */
#ifndef ADCENSUS_HARNESS_H_
#define ADCENSUS_HARNESS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Decode an 8-bit, non-interlaced PNG (colour types gray, RGB, palette, gray+alpha, RGBA).
 * channels = 3: interleaved B,G,R rows without padding -- the layout cv::imread hands the reference,
 *               which it repacks byte for byte (ADC/main.cpp:99-131); gray images are replicated;
 * channels = 1: gray, as cv::imread(path, 0) returns a gray file (ADC/performance_eval.cpp:221);
 *               colour files are rejected.
 * *data is allocated by the library (release with adc_image_free). Returns 0 on success. */
int adc_png_read(const char *path, int channels, uint8_t **data, int *width, int *height);
void adc_image_free(uint8_t *data);

/* Write an 8-bit gray PNG (cv::imwrite of the CV_8UC1 disparity image, ADC/main.cpp:470). */
int adc_png_write_gray(const char *path, const uint8_t *gray, int width, int height);

/* SaveDisparityMap's normalisation (ADC/main.cpp:443-468): |d|, min/max over the valid (non-inf)
 * values, invalid -> 0, else (uchar)((|d| - min) / (max - min) * 255) in float32. A map without two
 * distinct valid values gives all zeros (the reference divides by zero there). */
int adc_disparity_to_u8(const float *disp, int width, int height, uint8_t *out);

/* PerformanceEval0103 (ADC/performance_eval.cpp:218-253), the 2001/2003 Middlebury evaluation:
 * ground truth u8 with 0 = unknown (skipped), true disparity = gt / gt_div (4 for 2003, :15); an
 * invalid estimate counts as disparity 0; RMS and PEP (fraction with |error| > delta) are divided by
 * width*height, not by the number of known pixels; Mde = width*height*max(estimate)*1e-6 / seconds
 * (+inf as soon as one estimate is invalid, as in the reference). float32 accumulation in row-major
 * order, like the reference. */
int adc_eval_0103(const float *disp, const uint8_t *gt, int width, int height, float gt_div, float delta,
                  double seconds, float *rms, float *pep, float *mde);

/* Single-channel little-endian PFM ("Pf"), rows stored bottom-up, as LoadPFM / savePFM handle the
 * Middlebury 2021 ground truth (ADC/performance_eval.cpp:11-181). *data: float[height][width], top row
 * first, allocated by the library (release with adc_float_free). Colour ("PF") files are rejected. */
int adc_pfm_read(const char *path, float **data, int *width, int *height);
void adc_float_free(float *data);
int adc_pfm_write(const char *path, const float *data, int width, int height);

/* PerformanceEval (ADC/performance_eval.cpp:184-215), the 2021 evaluation against a PFM ground truth:
 * every pixel counts, an unknown (+inf) ground truth counts as disparity 0, error = estimate - truth (an
 * invalid estimate makes RMS infinite, as in the reference), PEP = fraction with |error| > delta,
 * Mde = width*height*max(estimate)*1e-6 / seconds. float32 accumulation in row-major order. */
int adc_eval_pfm(const float *disp, const float *gt, int width, int height, float delta, double seconds,
                 float *rms, float *pep, float *mde);

/* Disp2Depth2 (ADC/main.cpp:499-551): depth = focal*baseline/|d| + doffs where d is valid and non-zero,
 * 0 elsewhere; min / max over the whole map (starting from 114514 / 0, so any zero pulls the minimum to 0);
 * depth_u8 = (uchar)((depth - min) / (max - min) * 255), all zeros when max == min. depth and depth_u8 may
 * each be NULL. The reference takes focal and baseline from tables that repeat the calib.txt files
 * (ADC/main.cpp:9-20) and doffs = 0. */
int adc_disparity_to_depth(const float *disp, int width, int height, float focal, float baseline, float doffs,
                           float *depth, uint8_t *depth_u8, float *min_depth, float *max_depth);

/* focal length (cam0[0][0]), baseline and doffs of a Middlebury 2021 calib.txt. */
int adc_calib_read(const char *path, float *focal, float *baseline, float *doffs);

#ifdef __cplusplus
}
#endif
#endif /* ADCENSUS_HARNESS_H_ */
