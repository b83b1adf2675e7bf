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

#ifdef __cplusplus
}
#endif
#endif /* ADCENSUS_HARNESS_H_ */
