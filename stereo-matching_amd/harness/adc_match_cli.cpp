// Command-line harness: what the reference's main() does for one stereo pair (ADC/main.cpp:99-255,
// ADC/ = final_version_project/Stereo_Matching/ADCensus_CUDA/): read the two PNGs, repack to BGR,
// Initialize + Match, save the normalised disparity image, and, given a 2001/2003-style ground truth,
// print the evaluation block of PerformanceEval0103.
//
//   adc_match LEFT.png RIGHT.png OUT.png [--min N] [--max N] [--gt GT.png] [--gt-div F] [--delta F]
//             [--gt-pfm GT.pfm] [--calib calib.txt --depth DEPTH.png] [--raw OUT.f32] [--device N]
// --gt-pfm: the 2021 evaluation (PerformanceEval, ADC/performance_eval.cpp:184-215);
// --calib/--depth: Disp2Depth2 (ADC/main.cpp:499-551) with focal length and baseline from the scene's calib.txt.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ADCensusStereo.hpp"
#include "adcensus_harness.h"

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s LEFT.png RIGHT.png OUT.png [--min N] [--max N] [--gt GT.png] [--gt-div F] "
                             "[--delta F] [--raw OUT.f32] [--device N]\n", argv[0]);
        return 2;
    }
    ADCensusOption opt;               // defaults of ADC/adcensus_types.h:86-93
    std::string gt_path, raw_path, gt_pfm, calib_path, depth_path;
    float gt_div = 4.0f, delta = 2.0f;  // para[] for the 2003 set, delta_d (ADC/performance_eval.h:13-15)
    int device = 0;
    for (int i = 4; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        const char *v = argv[i + 1];
        if (k == "--min") opt.min_disparity = std::atoi(v);
        else if (k == "--max") opt.max_disparity = std::atoi(v);
        else if (k == "--gt") gt_path = v;
        else if (k == "--gt-div") gt_div = (float)std::atof(v);
        else if (k == "--delta") delta = (float)std::atof(v);
        else if (k == "--raw") raw_path = v;
        else if (k == "--gt-pfm") gt_pfm = v;
        else if (k == "--calib") calib_path = v;
        else if (k == "--depth") depth_path = v;
        else if (k == "--device") device = std::atoi(v);
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    uint8 *left = nullptr, *right = nullptr;
    int w = 0, h = 0, w2 = 0, h2 = 0;
    if (adc_png_read(argv[1], 3, &left, &w, &h) || adc_png_read(argv[2], 3, &right, &w2, &h2)) {
        std::fprintf(stderr, "cannot read the input images\n");            // ADC/main.cpp:105-108
        return -1;
    }
    if (w != w2 || h != h2) {
        std::fprintf(stderr, "left and right image sizes differ\n");       // ADC/main.cpp:109-112
        return -1;
    }
    ADCensusStereo matcher(device);
    if (!matcher.Initialize(w, h, opt)) {
        std::fprintf(stderr, "Initialize failed: %s\n", matcher.LastError());
        return -2;
    }
    std::vector<float32> disp((size_t)w * h);
    const auto t0 = std::chrono::steady_clock::now();
    if (!matcher.Match(left, right, disp.data())) {
        std::fprintf(stderr, "Match failed: %s\n", matcher.LastError());
        return -2;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("W = %d, H = %d, d = [%d,%d]\nmatching: %.6f s, speed %.1f Mpix*disp/s\n", w, h, opt.min_disparity,
                opt.max_disparity, secs, (double)w * h * (opt.max_disparity - opt.min_disparity) * 1e-6 / secs);
    std::vector<uint8> img((size_t)w * h);
    adc_disparity_to_u8(disp.data(), w, h, img.data());
    if (adc_png_write_gray(argv[3], img.data(), w, h)) {
        std::fprintf(stderr, "cannot write %s\n", argv[3]);
        return -3;
    }
    if (!raw_path.empty()) {
        FILE *f = std::fopen(raw_path.c_str(), "wb");
        if (!f || std::fwrite(disp.data(), sizeof(float), disp.size(), f) != disp.size()) return -3;
        std::fclose(f);
    }
    if (!gt_path.empty()) {
        uint8 *gt = nullptr;
        int gw = 0, gh = 0;
        if (adc_png_read(gt_path.c_str(), 1, &gt, &gw, &gh) || gw != w || gh != h) {
            std::fprintf(stderr, "cannot use ground truth %s\n", gt_path.c_str());
            return -4;
        }
        float rms = 0, pep = 0, mde = 0;
        adc_eval_0103(disp.data(), gt, w, h, gt_div, delta, secs, &rms, &pep, &mde);
        std::printf("\nEvalution:\n  RMS = %.3f\n  PEP = %.3f (bad %.1f)\n  Mde = %.3f\n", rms, pep, delta, mde);  // performance_eval.cpp:252
        std::printf("PEP_exact = %.9g\n", pep);
        adc_image_free(gt);
    }
    if (!gt_pfm.empty()) {
        float *gt = nullptr;
        int gw = 0, gh = 0;
        if (adc_pfm_read(gt_pfm.c_str(), &gt, &gw, &gh) || gw != w || gh != h) {
            std::fprintf(stderr, "cannot use ground truth %s\n", gt_pfm.c_str());
            return -4;
        }
        float rms = 0, pep = 0, mde = 0;
        adc_eval_pfm(disp.data(), gt, w, h, delta, secs, &rms, &pep, &mde);
        std::printf("\nEvalution:\n  RMS = %.3f\n  PEP = %.3f (bad %.1f)\n  Mde = %.3f\n", rms, pep, delta, mde);  // performance_eval.cpp:213
        std::printf("RMS_exact = %.9g\nPEP_exact = %.9g\n", rms, pep);
        adc_float_free(gt);
    }
    if (!depth_path.empty()) {
        float focal = 0, baseline = 0, doffs = 0;
        if (calib_path.empty() || adc_calib_read(calib_path.c_str(), &focal, &baseline, &doffs)) {
            std::fprintf(stderr, "--depth needs a readable --calib file\n");
            return -5;
        }
        float lo = 0, hi = 0;
        adc_disparity_to_depth(disp.data(), w, h, focal, baseline, doffs, nullptr, img.data(), &lo, &hi);
        std::printf("max_depth%g\nmin_depth%g\n", hi, lo);                // ADC/main.cpp:529-530
        if (adc_png_write_gray(depth_path.c_str(), img.data(), w, h)) return -3;
    }
    adc_image_free(left);
    adc_image_free(right);
    return 0;
}
