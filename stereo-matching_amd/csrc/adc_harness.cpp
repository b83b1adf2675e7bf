// Host-side harness functions (include/adcensus_harness.h): PNG in/out on top of zlib, the disparity
// image normalisation and the Middlebury 2001/2003 evaluation of the reference's caller.
// ADC/ = final_version_project/Stereo_Matching/ADCensus_CUDA/.
#include "../../include/adcensus_harness.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <limits>
#include <string>
#include <vector>

namespace {

uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
void put32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}

bool read_file(const char *path, std::vector<uint8_t> &buf) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) { std::fclose(f); return false; }
    buf.resize((size_t)n);
    const bool ok = std::fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
    std::fclose(f);
    return ok;
}

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

void chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &data) {
    put32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(out.size() - at)));
}

}  // namespace

extern "C" {

int adc_png_read(const char *path, int channels, uint8_t **data, int *width, int *height) {
    if (!path || !data || !width || !height || (channels != 1 && channels != 3)) return 1;
    std::vector<uint8_t> f;
    if (!read_file(path, f)) return 2;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 8 + 25 || std::memcmp(f.data(), sig, 8) != 0) return 3;
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    for (size_t p = 8; p + 12 <= f.size();) {
        const uint32_t len = be32(&f[p]);
        if (p + 12 + (size_t)len > f.size()) return 3;
        const uint8_t *t = &f[p + 4], *d = &f[p + 8];
        if (!std::memcmp(t, "IHDR", 4) && len >= 13) {
            W = be32(d); H = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
        } else if (!std::memcmp(t, "PLTE", 4)) {
            plte.assign(d, d + len);
        } else if (!std::memcmp(t, "IDAT", 4)) {
            idat.insert(idat.end(), d, d + len);
        } else if (!std::memcmp(t, "IEND", 4)) {
            break;
        }
        p += 12 + (size_t)len;
    }
    if (!W || !H || W > 65535 || H > 65535 || depth != 8 || interlace != 0) return 4;   // unsupported variant
    int spp;                                                                         // samples per pixel
    switch (ctype) {
        case 0: spp = 1; break;
        case 2: spp = 3; break;
        case 3: spp = 1; break;
        case 4: spp = 2; break;
        case 6: spp = 4; break;
        default: return 4;
    }
    if (channels == 1 && ctype != 0 && ctype != 4) return 4;
    const size_t rowb = (size_t)W * spp;
    std::vector<uint8_t> raw((rowb + 1) * H);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return 5;
    // undo the per-row filters in place
    std::vector<uint8_t> img(rowb * H);
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t ft = raw[(rowb + 1) * y];
        const uint8_t *in = &raw[(rowb + 1) * y + 1];
        uint8_t *cur = &img[rowb * y];
        const uint8_t *up = y ? &img[rowb * (y - 1)] : nullptr;
        for (size_t i = 0; i < rowb; i++) {
            const int a = i >= (size_t)spp ? cur[i - spp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)spp) ? up[i - spp] : 0;
            int v = in[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: return 5;
            }
            cur[i] = (uint8_t)v;
        }
    }
    uint8_t *out = (uint8_t *)std::malloc((size_t)W * H * channels);
    if (!out) return 6;
    for (size_t i = 0; i < (size_t)W * H; i++) {
        const uint8_t *s = &img[i * spp];
        uint8_t r, g, b;
        if (ctype == 3) {
            if ((size_t)s[0] * 3 + 2 >= plte.size()) { std::free(out); return 5; }
            r = plte[s[0] * 3]; g = plte[s[0] * 3 + 1]; b = plte[s[0] * 3 + 2];
        } else if (ctype == 0 || ctype == 4) {
            r = g = b = s[0];
        } else {
            r = s[0]; g = s[1]; b = s[2];
        }
        if (channels == 1) {
            out[i] = r;
        } else {
            out[i * 3] = b; out[i * 3 + 1] = g; out[i * 3 + 2] = r;
        }
    }
    *data = out;
    *width = (int)W;
    *height = (int)H;
    return 0;
}

void adc_image_free(uint8_t *data) { std::free(data); }

int adc_png_write_gray(const char *path, const uint8_t *gray, int width, int height) {
    if (!path || !gray || width <= 0 || height <= 0) return 1;
    std::vector<uint8_t> raw((size_t)(width + 1) * height);
    for (int y = 0; y < height; y++) {
        raw[(size_t)(width + 1) * y] = 0;   // filter: none
        std::memcpy(&raw[(size_t)(width + 1) * y + 1], gray + (size_t)width * y, (size_t)width);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(clen);
    if (compress2(z.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return 5;
    z.resize(clen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)width);
    put32(ihdr, (uint32_t)height);
    const uint8_t tail[5] = {8, 0, 0, 0, 0};   // 8-bit gray, deflate, adaptive filtering, not interlaced
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE *f = std::fopen(path, "wb");
    if (!f) return 2;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok ? 0 : 2;
}

int adc_disparity_to_u8(const float *disp, int width, int height, uint8_t *out) {
    if (!disp || !out || width <= 0 || height <= 0) return 1;
    const float inf = std::numeric_limits<float>::infinity();
    const size_t n = (size_t)width * height;
    float lo = (float)width, hi = -(float)width;    // ADC/main.cpp:447
    for (size_t i = 0; i < n; i++) {
        const float d = std::fabs(disp[i]);
        if (d != inf) {
            lo = d < lo ? d : lo;
            hi = d > hi ? d : hi;
        }
    }
    const float span = hi - lo;
    for (size_t i = 0; i < n; i++) {
        const float d = std::fabs(disp[i]);
        if (d == inf || !(span > 0.0f)) out[i] = 0;
        else out[i] = (uint8_t)((d - lo) / span * 255);
    }
    return 0;
}

int adc_eval_0103(const float *disp, const uint8_t *gt, int width, int height, float gt_div, float delta,
                  double seconds, float *rms, float *pep, float *mde) {
    if (!disp || !gt || width <= 0 || height <= 0 || !(gt_div > 0.0f)) return 1;
    const float inf = std::numeric_limits<float>::infinity();
    float sq = 0.0f, bad = 0.0f, dmax = 0.0f;
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            const size_t i = (size_t)y * width + x;
            dmax = dmax > disp[i] ? dmax : disp[i];        // std::max(d_max, estimate), :229
            const float g = (float)gt[i];
            if (g == 0) continue;
            const float d = disp[i] != inf ? disp[i] - g / gt_div : 0 - g / gt_div;
            sq += d * d;
            bad += (std::fabs(d) > delta);
        }
    }
    if (rms) *rms = std::sqrt(sq / (float)(height * width));
    if (pep) *pep = bad / (float)(height * width);
    if (mde) *mde = (float)((double)((float)(width * height) * dmax) / seconds * 1e-6);
    return 0;
}

int adc_pfm_read(const char *path, float **data, int *width, int *height) {
    if (!path || !data || !width || !height) return 1;
    std::vector<uint8_t> f;
    if (!read_file(path, f)) return 2;
    // header: "Pf\n<width> <height>\n<scale>\n"; a negative scale means little endian
    size_t p = 0;
    auto token = [&](std::string &t) {
        t.clear();
        while (p < f.size() && std::isspace(f[p])) p++;
        while (p < f.size() && !std::isspace(f[p])) t.push_back((char)f[p++]);
        return !t.empty();
    };
    std::string t;
    if (!token(t) || t != "Pf") return t == "PF" ? 4 : 3;
    int W = 0, H = 0;
    if (!token(t)) return 3;
    W = std::atoi(t.c_str());
    if (!token(t)) return 3;
    H = std::atoi(t.c_str());
    if (!token(t)) return 3;
    const double scale = std::atof(t.c_str());
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535 || !(scale < 0)) return 4;   // big-endian files: not supported
    p++;                                                                       // the single whitespace after the scale
    const size_t n = (size_t)W * H;
    if (f.size() < p + n * 4) return 5;
    float *out = (float *)std::malloc(n * sizeof(float));
    if (!out) return 6;
    for (int y = 0; y < H; y++)   // the file's first row is the image's bottom row
        std::memcpy(out + (size_t)(H - 1 - y) * W, f.data() + p + (size_t)y * W * 4, (size_t)W * 4);
    *data = out;
    *width = W;
    *height = H;
    return 0;
}

void adc_float_free(float *data) { std::free(data); }

int adc_pfm_write(const char *path, const float *data, int width, int height) {
    if (!path || !data || width <= 0 || height <= 0) return 1;
    FILE *f = std::fopen(path, "wb");
    if (!f) return 2;
    std::fprintf(f, "Pf\n%d %d\n-1.000000\n", width, height);   // ADC/performance_eval.cpp:120-140
    bool ok = true;
    for (int y = height - 1; y >= 0 && ok; y--)
        ok = std::fwrite(data + (size_t)y * width, sizeof(float), (size_t)width, f) == (size_t)width;
    std::fclose(f);
    return ok ? 0 : 2;
}

int adc_eval_pfm(const float *disp, const float *gt, int width, int height, float delta, double seconds,
                 float *rms, float *pep, float *mde) {
    if (!disp || !gt || width <= 0 || height <= 0) return 1;
    const float inf = std::numeric_limits<float>::infinity();
    float sq = 0.0f, bad = 0.0f, dmax = 0.0f;
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            const size_t i = (size_t)y * width + x;
            dmax = dmax > disp[i] ? dmax : disp[i];
            const float g = gt[i] == inf ? 0.0f : gt[i];
            const float d = disp[i] - g;
            sq += d * d;
            bad += (std::fabs(d) > delta);
        }
    }
    if (rms) *rms = std::sqrt(sq / (float)(height * width));
    if (pep) *pep = bad / (float)(height * width);
    if (mde) *mde = (float)((double)((float)(width * height) * dmax) * 1e-6 / seconds);
    return 0;
}

int adc_disparity_to_depth(const float *disp, int width, int height, float focal, float baseline, float doffs,
                           float *depth, uint8_t *depth_u8, float *min_depth, float *max_depth) {
    if (!disp || width <= 0 || height <= 0) return 1;
    const float inf = std::numeric_limits<float>::infinity();
    const size_t n = (size_t)width * height;
    std::vector<float> tmp;
    if (!depth) { tmp.resize(n); depth = tmp.data(); }
    float lo = 114514.0f, hi = 0.0f;                       // ADC/main.cpp:516-517
    for (size_t i = 0; i < n; i++) {
        float d = std::fabs(disp[i]);
        if (d == inf) d = 0;
        depth[i] = d == 0 ? 0.0f : (float)(focal * baseline / d + doffs);
    }
    for (size_t i = 0; i < n; i++) {
        const float z = std::fabs(depth[i]);
        if (z != inf) {
            lo = z < lo ? z : lo;
            hi = z > hi ? z : hi;
        }
    }
    if (depth_u8) {
        const float span = hi - lo;
        for (size_t i = 0; i < n; i++) {
            const float z = std::fabs(depth[i]);
            depth_u8[i] = (z == inf || !(span > 0.0f)) ? 0 : (uint8_t)((z - lo) / span * 255);
        }
    }
    if (min_depth) *min_depth = lo;
    if (max_depth) *max_depth = hi;
    return 0;
}

int adc_calib_read(const char *path, float *focal, float *baseline, float *doffs) {
    if (!path) return 1;
    std::vector<uint8_t> f;
    if (!read_file(path, f)) return 2;
    const std::string txt(f.begin(), f.end());
    auto value = [&](const char *key, float *out) {
        const size_t at = txt.find(key);
        if (at == std::string::npos) return false;
        size_t p = at + std::strlen(key);
        while (p < txt.size() && (txt[p] == '[' || txt[p] == ' ')) p++;
        if (out) *out = (float)std::atof(txt.c_str() + p);
        return true;
    };
    if (!value("cam0=", focal) || !value("baseline=", baseline)) return 3;
    if (!value("doffs=", doffs) && doffs) *doffs = 0.0f;
    return 0;
}

}  // extern "C"
