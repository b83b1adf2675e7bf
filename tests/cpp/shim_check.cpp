// The reference caller's sequence (main.cpp:136-180) against include/ADCensusStereo.hpp:
//   shim_check W H DMIN DMAX left.raw right.raw out.raw   -> 0 on success
//   shim_check --errors                                  -> checks the `return false` paths only (no GPU needed)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ADCensusStereo.hpp"

static bool read_all(const char *path, std::vector<uint8> &buf) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    const size_t n = std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    return n == buf.size();
}

int main(int argc, char **argv) {
    if (argc == 2 && std::strcmp(argv[1], "--errors") == 0) {
        ADCensusStereo m;
        ADCensusOption opt;
        float32 d = 0;
        uint8 px[3] = {0, 0, 0};
        if (m.Match(px, px, &d)) return 10;              // not initialised
        if (m.Initialize(0, 4, opt)) return 11;          // width <= 0
        opt.max_disparity = opt.min_disparity;
        if (m.Initialize(4, 4, opt)) return 12;          // empty disparity range
        if (!m.LastError() || !m.LastError()[0]) return 13;
        std::puts("shim errors ok");
        return 0;
    }
    if (argc != 8) return 2;
    const sint32 W = std::atoi(argv[1]), H = std::atoi(argv[2]);
    ADCensusOption opt;
    opt.min_disparity = std::atoi(argv[3]);
    opt.max_disparity = std::atoi(argv[4]);
    std::vector<uint8> L((size_t)W * H * 3), R((size_t)W * H * 3);
    if (!read_all(argv[5], L) || !read_all(argv[6], R)) return 3;
    std::vector<float32> disp((size_t)W * H);
    ADCensusStereo m;
    if (!m.Initialize(W, H, opt)) { std::fprintf(stderr, "Initialize: %s\n", m.LastError()); return 4; }
    if (!m.Match(L.data(), R.data(), disp.data())) { std::fprintf(stderr, "Match: %s\n", m.LastError()); return 5; }
    if (m.Match(nullptr, R.data(), disp.data())) return 6;   // null pointer -> false, handle stays usable
    if (!m.Match(L.data(), R.data(), disp.data())) return 7;
    FILE *f = std::fopen(argv[7], "wb");
    if (!f) return 8;
    std::fwrite(disp.data(), sizeof(float32), disp.size(), f);
    std::fclose(f);
    return 0;
}
