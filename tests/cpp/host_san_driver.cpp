// Host-side sanitizer run of the C-ABI (AddressSanitizer + UBSan on the host code only; the kernels are the product's):
// every entry point once, on small frames -- create / initialize / match from pageable and from pinned memory / reset to
// another size / taps / timers / device-pointer match / the entry point that allocates nothing / a batch over two host
// threads on one GPU / destroy. Exit code 0 and no sanitizer report = pass. Built by `make -C stereo-matching_amd san`.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "adcensus.h"

#define REQUIRE(x) do { if (!(x)) { std::printf("FAILED line %d: %s\n", __LINE__, #x); return 1; } } while (0)

static void make_pair(int W, int H, int shift, std::vector<uint8_t> &L, std::vector<uint8_t> &R) {
    L.resize((size_t)W * H * 3);
    R.resize((size_t)W * H * 3);
    uint32_t s = 12345u + (uint32_t)shift;
    std::vector<uint8_t> tex((size_t)(W + 64) * H * 3);
    for (auto &v : tex) { s = s * 1664525u + 1013904223u; v = (uint8_t)(s >> 24); }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < 3; c++) {
                L[((size_t)y * W + x) * 3 + c] = tex[((size_t)y * (W + 64) + x + 8) * 3 + c];
                R[((size_t)y * W + x) * 3 + c] = tex[((size_t)y * (W + 64) + x + 8 + 5) * 3 + c];   // disparity 5 (x - d)
            }
}

int main() {
    adc_option opt;
    adc_default_option(&opt);
    opt.max_disparity = 32;
    const int W = 97, H = 61;
    std::vector<uint8_t> L, R;
    make_pair(W, H, 0, L, R);
    std::vector<float> d0((size_t)W * H), d1((size_t)W * H), d2((size_t)W * H);

    adc_handle *h = adc_create(0);
    REQUIRE(h);
    REQUIRE(adc_match(h, L.data(), R.data(), d0.data()) == ADC_ERR_NOT_INIT);
    REQUIRE(adc_initialize(h, 0, H, &opt) == ADC_ERR_ARG);
    REQUIRE(adc_initialize(h, W, H, &opt) == ADC_OK);
    REQUIRE(adc_set_profiling(h, 1) == ADC_OK);
    REQUIRE(adc_match(h, L.data(), R.data(), d0.data()) == ADC_OK);
    float st[ADC_STAGE_COUNT_], kt[ADC_KERNEL_COUNT_], xt[4];
    REQUIRE(adc_stage_times(h, st, ADC_STAGE_COUNT_) == ADC_OK);
    REQUIRE(adc_kernel_times(h, kt, ADC_KERNEL_COUNT_) == ADC_OK);
    REQUIRE(adc_transfer_times(h, xt, 4) == ADC_OK);
    REQUIRE(adc_set_profiling(h, 0) == ADC_OK);

    // pinned host memory, debug taps
    uint8_t *pl = nullptr, *pr = nullptr;
    float *po = nullptr;
    REQUIRE(hipHostMalloc((void **)&pl, L.size()) == hipSuccess);
    REQUIRE(hipHostMalloc((void **)&pr, R.size()) == hipSuccess);
    REQUIRE(hipHostMalloc((void **)&po, d1.size() * sizeof(float)) == hipSuccess);
    std::memcpy(pl, L.data(), L.size());
    std::memcpy(pr, R.data(), R.size());
    REQUIRE(adc_set_debug_taps(h, 1) == ADC_OK);
    REQUIRE(adc_match(h, pl, pr, po) == ADC_OK);
    REQUIRE(std::memcmp(po, d0.data(), d0.size() * sizeof(float)) == 0);
    for (int b = 0; b < ADC_BUF_COUNT_; b++) {
        const size_t n = adc_buffer_bytes(h, (adc_buffer_id)b);
        if (!n) continue;
        std::vector<uint8_t> buf(n);
        REQUIRE(adc_get_buffer(h, (adc_buffer_id)b, buf.data(), n) == ADC_OK);
        REQUIRE(adc_get_buffer(h, (adc_buffer_id)b, buf.data(), n - 1) != ADC_OK);
    }
    REQUIRE(adc_set_debug_taps(h, 0) == ADC_OK);

    // device pointers, and the entry point that allocates nothing
    uint8_t *dl = nullptr, *dr = nullptr;
    float *dd = nullptr;
    REQUIRE(hipMalloc((void **)&dl, L.size()) == hipSuccess);
    REQUIRE(hipMalloc((void **)&dr, R.size()) == hipSuccess);
    REQUIRE(hipMalloc((void **)&dd, d1.size() * sizeof(float)) == hipSuccess);
    REQUIRE(hipMemcpy(dl, L.data(), L.size(), hipMemcpyHostToDevice) == hipSuccess);
    REQUIRE(hipMemcpy(dr, R.data(), R.size(), hipMemcpyHostToDevice) == hipSuccess);
    REQUIRE(adc_match_device(h, dl, dr, dd, nullptr) == ADC_OK);
    REQUIRE(adc_synchronize(h) == ADC_OK);
    REQUIRE(hipMemcpy(d1.data(), dd, d1.size() * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
    REQUIRE(std::memcmp(d1.data(), d0.data(), d0.size() * sizeof(float)) == 0);
    hipStream_t s;
    REQUIRE(hipStreamCreate(&s) == hipSuccess);
    REQUIRE(adc_match_device_static(0, W, H, &opt, dl, dr, dd, s) == ADC_OK);
    REQUIRE(hipStreamSynchronize(s) == hipSuccess);
    REQUIRE(hipMemcpy(d2.data(), dd, d2.size() * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
    REQUIRE(std::memcmp(d2.data(), d0.data(), d0.size() * sizeof(float)) == 0);
    REQUIRE(adc_match_device_static(0, 4000, 3000, &opt, dl, dr, dd, s) == ADC_ERR_UNSUPPORTED);

    // another size on the same handle
    const int W2 = 64, H2 = 40;
    std::vector<uint8_t> L2, R2;
    make_pair(W2, H2, 7, L2, R2);
    std::vector<float> e0((size_t)W2 * H2);
    opt.max_disparity = 70;
    REQUIRE(adc_reset(h, W2, H2, &opt) == ADC_OK);
    REQUIRE(adc_match(h, L2.data(), R2.data(), e0.data()) == ADC_OK);

    // a batch of six pairs over two host threads on one GPU, own handles and the caller's
    const int N = 6;
    std::vector<const uint8_t *> ls(N), rs(N);
    std::vector<std::vector<float>> outs(N, std::vector<float>((size_t)W2 * H2));
    std::vector<float *> os(N);
    for (int i = 0; i < N; i++) { ls[i] = L2.data(); rs[i] = R2.data(); os[i] = outs[i].data(); }
    int devices[2] = {0, 0};
    adc_batch_desc bd;
    std::memset(&bd, 0, sizeof(bd));
    bd.width = W2; bd.height = H2; bd.option = opt;
    bd.n_pairs = N; bd.left_bgr = ls.data(); bd.right_bgr = rs.data(); bd.disp_left = os.data();
    bd.n_devices = 2; bd.devices = devices;
    REQUIRE(adc_match_batch(&bd) == ADC_OK);
    for (int i = 0; i < N; i++) REQUIRE(std::memcmp(outs[i].data(), e0.data(), e0.size() * sizeof(float)) == 0);
    adc_handle *hs[2] = {adc_create(0), adc_create(0)};
    REQUIRE(hs[0] && hs[1]);
    REQUIRE(adc_initialize(hs[0], W2, H2, &opt) == ADC_OK && adc_initialize(hs[1], W2, H2, &opt) == ADC_OK);
    bd.handles = hs;
    REQUIRE(adc_match_batch(&bd) == ADC_OK);
    REQUIRE(adc_match_batch(&bd) == ADC_OK);
    for (int i = 0; i < N; i++) REQUIRE(std::memcmp(outs[i].data(), e0.data(), e0.size() * sizeof(float)) == 0);
    adc_destroy(hs[0]);
    adc_destroy(hs[1]);

    REQUIRE(hipStreamDestroy(s) == hipSuccess);
    REQUIRE(hipFree(dl) == hipSuccess && hipFree(dr) == hipSuccess && hipFree(dd) == hipSuccess);
    REQUIRE(hipHostFree(pl) == hipSuccess && hipHostFree(pr) == hipSuccess && hipHostFree(po) == hipSuccess);
    adc_destroy(h);
    std::puts("host sanitizer driver ok");
    // Everything of the library's has been released and checked by now. Leave without the exit handlers: the GPU
    // runtime's own teardown under the sanitizer's allocator (operator delete inside its unload path) fails now and
    // then with an allocator check that has nothing to do with the code under test.
    std::fflush(stdout);
    std::fflush(stderr);
    _exit(0);
}
