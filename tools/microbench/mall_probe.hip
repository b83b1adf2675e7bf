// Does the 256 MiB Infinity Cache keep freshly written data for the next kernel's reads, and does it absorb
// repeated overwrites of a small buffer? Alternates a write-only and a read-only kernel over buffers of
// 32 MiB ... 2 GiB and prints the bandwidth of each. Build: hipcc -O3 --offload-arch=gfx950 mall_probe.hip -o mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_write(float4 *p, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(v, v, v, v);
}
__global__ void __launch_bounds__(256) k_read(const float4 *p, size_t n4, float *out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 t = p[i];
        acc += t.x + t.y + t.z + t.w;
    }
    if (acc == 12345.678f) out[0] = acc;   // never true; keeps the loads
}
__global__ void __launch_bounds__(256) k_copy(const float4 *a, float4 *b, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

int main() {
    const size_t max_bytes = (size_t)2 << 30;
    float4 *buf = nullptr, *buf2 = nullptr;
    float *out = nullptr;
    CHECK(hipMalloc(&buf, max_bytes));
    CHECK(hipMalloc(&buf2, max_bytes));
    CHECK(hipMalloc(&out, 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    std::printf("%10s %14s %14s %14s\n", "MiB", "write GB/s", "read-after-write GB/s", "copy GB/s (r+w)");
    for (size_t mib : {32, 64, 96, 128, 160, 192, 224, 256, 384, 512, 1024, 2048}) {
        const size_t bytes = mib << 20, n4 = bytes / 16;
        const int reps = (int)(((size_t)8 << 30) / bytes) + 2;
        float tw = 0, tr = 0, tc = 0;
        for (int warm = 0; warm < 2; warm++) { k_write<<<grid, 256>>>(buf, n4, 1.f); k_read<<<grid, 256>>>(buf, n4, out); }
        CHECK(hipDeviceSynchronize());
        for (int r = 0; r < reps; r++) {
            float ms;
            CHECK(hipEventRecord(e0)); k_write<<<grid, 256>>>(buf, n4, (float)r); CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); tw += ms;
            CHECK(hipEventRecord(e0)); k_read<<<grid, 256>>>(buf, n4, out); CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); tr += ms;
        }
        for (int r = 0; r < reps; r++) {
            float ms;
            CHECK(hipEventRecord(e0)); k_copy<<<grid, 256>>>(buf, buf2, n4); CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); tc += ms;
        }
        std::printf("%10zu %14.0f %14.0f %14.0f\n", mib, bytes * 1e-6 * reps / tw, bytes * 1e-6 * reps / tr, 2.0 * bytes * 1e-6 * reps / tc);
    }
    return 0;
}
