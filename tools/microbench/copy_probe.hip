// Copy-bandwidth variants on a 2 GiB buffer: what does a read+write stream sustain on this chip, and does the
// access order matter? Build: hipcc -O3 --offload-arch=gfx950 copy_probe.hip -o copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// grid-stride, consecutive workgroups touch consecutive 4 KiB blocks
template <int NT>
__global__ void __launch_bounds__(256) k_stride(const f4 *a, f4 *b, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f4 v = (NT & 1) ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT & 2) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}
// every workgroup owns one contiguous piece of the buffer
template <int NT>
__global__ void __launch_bounds__(256) k_chunk(const f4 *a, f4 *b, size_t n4) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        f4 v = (NT & 1) ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT & 2) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}
// four independent loads in flight per thread
__global__ void __launch_bounds__(256) k_stride4(const f4 *a, f4 *b, size_t n4) {
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * step < n4; i += 4 * step) {
        f4 v0 = a[i], v1 = a[i + step], v2 = a[i + 2 * step], v3 = a[i + 3 * step];
        b[i] = v0; b[i + step] = v1; b[i + 2 * step] = v2; b[i + 3 * step] = v3;
    }
}

int main() {
    const size_t bytes = (size_t)2 << 30, n4 = bytes / 16;
    f4 *a, *b;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        for (int w = 0; w < 2; w++) launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 10;
        for (int r = 0; r < reps; r++) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-44s %8.0f GB/s (read + write)\n", name, 2.0 * bytes * 1e-6 * reps / ms);
    };
    for (int grid : {512, 1024, 2048, 4096, 16384}) {
        char nm[64];
        std::snprintf(nm, sizeof nm, "grid-stride, %d workgroups", grid);
        timeit(nm, [&] { k_stride<0><<<grid, 256>>>(a, b, n4); });
    }
    timeit("grid-stride 2048, nt store", [&] { k_stride<2><<<2048, 256>>>(a, b, n4); });
    timeit("grid-stride 2048, nt load + nt store", [&] { k_stride<3><<<2048, 256>>>(a, b, n4); });
    timeit("grid-stride 2048, 4 loads in flight", [&] { k_stride4<<<2048, 256>>>(a, b, n4); });
    timeit("contiguous piece per workgroup, 2048", [&] { k_chunk<0><<<2048, 256>>>(a, b, n4); });
    timeit("contiguous piece per workgroup, 16384", [&] { k_chunk<0><<<16384, 256>>>(a, b, n4); });
    timeit("contiguous piece per workgroup, 16384, nt", [&] { k_chunk<3><<<16384, 256>>>(a, b, n4); });
    timeit("hipMemcpyAsync device to device", [&] { CHECK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
