// Known byte counts for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this chip: each kernel below moves exactly
// 1 GiB per direction it touches, in the access shapes the volume kernels use. Run once per counter:
//   rocprofv3 --pmc FETCH_SIZE -- bin/pmc_calibration      rocprofv3 --pmc WRITE_SIZE -- bin/pmc_calibration
// Build: hipcc -O3 --offload-arch=gfx950 pmc_calibration.hip -o bin/pmc_calibration
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void cal_copy16(const f4 *__restrict__ a, f4 *__restrict__ b) {            // 16-byte loads, 16-byte stores
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    b[i] = a[i];
}
__global__ void cal_read16(const f4 *__restrict__ a, float *__restrict__ out) {       // 16-byte loads only
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const f4 v = a[i];
    if (v.x + v.y + v.z + v.w == -1.0f) out[0] = 1.0f;
}
__global__ void cal_write8(f2 *__restrict__ b) {                                       // 8-byte stores only (the aggregation's store)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    b[i] = f2{1.0f, 2.0f};
}
__global__ void cal_read4(const float *__restrict__ a, float *__restrict__ out) {      // 4-byte loads only
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a[i] == -1.0f) out[0] = 1.0f;
}
__global__ void __launch_bounds__(256) cal_dma16_store8(const float *__restrict__ a, float *__restrict__ b) {
    // 1 KiB rows by one 16-byte-per-lane LDS-DMA, stored as two 8-byte-per-lane stores: the fused aggregation's shapes
    __shared__ __align__(16) float lds[4 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t row = (size_t)blockIdx.x * 4 + wave;
    __builtin_amdgcn_global_load_lds(a + row * 256 + lane * 4, (__attribute__((address_space(3))) void *)(lds + wave * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const f2 v0 = *(volatile f2 *)(lds + wave * 256 + lane * 2), v1 = *(volatile f2 *)(lds + wave * 256 + 128 + lane * 2);
    *(f2 *)(b + row * 256 + lane * 2) = v0;
    *(f2 *)(b + row * 256 + 128 + lane * 2) = v1;
}
__global__ void __launch_bounds__(256) cal_dma4(const float *__restrict__ a, float *__restrict__ b) {
    // 256-byte rows by 4-byte-per-lane LDS-DMA (the single-pass kernels' fill), 4-byte stores
    __shared__ float lds[4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t row = (size_t)blockIdx.x * 4 + wave;
    __builtin_amdgcn_global_load_lds(a + row * 64 + lane, (__attribute__((address_space(3))) void *)(lds + wave * 64), 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    b[row * 64 + lane] = *(volatile float *)(lds + wave * 64 + lane);
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    float *a, *b, *out;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; rep++) {
        cal_copy16<<<(unsigned)(bytes / 16 / 256), 256>>>((const f4 *)a, (f4 *)b);
        cal_read16<<<(unsigned)(bytes / 16 / 256), 256>>>((const f4 *)a, out);
        cal_write8<<<(unsigned)(bytes / 8 / 256), 256>>>((f2 *)b);
        cal_read4<<<(unsigned)(bytes / 4 / 256), 256>>>(a, out);
        cal_dma16_store8<<<(unsigned)(bytes / 1024 / 4), 256>>>(a, b);
        cal_dma4<<<(unsigned)(bytes / 256 / 4), 256>>>(a, b);
    }
    CHECK(hipDeviceSynchronize());
    std::printf("each kernel moved 1 GiB = 1048576 KiB per direction it touches\n");
    return 0;
}
