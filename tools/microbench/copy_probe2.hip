// Which access shape gets a read+write stream closest to the HBM ceiling on this chip? Variants of a copy over
// buffers from 256 MiB to 4 GiB: one 16-byte element per thread without a loop, 2/4/8 elements per thread (block-
// contiguous), grid-stride loops of 256..2048 workgroups, 256- and 1024-thread workgroups, read-only and write-only
// streams, and the shape the volume kernels use (1 KiB rows by LDS-DMA into LDS, 8-byte-per-lane stores).
// Build: hipcc -O3 --offload-arch=gfx950 copy_probe2.hip -o bin/copy_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// E elements of 16 bytes per thread, the workgroup's elements contiguous (E x blockDim.x x 16 bytes), no loop
template <int E>
__global__ void k_once(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n4) {
    const size_t base = (size_t)blockIdx.x * blockDim.x * E + threadIdx.x;
    f4 v[E];
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = a[base + (size_t)e * blockDim.x];
#pragma unroll
    for (int e = 0; e < E; e++) b[base + (size_t)e * blockDim.x] = v[e];
}
__global__ void k_stride(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
template <int E>
__global__ void k_read(const f4 *__restrict__ a, float *__restrict__ out, size_t n4) {
    const size_t base = (size_t)blockIdx.x * blockDim.x * E + threadIdx.x;
    f4 s = {};
#pragma unroll
    for (int e = 0; e < E; e++) s += a[base + (size_t)e * blockDim.x];
    if (s.x + s.y + s.z + s.w == -1.0f) out[0] = 1.0f;
}
template <int E>
__global__ void k_write(f4 *__restrict__ b, size_t n4) {
    const size_t base = (size_t)blockIdx.x * blockDim.x * E + threadIdx.x;
    const f4 v = {1.0f, 2.0f, 3.0f, 4.0f};
#pragma unroll
    for (int e = 0; e < E; e++) b[base + (size_t)e * blockDim.x] = v;
}
// the volume kernels' shape: each wave brings rows of 1 KiB by one 16-byte-per-lane LDS-DMA, DEPTH rows ahead, and
// stores them from LDS as two 8-byte-per-lane stores (512 B each); a workgroup of 16 waves walks a contiguous piece
template <int DEPTH, int STORE_B>
__global__ void __launch_bounds__(1024) k_dma(const float *__restrict__ a, float *__restrict__ b, size_t rows_per_wg) {
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *my = lds + wave * (DEPTH + 1) * 256;
    const size_t r0 = (size_t)blockIdx.x * rows_per_wg;
    const float *src = a + r0 * 256 + lane * 4;
    float *dst = b + r0 * 256;
    const size_t n = rows_per_wg / 16;   // rows of this wave: r = wave + 16 * i
    for (int d = 0; d < DEPTH; d++)
        __builtin_amdgcn_global_load_lds(src + ((size_t)wave + 16 * d) * 256, (__attribute__((address_space(3))) void *)(my + d * 256), 16, 0, 0);
    int slot = 0, fslot = DEPTH;
    for (size_t i = 0; i < n; i++) {
        const size_t nx = i + DEPTH < n ? i + DEPTH : n - 1;
        __builtin_amdgcn_global_load_lds(src + ((size_t)wave + 16 * nx) * 256, (__attribute__((address_space(3))) void *)(my + fslot * 256), 16, 0, 0);
        // row i's DMA is older than DEPTH later DMAs and the stores of DEPTH iterations (none yet in the first DEPTH)
        if (i < DEPTH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH * (1 + (STORE_B == 8 ? 2 : 1))) : "memory");
        float *o = dst + ((size_t)wave + 16 * i) * 256;
        if (STORE_B == 8) {
            const f2 v0 = *(volatile f2 *)(my + slot * 256 + lane * 2), v1 = *(volatile f2 *)(my + slot * 256 + 128 + lane * 2);
            *(f2 *)(o + lane * 2) = v0;
            *(f2 *)(o + 128 + lane * 2) = v1;
        } else {
            const f4 v = *(volatile f4 *)(my + slot * 256 + lane * 4);
            *(f4 *)(o + lane * 4) = v;
        }
        slot = slot == DEPTH ? 0 : slot + 1;
        fslot = fslot == DEPTH ? 0 : fslot + 1;
    }
}

// grid-stride with K loads in flight per thread and counted waits: a store never waits for an earlier store's
// acknowledgement, only for its own load (loads and stores retire in issue order on one counter)
template <int K>
__global__ void __launch_bounds__(256) k_pipe(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n4) {
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (n4 - i + step - 1) / step;     // elements of this thread
    f4 v[K];
    const f4 *pa = a + i;
    f4 *pb = b + i;
#pragma unroll
    for (int k = 0; k < K; k++) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[k]) : "v"(pa + (size_t)(k < (int)n ? k : 0) * step) : "memory");
    size_t j = 0;
    for (; j + K <= n; j += K) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            // in flight behind v[k]'s load: K - 1 later loads and the stores issued since: at least K - 1 operations
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v[k]) : "n"(K - 1) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(pb + (j + k) * step), "v"(v[k]) : "memory");
            const size_t nx = j + k + K < n ? j + k + K : n - 1;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[k]) : "v"(pa + nx * step) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (the last n % K elements are left out: a bandwidth probe)
}
// one row of 1 KiB per wave and no loop: LDS-DMA in, 16-byte stores out (the volume kernels' instructions, the
// one-shot kernel's life cycle)
__global__ void __launch_bounds__(256) k_dma_once(const float *__restrict__ a, float *__restrict__ b) {
    __shared__ __align__(16) float lds[4 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t row = (size_t)blockIdx.x * 4 + wave;
    __builtin_amdgcn_global_load_lds(a + row * 256 + lane * 4, (__attribute__((address_space(3))) void *)(lds + wave * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const f4 v = *(volatile f4 *)(lds + wave * 256 + lane * 4);
    *(f4 *)(b + row * 256 + lane * 4) = v;
}
// k_dma with the rows dealt round-robin over all waves of the grid (a compact front instead of one stream per workgroup)
template <int DEPTH>
__global__ void __launch_bounds__(1024) k_dma_rr(const float *__restrict__ a, float *__restrict__ b, size_t rows) {
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *my = lds + wave * (DEPTH + 1) * 256;
    const size_t gw = (size_t)blockIdx.x * 16 + wave, nw = (size_t)gridDim.x * 16;
    const size_t n = rows / nw;
    const float *src = a + gw * 256 + lane * 4;
    float *dst = b + gw * 256 + lane * 4;
    for (int d = 0; d < DEPTH; d++)
        __builtin_amdgcn_global_load_lds(src + (size_t)d * nw * 256, (__attribute__((address_space(3))) void *)(my + d * 256), 16, 0, 0);
    int slot = 0, fslot = DEPTH;
    for (size_t i = 0; i < n; i++) {
        const size_t nx = i + DEPTH < n ? i + DEPTH : n - 1;
        __builtin_amdgcn_global_load_lds(src + nx * nw * 256, (__attribute__((address_space(3))) void *)(my + fslot * 256), 16, 0, 0);
        if (i < DEPTH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH * 2) : "memory");
        const f4 v = *(volatile f4 *)(my + slot * 256 + lane * 4);
        *(f4 *)(dst + i * nw * 256) = v;
        slot = slot == DEPTH ? 0 : slot + 1;
        fslot = fslot == DEPTH ? 0 : fslot + 1;
    }
}

// loads and stores on different waves: waves 0..3 of a workgroup only bring rows in (LDS-DMA, DEPTH steps ahead,
// their wait counter sees nothing but loads), waves 4..7 only store them (never a wait on the vector-memory
// counter); one workgroup barrier per step. CONTIG: a workgroup walks a contiguous piece, else the 4 KiB steps are
// dealt round-robin over the workgroups.
template <int DEPTH, bool CONTIG>
__global__ void __launch_bounds__(512) k_split(const float *__restrict__ a, float *__restrict__ b, size_t rows) {
    constexpr int R = DEPTH + 2;
    __shared__ __align__(16) float lds[R * 4 * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), w4 = wave & 3;
    const size_t n = rows / 4 / gridDim.x;            // steps of this workgroup
    const size_t first = CONTIG ? (size_t)blockIdx.x * n : blockIdx.x, stride = CONTIG ? 1 : gridDim.x;
    const float *src = a + (first * 4 + w4) * 256 + lane * 4;
    float *dst = b + (first * 4 + w4) * 256 + lane * 4;
    const size_t step = stride * 4 * 256;
    if (wave < 4) {
        for (int d = 0; d < DEPTH; d++)
            __builtin_amdgcn_global_load_lds(src + (size_t)d * step, (__attribute__((address_space(3))) void *)(lds + (d * 4 + w4) * 256), 16, 0, 0);
        int fslot = DEPTH;
        for (size_t i = 0; i < n; i++) {
            const size_t nx = i + DEPTH < n ? i + DEPTH : n - 1;
            __builtin_amdgcn_global_load_lds(src + nx * step, (__attribute__((address_space(3))) void *)(lds + (fslot * 4 + w4) * 256), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
            __builtin_amdgcn_s_barrier();
            fslot = fslot == R - 1 ? 0 : fslot + 1;
        }
    } else {
        int slot = 0;
        for (size_t i = 0; i < n; i++) {
            __builtin_amdgcn_s_barrier();
            const f4 v = *(volatile f4 *)(lds + (slot * 4 + w4) * 256 + lane * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            *(f4 *)(dst + i * step) = v;
            slot = slot == R - 1 ? 0 : slot + 1;
        }
    }
}

int main() {
    const size_t maxb = (size_t)4 << 30;
    f4 *a, *b;
    float *out;
    CHECK(hipMalloc(&a, maxb));
    CHECK(hipMalloc(&b, maxb));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(a, 1, maxb));
    CHECK(hipMemset(b, 0, maxb));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double bytes_moved, auto launch) {
        for (int w = 0; w < 2; w++) launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 10;
        for (int r = 0; r < reps; r++) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-64s %8.0f GB/s\n", name, bytes_moved * 1e-6 * reps / ms);
        std::fflush(stdout);
    };
    char nm[128];
    for (size_t mib : {256, 512, 1024, 2048, 4096}) {
        const size_t bytes = mib << 20, n4 = bytes / 16;
        for (int T : {256, 1024}) {
            std::snprintf(nm, sizeof nm, "%zu MiB copy, 1 x 16 B per thread, %d threads", mib, T);
            timeit(nm, 2.0 * bytes, [&] { k_once<1><<<(unsigned)(n4 / T), T>>>(a, b, n4); });
            std::snprintf(nm, sizeof nm, "%zu MiB copy, 4 x 16 B per thread, %d threads", mib, T);
            timeit(nm, 2.0 * bytes, [&] { k_once<4><<<(unsigned)(n4 / T / 4), T>>>(a, b, n4); });
        }
        std::snprintf(nm, sizeof nm, "%zu MiB copy, 8 x 16 B per thread, 256 threads", mib);
        timeit(nm, 2.0 * bytes, [&] { k_once<8><<<(unsigned)(n4 / 256 / 8), 256>>>(a, b, n4); });
    }
    const size_t bytes = (size_t)2 << 30, n4 = bytes / 16;
    for (int grid : {256, 512, 768, 1024, 1280, 1536, 2048}) {
        std::snprintf(nm, sizeof nm, "2048 MiB copy, grid-stride, %d workgroups of 256", grid);
        timeit(nm, 2.0 * bytes, [&] { k_stride<<<grid, 256>>>(a, b, n4); });
    }
    for (int grid : {256, 512, 1024}) {
        std::snprintf(nm, sizeof nm, "2048 MiB copy, grid-stride, %d workgroups of 1024", grid);
        timeit(nm, 2.0 * bytes, [&] { k_stride<<<grid, 1024>>>(a, b, n4); });
    }
    timeit("2048 MiB read only, 4 x 16 B per thread, 256 threads", 1.0 * bytes, [&] { k_read<4><<<(unsigned)(n4 / 256 / 4), 256>>>(a, out, n4); });
    timeit("2048 MiB read only, 1 x 16 B per thread, 256 threads", 1.0 * bytes, [&] { k_read<1><<<(unsigned)(n4 / 256), 256>>>(a, out, n4); });
    timeit("2048 MiB write only, 4 x 16 B per thread, 256 threads", 1.0 * bytes, [&] { k_write<4><<<(unsigned)(n4 / 256 / 4), 256>>>(b, n4); });
    timeit("2048 MiB write only, 1 x 16 B per thread, 256 threads", 1.0 * bytes, [&] { k_write<1><<<(unsigned)(n4 / 256), 256>>>(b, n4); });
    // LDS-DMA rows + stores from LDS, one workgroup of 16 waves per CU x k
    const size_t rows = bytes / 1024;
    for (int wgs : {256, 512, 1024, 2048, 4096}) {
        const size_t rpw = rows / wgs;
        std::snprintf(nm, sizeof nm, "2048 MiB LDS-DMA rows, depth 2, 8-byte stores, %d workgroups", wgs);
        timeit(nm, 2.0 * bytes, [&] { k_dma<2, 8><<<wgs, 1024, 16 * 3 * 1024>>>((const float *)a, (float *)b, rpw); });
        std::snprintf(nm, sizeof nm, "2048 MiB LDS-DMA rows, depth 2, 16-byte stores, %d workgroups", wgs);
        timeit(nm, 2.0 * bytes, [&] { k_dma<2, 16><<<wgs, 1024, 16 * 3 * 1024>>>((const float *)a, (float *)b, rpw); });
        std::snprintf(nm, sizeof nm, "2048 MiB LDS-DMA rows, depth 4, 16-byte stores, %d workgroups", wgs);
        timeit(nm, 2.0 * bytes, [&] { k_dma<4, 16><<<wgs, 1024, 16 * 5 * 1024>>>((const float *)a, (float *)b, rpw); });
    }
    for (int grid : {512, 1024, 2048}) {
        std::snprintf(nm, sizeof nm, "2048 MiB copy, grid-stride, 2 loads in flight, counted waits, %d workgroups", grid);
        timeit(nm, 2.0 * bytes, [&] { k_pipe<2><<<grid, 256>>>(a, b, n4); });
        std::snprintf(nm, sizeof nm, "2048 MiB copy, grid-stride, 4 loads in flight, counted waits, %d workgroups", grid);
        timeit(nm, 2.0 * bytes, [&] { k_pipe<4><<<grid, 256>>>(a, b, n4); });
    }
    timeit("2048 MiB copy, one 1 KiB row per wave by LDS-DMA, no loop", 2.0 * bytes, [&] { k_dma_once<<<(unsigned)(rows / 4), 256>>>((const float *)a, (float *)b); });
    for (int wgs : {128, 256, 512, 1024}) {
        std::snprintf(nm, sizeof nm, "2048 MiB LDS-DMA rows round-robin over waves, depth 2, %d workgroups", wgs);
        timeit(nm, 2.0 * bytes, [&] { k_dma_rr<2><<<wgs, 1024, 16 * 3 * 1024>>>((const float *)a, (float *)b, rows); });
        std::snprintf(nm, sizeof nm, "2048 MiB LDS-DMA rows round-robin over waves, depth 6, %d workgroups", wgs);
        timeit(nm, 2.0 * bytes, [&] { k_dma_rr<6><<<wgs, 1024, 16 * 7 * 1024>>>((const float *)a, (float *)b, rows); });
    }
    for (int wgs : {512, 1024, 2048}) {
        std::snprintf(nm, sizeof nm, "2048 MiB loader waves + storer waves, depth 4, round-robin steps, %d workgroups", wgs);
        timeit(nm, 2.0 * bytes, [&] { k_split<4, false><<<wgs, 512>>>((const float *)a, (float *)b, rows); });
        std::snprintf(nm, sizeof nm, "2048 MiB loader waves + storer waves, depth 4, contiguous pieces, %d workgroups", wgs);
        timeit(nm, 2.0 * bytes, [&] { k_split<4, true><<<wgs, 512>>>((const float *)a, (float *)b, rows); });
        std::snprintf(nm, sizeof nm, "2048 MiB loader waves + storer waves, depth 8, contiguous pieces, %d workgroups", wgs);
        timeit(nm, 2.0 * bytes, [&] { k_split<8, true><<<wgs, 512>>>((const float *)a, (float *)b, rows); });
    }
    return 0;
}
