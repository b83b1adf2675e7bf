// Can the matrix pipe do the aggregation's ordered window sums?
//   v_mfma_f32_4x4x1_16b_f32  D = A x B + C  with K = 1: every element of D receives exactly ONE product per
//   instruction, D[i][j] = C[i][j] + a[i] * b[j]. With b[j] in {0.0, 1.0} that is "add the term a[i] to window j if the
//   term belongs to it": sixteen blocks x 4 rows = the 64 disparities of a term, 4 columns = four neighbouring positions.
// Part 1 (exactness): N terms accumulated that way must be the bits of the sequential fp32 sum  c = c + (b ? a : 0)
//   in the same order (one rounding per term, round to nearest even, no flushing of anything that occurs here).
// Part 2 (rate): cycles per MFMA on one SIMD with 1 / 2 / 4 waves per SIMD, one and two accumulator sets per wave,
//   with a ds_read_b32 per term feeding A, counted waits.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off mfma_add_probe.hip -o mfma_add_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// ---- part 1 ------------------------------------------------------------------------------------------------
// a: [n][64] term values, b: [n][64] masks (0 / 1); out: [64][4] accumulators (lane, register)
__global__ void __launch_bounds__(64) k_exact(const float *a, const float *b, float *out, int n) {
    const int lane = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < n; t++) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[t * 64 + lane], b[t * 64 + lane], c, 0, 0, 0);
    for (int i = 0; i < 4; i++) out[lane * 4 + i] = c[i];
}

// ---- part 2 ------------------------------------------------------------------------------------------------
// MODE 0: one accumulator set, MFMAs back to back (B = 1.0), A from registers
//      1: two accumulator sets interleaved, A from registers
//      2: one set, A by ds_read_b32, 8 reads in flight, counted waits
//      3: two sets (two 64-disparity halves of a slot), A by ds_read_b32, 16 reads in flight, counted waits
template <int MODE>
__global__ void __launch_bounds__(1024) k_rate(float *out, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 128; i += blockDim.x) lds[i] = 1.0f + (float)(i & 7);
    __syncthreads();
    f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
    const float one = 1.0f;
    float a0 = (float)lane, a1 = (float)(lane + 1);
    const uint32_t addr = (uint32_t)(uintptr_t)lds + (uint32_t)lane * 4u;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 16; u++) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, one, c0, 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, one, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, one, c1, 0, 0, 0);
            }
        } else if (MODE == 2) {
            float r[16];
#pragma unroll
            for (int u = 0; u < 16; u++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[u]) : "v"(addr), "n"(u * 512));
#pragma unroll
            for (int u = 0; u < 16; u++) {
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r[u]) : "n"(15 - u));
                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(r[u], one, c0, 0, 0, 0);
            }
        } else {
            float r[16];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[2 * u]) : "v"(addr), "n"(u * 512));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[2 * u + 1]) : "v"(addr), "n"(u * 512 + 256));
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r[2 * u]), "+v"(r[2 * u + 1]) : "n"(14 - 2 * u));
                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(r[2 * u], one, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(r[2 * u + 1], one, c1, 0, 0, 0);
            }
        }
    }
    if (c0[0] + c1[0] == -1.f) out[threadIdx.x] = c0[1] + c1[2] + c0[3];
}

template <int MODE>
static void rate(const char *what, int waves_per_simd, float *d_out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int threads = 64 * 4 * waves_per_simd;
    k_rate<MODE><<<256, threads, 16 * 128 * 4>>>(d_out, 10);
    CHECK(hipEventRecord(e0));
    k_rate<MODE><<<256, threads, 16 * 128 * 4>>>(d_out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_simd = 16.0 * iters * waves_per_simd;
    std::printf("%-58s %d waves/SIMD  %7.3f ms  %5.2f cycles per MFMA per SIMD at 2400 MHz (256 adds each: %5.1f adds per cycle)\n",
                what, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd, 256.0 * mfma_per_simd / (ms * 1e-3 * 2.4e9));
}

int main() {
    // ---- exactness
    const int n = 70, trials = 4000;
    std::vector<float> a((size_t)n * 64), b((size_t)n * 64), got(256), want(256);
    float *da, *db, *dout;
    CHECK(hipMalloc(&da, a.size() * 4)); CHECK(hipMalloc(&db, b.size() * 4)); CHECK(hipMalloc(&dout, 4096 * 4));
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
    long long bad = 0, checked = 0;
    for (int tr = 0; tr < trials; tr++) {
        const int kind = tr % 4;   // 0: costs in [0, 2)   1: wide exponent range   2: tiny and zero values   3: large sums
        for (size_t i = 0; i < a.size(); i++) {
            const uint32_t r = rnd();
            float v;
            if (kind == 0) v = (float)(r & 0xffffff) / (float)(1 << 23);
            else if (kind == 1) { uint32_t bits = ((r % 60 + 97) << 23) | (rnd() & 0x7fffff); std::memcpy(&v, &bits, 4); }
            else if (kind == 2) { uint32_t bits = (r % 3 == 0) ? 0u : (((r % 20 + 1) << 23) | (rnd() & 0x7fffff)); std::memcpy(&v, &bits, 4); }
            else v = (float)(r & 0xffffff) * 37.0f;
            a[i] = v;
            b[i] = (rnd() % 5) ? 1.0f : 0.0f;
        }
        CHECK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        k_exact<<<1, 64>>>(da, db, dout, n);
        CHECK(hipMemcpy(got.data(), dout, 256 * 4, hipMemcpyDeviceToHost));
        for (int lane = 0; lane < 64; lane++) {
            const int k = lane >> 2, j = lane & 3;
            for (int i = 0; i < 4; i++) {
                volatile float c = 0.0f;
                for (int t = 0; t < n; t++) {
                    const float term = b[(size_t)t * 64 + 4 * k + j] != 0.0f ? a[(size_t)t * 64 + 4 * k + i] : 0.0f;
                    c = c + term;
                }
                want[lane * 4 + i] = c;
            }
        }
        for (int i = 0; i < 256; i++) {
            checked++;
            if (std::memcmp(&got[i], &want[i], 4) != 0) {
                if (bad < 10) std::printf("trial %d (kind %d) element %d: mfma %.9g (%08x)  sequential %.9g (%08x)\n", tr, kind, i, got[i],
                                          *(uint32_t *)&got[i], want[i], *(uint32_t *)&want[i]);
                bad++;
            }
        }
    }
    std::printf("exactness: %lld of %lld accumulators differ from the sequential fp32 sums (%d terms each, %d trials)\n", bad, checked, n, trials);

    // ---- rate
    rate<0>("one accumulator set, A from a register", 1, dout);
    rate<0>("one accumulator set, A from a register", 4, dout);
    rate<1>("two accumulator sets, A from registers", 1, dout);
    rate<1>("two accumulator sets, A from registers", 4, dout);
    rate<2>("one set, A by ds_read_b32 (16 in flight, counted waits)", 1, dout);
    rate<2>("one set, A by ds_read_b32 (16 in flight, counted waits)", 2, dout);
    rate<2>("one set, A by ds_read_b32 (16 in flight, counted waits)", 4, dout);
    rate<3>("two sets, A by ds_read_b32 (16 in flight, counted waits)", 1, dout);
    rate<3>("two sets, A by ds_read_b32 (16 in flight, counted waits)", 2, dout);
    rate<3>("two sets, A by ds_read_b32 (16 in flight, counted waits)", 4, dout);
    return bad ? 1 : 0;
}
