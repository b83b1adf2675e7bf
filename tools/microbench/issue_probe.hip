// What does a SIMD of this chip issue per cycle when every wave runs the aggregation kernels' inner pattern?
// One workgroup of 16 waves per CU (4 waves per SIMD, as the fused aggregation kernel runs), every wave a loop of
// N instructions of one kind; prints cycles per wave-instruction per SIMD (4.0 = one instruction per quad-cycle).
// Build: hipcc -O3 --offload-arch=gfx950 issue_probe.hip -o issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP8(x) x x x x x x x x

// MODE 0: one dependent chain of v_pk_add_f32          1: two interleaved chains of v_pk_add_f32
//      2: one dependent chain of v_add_f32             3: four independent chains of v_pk_add_f32
//      4: ds_read_b64 only (16 in flight, then wait)   5: 16 x ds_read_b64, then 16 x (2 pk adds) with counted waits
//      7: pattern 5 plus 16 SALU instructions in between
//      8: pattern 5 plus 8 taken branches (compare + s_cbranch over one instruction)     9: the same, not taken
//     10: pattern 5 plus 16 v_cndmask_b32     11: pattern 5 with s_waitcnt lgkmcnt(0) before the adds (as the compiler emits)
template <int MODE>
__global__ void __launch_bounds__(1024) k_probe(float *out, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    f2 a = {1.f, 2.f}, b = {3.f, 4.f}, c = {5.f, 6.f}, d = {7.f, 8.f};
    const f2 v = {(float)lane, 1.f};
    float s = lane;
    uint32_t addr = (uint32_t)((threadIdx.x & 1023) * 8);
    lds[threadIdx.x * 2] = 1.f;
    lds[threadIdx.x * 2 + 1] = 2.f;
    __syncthreads();
    uint32_t sacc = 0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            asm volatile(REP16("v_pk_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(v));
        } else if (MODE == 1) {
            asm volatile(REP16("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n") : "+v"(a), "+v"(b) : "v"(v));
        } else if (MODE == 2) {
            asm volatile(REP16("v_add_f32 %0, %0, %1\n") : "+v"(s) : "v"(v.x));
        } else if (MODE == 3) {
            asm volatile(REP16("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(v));
        } else if (MODE == 4) {
            f2 r[16];
#pragma unroll
            for (int u = 0; u < 16; u++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[u]) : "v"(addr), "n"(u * 512));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 16; u++) asm volatile("" : : "v"(r[u]));
        } else if (MODE == 5 || MODE == 7 || MODE == 8 || MODE == 9 || MODE == 10 || MODE == 11) {
            f2 r[16];
#pragma unroll
            for (int u = 0; u < 16; u++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[u]) : "v"(addr), "n"(u * 512));
            if (MODE == 7) asm volatile(REP16("s_add_u32 %0, %0, 1\n") : "+s"(sacc));
            if (MODE == 8) asm volatile(REP8("s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n 1:\n") : "+s"(sacc) : : "scc");
            if (MODE == 9) asm volatile(REP8("s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n 1:\n") : "+s"(sacc) : : "scc");
            if (MODE == 10) asm volatile(REP16("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(s) : "v"(v.x) : );
            if (MODE == 11) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 16; u++) {
                asm volatile("s_waitcnt lgkmcnt(%3)\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n"
                             : "+v"(a), "+v"(b) : "v"(r[u]), "n"(15 - u));
            }
        }
    }
    a += b + c + d;
    if (a.x + a.y + s + (float)sacc == -1.f) out[0] = a.x;   // never true
}

template <int MODE>
static void run(const char *name, int per_iter, int cus, float *out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void *)k_probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    k_probe<MODE><<<cus, 1024, 96 * 1024>>>(out, 100);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_probe<MODE><<<cus, 1024, 96 * 1024>>>(out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    int khz = 0;
    CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double cycles = ms * 1e-3 * khz * 1e3;
    const double instr_per_simd = (double)iters * per_iter * 4;   // 4 waves per SIMD
    std::printf("%-58s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %d MHz)\n", name, ms, cycles / instr_per_simd, khz / 1000);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out;
    CHECK(hipMalloc(&out, 256));
    run<0>("v_pk_add_f32, one dependent chain", 16, cus, out);
    run<1>("v_pk_add_f32, two interleaved chains", 32, cus, out);
    run<3>("v_pk_add_f32, four interleaved chains", 64, cus, out);
    run<2>("v_add_f32, one dependent chain", 16, cus, out);
    run<4>("ds_read_b64 x16, wait for all", 16, cus, out);
    run<5>("ds_read_b64 x16 + 32 v_pk_add_f32 (counted waits)", 48, cus, out);
    run<7>("the same + 16 s_add_u32", 64, cus, out);
    run<8>("the same + 8 x (s_cmp, taken s_cbranch)", 64, cus, out);
    run<9>("the same + 8 x (s_cmp, untaken s_cbranch, s_add)", 72, cus, out);
    run<10>("the same + 16 v_cndmask_b32", 64, cus, out);
    run<11>("the same, one wait for all 16 reads before the adds", 48, cus, out);
    return 0;
}
