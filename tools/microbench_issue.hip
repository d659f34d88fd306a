// Calibration microbenchmark (measurement aid, not product): issue cost per instruction of one wave's instruction stream on
// one SIMD -- v_fma_f32, v_pk_fma_f32, v_rsq_f32, v_fma_f64, s_mov/s_nop -- alone and with a second wave on the same SIMD,
// in independent chains (8 accumulators) and in one dependent chain.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/microbench_issue tools/microbench_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *stamps, int iters) {
    const int tid = threadIdx.x;
    float a0 = tid * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const float m = 0.999f, c = 1e-3f;
    const f2 pm = {0.999f, 0.998f}, pc = {1e-3f, 2e-3f};
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {  // 64 independent v_fma_f32 (8 chains)
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        }
        if (MODE == 1) {  // 64 dependent v_fma_f32
            REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(m), "v"(c));)
        }
        if (MODE == 2) {  // 64 independent v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));)
        }
        if (MODE == 3) {  // 64 dependent v_pk_fma_f32 (with the s_nop the hazard needs)
            REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n" : "+v"(p0) : "v"(pm), "v"(pc));)
        }
        if (MODE == 4) {  // 64 independent v_rsq_f32
            REP8(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                              "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
        if (MODE == 5) {  // 64 independent v_fma_f64 (4 chains)
            REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                              "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)m), "v"((double)c));)
        }
        if (MODE == 6) {  // 32 independent v_fma_f32 interleaved with 32 s_mov_b32
            REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n s_mov_b32 s20, 0x3f000001\n v_fma_f32 %1, %1, %4, %5\n s_mov_b32 s21, 0x3f000002\n"
                              "v_fma_f32 %2, %2, %4, %5\n s_mov_b32 s22, 0x3f000003\n v_fma_f32 %3, %3, %4, %5\n s_mov_b32 s23, 0x3f000004\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c) : "s20", "s21", "s22", "s23");)
        }
        if (MODE == 7) {  // 32 independent v_fma_f32 interleaved with 32 s_nop 0
            REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n s_nop 0\n v_fma_f32 %1, %1, %4, %5\n s_nop 0\n"
                              "v_fma_f32 %2, %2, %4, %5\n s_nop 0\n v_fma_f32 %3, %3, %4, %5\n s_nop 0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));)
        }
        if (MODE == 8) {  // 32 independent v_fma_f32 interleaved with 32 independent v_pk_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %4, %4, %10, %11\n v_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %5, %5, %10, %11\n"
                              "v_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %6, %6, %10, %11\n v_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %7, %7, %10, %11\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m), "v"(c), "v"(pm), "v"(pc));)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x + (float)(d0 + d1 + d2 + d3) == 123.456f) out[0] = a0;
    if ((tid & 63) == 0) stamps[(blockIdx.x * blockDim.x + tid) >> 6] = t1 - t0;
}

template <int MODE>
double run(int waves_per_simd) {
    const int blocks = 256, threads = 256 * waves_per_simd, waves = blocks * threads / 64, iters = 200;
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&st, waves * sizeof(unsigned long long));
    for (int r = 0; r < 3; r++) { k<MODE><<<blocks, threads>>>(out, st, iters); (void)hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), st, waves * sizeof(h[0]), hipMemcpyDeviceToHost);
    double cyc = 0; for (auto v : h) cyc += v;
    (void)hipFree(out); (void)hipFree(st);
    return cyc / waves / iters / 64.0;  // s_memtime ticks per instruction (64 per iteration)
}

int main() {
    printf("s_memtime ticks per instruction of a 64-instruction body (one wave per SIMD | two waves per SIMD); ticks run at the\n");
    printf("constant 100 MHz reference on this part if the figures are ~20x smaller than expected cycles\n");
#define ROW(M, what) printf("%-72s %7.3f | %7.3f\n", what, run<M>(1), run<M>(2));
    ROW(0, "v_fma_f32, 8 independent chains")
    ROW(1, "v_fma_f32, one dependent chain")
    ROW(2, "v_pk_fma_f32, 8 independent chains")
    ROW(3, "v_pk_fma_f32 + s_nop 0, one dependent chain (per pair of instructions x2)")
    ROW(4, "v_rsq_f32, 8 independent")
    ROW(5, "v_fma_f64, 4 independent chains")
    ROW(6, "v_fma_f32 / s_mov_b32 alternating")
    ROW(7, "v_fma_f32 / s_nop 0 alternating")
    ROW(8, "v_fma_f32 / v_pk_fma_f32 alternating")
    return 0;
}
