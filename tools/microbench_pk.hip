// Calibration microbenchmark (measurement aid, not product): is packed fp32 (v_pk_fma_f32) a lever on gfx950?
// Aggregate issue cost of scalar vs packed fp32 FMA at 1, 2, 4 and 8 waves per SIMD (256 threads/block, 256 CUs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void chain(float *out, unsigned long long *stamps, int iters) {
    const float s = threadIdx.x * 1e-3f + 1.0f;
    float a = s, d = s + 1.f, e = s + 2.f, f = s + 3.f, g = s + 4.f, h = s + 5.f, k = s + 6.f, l = s + 7.f;
    const float b = 0.999f, c = 1e-3f;
    f2 A = {a, d}, D = {e, f}, E = {g, h}, F = {k, l};
    const f2 Bv = {b, b * 0.5f}, Cv = {c, c * 2.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 32; j++) {
#define SFMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c))  // (plain fmaf gets SLP-packed by the compiler)
            if (MODE == 0) { SFMA(a); SFMA(d); SFMA(e); SFMA(f); SFMA(g); SFMA(h); SFMA(k); SFMA(l); }  // 8 scalar FMAs
            if (MODE == 1) { A = __builtin_elementwise_fma(A, Bv, Cv); D = __builtin_elementwise_fma(D, Bv, Cv);
                             E = __builtin_elementwise_fma(E, Bv, Cv); F = __builtin_elementwise_fma(F, Bv, Cv); }  // 4 packed = 8 FMAs
            if (MODE == 2) { SFMA(a); SFMA(a); }                                              // dependent scalar, 2 FMAs
            if (MODE == 3) { A = __builtin_elementwise_fma(A, Bv, Cv); }                    // dependent packed, 2 FMAs
            if (MODE == 4) { A = A * Bv; D = D + Cv; E = E * Bv; F = F + Cv; }              // packed mul / add
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a + d + e + f + g + h + k + l + A.x + A.y + D.x + D.y + E.x + E.y + F.x + F.y;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks, int fmas_per_iter) {
    float *out; unsigned long long *st;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&st, blocks * 4 * sizeof(unsigned long long));
    const int iters = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    chain<MODE><<<blocks, 256>>>(out, st, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    chain<MODE><<<blocks, 256>>>(out, st, iters);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), st, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto v : h) cyc += v;
    cyc /= h.size();
    const double waves_per_simd = blocks / 256.0;
    // SIMD cycles per lane-FMA-pair of one wave64 instruction's worth of work: wave cycles / (FMAs per wave) * ... aggregate
    printf("%-30s %2.0f waves/SIMD: %8.1f us, wave %9.0f cycles, SIMD cycles per wave-FMA (aggregate) %.2f\n", name,
           waves_per_simd, ms * 1e3, cyc, cyc / ((double)iters * 32 * fmas_per_iter) / waves_per_simd);
    hipFree(out); hipFree(st);
}

int main() {
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0>("8 independent v_fma_f32", blocks, 8);
        run<1>("4 independent v_pk_fma_f32", blocks, 8);
        run<2>("dependent v_fma_f32", blocks, 2);
        run<3>("dependent v_pk_fma_f32", blocks, 2);
        run<4>("v_pk_mul_f32 / v_pk_add_f32", blocks, 8);
    }
    return 0;
}
