// Calibration microbenchmark (measurement aid, not product): issue cost per VALU instruction of ONE wave per SIMD as a
// function of the number of interleaved independent dependency chains (1..4, 8), for v_fma_f32, v_pk_fma_f32 and a
// v_rsq_f32 + 3 dependent multiplies pattern (the per-planet chain of the RHS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define SFMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c))
#define PFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(B2), "v"(C2))
#define RSQ(x) asm volatile("v_rsq_f32 %0, %0" : "+v"(x))
#define MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b))
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void chain(float *out, unsigned long long *stamps, int iters) {
    const float s = threadIdx.x * 1e-3f + 1.0f, b = 0.999f, c = 1e-3f;
    float x0 = s, x1 = s + 1, x2 = s + 2, x3 = s + 3, x4 = s + 4, x5 = s + 5, x6 = s + 6, x7 = s + 7;
    f2 p0 = {s, s}, p1 = {s + 1, s}, p2 = {s + 2, s}, p3 = {s + 3, s};
    const f2 B2 = {b, b}, C2 = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (MODE == 1) { SFMA(x0); }
            if (MODE == 2) { SFMA(x0); SFMA(x1); }
            if (MODE == 3) { SFMA(x0); SFMA(x1); SFMA(x2); }
            if (MODE == 4) { SFMA(x0); SFMA(x1); SFMA(x2); SFMA(x3); }
            if (MODE == 8) { SFMA(x0); SFMA(x1); SFMA(x2); SFMA(x3); SFMA(x4); SFMA(x5); SFMA(x6); SFMA(x7); }
            if (MODE == 11) { PFMA(p0); }
            if (MODE == 12) { PFMA(p0); PFMA(p1); }
            if (MODE == 13) { PFMA(p0); PFMA(p1); PFMA(p2); }
            if (MODE == 14) { PFMA(p0); PFMA(p1); PFMA(p2); PFMA(p3); }
            if (MODE == 21) { RSQ(x0); MUL(x0); MUL(x0); MUL(x0); }                                   // one planet chain
            if (MODE == 23) { RSQ(x0); RSQ(x1); RSQ(x2); MUL(x0); MUL(x1); MUL(x2); MUL(x0); MUL(x1); MUL(x2); MUL(x0); MUL(x1); MUL(x2); }  // three, interleaved
            if (MODE == 24) { RSQ(x0); MUL(x0); MUL(x0); MUL(x0); RSQ(x1); MUL(x1); MUL(x1); MUL(x1); RSQ(x2); MUL(x2); MUL(x2); MUL(x2); }  // three, one after the other
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.x + p2.x + p3.x + p0.y + p1.y + p2.y + p3.y;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE>
void run(const char *name, int per_iter) {
    const int blocks = 256, iters = 400;
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, blocks * 256 * sizeof(float)); (void)hipMalloc(&st, blocks * 4 * sizeof(unsigned long long));
    chain<MODE><<<blocks, 256>>>(out, st, iters); (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a); chain<MODE><<<blocks, 256>>>(out, st, iters); (void)hipEventRecord(b); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), st, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double cyc = 0; for (auto v : h) cyc += v; cyc /= h.size();
    printf("%-44s %6.2f cycles per instruction (%.1f us)\n", name, cyc / ((double)iters * 16 * per_iter), ms * 1e3);
    (void)hipFree(out); (void)hipFree(st);
}
int main() {
    run<1>("v_fma_f32, 1 chain", 1); run<2>("v_fma_f32, 2 chains interleaved", 2); run<3>("v_fma_f32, 3 chains", 3);
    run<4>("v_fma_f32, 4 chains", 4); run<8>("v_fma_f32, 8 chains", 8);
    run<11>("v_pk_fma_f32, 1 chain", 1); run<12>("v_pk_fma_f32, 2 chains", 2); run<13>("v_pk_fma_f32, 3 chains", 3); run<14>("v_pk_fma_f32, 4 chains", 4);
    run<21>("rsq + 3 mul, 1 chain", 4); run<23>("rsq + 3 mul, 3 chains interleaved", 12); run<24>("rsq + 3 mul, 3 chains back to back", 12);
    return 0;
}
