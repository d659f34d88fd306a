// Calibration microbenchmark (measurement aid, not product): shader clock under this launch geometry, and the issue
// cost of dependent / independent fp32 FMA chains, rsq, f64 FMA, u32 mul_hi at 1 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void chain(float *out, unsigned long long *stamps, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = 0.999f, c = 1e-3f, d = a + 1.0f, e = a + 2.0f, f = a + 3.0f;
    double da = a, db = 0.999, dc = 1e-3;
    unsigned ua = threadIdx.x + 12345u, ub = 0xD2511F53u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++) {
            if (MODE == 0) a = fmaf(a, b, c);                       // dependent fp32 FMA chain
            if (MODE == 1) { a = fmaf(a, b, c); d = fmaf(d, b, c); e = fmaf(e, b, c); f = fmaf(f, b, c); }  // 4 independent chains
            if (MODE == 2) a = __builtin_amdgcn_rsqf(a) + 1.0f;     // rsq + add, dependent
            if (MODE == 3) da = fma(da, db, dc);                    // dependent f64 FMA
            if (MODE == 4) ua = __umulhi(ua, ub) ^ ua;              // mul_hi + xor, dependent
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a + d + e + f + (float)da + (float)ua;
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int MODE>
void run(const char *name, int blocks, int ops_per_iter) {
    float *out; unsigned long long *st;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&st, blocks * 4 * 2 * sizeof(unsigned long long));
    const int iters = 200;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    chain<MODE><<<blocks, 256>>>(out, st, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    chain<MODE><<<blocks, 256>>>(out, st, iters);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), st, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int w = 0; w < blocks * 4; w++) { cyc += h[2 * w]; real += h[2 * w + 1]; }
    cyc /= blocks * 4; real /= blocks * 4;
    double ghz = cyc / (real * 10.0);  // s_memrealtime ticks at 100 MHz
    printf("%-34s blocks %5d: %7.1f us, wave: %9.0f cycles, clock %.2f GHz, %.2f cycles per instruction-group\n",
           name, blocks, ms * 1e3, cyc, ghz, cyc / (double)(iters * 64) / ops_per_iter * ops_per_iter);
    hipFree(out); hipFree(st);
}

int main() {
    for (int blocks : {256, 2048}) {
        run<0>("dependent v_fma_f32", blocks, 1);
        run<1>("4 independent v_fma_f32 chains", blocks, 4);
        run<2>("v_rsq_f32 + v_add (dependent)", blocks, 2);
        run<3>("dependent v_fma_f64", blocks, 1);
        run<4>("v_mul_hi_u32 + xor (dependent)", blocks, 2);
    }
    return 0;
}
