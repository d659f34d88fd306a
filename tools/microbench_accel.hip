// Measurement aid: cycles per RHS evaluation (sg::accel<3,3>) for a lone wave per SIMD, in the dependent pattern of the
// RK stages (next position depends on the previous acceleration), plus a hand-interleaved variant of the same math.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../space_gym_amd/csrc/sg_device.hpp"
using namespace sg;

// same arithmetic as sg::accel<3,3>, written "row by row" across the three planets and the thrust chain
__device__ __forceinline__ void accel_rows(const float (&cqx)[3], const float (&cqy)[3], float gm, float F, float C0, float S0,
                                           float om, float t, float X, float Y, float &ax, float &ay) {
    const float d = om * t, z = d * d;
    const float dx0 = cqx[0] - X, dx1 = cqx[1] - X, dx2 = cqx[2] - X, dy0 = cqy[0] - Y, dy1 = cqy[1] - Y, dy2 = cqy[2] - Y;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    const float r0 = fmaf(dx0, dx0, dy0 * dy0), r1 = fmaf(dx1, dx1, dy1 * dy1), r2 = fmaf(dx2, dx2, dy2 * dy2);
    ps = fmaf(ps, z, -1.6666654611e-1f); pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float i0 = rsq(r0), i1 = rsq(r1), i2 = rsq(r2);
    const float sd = fmaf(ps * z, d, d), cd = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const float w0 = gm * i0 * i0 * i0, w1 = gm * i1 * i1 * i1, w2 = gm * i2 * i2 * i2;
    const float c = fmaf(C0, cd, -S0 * sd), s = fmaf(S0, cd, C0 * sd);
    ax = fmaf(dx2, w2, fmaf(dx1, w1, fmaf(dx0, w0, -c * F)));
    ay = fmaf(dy2, w2, fmaf(dy1, w1, fmaf(dy0, w0, -s * F)));
}

template <int MODE>
__global__ __launch_bounds__(256) void chain(float *out, unsigned long long *stamps, int iters) {
    float cqx[3] = {0.5f + threadIdx.x * 1e-4f, -0.7f, 0.2f}, cqy[3] = {0.3f, 0.6f, -0.9f};
    float X = 0.0f, Y = 0.0f, ax, ay;
    const float gm = 0.0222f, F = 0.3f, C0 = 0.8f, S0 = 0.6f, om = 2.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float t = 0.004f * j;
            if (MODE == 0) accel<3, 3>(cqx, cqy, gm, F, C0, S0, om, t, X, Y, ax, ay);
            else accel_rows(cqx, cqy, gm, F, C0, S0, om, t, X, Y, ax, ay);
            X = fmaf(1e-3f, ax, X); Y = fmaf(1e-3f, ay, Y);  // the next evaluation depends on this one, as in rk_step
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = X + Y;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks) {
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, blocks * 256 * sizeof(float));
    (void)hipMalloc(&st, blocks * 4 * sizeof(unsigned long long));
    const int iters = 100;
    chain<MODE><<<blocks, 256>>>(out, st, iters);
    (void)hipDeviceSynchronize();
    chain<MODE><<<blocks, 256>>>(out, st, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), st, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto v : h) cyc += v;
    printf("%-28s blocks %5d: %.1f cycles per RHS evaluation (+2 FMAs)\n", name, blocks, cyc / h.size() / (iters * 16.0));
    (void)hipFree(out); (void)hipFree(st);
}

int main() {
    for (int blocks : {256, 1024}) {
        run<0>("sg::accel<3,3>", blocks);
        run<1>("hand-interleaved rows", blocks);
    }
    return 0;
}
