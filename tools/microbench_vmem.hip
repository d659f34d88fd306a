// Calibration microbenchmark (measurement aid, not product): what a vector-memory instruction costs the ISSUING wave when
// every SIMD of the chip runs two waves that alternate arithmetic with a few stores / loads per "step", as the rollout
// kernels do.  Per pattern: cycles per step with S memory instructions minus cycles per step with none.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/microbench_vmem tools/microbench_vmem.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

// MODE 0: no memory instruction; 1: row-per-lane 60-byte rows (3 x dwordx4 + dwordx3), 2: the same bytes as 4 coalesced
// dwordx4 stores (lane * 16 + k * 1024), 3: one dword store per lane (coalesced), 4: one byte store per lane,
// 5: 4 coalesced dwordx4 loads (consumed a step later), 6: 4 dwordx4 loads under an exec mask with 2 active lanes
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, const float *in, unsigned long long *stamps, int steps, int chain) {
    const int tid = threadIdx.x, wave = (blockIdx.x * 512 + tid) >> 6, lane = tid & 63;
    float x = tid * 1e-3f + 1.0f, acc = 0.f;
    v4f l0 = {0, 0, 0, 0}, l1 = l0, l2 = l0, l3 = l0;
    char *base = reinterpret_cast<char *>(out) + (size_t)wave * 64 * 64 * 4;   // 16 KiB per wave
    const char *ibase = reinterpret_cast<const char *>(in) + (size_t)wave * 64 * 64 * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s++) {
        for (int j = 0; j < chain; j++) x = fmaf(x, 0.999f, 1e-3f);  // the step's arithmetic (dependent chain)
        const v4f v = {x, x + 1, x + 2, x + 3};
        if (MODE == 1) {
            float *row = reinterpret_cast<float *>(base + lane * 60);
            *reinterpret_cast<v4f *>(row) = v; *reinterpret_cast<v4f *>(row + 4) = v; *reinterpret_cast<v4f *>(row + 8) = v;
            row[12] = x; row[13] = x; row[14] = x;
        }
        if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<v4f *>(base + q * 1024 + lane * 16) = v;
        }
        if (MODE == 3) *reinterpret_cast<float *>(base + lane * 4) = x;
        if (MODE == 4) *reinterpret_cast<unsigned char *>(base + lane) = (unsigned char)s;
        if (MODE == 9 && lane < 2) *reinterpret_cast<v4f *>(base + lane * 16) = v;
        if (MODE == 10 && lane < 2) { acc += l0.x; l0 = *reinterpret_cast<const v4f *>(ibase + lane * 16 + ((s & 3) << 12)); }
        if (MODE == 11) { if (__any(lane < 2 && s >= 0)) { acc += l0.x + l1.y + l2.z + l3.w;   // wave-uniform branch, all lanes load
            l0 = *reinterpret_cast<const v4f *>(ibase + lane * 16 + ((s & 3) << 12));
            l1 = *reinterpret_cast<const v4f *>(ibase + 1024 + lane * 16 + ((s & 3) << 12));
            l2 = *reinterpret_cast<const v4f *>(ibase + 2048 + lane * 16 + ((s & 3) << 12));
            l3 = *reinterpret_cast<const v4f *>(ibase + 3072 + lane * 16 + ((s & 3) << 12)); } }
        if (MODE == 5 || (MODE == 6 && lane < 2) || (MODE == 7 && lane < 32) || (MODE == 8 && (lane & 1))) {
            acc += l0.x + l1.y + l2.z + l3.w;
            l0 = *reinterpret_cast<const v4f *>(ibase + lane * 16 + ((s & 3) << 12));
            l1 = *reinterpret_cast<const v4f *>(ibase + 1024 + lane * 16 + ((s & 3) << 12));
            l2 = *reinterpret_cast<const v4f *>(ibase + 2048 + lane * 16 + ((s & 3) << 12));
            l3 = *reinterpret_cast<const v4f *>(ibase + 3072 + lane * 16 + ((s & 3) << 12));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (x + acc == 123.456f) out[0] = x;
    if (lane == 0) stamps[wave] = t1 - t0;
}

template <int MODE>
double run(int steps, int chain) {
    const int blocks = 256, waves = blocks * 8;
    float *out, *in;
    unsigned long long *st;
    (void)hipMalloc(&out, (size_t)waves * 16384 + 4096); (void)hipMalloc(&in, (size_t)waves * 16384 + 4096);
    (void)hipMemset(in, 0, (size_t)waves * 16384 + 4096);
    (void)hipMalloc(&st, waves * sizeof(unsigned long long));
    for (int r = 0; r < 2; r++) { k<MODE><<<blocks, 512>>>(out, in, st, steps, chain); (void)hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), st, waves * sizeof(h[0]), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto v : h) cyc += v;
    (void)hipFree(out); (void)hipFree(in); (void)hipFree(st);
    return cyc / waves / steps;
}

int main() {
    const int steps = 400;
    for (int chain : {250, 1000}) {
        const double base = run<0>(steps, chain);
        printf("arithmetic only (%d dependent FMAs per step, two waves per SIMD): %.0f cycles per step\n", chain, base);
        printf("  + obs row per lane (3 x dwordx4 + 3 dwords, 60-byte stride)  %+7.0f cycles per step\n", run<1>(steps, chain) - base);
        printf("  + the same bytes as 4 coalesced dwordx4 stores               %+7.0f\n", run<2>(steps, chain) - base);
        printf("  + one coalesced dword store                                  %+7.0f\n", run<3>(steps, chain) - base);
        printf("  + one byte store                                             %+7.0f\n", run<4>(steps, chain) - base);
        printf("  + 4 coalesced dwordx4 loads, used a step later               %+7.0f\n", run<5>(steps, chain) - base);
        printf("  + 4 dwordx4 loads by 2 lanes, used a step later              %+7.0f\n", run<6>(steps, chain) - base);
        printf("  + 4 dwordx4 loads by lanes 0..31                             %+7.0f\n", run<7>(steps, chain) - base);
        printf("  + 4 dwordx4 loads by odd lanes                               %+7.0f\n", run<8>(steps, chain) - base);
        printf("  + 1 dwordx4 store by 2 lanes                                 %+7.0f\n", run<9>(steps, chain) - base);
        printf("  + 1 dwordx4 load by 2 lanes                                  %+7.0f\n", run<10>(steps, chain) - base);
        printf("  + 4 dwordx4 loads by all lanes under a wave-uniform branch   %+7.0f\n", run<11>(steps, chain) - base);
    }
    return 0;
}
