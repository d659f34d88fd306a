// Measurement aid: which SIMD / CU / XCC does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID, gfx9 layout)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void probe(unsigned *out) {
    unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20); // XCC_ID (gfx940+)
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = id; out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    // keep the waves alive for a moment so that the whole grid is co-resident
    float a = threadIdx.x;
    for (int i = 0; i < 20000; i++) a = fmaf(a, 0.999f, 0.001f);
    if (a == 1234.5f) out[0] = 0;
}
int main() {
    const int blocks = 256;
    unsigned *d; hipMalloc(&d, blocks * 8 * 2 * sizeof(unsigned));
    probe<<<blocks, 512>>>(d); hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 16); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int same = 0, total = 0, hist[4][2] = {};
    for (int b = 0; b < blocks; b++) {
        if (b < 6) { printf("block %3d:", b); }
        for (int w = 0; w < 8; w++) {
            unsigned id = h[(b * 8 + w) * 2];
            unsigned simd = (id >> 4) & 3, cu = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7, wave = id & 15;
            if (b < 6) printf("  w%d:simd%u cu%u sh%u se%u slot%u xcc%u", w, simd, cu, sh, se, wave, h[(b * 8 + w) * 2 + 1] & 15);
            hist[simd][w >> 2]++;
            if (w < 4) { unsigned id2 = h[(b * 8 + w + 4) * 2]; total++; same += (((id2 >> 4) & 3) == simd) && (((id2 >> 8) & 0xff) == ((id >> 8) & 0xff)); }
        }
        if (b < 6) printf("\n");
    }
    printf("pairs (w, w+4) on the same SIMD of the same CU: %d of %d\n", same, total);
    for (int s = 0; s < 4; s++) printf("simd %d: %d waves w<4, %d waves w>=4\n", s, hist[s][0], hist[s][1]);
    return 0;
}
