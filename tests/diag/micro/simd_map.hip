// Diagnostic: which SIMD does wave w of a 4-wave workgroup land on?  (k_scan's replay wave is wave 3 of every workgroup:
// if wave index -> SIMD is the same for every workgroup, the four replay waves of a CU share ONE SIMD.)
// hipcc --offload-arch=gfx950 -O2 simd_map.hip -o simd_map && ./simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4); // HW_REG_HW_ID bits 0..15
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
    }
    // keep the workgroup resident for a while so that several share a CU
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000) {}
}
int main() {
    const int nwg = 1024;
    unsigned* d; hipMalloc(&d, nwg * 4 * 4);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 20000, 0, d);
    std::vector<unsigned> h(nwg * 4);
    hipMemcpy(h.data(), d, nwg * 16, hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    for (int b = 0; b < nwg; ++b) for (int w = 0; w < 4; ++w) hist[w][(h[b * 4 + w] >> 4) & 3]++;
    for (int w = 0; w < 4; ++w) printf("wave %d -> simd counts: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 8; ++b) { printf("wg %d: cu %u se %u simds", b, (h[b*4] >> 8) & 15, (h[b*4] >> 13) & 7); for (int w = 0; w < 4; ++w) printf(" %u", (h[b*4+w] >> 4) & 3); printf("\n"); }
    return 0;
}
