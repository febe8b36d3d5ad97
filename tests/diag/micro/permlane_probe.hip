// Diagnostic: prints what v_permlane16_swap / v_permlane32_swap and the DPP forms used by fht_wave do to lane ids (gfx950).
// hipcc --offload-arch=gfx950 -O2 tests/diag/micro/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    const int a = lane, b = 100 + lane;
    auto r16 = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
    auto r32 = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    out[lane] = (int)r16[0]; out[64 + lane] = (int)r16[1];
    out[128 + lane] = (int)r32[0]; out[192 + lane] = (int)r32[1];
    int t = __builtin_amdgcn_update_dpp(-1, a, 0x104, 0xf, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, a, 0x114, 0xf, 0xa, false);
    out[256 + lane] = t;                                                     // expect lane ^ 4
    out[320 + lane] = __builtin_amdgcn_update_dpp(-1, a, 0x128, 0xf, 0xf, false); // expect lane ^ 8
    out[384 + lane] = __builtin_amdgcn_update_dpp(-1, a, 0xB1, 0xf, 0xf, false);  // expect lane ^ 1
    out[448 + lane] = __builtin_amdgcn_update_dpp(-1, a, 0x4E, 0xf, 0xf, false);  // expect lane ^ 2
}
int main() {
    int* d; hipMalloc(&d, 512 * 4);
    k<<<1, 64>>>(d);
    int h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"p16swap[0]", "p16swap[1]", "p32swap[0]", "p32swap[1]", "xor4", "xor8", "xor1", "xor2"};
    for (int r = 0; r < 8; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %d", h[r * 64 + i]); printf("\n"); }
    return 0;
}
