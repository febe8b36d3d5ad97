// Stand-alone check of the observation recorded in rank_mfma.hpp: do waves that execute the gfx950 double-rate
// v_mfma_f32_32x32x16_bf16 disturb division/rounding results of OTHER kernels resident on the same SIMDs?
//   victim: every thread quantises 16 values the way k_prep does (IEEE divide + roundf), results to memory
//   noise : 256 workgroups x 4 waves of back-to-back MFMAs (x16 form, or the K=8 form with -DUSE_X8)
// The victim's output of a quiet launch is the reference; launches beside the noise kernel are compared with it.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/repro/mfma_x16_coexec.hip -o gpurun_out/mfma_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void noise(float* sink, int iters) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(float)(threadIdx.x % 7 + k); b[k] = (__bf16)(float)(threadIdx.x % 5 - k); }
    for (int i = 0; i < iters; ++i) {
#ifdef USE_X8
        const s16x4 a0 = {(short)i, 1, 2, 3}, b0 = {1, (short)i, 2, 3};
        acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(b0, a0, acc, 0, 0, 0);
#else
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#endif
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) sink[0] = s; // keep the loop
}

__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, float vl, float delta, unsigned* __restrict__ out, int reps) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    float l[16];
    for (int j = 0; j < 16; ++j) l[j] = in[(size_t)t * 16 + j];
    unsigned w[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        w[0] = w[1] = w[2] = w[3] = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = roundf((l[j] - vl) / delta);
            v = v >= 0.0f ? v : 0.0f;
            v = v > 255.0f ? 255.0f : v;
            w[j >> 2] |= (unsigned)v << (8 * (j & 3));
        }
        asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
        __syncthreads();
    }
    for (int k = 0; k < 4; ++k) out[(size_t)t * 4 + k] = w[k];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    const int nwg = 1024, nt = nwg * 256;
    std::vector<float> h(nt * 16);
    srand(1);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 40.0f - 20.0f;
    float *d_in, *d_sink; unsigned *d_out;
    CK(hipMalloc(&d_in, h.size() * 4)); CK(hipMalloc(&d_sink, 64)); CK(hipMalloc(&d_out, (size_t)nt * 16));
    CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    std::vector<unsigned> ref(nt * 4), got(nt * 4);
    hipLaunchKernelGGL(victim, dim3(nwg), dim3(256), 0, s1, d_in, -20.0f, 40.0f / 255.0f, d_out, 20);
    CK(hipStreamSynchronize(s1));
    CK(hipMemcpy(ref.data(), d_out, ref.size() * 4, hipMemcpyDeviceToHost));
    long bad_launches = 0, bad_words = 0;
    for (int it = 0; it < launches; ++it) {
        CK(hipMemsetAsync(d_out, 0, (size_t)nt * 16, s1));
        hipLaunchKernelGGL(noise, dim3(256), dim3(256), 0, s2, d_sink, 4000);
        hipLaunchKernelGGL(victim, dim3(nwg), dim3(256), 0, s1, d_in, -20.0f, 40.0f / 255.0f, d_out, 20);
        hipLaunchKernelGGL(noise, dim3(256), dim3(256), 0, s2, d_sink, 4000);
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
        long nb = 0;
        for (size_t i = 0; i < got.size(); ++i) nb += got[i] != ref[i];
        if (nb) { ++bad_launches; bad_words += nb; if (bad_launches <= 3) printf("launch %d: %ld words differ\n", it, nb); }
    }
    printf("launches %d, with mismatches %ld, words differing %ld\n", launches, bad_launches, bad_words);
    return 0;
}
