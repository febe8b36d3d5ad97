// k_scanw.hip — translation unit of the wave-per-query scan (scanw.hpp) and its launcher.  gfx950 only.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "launch.hpp"
#include "scanw.hpp"

namespace rbq {

namespace {

constexpr int kScanwMaxScratch = 16; // bytes per lane

template <int DT, int EX, int TR>
hipError_t launch_scanw_r(const ScanParams& P, uint32_t nq, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const size_t lds = scanw_lds_bytes(P.Dc, P.D, P.ex_bits, TR); // (always below the default 48 KB dynamic limit: D <= 2048)
    (void)device;
    // An instantiation that spills registers is not used: its scratch traffic travels through the same in-order vector-memory
    // queue as the loads the wave is waiting for (measured: 190 bytes of scratch per lane made the kernel 2.5x slower).  The
    // code object says how much it got; k_scan serves the call instead (identical results).
    static const int spills = [] {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_scanw<DT, EX, TR>)) != hipSuccess) { (void)hipGetLastError(); return 1; }
        return fa.localSizeBytes > (size_t)kScanwMaxScratch ? 1 : 0;
    }();
    if (spills) return hipErrorNotSupported;
    if (probe_stage(3, reinterpret_cast<const void*>(&k_scanw<DT, EX, TR>), dim3(nq), 64, lds)) return hipSuccess;
    if (ev0) hipExtLaunchKernelGGL((k_scanw<DT, EX, TR>), dim3(nq), dim3(64), lds, s, ev0, ev1, 0, P);
    else hipLaunchKernelGGL((k_scanw<DT, EX, TR>), dim3(nq), dim3(64), lds, s, P);
    return hipGetLastError();
}
// top_k <= 63: one register per lane holds the top-k (sorted run, or the exact heap after a distance tie); 64..127: two
// registers (RankRun; the reference benchmarks top_k = 100); ..255: four
template <int DT, int EX>
hipError_t launch_scanw_t(const ScanParams& P, uint32_t nq, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (P.top_k < 64u) return launch_scanw_r<DT, EX, 1>(P, nq, device, s, ev0, ev1);
    if (P.top_k < 128u) return launch_scanw_r<DT, EX, 2>(P, nq, device, s, ev0, ev1);
    return launch_scanw_r<DT, EX, 4>(P, nq, device, s, ev0, ev1);
}
template <int DT>
hipError_t launch_scanw_d(const ScanParams& P, uint32_t nq, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if constexpr (DT == 0) {
        return launch_scanw_t<0, 0>(P, nq, device, s, ev0, ev1); // runtime dimension and ex_bits
    } else {
        switch (P.ex_bits) {
            case 0: return launch_scanw_t<DT, 0>(P, nq, device, s, ev0, ev1);
            case 2: return launch_scanw_t<DT, 2>(P, nq, device, s, ev0, ev1);
            default: return launch_scanw_t<DT, 6>(P, nq, device, s, ev0, ev1);
        }
    }
}

} // namespace

#ifndef RBQ_SCANW_TU2
hipError_t launch_scanw_more_dims(const ScanParams& P, uint32_t nq, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1,
                                  bool* handled); // k_scanw2.hip

bool scanw_serves(const ScanParams& P) {
    return !P.mstg && !P.heap_ws && P.top_k >= 1u && P.top_k <= 255u && P.Dc <= 2048u;
}
hipError_t launch_scanw(const ScanParams& P, uint32_t nq, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const uint32_t D = P.D, Dc = P.Dc;
    if (D == Dc && D == 960) return launch_scanw_d<960>(P, nq, device, s, ev0, ev1);
    if (D == Dc && D == 768) return launch_scanw_d<768>(P, nq, device, s, ev0, ev1);
    if (D == Dc) {
        bool handled = false;
        const hipError_t e = launch_scanw_more_dims(P, nq, device, s, ev0, ev1, &handled);
        if (handled) return e;
    }
    return launch_scanw_d<0>(P, nq, device, s, ev0, ev1); // any other dimension: runtime-dimension kernel
}
#else
hipError_t launch_scanw_more_dims(const ScanParams& P, uint32_t nq, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1,
                                  bool* handled) {
    *handled = true;
    switch (P.D) {
        case 128: return launch_scanw_d<128>(P, nq, device, s, ev0, ev1);
        case 256: return launch_scanw_d<256>(P, nq, device, s, ev0, ev1);
        case 384: return launch_scanw_d<384>(P, nq, device, s, ev0, ev1);
        case 512: return launch_scanw_d<512>(P, nq, device, s, ev0, ev1);
        case 1024: return launch_scanw_d<1024>(P, nq, device, s, ev0, ev1);
        case 1536: return launch_scanw_d<1536>(P, nq, device, s, ev0, ev1);
        default: *handled = false; return hipSuccess;
    }
}
#endif

} // namespace rbq
