// k_scan.hip — translation unit of the roofline kernel (scan.hpp) and its launcher.  gfx950 only.
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <mutex>

#include "launch.hpp"
#include "scan.hpp"

namespace rbq {

#ifndef RBQ_SCAN_TU2
StageProbes*& stage_probes() { static thread_local StageProbes* p = nullptr; return p; }
hipError_t LdsAttrCache::ensure(const void* fn, size_t lds, int device) {
    if (lds <= 48 * 1024) return hipSuccess; // default dynamic-LDS limit
    const int d = device & 15;
    if (lds <= set[d].load(std::memory_order_acquire)) return hipSuccess;
    static std::mutex mu; // one lock for all kernels: taken a handful of times per process
    std::lock_guard<std::mutex> g(mu);
    if (lds <= set[d].load(std::memory_order_relaxed)) return hipSuccess; // raised meanwhile by another caller
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) set[d].store(lds, std::memory_order_release);
    return e;
}
#endif

namespace {

template <int DT, int EX, int TR>
hipError_t launch_scan_r(const ScanParams& P, uint32_t nq, size_t lds, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    static LdsAttrCache attr;
    if (probe_stage(3, reinterpret_cast<const void*>(&k_scan<DT, EX, TR>), dim3(nq), kScanThreads, lds)) return hipSuccess;
    hipError_t e = attr.ensure(reinterpret_cast<const void*>(&k_scan<DT, EX, TR>), lds, device);
    if (e != hipSuccess) return e;
    if (ev0) hipExtLaunchKernelGGL((k_scan<DT, EX, TR>), dim3(nq), dim3(kScanThreads), lds, s, ev0, ev1, 0, P);
    else hipLaunchKernelGGL((k_scan<DT, EX, TR>), dim3(nq), dim3(kScanThreads), lds, s, P);
    return hipGetLastError();
}
// top_k <= 63: one register per lane of the replay wave holds the top-k (bag, or the exact heap after a distance tie);
// 64..128: two registers (the reference benchmarks top_k = 100); ..kTopKRegMax = 256: four; above that the exact heap in LDS
template <int DT, int EX>
hipError_t launch_scan_t(const ScanParams& P, uint32_t nq, size_t lds, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (P.top_k >= 64u && P.top_k <= 128u) return launch_scan_r<DT, EX, 2>(P, nq, lds, device, s, ev0, ev1);
    if (P.top_k > 128u && P.top_k <= kTopKRegMax) return launch_scan_r<DT, EX, 4>(P, nq, lds, device, s, ev0, ev1);
    return launch_scan_r<DT, EX, 1>(P, nq, lds, device, s, ev0, ev1);
}
template <int DT>
hipError_t launch_scan_d(const ScanParams& P, uint32_t nq, size_t lds, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if constexpr (DT == 0) {
        return launch_scan_t<0, 0>(P, nq, lds, device, s, ev0, ev1); // runtime dimension and ex_bits
    } else {
        switch (P.ex_bits) {
            case 0: return launch_scan_t<DT, 0>(P, nq, lds, device, s, ev0, ev1);
            case 2: return launch_scan_t<DT, 2>(P, nq, lds, device, s, ev0, ev1);
            default: return launch_scan_t<DT, 6>(P, nq, lds, device, s, ev0, ev1);
        }
    }
}

} // namespace

#ifndef RBQ_SCAN_TU2
hipError_t launch_scan_more_dims(const ScanParams& P, uint32_t nq, size_t lds, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1,
                                 bool* handled); // k_scan2.hip

hipError_t launch_scan(const ScanParams& P, uint32_t nq, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (P.wave_kernel && scanw_serves(P)) {
        const hipError_t e = launch_scanw(P, nq, device, s, ev0, ev1);
        if (e != hipErrorNotSupported) return e; // (not supported: that instantiation spills registers — k_scan serves the call)
    }
    const uint32_t D = P.D, Dc = P.Dc;
    const int nb = D == Dc ? scan_nb(D == 128 || D == 256 || D == 384 || D == 512 || D == 768 || D == 960 || D == 1024 || D == 1536 ? D : 0u) : 1;
    // (RBQ_SCAN_LDS_PAD: diagnostic — extra dynamic LDS per workgroup, occupancy experiments)
    static const size_t pad = [] { const char* e = std::getenv("RBQ_SCAN_LDS_PAD"); return e ? (size_t)std::atol(e) : (size_t)0; }();
    const size_t lds = scan_lds_bytes(Dc, D, P.ex_bits, P.top_k, P.heap_ws == nullptr, nb) + pad;
    if (D == Dc && D == 960) return launch_scan_d<960>(P, nq, lds, device, s, ev0, ev1);
    if (D == Dc && D == 768) return launch_scan_d<768>(P, nq, lds, device, s, ev0, ev1);
    if (D == Dc && D == 128) return launch_scan_d<128>(P, nq, lds, device, s, ev0, ev1);
    if (D == Dc) { // the other common padded dimensions (k_scan2.hip): 256, 384, 512, 1024, 1536
        bool handled = false;
        const hipError_t e = launch_scan_more_dims(P, nq, lds, device, s, ev0, ev1, &handled);
        if (handled) return e;
    }
    return launch_scan_d<0>(P, nq, lds, device, s, ev0, ev1); // any other dimension: runtime-dimension kernel
}
#else
// second translation unit of the same kernel template (compiled in parallel with the first): the compile-time dimension
// buys the rolling two-granule code window and immediate LUT offsets — the reference runs every multiple of 64 through one
// SIMD body (src/simd.rs:972-1014), so the common embedding sizes get the same treatment as 128 / 768 / 960
hipError_t launch_scan_more_dims(const ScanParams& P, uint32_t nq, size_t lds, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1,
                                 bool* handled) {
    *handled = true;
    switch (P.D) {
        case 256: return launch_scan_d<256>(P, nq, lds, device, s, ev0, ev1);
        case 384: return launch_scan_d<384>(P, nq, lds, device, s, ev0, ev1);
        case 512: return launch_scan_d<512>(P, nq, lds, device, s, ev0, ev1);
        case 1024: return launch_scan_d<1024>(P, nq, lds, device, s, ev0, ev1);
        case 1536: return launch_scan_d<1536>(P, nq, lds, device, s, ev0, ev1);
        default: *handled = false; return hipSuccess;
    }
}
#endif

} // namespace rbq
