// k_query.hip — translation unit of the per-batch query stages in front of the scan: k_prep / k_prep_wave
// (rotate + constants + LUT), the centroid-ranking GEMMs, probe selection and the MSTG probe list.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "launch.hpp"
#include "query_kernels.hpp"
#include "rank_mfma.hpp"
#include "latency.hpp"

namespace rbq {

namespace {
uint32_t next_pow2(uint32_t x) { uint32_t p = 1; while (p < x) p <<= 1; return p; }
} // namespace

hipError_t launch_prep(const PrepParams& p, int device, hipStream_t s) {
    if (p.rotator == 0 /* matrix: O(D^2) per query */ || p.wg_prep) {
        static LdsAttrCache attr;
        const size_t lds = (size_t)p.D * 4 * 2;
        if (probe_stage(0, reinterpret_cast<const void*>(&k_prep), dim3(p.nq), kThreads, lds)) return hipSuccess;
        hipError_t e = attr.ensure(reinterpret_cast<const void*>(&k_prep), lds, device);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_prep, dim3(p.nq), dim3(kThreads), lds, s, p.queries, p.dim, p.D, p.Dc, p.rotator, p.rot_blob, p.trunc,
                           p.fac, p.ex_bits, p.rot, p.lut, p.consts, p.rot_hi, p.rot_lo);
    } else { // FHT-Kac / identity: one wave per query
        static LdsAttrCache attr;
        const uint32_t qpw = kThreads / 64;
        const size_t lds = (size_t)p.D * 4 * 2 * qpw + p.D / 2;
        if (probe_stage(0, reinterpret_cast<const void*>(&k_prep_wave), dim3((p.nq + qpw - 1) / qpw), kThreads, lds)) return hipSuccess;
        hipError_t e = attr.ensure(reinterpret_cast<const void*>(&k_prep_wave), lds, device); // 66.5 KB at D = 2048
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_prep_wave, dim3((p.nq + qpw - 1) / qpw), dim3(kThreads), lds, s, p.queries, p.nq, p.dim, p.D, p.Dc,
                           p.rotator, p.rot_blob, p.trunc, p.fac, p.ex_bits, p.rot, p.lut, p.consts, p.rot_hi, p.rot_lo);
    }
    return hipGetLastError();
}

// latency-first front of a small call (latency.hpp): rotation + constants + LUT + the exact canonical score of every list, one launch
hipError_t launch_lat_front(const PrepParams& p, const RankParams& r, int device, hipStream_t s) {
    static LdsAttrCache attr;
    LatFrontParams P;
    P.queries = p.queries; P.nq = p.nq; P.dim = p.dim; P.D = p.D; P.Dc = p.Dc; P.rotator = p.rotator; P.rot_blob = p.rot_blob;
    P.trunc = p.trunc; P.fac = p.fac; P.ex_bits = p.ex_bits; P.rot = p.rot; P.lut = p.lut; P.consts = p.consts;
    P.cent = r.cent; P.nlist = r.nlist; P.metric = r.metric; P.scores = r.scores;
    P.rot_hi = p.rot_hi; P.rot_lo = p.rot_lo; P.scorers = (r.scores && !r.ksplit) ? (r.nlist + kLatLists - 1) / kLatLists : 0u;
    P.zero_scores = r.ksplit > 1 ? r.scores : nullptr; // (split-K ranking GEMM behind this preparation: its parts are added to a zeroed row)
    const dim3 grid(lat_front_grid(P.scorers, p.nq));
    const size_t lds = (size_t)p.D * 4 * 2 + p.D / 2;
    if (probe_stage(0, reinterpret_cast<const void*>(&k_lat_front), grid, kThreads, lds)) return hipSuccess;
    hipError_t e = attr.ensure(reinterpret_cast<const void*>(&k_lat_front), lds, device);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_lat_front, grid, dim3(kThreads), lds, s, P);
    return hipGetLastError();
}

hipError_t launch_rank_exact(const RankParams& p, hipStream_t s) {
    dim3 grid((p.nlist + 31) / 32, (p.nq + 31) / 32);
    if (probe_stage(1, p.metric == 0 ? reinterpret_cast<const void*>(&k_rank_scores<0>) : reinterpret_cast<const void*>(&k_rank_scores<1>), grid, kThreads, 0)) return hipSuccess;
    if (p.metric == 0) hipLaunchKernelGGL(k_rank_scores<0>, grid, dim3(kThreads), 0, s, p.rot, p.cent, p.nq, p.nlist, p.D, p.scores);
    else hipLaunchKernelGGL(k_rank_scores<1>, grid, dim3(kThreads), 0, s, p.rot, p.cent, p.nq, p.nlist, p.D, p.scores);
    return hipGetLastError();
}

namespace {
// big problems: 128x128 tiles, 8 waves (each 64x32) — the tile traffic of the 4-wave form with twice the waves to
// hide the staging behind the MFMAs; small ones: 64x64 tiles, 4 waves
template <int M, int TM, int TN, int WM, int WN>
hipError_t launch_rank_split(const RankParams& p, dim3 grid, int device, hipStream_t s) {
    static LdsAttrCache attr;
    const size_t lds = (size_t)(2 * 32 * TM * WM + 2 * 32 * TN * WN) * 80 * 2; // two slabs of 32, rows of 80 B
    if (probe_stage(1, reinterpret_cast<const void*>(&k_rank_bf16_db<M, TM, TN, WM, WN>), grid, 64 * WM * WN, lds)) return hipSuccess;
    hipError_t e = attr.ensure(reinterpret_cast<const void*>(&k_rank_bf16_db<M, TM, TN, WM, WN>), lds, device);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_rank_bf16_db<M, TM, TN, WM, WN>), grid, dim3(64 * WM * WN), lds, s, p.rot_hi, p.rot_lo, p.cent_hi, p.cent_lo,
                       p.consts, p.cnorm2, p.nq, p.nlist, p.D, p.scores);
    return hipGetLastError();
}
template <int M, int TW>
hipError_t launch_rank_f32(const RankParams& p, dim3 grid, hipStream_t s) {
    if (probe_stage(1, reinterpret_cast<const void*>(&k_rank_mfma<M, TW>), grid, 256, 0)) return hipSuccess;
    hipLaunchKernelGGL((k_rank_mfma<M, TW>), grid, dim3(256), 0, s, p.rot, p.cent, p.consts, p.cnorm2, p.nq, p.nlist, p.D, p.scores);
    return hipGetLastError();
}
} // namespace

hipError_t launch_rank_gemm(const RankParams& p, int device, hipStream_t s) {
    if (p.split && p.wide) {
        // very large problems (cfg5: 16 384 x 65 536): 128 x 256 tiles, each wave a 64 x 64 sub-tile — 8 LDS fragment reads per 12 MFMAs
        // instead of 6 per 6, one workgroup (123 KB of LDS) per compute unit
        dim3 grid((p.nlist + 255) / 256, (p.nq + 127) / 128);
        return p.metric == 0 ? launch_rank_split<0, 2, 2, 2, 4>(p, grid, device, s) : launch_rank_split<1, 2, 2, 2, 4>(p, grid, device, s);
    }
    const uint32_t T = p.big ? 128u : 64u;
    dim3 grid((p.nlist + T - 1) / T, (p.nq + T - 1) / T, p.split && p.ksplit > 1 ? p.ksplit : 1u); // (z: split-K, the row zeroed by the preparation)
    if (p.split) {
        if (p.metric == 0) return p.big ? launch_rank_split<0, 2, 1, 2, 4>(p, grid, device, s) : launch_rank_split<0, 1, 1, 2, 2>(p, grid, device, s);
        return p.big ? launch_rank_split<1, 2, 1, 2, 4>(p, grid, device, s) : launch_rank_split<1, 1, 1, 2, 2>(p, grid, device, s);
    }
    if (p.metric == 0) return p.big ? launch_rank_f32<0, 2>(p, grid, s) : launch_rank_f32<0, 1>(p, grid, s);
    return p.big ? launch_rank_f32<1, 2>(p, grid, s) : launch_rank_f32<1, 1>(p, grid, s);
}

uint32_t select_exact_np2(uint32_t nprobe) { return next_pow2(nprobe); }

hipError_t launch_select_exact(const SelectParams& p, int device, hipStream_t s, uint64_t* key_window) {
    const uint32_t np2 = next_pow2(p.nprobe);
    // key_window != null: [nq][np2] u64 in global memory (nprobe > kNprobeMax)
    const size_t lds = (key_window ? 0 : (size_t)np2 * 8) + (size_t)p.D * 4 + kThreads * 4;
    static LdsAttrCache attr; // nprobe > 4096: more than the default 64 KB of dynamic LDS
    if (probe_stage(2, reinterpret_cast<const void*>(&k_select), dim3(p.nq), kThreads, lds)) return hipSuccess;
    hipError_t e = attr.ensure(reinterpret_cast<const void*>(&k_select), lds, device);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_select, dim3(p.nq), dim3(kThreads), lds, s, (const float*)p.scores, p.nlist, p.nprobe, np2, p.metric, p.rot,
                       p.cent, p.D, p.list_gb0, p.list_n, p.probe, p.wl, p.wl_stride, p.nstream, p.nvec, p.prof_total, p.consts,
                       p.bsum, key_window);
    return hipGetLastError();
}

namespace {
template <int RM>
hipError_t launch_select_rm(const SelectParams& p, const SelectGeom& g, size_t lds, int device, hipStream_t s) {
    static LdsAttrCache attr;
    if (probe_stage(2, reinterpret_cast<const void*>(&k_select_mfma<RM>), dim3(p.nq), kThreads, lds)) return hipSuccess;
    hipError_t e = attr.ensure(reinterpret_cast<const void*>(&k_select_mfma<RM>), lds, device);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_select_mfma<RM>, dim3(p.nq), dim3(kThreads), lds, s, p, g);
    return hipGetLastError();
}
} // namespace

// Static __shared__ of k_select_mfma<RM> (s_todo, s_zone, s_kept ...: 7424 B in the round-3 build), read once from the code
// object: the dynamic-LDS decisions below must leave room for it (with the row in LDS and a 128 KB shortlist
// window, dynamic alone reached 160 KB and the launch failed for n_lists in (6016, 7872] at D = 64).
template <int RM>
size_t select_static_lds() {
    static const size_t v = [] {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_select_mfma<RM>)) != hipSuccess) { (void)hipGetLastError(); return (size_t)8192; }
        return (size_t)fa.sharedSizeBytes;
    }();
    return v;
}

hipError_t launch_select_mfma(const SelectParams& p, int device, hipStream_t s) {
    SelectGeom g;
    g.cap2 = next_pow2(2 * p.nprobe) < 64u ? 64u : next_pow2(2 * p.nprobe);
    const size_t lds0 = (size_t)g.cap2 * 8 + (size_t)p.D * 4 + kThreads * 4;
    // <= 4096 lists: the score row lives in registers; else in LDS while it fits beside the shortlist window AND the
    // kernel's static LDS
    g.row_in_lds = (p.nlist > 4096 && (size_t)p.nlist * 4 <= 65536 &&
                    lds0 + (size_t)p.nlist * 4 + select_static_lds<1>() <= kLdsPerWorkgroupMax) ? 1 : 0;
    size_t lds = lds0 + (g.row_in_lds ? (size_t)p.nlist * 4 : 0);
    g.stage = lds + (size_t)p.nprobe * 16 <= 48 * 1024 ? 1 : 0; // per-probe geometry staged in LDS
    if (g.stage) lds += (size_t)p.nprobe * 16;
    // LDS scorer of the lazy selection: RBQ_SEL_ROWS centroid rows at a time while the workgroup stays below 64 KB (>= 2 per CU)
#ifndef RBQ_SEL_ROWS
#define RBQ_SEL_ROWS 4
#endif
    const size_t rowsN = RBQ_SEL_ROWS * ((size_t)p.D + 8) * 4 + 16;
    g.stage_rows = (p.lazy && lds + rowsN <= 64 * 1024) ? (uint32_t)RBQ_SEL_ROWS : 0u;
    // exact head evaluation (rank_mfma.hpp, step 3b): its scratch shares the rows' region (used before the scoring round)
    g.hx_nv = 0;
    size_t regionN = g.stage_rows ? rowsN : 0;
    if (p.lazy && p.head_exact && p.lut && p.top_k <= kHxMaxVec && p.D % 16 == 0) {
#ifndef RBQ_HX_BUDGET
#define RBQ_HX_BUDGET (16384u * 8u)
#endif
        uint32_t nv = (uint32_t)(((uint32_t)RBQ_HX_BUDGET / p.Dc) / 32u * 32u); // vectors evaluated: ~128 K code dimensions per query
        nv = nv < 64u ? 64u : (nv > kHxMaxVec ? kHxMaxVec : nv);
        const size_t hxN = (hx_scratch_bytes(p.D, p.Dc, p.ex_bits, nv) + 15) & ~(size_t)15;
        if (lds + std::max(regionN, hxN) <= 64 * 1024) { g.hx_nv = nv; regionN = std::max(regionN, hxN); }
    }
    lds += regionN;
    g.cand_cap = (uint32_t)(((g.stage ? (size_t)p.nprobe * 16 : 0) + (g.stage_rows ? rowsN - 16 : 0)) / 8); // (RM == 0 prefilter window)
    {   // diagnostic: extra dynamic LDS per workgroup (occupancy experiments)
        static const size_t pad = [] { const char* e = std::getenv("RBQ_SEL_LDS_PAD"); return e ? (size_t)std::atol(e) : (size_t)0; }();
        lds += pad;
    }
    if (p.nlist <= 4096) return launch_select_rm<2>(p, g, lds, device, s);
    if (g.row_in_lds) return launch_select_rm<1>(p, g, lds, device, s);
    return launch_select_rm<0>(p, g, lds, device, s);
}

hipError_t launch_probes_given(const ProbesGivenParams& p, hipStream_t s) {
    hipLaunchKernelGGL(k_probes_given, dim3(p.nq), dim3(kThreads), (size_t)p.D * 4 + kThreads * 4, s, p.list_ids, p.list_counts,
                       p.max_lists, p.nlist, p.metric, p.rot, p.cent, p.D, p.list_gb0, p.list_n, p.probe, p.wl, p.wl_stride, p.nstream,
                       p.consts, p.bsum);
    return hipGetLastError();
}

} // namespace rbq
