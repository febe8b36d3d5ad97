// launch.hpp — host-callable launchers of the kernel translation units.  rbq_api.hip (pure host code) sees the
// kernels only through these; each .hip file below compiles on its own, so a change to the host side does not
// rebuild k_scan's instantiations and vice versa.
//   k_query.hip  k_prep / k_prep_wave, k_rank_* , k_select*, k_probes_given     (kernels.hpp, rank_mfma.hpp)
//   k_scan.hip   k_scan<DT, EX, TR>                                               (scan.hpp)
//   k_build.hip  encoder, reference-layout -> device-layout converters, sorting   (encode.hpp)
#pragma once
#include <atomic>

#include "types.hpp"

namespace rbq {

// Raises a kernel's dynamic-LDS limit once per (kernel, device, size): hipFuncSetAttribute is slow and serialises
// launches, so the largest value set so far is remembered per device.  The limit is only ever RAISED, and the check-and-set
// is serialised: two caller threads that need different sizes cannot leave the kernel with the smaller one.
struct LdsAttrCache {
    std::atomic<size_t> set[16];
    LdsAttrCache() { for (auto& v : set) v.store(0, std::memory_order_relaxed); }
    hipError_t ensure(const void* fn, size_t lds, int device);
};

// Launch probe (diagnostic: rbq_debug_stage_resources).  While the calling thread has a probe installed, the four stage
// launchers record WHICH kernel instantiation they would launch, and its geometry, instead of launching it.
struct KernelProbe { const void* fn = nullptr; uint32_t grid_x = 0, grid_y = 0, block = 0; size_t dyn_lds = 0; };
struct StageProbes { KernelProbe k[4]; }; // prep, rank, select, scan
StageProbes*& stage_probes(); // this thread's probe (null: launch normally)
inline bool probe_stage(int stage, const void* fn, dim3 grid, uint32_t block, size_t lds) {
    StageProbes* p = stage_probes();
    if (!p) return false;
    p->k[stage].fn = fn; p->k[stage].grid_x = grid.x; p->k[stage].grid_y = grid.y; p->k[stage].block = block; p->k[stage].dyn_lds = lds;
    return true;
}

struct PrepParams {
    const float* queries; // [nq][dim]
    uint32_t nq, dim, D, Dc;
    int rotator;
    const uint8_t* rot_blob;
    uint32_t trunc;
    float fac;
    uint32_t ex_bits;
    float* rot;          // [nq][D]
    uint8_t* lut;        // [nq][4Dc]
    QueryConsts* consts; // [nq]
    uint16_t *rot_hi, *rot_lo; // split-bf16 image or null
    bool wg_prep;        // one workgroup per query (always for the matrix rotator)
};
hipError_t launch_prep(const PrepParams& p, int device, hipStream_t s);

struct RankParams {
    int metric;
    const float* rot;
    const uint16_t *rot_hi, *rot_lo;
    const float* cent;
    const uint16_t *cent_hi, *cent_lo;
    const QueryConsts* consts;
    const float* cnorm2;
    uint32_t nq, nlist, D;
    float* scores; // [nq][nlist]
    bool split;    // split-bf16 MFMA GEMM (else f32 MFMA)
    bool big;      // 128x128 tiles
    bool wide;     // 128x256 tiles (split-bf16 only): problems of at least 2048 such tiles
    uint32_t ksplit = 0; // > 1: split-K over grid.z (split-bf16, 64 / 128 tiles): parts added atomically to a row the preparation zeroed
};
hipError_t launch_rank_exact(const RankParams& p, hipStream_t s);
hipError_t launch_rank_gemm(const RankParams& p, int device, hipStream_t s);
// latency-first front of a small call (latency.hpp): prep + the exact canonical score of every list in ONE launch (FhtKac / no rotator)
hipError_t launch_lat_front(const PrepParams& p, const RankParams& r, int device, hipStream_t s);

constexpr uint32_t kAuditCap = 1023; // dead lists exported per query under lazy_audit (a shortlist holds fewer)
struct SelectParams {
    float* scores;
    uint32_t nq, nlist, nprobe;
    int metric;
    const float* rot;
    const float* cent;
    uint32_t D;
    const QueryConsts* consts;
    float cnorm2_max;
    const uint32_t *list_gb0, *list_n;
    ProbeInfo* probe;
    StreamItem* wl;
    uint64_t wl_stride;
    uint32_t* nstream;
    unsigned long long* nvec;
    unsigned long long* prof_total; // null unless a profile is open: slot kProfVectorsProbed of the striped counters (types.hpp)
    unsigned int* fallback_count;
    int force_fallback;
    const BlockSummary* bsum;
    // lazy selection (k_select_mfma only): lists that are provably skipped as a whole are neither scored nor streamed
    const float* cnorm2;          // [nlist] squared centroid norms (inner-product metric: distance from the approximate dot)
    const BlockSummary* lsum;     // [nlist] factor ranges of every list
    const BlockSummaryEx* bsumx;  // [n_blocks] ex-factor ranges of every block
    uint32_t* dead_skipped;       // [4][nq] out: vectors of probed lists dropped that way (exact when exact_members, else 0) |
                                  //              number of lists that go to the scan | two diagnostics taps (T_ub bits; z0, n, scored)
    uint32_t top_k, ex_bits;
    int lazy;                     // 0: every probed list is scored and streamed (round-2 behaviour)
    int exact_members;            // diagnostics: dead_skipped and the probed-vector count must be exact
    // exact head evaluation (round 4): a select-time bound of the k-th distance from REAL estimates of the nearest list's first vectors
    const uint8_t* lut;           // [nq][4Dc] the queries' u8 LUTs (k_prep), device codebook order
    const uint8_t* blocks;        // the index's block records / ex codes / ex factors, as k_scan reads them
    const uint8_t* ex_codes;
    const float *f_add_ex, *f_rescale_ex;
    uint32_t Dc;
    uint32_t n_blocks;            // blocks of the index (the exact head evaluation checks its geometry against it)
    const uint32_t* filter;       // dense id filter of search_filtered (or null) and the ids it tests: under a filter only the exact
    uint64_t filter_nbits;        // head evaluation — which counts filter-passing vectors only — can bound the k-th distance
    const uint64_t* ids;
    int head_exact;               // 0: Cauchy-Schwarz bound only (round 3)
    SlackMul slack;               // TEST ONLY: multipliers of block_ub()'s rounding-slack terms (all 1 in the product)
    uint32_t* audit_dead;         // null, or (option lazy_audit) [nq][kAuditCap + 1] u32: the number of lists this query's selection dropped as
                                  // a whole, then their ids — exported WITHOUT changing any decision, with or without diagnostics or a
                                  // filter, so that a test can ask the oracle what the reference did with exactly those lists
    int fault_dead_all;           // TEST ONLY (debug option lazy_fault_inject, default 0): T_ub := -inf — every list behind the head is
                                  // declared dead whatever its bounds say: a deliberately WRONG selection, so that the
                                  // bound_violations audit can be shown to catch one
};
// key_window: null, or [nq][select_exact_np2(nprobe)] u64 in global memory (nprobe > kNprobeMax: the exact path with its key
// window outside the LDS — slow, but every nprobe up to n_lists is served, as the reference does, src/ivf.rs:1791)
uint32_t select_exact_np2(uint32_t nprobe);
hipError_t launch_select_exact(const SelectParams& p, int device, hipStream_t s, uint64_t* key_window);
hipError_t launch_select_mfma(const SelectParams& p, int device, hipStream_t s);

struct ProbesGivenParams {
    const uint32_t *list_ids, *list_counts;
    uint32_t max_lists, nq, nlist;
    int metric;
    const float* rot;
    const float* cent;
    uint32_t D;
    const uint32_t *list_gb0, *list_n;
    ProbeInfo* probe;
    StreamItem* wl;
    uint64_t wl_stride;
    uint32_t* nstream;
    const QueryConsts* consts;
    const BlockSummary* bsum;
};
hipError_t launch_probes_given(const ProbesGivenParams& p, hipStream_t s);

// ev0/ev1: null, or an event pair carried by the dispatch packet itself (hipExtLaunchKernelGGL)
hipError_t launch_scan(const ScanParams& P, uint32_t nq, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
// k_scanw (scanw.hpp): the same scan with one wave per query; scanw_serves(): the call shapes it takes (launch_scan routes
// to it when ScanParams::wave_kernel is set and it serves the call)
bool scanw_serves(const ScanParams& P);
hipError_t launch_scanw(const ScanParams& P, uint32_t nq, int device, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);

// ---- k_build.hip -------------------------------------------------------------------------------------------------
hipError_t launch_rotate_rows(const float* src, const uint32_t* map, uint32_t nrows, uint32_t dim, uint32_t D, int rotator,
                              const uint8_t* rot_blob, uint32_t trunc, float fac, float* rows, hipStream_t s);
hipError_t launch_encode(const EncodeParams& P, hipStream_t s);
// raw ex codes [row][D] u8 -> lane-major units of slot (row_slot ? row_slot[row] : row) in `ex`
hipError_t launch_pack_ex(const uint8_t* raw, const uint32_t* slot_src, const uint32_t* row_slot, uint32_t nrows, uint32_t D,
                          uint32_t ex_bits, uint8_t* ex, hipStream_t s);
hipError_t launch_block_summary(const uint8_t* blocks, const uint32_t* block_nv, uint32_t nblocks, uint32_t Dc, BlockSummary* bsum,
                                hipStream_t s);
// per-block Cauchy-Schwarz bound terms (BlockSummaryEx) and per-list factor ranges (lazy probe selection); one workgroup per list
hipError_t launch_list_summaries(const uint8_t* blocks, const uint8_t* ex, const float* fadd_ex, const float* fres_ex, const float* cent,
                                 const BlockSummary* bsum, const uint32_t* list_gb0, const uint32_t* list_n, uint32_t nlist, uint32_t D,
                                 uint32_t Dc, uint32_t ex_bits, BlockSummaryEx* bsumx, BlockSummary* lsum, hipStream_t s);
hipError_t launch_count_assign(const uint32_t* assign, uint64_t n, uint32_t nlist, uint32_t* counts, uint32_t* err, hipStream_t s);
hipError_t launch_iota(uint32_t* x, uint64_t n, hipStream_t s);
hipError_t launch_scatter_slots(const uint32_t* sorted_list, const uint32_t* sorted_src, uint64_t n, const uint32_t* list_gb0,
                                const uint64_t* vstart, uint32_t* slot_src, hipStream_t s);
// stable radix sort of (key, value) u32 pairs on the low `bits` bits; tmp == null returns the scratch size in *tmp_bytes
hipError_t sort_pairs_u32(void* tmp, size_t* tmp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                          uint32_t* vals_out, size_t n, unsigned bits, hipStream_t s);
// streamed build: rows of a pushed chunk sorted by (list, source index) -> first row per list, global slot per row
// (first slot of the list + vectors pushed by earlier chunks + rank in this chunk), then advance the cursors
hipError_t launch_chunk_first(const uint32_t* sorted_list, uint32_t n, uint32_t* chunk_first, hipStream_t s);
hipError_t launch_chunk_slots(const uint32_t* sorted_list, const uint32_t* sorted_src, uint32_t n, const uint32_t* list_gb0,
                              const uint32_t* list_cursor, const uint32_t* chunk_first, uint32_t* row_src, uint32_t* row_slot,
                              hipStream_t s);
hipError_t launch_chunk_advance(const uint32_t* sorted_list, uint32_t n, const uint32_t* chunk_first, uint32_t* list_cursor,
                                hipStream_t s);
// reference layout -> device layout (rbq_index_create / load_rbq1): `recs` = the reference's batch records of nb
// blocks back to back; `exsrc` = the packed ex codes of the chunk's vectors, dense, in list order; block b's first
// vector is entry block_dense0[b] of the dense arrays
hipError_t launch_relayout_blocks(const uint8_t* recs, uint32_t nb, uint32_t D, uint32_t Dc, uint8_t* blocks, hipStream_t s);
hipError_t launch_relayout_ex(const uint8_t* exsrc, const uint64_t* block_dense0, const uint32_t* block_nv,
                              uint32_t nb, uint32_t D, uint32_t ex_bits, uint8_t* ex, hipStream_t s);
// per-slot arrays from per-vector (dense, list order) arrays: dst[b*32+v] = v < nv[b] ? src[dense0[b]+v] : fill
hipError_t launch_spread_u64(const uint64_t* src, const uint64_t* block_dense0, const uint32_t* block_nv, uint32_t nb, uint64_t fill,
                             uint64_t* dst, hipStream_t s);
hipError_t launch_spread_f32(const float* src, const uint64_t* block_dense0, const uint32_t* block_nv, uint32_t nb, float fill,
                             float* dst, hipStream_t s);
// centroid-derived arrays: squared norms (f64 accumulate, as the host did) and the split-bf16 image
hipError_t launch_centroid_arrays(const float* cent, uint32_t nlist, uint32_t D, float* cnorm2, uint16_t* hi, uint16_t* lo,
                                  hipStream_t s);
// exact re-scoring of the returned ids against caller-supplied raw vectors (optional rerank, default off)
hipError_t launch_rerank(const float* queries, uint32_t nq, uint32_t dim, const float* raw, uint64_t n_raw, int metric,
                         uint32_t top_k, uint64_t* ids, float* scores, const uint32_t* counts, hipStream_t s);

} // namespace rbq
