// types.hpp — plain data shared by the host side (rbq_api.hip) and the kernel translation units
// (k_query.hip, k_scan.hip, k_build.hip): kernel parameter blocks, device record layouts, launch geometry.
// No device code in here, so the host TU can include it without instantiating a kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rbq {

constexpr int kThreads = 256;

struct QueryConsts {
    float delta, sum_vl, k1x, kbx, scale, qnorm;
    float qnorm2;     // |q|^2 of the rotated query (approximate ranking, rank_mfma.hpp)
    float exlo, exhi; // the ex-code dot product sum_i code_i * q_i of ANY code (code_i in [0, 2^ex - 1]) lies in [exlo, exhi],
                      // rounding of the kernel's own summation included (lazy probe selection, rank_mfma.hpp)
    float q1norm;     // sum_i |q_i| of the rotated query (rounding slack of that selection)
    float amin, amax; // sum over codebooks of the smallest / largest u8 entry: accu of ANY code lies in [amin, amax]
};
struct ProbeInfo {
    float g_add, g_err, dotqc;
    uint32_t cid;
};
struct WorkItem {
    uint32_t gblock;      // global 32-vector block index
    uint32_t rank_nvalid; // (probe rank << 6) | number of real vectors in the block (1..32)
};
// Per-block factor ranges over the block's REAL vectors, computed once at index creation.  Every float op of
// the epilogue is monotone in each operand, so evaluating it on these extremes brackets every lane's lower
// bound: lbmin <= lb_v <= lbmax.  `usable` is 0 when any factor is non-finite (the block is then never skipped).
struct BlockSummary {
    float fadd_min, fadd_max, fres_min, fres_max, ferr_min, ferr_max;
    uint32_t usable, pad;
};
// Per-block maxima over the block's REAL vectors of the quantities that bound, by Cauchy-Schwarz around the list's
// centroid, the refined distance and the 1-bit estimate of every vector for ANY query (k_list_summaries, encode.hpp):
//   refined distance <= S + g_add + B * g_err,   1-bit estimate <= S1 + g_add + B1 * g_err   (+ rounding slack, block_ub()).
// fadd_ex_abs / fres_ex_abs: largest |f_add_ex| / |f_rescale_ex| (magnitudes for that slack).  usable = 0: a factor of the
// block is not finite (the block then proves nothing).
struct BlockSummaryEx {
    float S, B, S1, B1, fadd_ex_abs, fres_ex_abs;
    uint32_t usable, pad;
};
// TEST ONLY: multipliers of the rounding-slack terms of block_ub() (kernels.hpp), all 1 in the product.  A test scales one term at a
// time (debug options slack_term / slack_milli) and asks the lazy_audit check whether the selection still drops only lists the
// reference skips: a term scaled far below zero must make the audit fire — the audit sees every term.
struct SlackMul {
    float ge = 1.0f;   // 0: |q - c| <= g_err (1 + 1e-4)
    float eip = 1.0f;  // 1: E_ip, the u8 quantisation of the LUT + the sequential f32 sums
    float est = 1.0f;  // 2: 1e-5 of the magnitudes of the 1-bit estimate's operations
    float lb = 1.0f;   // 3: 1e-5 |f_error| g_err of the lower bound's last operation
    float et = 1.0f;   // 4: 1e-3 (2^ex - 1) |q|_1, the f32 summation of the ex-code dot
    float dist = 1.0f; // 5: 1e-5 of the magnitudes of the refined distance's operations
};
// One entry of a query's block stream (probe order, block order within a list).  `lbmin` is the block-level
// lower bound of this (query, block) pair — everything in it but the running threshold is known when the
// stream is written, so the scan's fill step is one 16-byte load and one compare per block.
struct StreamItem {
    uint32_t gblock, rank_nvalid;
    float lbmin; // -inf: never skip
    uint32_t pad;
};

// round-to-nearest-even f32 -> bf16 bits (inf stays inf; NaN stays NaN)
__host__ __device__ inline uint16_t bf16_rne(float x) {
    uint32_t u;
    __builtin_memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u); // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
// x = hi + lo + r: the subtraction is exact (hi is x rounded to 8 significant bits)
__host__ __device__ inline void bf16_split(float x, uint16_t& hi, uint16_t& lo) {
    hi = bf16_rne(x);
    lo = bf16_rne(x - bf16_to_f32(hi));
}

// ---- k_scan ---------------------------------------------------------------------------------------------------
struct ScanParams {
    const uint8_t* blocks;   // [n_blocks][4Dc + 384]: lane-major sign codes | f_add[32] | f_rescale[32] | f_error[32]
    const uint64_t* ids;     // [n_blocks*32]
    const uint8_t* ex_codes; // [n_blocks*32][ex_bytes_dev]: lane-major ex codes, see ex_w4()
    const float* f_add_ex;   // [n_blocks*32]
    const float* f_rescale_ex;
    const uint8_t* lut;      // [nq][4Dc] (pair-swapped codebook order); Dc = D rounded up to x64
    const float* rot;        // [nq][D]
    const QueryConsts* consts;
    const ProbeInfo* probe;  // [nq][nprobe]
    const StreamItem* wl;    // [nq][wl_stride]
    const uint32_t* nstream; // [nq]
    const uint32_t* filter;  // dense bitset or null
    uint64_t filter_nbits;
    uint64_t wl_stride;
    uint64_t* out_ids;
    float* out_scores;
    uint32_t* out_counts;
    unsigned long long* diag; // [nq][3] or null
    uint32_t D, Dc, nprobe, top_k, metric, ex_bits;
    uint32_t no_block_bound; // diagnostic: stream every probed block (measures the pure streaming rate)
    uint32_t exact_heap;     // diagnostic: emulate the reference's BinaryHeap from the start (no sorted fast path)
    unsigned int* heap_restarts; // counter of queries re-run with the exact heap after a distance tie (or null)
    uint32_t mstg;           // MSTG posting-list semantics (src/mstg/index.rs:216-330): distance = binary estimate,
                             // non-finite dropped, L2 clamped to >= 0, no error-bound term
    unsigned long long* prof; // null, or the traffic counters [kProfStripes][kProfSlots] of kProf* below (added once per workgroup, at exit)
    uint32_t* heap_ws;            // null, or [nq][2 * (top_k + 1)] words: the exact heap in global memory (top_k beyond the LDS)
    const uint32_t* dead_skipped; // [nq] vectors of probed lists that the probe selection proved skipped as a whole (they
                                  // never enter the stream; diagnostics add them to skipped_by_lower_bound), or null
    uint32_t wave_kernel;         // 1: k_scanw (one wave per query, scanw.hpp) where it serves the call; 0: k_scan (one workgroup per query)
    // k_scan: log of the candidates the fast pass refined (lower bound, distance, slot — in stream order, tie_log_cap entries per
    // query), or null.  A query whose result depends on the layout of the reference's BinaryHeap (equal distances) then replays the
    // LOG through the exact heap instead of scanning its lists again; a log that overflowed falls back to the re-scan.
    uint32_t* tie_log;
    uint32_t tie_log_cap;
    unsigned int* tie_stats;      // [4] counters: replays, entries replayed, real heap operations, overflowed logs (or null)
};
// traffic counters kept while a profile is open (rbq_profile_begin/end); [0] is written by the select kernels
enum { kProfVectorsProbed = 0, kProfCodeBlocks = 1, kProfMetaBlocks = 2, kProfStreamEntries = 3, kProfExEvals = 4,
       kProfQueries = 5, kProfSlots = 8 };
// The counters are kept in kProfStripes copies (one 64-byte line each), a query adds to copy (query index mod kProfStripes) and
// the host sums them: six atomics per query on ONE address serialise in the L2 and were 9 % of the pipelined rate (round 3).
constexpr uint32_t kProfStripes = 64;
__host__ __device__ inline uint32_t prof_stripe(uint32_t q) { return (q % kProfStripes) * (uint32_t)kProfSlots; }

#ifndef RBQ_NSCAN
#define RBQ_NSCAN 3         // scanner waves per workgroup (3 + replay wave = 256 threads: 4 workgroups per CU)
#endif
#ifndef RBQ_FILL_K
#define RBQ_FILL_K 2
#endif
constexpr int kNScan = RBQ_NSCAN;
constexpr int kScanThreads = (kNScan + 1) * 64; // scanner waves + 1 replay wave
constexpr int kTileBlocks = 2 * kNScan;         // 32-vector blocks per tile (one per scanner half-wave)
constexpr int kTileCand = kTileBlocks * 32;     // candidates per tile
constexpr int kFillK = RBQ_FILL_K;                 // stream entries per scanner lane and fill step
constexpr int kWindow = kNScan * 64 * kFillK;      // largest fill window
constexpr int kQueueCap = kTileBlocks - 1 + kWindow <= 512 ? 512 : 1024; // live-block FIFO (>= kTileBlocks - 1 + kWindow)
// blocks per scanner half-wave and tile of k_scan<DT, ...>: two for the compile-time dimensions up to 256 (DT = 0: runtime dimension)
// (Measured in round 3 and left OFF: -DRBQ_SCAN_NB_MAXDIM=256 gives two blocks per half-wave and tile at D <= 256 — parity green,
// but cfg2 (d = 128) 5.78 M queries/s against 6.25 M with one block, streaming roofline 0.312 against 0.318: the tile needs
// 120 registers (4 waves per SIMD instead of 5) and the per-tile hand-over was not what bounds it.)
#ifndef RBQ_SCAN_NB_MAXDIM
#define RBQ_SCAN_NB_MAXDIM 0
#endif
#ifndef RBQ_SCAN_NB
#define RBQ_SCAN_NB 2
#endif
__host__ __device__ constexpr int scan_nb(uint32_t DT) { return (DT != 0 && DT <= RBQ_SCAN_NB_MAXDIM) ? RBQ_SCAN_NB : 1; }
static_assert(RBQ_SCAN_NB * kTileBlocks - 1 + kWindow <= kQueueCap, "live queue too small");
constexpr uint32_t kTopKRegMax = 256;             // largest top_k that lives in the replay wave's registers
constexpr uint32_t kNprobeMax = 8192;             // largest nprobe of the MFMA-shortlist selector (2 x nprobe u64 keys in LDS); above it the
                                                  // exact all-pairs ranking with its key window in global memory serves the call
constexpr uint32_t kTopKMax = 16384;              // largest top_k whose exact heap (top_k + 1 entries) fits the LDS; above it the heap of
                                                  // each query lives in global memory (ScanParams::heap_ws)
constexpr uint32_t kTopKHardMax = 1u << 20;       // largest top_k at all
constexpr size_t kLdsPerWorkgroupMax = 160 * 1024; // LDS of one compute unit (gfx950)

// Device layout of the ex codes ("lane-major"): per vector 16 lanes x W4 units of 16 B, stored [unit][lane][16 B].
// Lane l holds the codes of dims 16t+l (t = 0..D/16-1); unit j packs codes t = j*CPU .. j*CPU+CPU-1 as a
// little-endian 128-bit string, ex bits each, with no code straddling two units (CPU = 128/ex: 21 for 6-bit,
// 64 for 2-bit).  Unused code slots are zero.
__host__ __device__ constexpr uint32_t ex_cpu(uint32_t ex_bits) { return ex_bits ? 128u / ex_bits : 1u; }
__host__ __device__ constexpr uint32_t ex_w4(uint32_t D, uint32_t ex_bits) { // 16-byte units per lane
    return ex_bits ? (D / 16 + ex_cpu(ex_bits) - 1) / ex_cpu(ex_bits) : 0u;
}
__host__ __device__ inline uint32_t ex_bytes_dev(uint32_t D, uint32_t ex_bits) { return ex_w4(D, ex_bits) * 256u; }
// length of the zero-padded rotated query in LDS: every code slot of every unit has a (zero) partner
__host__ __device__ inline uint32_t ex_qlen(uint32_t D, uint32_t ex_bits) {
    const uint32_t n = ex_w4(D, ex_bits) * ex_cpu(ex_bits) * 16u;
    return n > D ? n : D;
}
// LDS carve-up of k_scan (dynamic only, LUT at byte 0):
//   lut[4Dc] u8 | qrot[D] f32 | heap_d[k+1] f32 | heap_s[k+1] u32 | q_slot,q_lb,q_ip,q_gadd,q_d [2][kTileCand] |
//   list[kTileCand] u32 | mask[2][kTileBlocks] u32 | queue[kQueueCap] WorkItem | fmask[kNScan] u64 |
//   T, len, nskip, nbatch | batch[kScanThreads/16] u32
// (heap_in_lds = false: the heap lives in global memory, ScanParams::heap_ws)
// nb = scan_nb(DT) of the instantiation that will run
__host__ __device__ inline size_t scan_lds_bytes(uint32_t Dc, uint32_t D, uint32_t ex_bits, uint32_t top_k, bool heap_in_lds = true, int nb = 1) {
    return (size_t)Dc * 4 + (size_t)ex_qlen(D, ex_bits) * 4 + (heap_in_lds ? (size_t)(top_k + 1) * 8 : (size_t)0) + (size_t)nb * kTileCand * 2 * 20 +
           (size_t)nb * kTileCand * 4 + 2 * (size_t)nb * kTileBlocks * 4 + kQueueCap * 8 + kNScan * kFillK * 8 + 32 + (kScanThreads / 16) * 8;
}

// ---- encoder ----------------------------------------------------------------------------------------------------
constexpr uint32_t kNoSrc = 0xffffffffu;
struct EncodeParams {
    const float* rows;          // [nrows][D] rotated vectors of this chunk
    const float* centroids;     // [nlist][D] rotated
    const uint32_t* slot_src;   // block-ordered mode: [nslots] source vector index or kNoSrc (chunk-local view)
    const uint32_t* block_list; // list of every block (block-ordered mode: chunk-local view; scatter mode: global)
    const uint32_t* row_slot;   // scatter mode: [nrows] global slot of row r (null = block-ordered mode, slot = r)
    uint8_t* blocks;            // block records (block-ordered mode: chunk-local view; scatter mode: whole array)
    uint8_t* raw_ex;            // [nrows][D] u8 scratch (ex_bits > 0)
    float* f_add_ex;            // like `blocks`
    float* f_rescale_ex;
    uint64_t* ids;
    uint64_t src_base;          // ids[slot] = src_base + source index (scatter mode: src_base + row)
    uint32_t nslots, D, Dc, ex_bits, metric;
    float t_const;
};
constexpr int kEncThreads = 64;          // 64 vectors = 2 blocks per workgroup

} // namespace rbq
