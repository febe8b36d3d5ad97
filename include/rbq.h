/*
 * rbq.h — C ABI of the MI355X-native IVF+RaBitQ candidate-scan engine.
 *
 * This is the drop-in boundary for ONE path of lqhl/rabitq-rs: the query-time
 * scan `IvfRabitqIndex::search` / `batch_search` / `search_filtered`
 * (reference src/ivf.rs:1705-1752) whose body is `search_fastscan`
 * (src/ivf.rs:1754-1895) + `search_cluster_v2_batched` (src/ivf.rs:1901-2129).
 * Everything from `rotator.rotate(query)` to the sorted result vector runs on
 * the GPU behind `rbq_search_batch`.  Index training stays with the caller.
 *
 * The reference has no FFI seam of its own (search_fastscan is a private
 * method over private fields, src/ivf.rs:935-946), so the boundary is cut at
 * the two places a maintainer can reach without touching the algorithm:
 *   - the in-memory `ClusterData` arrays (src/ivf.rs:205-242) -> rbq_index_create
 *   - the persisted RBQ1-v3 byte stream (src/ivf.rs:1317-1474) -> rbq_index_load_rbq1
 * INTEGRATION.md shows the Rust-side `extern "C"` binding for both.
 *
 * Plain pointers and sizes only; no torch / HIP types in any signature.
 */
#ifndef RBQ_H
#define RBQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: 1:1 with `RabitqError` (reference src/lib.rs:39-57) ---- */
#define RBQ_OK                   0
#define RBQ_DIMENSION_MISMATCH   1 /* RabitqError::DimensionMismatch{expected,got} */
#define RBQ_INVALID_CONFIG       2 /* RabitqError::InvalidConfig(msg)             */
#define RBQ_EMPTY_INDEX          3 /* RabitqError::EmptyIndex                     */
#define RBQ_IO                   4 /* RabitqError::Io                             */
#define RBQ_INVALID_PERSISTENCE  5 /* RabitqError::InvalidPersistence(msg)        */
#define RBQ_DEVICE               6 /* hipError_t / launch failure (new: no CPU fallback exists) */

#define RBQ_METRIC_L2 0            /* metric_to_tag, src/ivf.rs:122-127 */
#define RBQ_METRIC_IP 1
#define RBQ_ROTATOR_MATRIX 0       /* RotatorType, src/rotation.rs:10-15 */
#define RBQ_ROTATOR_FHT_KAC 1
#define RBQ_ROTATOR_NONE 2         /* no query-time rotation: MSTG posting lists (src/mstg/posting_list.rs:66-101)
                                      are quantised on raw residuals; padded_dim == dim, rotator_len == 0.
                                      Not a tag of the RBQ1 format (rbq_index_load_rbq1 rejects it). */

/* FASTSCAN_BATCH_SIZE, src/simd.rs:768 */
#define RBQ_BATCH 32

typedef struct rbq_index rbq_index;     /* opaque; owns device memory on one or N GPUs (replicas of one index) */
typedef struct rbq_builder rbq_builder; /* opaque; an index under construction by the streamed GPU encoder   */

/* Mirrors the scalar fields of `IvfRabitqIndex` (src/ivf.rs:935-946) and the
 * RBQ1-v3 header (src/ivf.rs:1324-1373). */
typedef struct {
    uint32_t dim;          /* query dimensionality                                  */
    uint32_t padded_dim;   /* rotator.padded_dim(): dim rounded up to x64 for FhtKac */
    uint8_t  metric;       /* RBQ_METRIC_*                                          */
    uint8_t  rotator;      /* RBQ_ROTATOR_*                                         */
    uint8_t  ex_bits;      /* total_bits-1; only 0, 2, 6 (src/simd.rs:3205-3215)    */
    uint8_t  reserved;
    uint64_t n_vectors;    /* sum of list sizes                                     */
    uint64_t n_lists;      /* clusters.len()                                        */
    const uint8_t* rotator_blob; /* DynamicRotator::serialize(): FhtKac = 4*D/8 flip
                                    bytes (src/rotation.rs:486-489); Matrix = D*D f32 LE */
    uint64_t rotator_len;
} rbq_header;

/* One `ClusterData` (src/ivf.rs:205-242), borrowed for the duration of the call. */
typedef struct {
    const float*    centroid;     /* [padded_dim], rotated space                     */
    uint64_t        n;            /* num_vectors                                     */
    const uint64_t* ids;          /* [n]                                             */
    const uint8_t*  batch_data;   /* exactly ClusterData.batch_data: ceil(n/32) records
                                     of [D*4 B FastScan codes | f_add[32] | f_rescale[32]
                                     | f_error[32]] (src/ivf.rs:247-316)             */
    uint64_t        batch_len;    /* ceil(n/32) * (D*4 + 384)                        */
    const uint8_t*  ex_codes;     /* [n][D*ex_bits/8] flattened ex_codes_packed; may be
                                     NULL when ex_bits == 0                          */
    const float*    f_add_ex;     /* [n] (ignored when ex_bits == 0)                 */
    const float*    f_rescale_ex; /* [n]                                             */
} rbq_list_view;

/* `SearchDiagnostics` (src/ivf.rs:150-155), one per query. */
typedef struct {
    uint64_t estimated;
    uint64_t skipped_by_lower_bound;
    uint64_t extended_evaluations;
} rbq_diag;

/* Build a device-resident index from ClusterData-shaped host arrays.  The reference bytes are uploaded as they
 * are and re-laid into the device layout by the GPU.
 * n_devices >= 1 replicas: devices[i] = HIP device ordinal of replica i (NULL = the current device for one
 * replica, devices 0..n-1 otherwise; an ordinal may repeat).  Replica 0 is built from the host arrays, the others
 * are device-to-device copies of it.  `rbq_search_batch` shards a batch over the replicas (the reference's
 * batch_search is a par_iter over queries, src/ivf.rs:1743-1752).  Inputs are copied; nothing is retained.
 * Test status: several replicas on ONE device (a repeated ordinal) are covered by the GPU suite; replicas on DISTINCT
 * devices (peer copy or pinned bounce, per-device launch state) have only run where tests/test_gpu_round3.py::
 * test_replicas_on_two_devices found two GPUs — the builder's boxes had one. */
int rbq_index_create(const rbq_header* hdr, const rbq_list_view* lists,
                     int n_devices, const int* devices, rbq_index** out);

/* Build a device-resident index straight from an RBQ1 v3 byte stream as
 * written by `IvfRabitqIndex::save_to_writer` (src/ivf.rs:1317-1474); applies
 * the same validation as `load_from_reader` (src/ivf.rs:1484-1702) incl. CRC32. */
int rbq_index_load_rbq1(const void* bytes, size_t len,
                        int n_devices, const int* devices, rbq_index** out);

/* GPU-side encoder: build the index on the device from raw vectors — the quantisation loop of
 * `IvfRabitqIndex::train_with_clusters` / `build_from_rotated` (src/ivf.rs:1025-1215: rotate, group by cluster in
 * ascending vector order, `quantize_with_centroid` src/quantizer.rs:140-262 with the constant rescale factor of
 * `RabitqConfig::faster`, src/quantizer.rs:33-46,563-592) — writing the device layout directly.  Clustering stays
 * with the caller (the reference accepts any clustering here).
 *   hdr        dim, padded_dim, metric, rotator (+ rotator_blob/len), ex_bits, n_lists; n_vectors is ignored
 *   centroids  HOST  [n_lists][dim]  cluster centroids in the input space
 *   d_data     DEVICE [n][dim]       vectors; the id of vector i is i
 *   d_assign   DEVICE [n]            cluster of every vector (< n_lists, else RBQ_INVALID_CONFIG)
 *   t_const    the constant scaling factor (`compute_const_scaling_factor`); required when ex_bits > 0
 * The result is identical, array for array, to rbq_index_create over the CPU path's ClusterData. */
int rbq_index_build_device(const rbq_header* hdr, const float* centroids, const float* d_data, const uint32_t* d_assign,
                           uint64_t n, float t_const, int device, rbq_index** out);

/* The same encoder fed chunk by chunk, for indexes whose raw vectors do not fit in HBM at once (100 M x 768 f32 =
 * 307 GB): `train_with_clusters` (src/ivf.rs:1025-1215) builds cluster by cluster and never needs all vectors
 * resident either.  Protocol:
 *   begin   hdr / centroids / t_const as for rbq_index_build_device; list_sizes HOST [n_lists] = number of vectors
 *           each cluster will receive (the caller's count pass over its assignment).  All device arrays of the
 *           index are allocated here.
 *   push    `count` vectors with ids first_id .. first_id+count-1 and their cluster ids.  vectors [count][dim] f32
 *           and assign [count] u32 may each be HOST or DEVICE pointers (detected).  Chunks must arrive in ascending
 *           id order (list membership order = ascending vector index, src/ivf.rs:1141-1149).  A list that receives
 *           more vectors than announced, or a cluster id >= n_lists, is RBQ_INVALID_CONFIG; after an error the builder
 *           can only be aborted.
 *   finish  requires every announced vector to have been pushed; returns the index (identical, array for array,
 *           to rbq_index_build_device over the same data), replicated on n_devices (devices[0] must be the
 *           builder's device; NULL / 1 = that device only) and frees the builder.
 *   abort   frees a builder that was not finished. */
int rbq_build_stream_begin(const rbq_header* hdr, const float* centroids, const uint32_t* list_sizes, float t_const,
                           int device, rbq_builder** out);
int rbq_build_stream_push(rbq_builder* b, const float* vectors, const uint32_t* assign, uint64_t first_id, uint64_t count);
int rbq_build_stream_finish(rbq_builder* b, int n_devices, const int* devices, rbq_index** out);
void rbq_build_stream_abort(rbq_builder* b);

void rbq_index_destroy(rbq_index* idx);

/* Accessors (IvfRabitqIndex::len / cluster_count, src/ivf.rs:1218-1230). */
uint64_t rbq_index_len(const rbq_index* idx);
uint64_t rbq_index_cluster_count(const rbq_index* idx);
uint32_t rbq_index_dim(const rbq_index* idx);
uint32_t rbq_index_padded_dim(const rbq_index* idx);
uint32_t rbq_index_device_count(const rbq_index* idx); /* number of replicas */

/* `batch_search` (src/ivf.rs:1743-1752); nq = 1 is `search` (:1705);
 * filter_words != NULL is `search_filtered` (:1723): a dense bitset over the
 * u32 id space, bit i set <=> RoaringBitmap::contains(i) (src/ivf.rs:2018-2022).
 *
 * queries:    [nq][query_dim] row-major host f32
 * out_ids:    [nq][top_k]  (unused slots = UINT64_MAX)
 * out_scores: [nq][top_k]  (unused slots = NaN); L2: distance asc, IP: score desc
 * out_counts: [nq]         number of valid results of each query (<= top_k)
 * diag:       NULL or [nq]
 *
 * Errors follow search_fastscan: EMPTY_INDEX is checked before
 * DIMENSION_MISMATCH (src/ivf.rs:1761-1769); top_k == 0 returns RBQ_OK with all
 * counts 0 (:1792-1794); nprobe is clamped to [1, n_lists] (:1791).
 * Every nprobe (clamped to n_lists) and every top_k <= 2^20 is served.  Fast paths: nprobe <= 8192 (MFMA shortlist with
 * its key window in LDS; above it the exact all-pairs ranking with the window in global memory), top_k <= 256 (top-k in
 * registers), top_k <= 16384 (exact heap in the LDS of one compute unit: one workgroup per CU from top_k ~ 7000; above it
 * the heap lives in global memory).  top_k > 2^20, or nq * top_k * 8 bytes of heap workspace > 8 GiB in one device call, is
 * RBQ_INVALID_CONFIG; nq < 2^31 per call of the device entry.
 * Re-entrant on one handle.  The batch is cut into sub-batches that are pipelined over a few streams: host
 * staging and PCIe copies of one sub-batch overlap the kernels of the others.  Buffers that are page-locked
 * (rbq_host_alloc, hipHostMalloc, hipHostRegister) are DMA-ed directly; pageable ones are staged through the
 * handle's pinned buffers.  With N replicas the queries are split into N contiguous shards, one per device. */
int rbq_search_batch(const rbq_index* idx, const float* queries, uint64_t nq,
                     uint32_t query_dim, uint32_t top_k, uint32_t nprobe,
                     const uint32_t* filter_words, uint64_t filter_nbits,
                     uint64_t* out_ids, float* out_scores, uint32_t* out_counts,
                     rbq_diag* diag);

/* MSTG posting-list scan: the caller (MstgIndex::search, src/mstg/index.rs:149-213) has
 * already chosen the posting lists of each query (HNSW centroid search + dynamic pruning stay on the
 * CPU); this scans them like search_posting_list_fastscan (src/mstg/index.rs:216-330) — binary FastScan
 * estimate only, f_error/g_error = 0, non-finite estimates dropped, L2 estimates clamped to >= 0 — and
 * keeps the top_k smallest distances like the partial sort of MstgIndex::search (:185-205; ties there
 * are unordered, here the earlier candidate in (list order, vector order) wins).
 *
 * list_ids:    [nq][max_lists] posting-list (cluster) ids per query, scanned in the given order
 * list_counts: [nq]            number of valid entries of each row (<= max_lists)
 * out_*:       as rbq_search_batch; out_scores are distances (ascending) for both metrics
 * The index must have been created with rotator = RBQ_ROTATOR_NONE. */
int rbq_posting_scan_batch(const rbq_index* idx, const float* queries, uint64_t nq, uint32_t query_dim,
                           uint32_t top_k, const uint32_t* list_ids, const uint32_t* list_counts,
                           uint32_t max_lists, uint64_t* out_ids, float* out_scores, uint32_t* out_counts);

/* Same operation on DEVICE pointers (queries and outputs already in HBM of the index's device), ENQUEUED on
 * `hip_stream` (a hipStream_t passed as void*, NULL = default stream) and returning without host
 * synchronisation: results are valid once the stream reaches this point. Each stream gets its own scratch
 * workspace inside the handle; issue calls for one stream from one host thread at a time.
 * d_filter_words may be NULL. d_diag is NULL or [nq] rbq_diag in device memory.
 * A handle with several replicas serves the call from the replica on the device that owns d_queries. */
int rbq_search_batch_device(const rbq_index* idx, const float* d_queries, uint64_t nq,
                            uint32_t query_dim, uint32_t top_k, uint32_t nprobe,
                            const uint32_t* d_filter_words, uint64_t filter_nbits,
                            uint64_t* d_out_ids, float* d_out_scores,
                            uint32_t* d_out_counts, rbq_diag* d_diag,
                            void* hip_stream);

/* Frees the scratch workspace the handle keeps for `hip_stream` (rbq_search_batch_device creates one per caller
 * stream and keeps it until the index is destroyed); call it before destroying a stream that will not be used
 * with this index again.  The stream must be idle. */
int rbq_release_stream(rbq_index* idx, void* hip_stream);

/* Page-locked host memory for query / result buffers: rbq_search_batch DMA-s such buffers directly. */
void* rbq_host_alloc(size_t bytes);
void rbq_host_free(void* p);

/* OPTIONAL full-precision rerank — an extension (BASELINE north_star item 3), NOT reference behaviour: the
 * reference index stores no raw vectors (src/ivf.rs:207-242) and never re-scores.  Default OFF; results are
 * reference-identical only while it is off.  `vectors` [n][dim] f32 (HOST: copied to every replica; DEVICE pointer
 * on the replica's device: borrowed, must outlive the index) indexed by id; passing NULL detaches.  While enabled,
 * every search re-scores its returned ids with the exact l2_distance_sqr / dot (canonical order of
 * src/math.rs:154-245) against these vectors and re-sorts them; ids >= n get NaN.  top_k <= 1024.
 * Toggle with rbq_debug_set_option(idx, "rerank", 0/1) once vectors are attached (attaching switches it on). */
int rbq_index_set_rerank_vectors(rbq_index* idx, const float* vectors, uint64_t n);

/* Timing taps for bench.py: average duration (ms) of each stage kernel between
 * rbq_profile_begin/end, measured with hipEvents on the stream the kernels run
 * on. stage names: "prep", "rank", "select", "scan". Returns <0 for an unknown stage. */
void   rbq_profile_begin(rbq_index* idx);
void   rbq_profile_end(rbq_index* idx);
double rbq_profile_stage_ms(const rbq_index* idx, const char* stage, uint64_t* launches);
/* Durations (ms) of the individual timed launches of a stage, in launch order (replica after replica); returns their
 * number, writes at most `cap` of them to `out` (which may be NULL). */
uint64_t rbq_profile_stage_samples(const rbq_index* idx, const char* stage, float* out, uint64_t cap);
/* Algorithmic bytes (sum over probed lists of n_c*(D/8+12)) of the
 * scan launches between rbq_profile_begin/end. */
uint64_t rbq_profile_scan_bytes(const rbq_index* idx);
/* Traffic counters of the scan launches between rbq_profile_begin/end (summed over replicas), out[0..n):
 *   [0] vectors probed (sum of n_c over probed lists)   [1] block records whose sign codes were requested
 *   [2] block records whose factor rows were requested  [3] block-stream entries read (16 B each)
 *   [4] candidates whose ex codes were fetched          [5] queries scanned                */
int rbq_profile_counters(const rbq_index* idx, uint64_t* out, uint32_t n);
/* Which stages rbq_profile_begin/end time: bit 0 prep, 1 rank, 2 select, 3 scan (default: all four). Every timed
 * stage costs two event records per launch; a throughput measurement that only needs the dominant kernel's
 * duration selects that stage alone. */
void rbq_profile_select_stages(rbq_index* idx, uint32_t mask);
/* Time only every n-th launch of each selected stage (default 1 = every launch). */
void rbq_profile_set_sampling(rbq_index* idx, uint32_t every);
/* Number of queries (since creation) whose probe selection fell back from the MFMA shortlist to the
 * all-lists canonical ranking (shortlist overflow / non-finite scores). Diagnostic. */
uint64_t rbq_debug_rank_fallbacks(const rbq_index* idx);
/* tie log of k_scan (queries whose result depends on the layout of the reference's BinaryHeap, src/ivf.rs:904-931, replay their logged
   candidates instead of their lists): out4 = replays, log entries replayed, real heap operations among them, logs that overflowed */
void rbq_debug_tie_log_stats(const rbq_index* idx, uint64_t* out4);
/* Diagnostic: which kernel instantiation each stage (prep, rank, select, scan) launches for a call of nq queries with this top_k /
 * nprobe on this index, and what it occupies: out[stage][6] = workgroups, threads per workgroup, VGPRs per lane, LDS bytes per
 * workgroup (static + dynamic), scratch bytes per lane, 0.  Nothing is launched.  bench.py's `regime` object is built from it. */
int rbq_debug_stage_resources(rbq_index* idx, uint64_t nq, uint32_t top_k, uint32_t nprobe, uint32_t* out);
/* Replica arrays (process-wide, since start) that were copied between devices through the page-locked bounce buffer instead of
 * hipMemcpyPeer: the path taken when the runtime refuses the peer copy, or for every replica copy when RBQ_FORCE_NO_PEER=1 is in
 * the environment (test switch: the only way the path can run on a one-GPU box). Diagnostic. */
uint64_t rbq_debug_bounce_copies(void);
/* Exact head evaluation of the lazy probe selection (since creation): queries for which it ran / times its geometry guard tripped
 * (must stay 0). Diagnostic. */
uint64_t rbq_debug_head_exact_evaluations(const rbq_index* idx);
uint64_t rbq_debug_head_exact_guard_trips(const rbq_index* idx);
/* Number of queries (since creation) that met two bit-identical distances in their top-k and were therefore
 * re-run inside the scan kernel with the exact BinaryHeap emulation (src/ivf.rs:2078-2105 pushes into a
 * std BinaryHeap, whose tie behaviour depends on its layout). Diagnostic. */
uint64_t rbq_debug_heap_restarts(const rbq_index* idx);
/* Diagnostic: copy one of the index's device arrays ("blocks", "ids", "ex", "fadd_ex", "fres_ex", "bsum", "lsum", "bsumx",
 * "centroids", "list_gb0", "list_n") to the host; `bytes` must be the array's exact size. */
int rbq_debug_copy_index(rbq_index* idx, const char* name, void* dst, uint64_t bytes);
/* Diagnostic: copy an intermediate buffer ("rot", "lut", "consts", "scores", "probe", "nstream", "wl", "nvec", "dead_skipped") of the
 * workspace that rbq_search_batch_device bound to `hip_stream`; the caller has synchronised that stream. */
int rbq_debug_copy_workspace(rbq_index* idx, void* hip_stream, const char* name, void* dst, uint64_t bytes);
/* Diagnostic switches; results are identical under every setting, only the work done changes:
 *   "block_bound" 0      stream every probed block (no block-level lower-bound skipping)
 *   "exact_rank" 1       rank all nq x nlist pairs in canonical order instead of the MFMA shortlist
 *   "force_rank_fallback" 1   send every query through the shortlist's all-lists fallback
 *   "f32_rank" 1         approximate list scores from the f32 MFMA GEMM instead of the split-bf16 one
 *   "wg_prep" 1          workgroup-per-query query preparation for every rotator (default: one wave per query)
 *   "exact_heap" 1       keep the top-k in the BinaryHeap emulation from the first candidate (no sorted fast path)
 *   "lazy_select" 0      score and stream every probed list (default 1: lists that are provably skipped as a whole —
 *                        every lower bound of the list at or above a select-time upper bound of the k-th distance — are
 *                        neither scored exactly nor streamed; their sizes still count in skipped_by_lower_bound)
 *   "profile_counters" 0 an open profile (rbq_profile_begin) keeps its stage timings but not the traffic counters (default 1; the
 *                        counters cost a pipelined caller 2-3 %: bench.py times without them and counts in a pass of its own)
 *   "latency_path" 0     small calls through the batch kernels (default 1: calls of up to 4 queries rotate the query, build its LUT and
 *                        score every list exactly in ONE launch; batches up to 512 queries are prepared by a workgroup per query)
 *   "tie_log" 0          a query with equal distances in its top-k is scanned again with the BinaryHeap emulation (default 1: k_scan
 *                        logs the candidates it refines — 12 B each, up to 8192 per query in the calling stream's workspace — and the
 *                        tied query replays the log; rbq_debug_tie_log_stats)
 *   "scan_wave" 0/1/2    which scan kernel serves a call
 *   "rank_ksplit" 0      never split the K loop of the ranking GEMM (default 1: calls of up to 256 queries split it 2-4 ways over
 *                        grid.z, the parts added atomically to a cleared row; n > 1 forces n parts)
 * and two that are not result-neutral:
 *   "rerank" 0/1         the optional full-precision rerank (needs rbq_index_set_rerank_vectors)
 *   "debug_replica" r    which replica rbq_debug_copy_index / rbq_debug_copy_workspace read */
int rbq_debug_set_option(rbq_index* idx, const char* name, int value);

/* OPT-IN process-wide defaults, for a host that wants the library's recommendation applied: sets GPU_MAX_HW_QUEUES=16 in the process
 * environment unless the variable is already set (the HIP runtime reads it when it initialises and multiplexes all streams of the
 * process over 4 hardware queues otherwise; two batches on one queue serialise: 5-15 % of the pipelined rate).  Call it
 * BEFORE the first HIP call of the process and before other threads exist (setenv is not thread-safe); it changes the queue
 * configuration of every HIP user in the process, which is why the library never does it by itself.  Returns RBQ_OK or RBQ_IO. */
int rbq_process_defaults(void);
const char* rbq_strerror(int code);
/* Copies the calling thread's last error detail (e.g. "checksum mismatch",
 * "expected 960, got 128") into buf; returns its full length. */
int rbq_last_error_detail(char* buf, size_t n);

/* Library/ABI version (major<<16 | minor). */
uint32_t rbq_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RBQ_H */
