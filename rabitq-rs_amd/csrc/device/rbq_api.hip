// rbq_api.hip — C ABI (include/rbq.h) over the HIP kernels.  Host code only: the kernels live in k_query.hip,
// k_scan.hip and k_build.hip and are reached through launch.hpp.  gfx950 only.
//
// Host responsibilities: validate like the reference (src/ivf.rs:1754-1769,1484-1702), upload the reference's
// ClusterData bytes and have the GPU re-lay them into the device layout (one-time, at create/load), own HBM on one
// or N devices (replicas), and enqueue prep -> rank -> select -> scan for each query batch.  There is no CPU
// compute path: every failure to reach the GPU surfaces as RBQ_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rbq.h"
#include "launch.hpp"
#include "../host/rbq_host_logic.hpp"

using namespace rbq;
using rbq_host::ListSrc;
using rbq_host::OutPack;
using rbq_host::align_up;

// The HIP runtime multiplexes a process's streams over FOUR hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and two of this
// library's lanes (or two caller streams) on one queue serialise their kernels: 16 queues are worth 5-15 % of the pipelined rate
// (DESIGN 5).  The variable is read when the runtime initialises.  The library does NOT touch the environment on its own (round 4 did,
// from a load-time constructor: a silent change for every other HIP user of the process, and setenv under dlopen races with getenv in
// a threaded host); rbq_process_defaults() is the explicit opt-in, for a host that wants the library's recommendation applied.
extern "C" int rbq_process_defaults(void) {
    return setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite=*/0) == 0 ? RBQ_OK : RBQ_IO;
}

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& detail) {
    g_err = detail;
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return fail(RBQ_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

// No C++ exception may cross the C boundary (the Rust host is panic = "abort", Cargo.toml:59; ctypes would
// terminate): every entry point runs inside this guard.
#define RBQ_GUARD_BEGIN try {
#define RBQ_GUARD_END                                                                               \
    } catch (const std::bad_alloc&) { return fail(RBQ_IO, "out of host memory"); }                 \
    catch (const std::exception& e) { return fail(RBQ_IO, std::string("internal error: ") + e.what()); } \
    catch (...) { return fail(RBQ_IO, "internal error"); }

// Entry points switch to the index's device and put the caller's device back on exit.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

uint32_t floor_log2_u32(uint32_t x) { uint32_t r = 0; while (x >>= 1) ++r; return r; }

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return RBQ_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return RBQ_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct PinBuf { // page-locked host staging
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return RBQ_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 4096;
        HIP_TRY(hipHostMalloc(&p, want, hipHostMallocPortable)); // portable: mapped for every device, not only the current one
        cap = want;
        return RBQ_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct Workspace {
    hipStream_t stream = nullptr;
    DevBuf queries, rot, lut, consts, scores, probe, wl, nstream, nvec, out_pack, filter, rot_hi, rot_lo, dead_skipped, heap_ws, key_window, audit_dead, tie_log;
    PinBuf h_in, h_out;      // rbq_search_batch: staging of one sub-batch
    hipEvent_t done = nullptr; // results of the sub-batch in flight have reached h_out / the caller's buffers
    uint64_t call_nq = 0;       // queries of the WHOLE host call this launch chain belongs to (0: a device-entry call — its own nq counts)
    bool latency_first = false; // this launch chain belongs to a host call that waits for it (rbq_search_batch below kHostWaveMinQueries)
    void release() {
        for (DevBuf* b : {&queries, &rot, &lut, &consts, &scores, &probe, &wl, &nstream, &nvec, &rot_hi, &rot_lo, &out_pack, &filter, &dead_skipped,
                          &heap_ws, &key_window, &audit_dead, &tie_log})
            b->release();
        h_in.release(); h_out.release();
        if (done) (void)hipEventDestroy(done);
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr; done = nullptr;
    }
};

struct StageProf {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; // pairs recorded since rbq_profile_begin
    double ms = 0;
    uint64_t launches = 0;
    std::vector<float> samples; // duration of every timed launch, in launch order
};
// Event pairs are created once and recycled: hipEventCreate inside the launch path cost ~15 % of the
// overlapped throughput and broke down beyond three caller streams.
struct EventPool {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_pairs;
    bool take(std::pair<hipEvent_t, hipEvent_t>& p) {
        if (!free_pairs.empty()) { p = free_pairs.back(); free_pairs.pop_back(); return true; }
        // timing only: without the system-scope fence a default event performs when it completes (that fence sits
        // between the kernels of a stream and shows up in the overlapped throughput)
        if (hipEventCreateWithFlags(&p.first, hipEventDisableSystemFence) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&p.second, hipEventDisableSystemFence) != hipSuccess) { (void)hipEventDestroy(p.first); return false; }
        return true;
    }
    void give(const std::pair<hipEvent_t, hipEvent_t>& p) { free_pairs.push_back(p); }
    void destroy() {
        for (auto& e : free_pairs) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        free_pairs.clear();
    }
};

struct Arr { // one device array of a replica (size kept for cloning)
    void* p = nullptr;
    size_t bytes = 0;
};

// Persistent staging helpers of a replica (rbq_search_batch with PAGEABLE queries).  The queries of a call must be copied once
// into page-locked memory before the GPU can read them; one thread copies 3.9 MB (1024 x 960 f32) in ~165 us — more than half of the
// ~280 us the same call takes from page-locked buffers.  The sub-batches of a call have their own lanes and staging buffers, so
// their copies are independent: the caller copies sub-batch 0 (and launches it at once), the helpers copy the others meanwhile.
// Started on the first pageable call, joined when the replica is freed; a helper that has just finished polls ~100 us for the
// next job before it blocks (callers that issue calls back to back find the helpers hot).
struct StageHelpers {
    static constexpr int kThreads = 3;
    struct Job { void* dst; const void* src; size_t bytes; std::atomic<int>* done; };
    std::thread th[kThreads];
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> jobs;
    std::atomic<uint32_t> pending{0};
    bool stop = false;
    void run() {
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return stop || !jobs.empty(); });
                if (jobs.empty()) return;
                job = jobs.front();
                jobs.pop_front();
            }
            std::memcpy(job.dst, job.src, job.bytes);
            job.done->store(1, std::memory_order_release);
            pending.fetch_sub(1, std::memory_order_release);
            const auto t0 = std::chrono::steady_clock::now();
            while (pending.load(std::memory_order_acquire) == 0 && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(100)) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
        }
    }
    // false: a thread could not be created (the ones that did start are shut down again; the caller stages inline)
    bool start() {
        try {
            for (auto& t : th) t = std::thread([this] { run(); });
        } catch (...) {
            shutdown();
            return false;
        }
        return true;
    }
    void post(const Job& j) {
        pending.fetch_add(1, std::memory_order_release);
        { std::lock_guard<std::mutex> lk(mu); jobs.push_back(j); }
        cv.notify_one();
    }
    void shutdown() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto& t : th) if (t.joinable()) t.join();
    }
};

// Default of the `scan_wave` option; RBQ_SCAN_WAVE=0|1|2 in the environment overrides it (A/B runs of the whole test suite).
inline int scan_wave_default() {
    static const int v = [] { const char* e = std::getenv("RBQ_SCAN_WAVE"); return e && *e ? std::atoi(e) : 2; }();
    return v;
}
// scan_wave = 2: batches of at least this many queries go to k_scanw (a query costs one resident wave instead of four: the
// pipelined rate), smaller ones to k_scan (four waves shorten ONE query's chain: the latency of a small call)
constexpr uint64_t kScanWaveMinQueries = 128;
// ... and rbq_search_batch (host buffers; the caller waits for the call) from this many queries per call
constexpr uint64_t kHostWaveMinQueries = 4096;
// Latency-first front (latency.hpp): one launch rotates the query, builds its LUT and scores EVERY list exactly, spread over
// n_lists / 32 workgroups per query.  It re-reads the centroid table once per query, so it serves calls of a few queries only
// (every query's pass over the table must stay a few microseconds: n_lists x D x 4 bytes x nq within kLatMaxBytes).
constexpr uint64_t kLatMaxQueries = 4; // (measured, GIST-1M shape: 1 / 2 / 4 queries per call -12 / -10 / -12 us against prep + GEMM; 8: no gain)
constexpr uint64_t kLatMaxBytes = 96ull << 20;
constexpr uint64_t kSplitKMaxCallQueries = 256; // calls up to here split the ranking GEMM's K loop over grid.z
constexpr uint64_t kPrepWgMaxQueries = 512; // up to here the preparation runs one WORKGROUP per query (latency.hpp without the scorers)

// One device-resident copy of the index.
struct Replica {
    int device = 0;
    uint32_t dim = 0, D = 0, Dc = 0;
    uint8_t metric = 0, rotator = 0, ex_bits = 0;
    uint64_t n_vectors = 0, n_lists = 0, n_blocks = 0;
    uint32_t trunc = 0;
    float fac = 1.0f;
    Arr rot_blob, centroids, blocks, ids, ex, fadd_ex, fres_ex, list_gb0, list_n, prof, bsum, cnorm2, fallbacks, cent_hi, cent_lo, raw,
        bsumx, lsum; // ex-factor ranges per block, factor ranges per list (lazy probe selection)
    Arr* arrays[18] = {&rot_blob, &centroids, &blocks, &ids, &ex, &fadd_ex, &fres_ex, &list_gb0, &list_n, &prof, &bsum, &cnorm2,
                       &fallbacks, &cent_hi, &cent_lo, &raw, &bsumx, &lsum};
    float cnorm2_max = 0.0f;
    uint64_t n_raw = 0;      // raw vectors attached for the optional rerank
    bool raw_borrowed = false;
    bool rerank = false;
    uint32_t host_lanes = 0, host_subbatch = 0, host_trace = 0; // rbq_debug_set_option: pipeline shape of rbq_search_batch (0 = default)
    int rank_ksplit = 1;              // option rank_ksplit: 0 = never split the ranking GEMM's K loop, 1 = by batch size, n > 1 = forced
    int host_taper = 0;               // option host_taper: weights of a call's sub-batches (rbq_host_logic.hpp; A/B runs)
    uint32_t host_zero_copy_min = 5;  // option host_zero_copy_min: smallest call whose queries are read in place (up to 4 queries take the
                                      // latency-first front: a hundred workgroups per query would each read it over PCIe)
    bool host_zero_copy = true; // rbq_search_batch: k_prep reads the queries from page-locked host memory in place (no H2D copy command)
    bool host_stage_helpers = true; // rbq_search_batch: pageable queries of a call's later sub-batches are staged by helper threads
    std::unique_ptr<StageHelpers> stagers; // (created on the first pageable call, under `mu`; published only when all its threads run)
    bool stagers_failed = false;           // thread creation failed once: pageable calls stage inline from then on
    bool no_block_bound = false, f32_rank = false, small_rank_tiles = false, wg_prep = false, exact_heap = false,
         force_rank_fallback = false, exact_rank = false; // rbq_debug_set_option
    bool head_exact = true;  // lazy selection: a bound of the k-th distance from real estimates of the nearest list's first vectors
    bool lazy_filter = true; // search_filtered: lazy selection on the exact head evaluation's bound (filter-passing vectors only)
    bool lazy_fault_inject = false; // TEST ONLY: makes the lazy selection wrong on purpose (tests/test_gpu_round4.py: the audit must notice)
    SlackMul slack;          // TEST ONLY (options slack_term / slack_milli): multipliers of block_ub()'s rounding-slack terms
    int slack_term = 0;
    bool lazy_audit = false; // DIAGNOSTIC (option lazy_audit): the select kernel exports the lists it drops as a whole (workspace "audit_dead")
    bool lazy_select = true; // probe selection drops lists that are provably skipped as a whole (rank_mfma.hpp)
    int host_wave_policy = 0;  // option host_wave_policy (see search_host)
    uint32_t tie_log_cap = 0;  // TEST ONLY (option tie_log_cap): entries per query of the tie log (0 = the default sizes)
    bool tie_log = true;       // option tie_log: k_scan logs the candidates it refines; a tied query replays the log (scan.hpp)
    int latency_path = 1;      // option latency_path: small calls (see kLatMaxQueries) take the latency-first front (latency.hpp); 0 = never
    int rank_tile = 0;         // option rank_tile: tile of the split-bf16 ranking GEMM (0 = by problem size)
    uint32_t stage_mask = 0xf; // DIAGNOSTIC (option stage_mask): bit s = launch stage s (prep, rank, select, scan); a skipped stage leaves the
                               // workspace of the stream as the last full call wrote it — results are then those of THAT batch (rate probes only)
    int scan_wave = scan_wave_default(); // which scan kernel serves a call: 0 = k_scan (one workgroup per query), 1 = k_scanw (one wave per
                                         // query) wherever it serves the call shape, 2 = by batch size (kScanWaveMinQueries)
    bool profile_counters = true; // an open profile keeps the traffic counters (option profile_counters = 0: stage timings only —
                                  // the counters cost the pipelined run 2-3 %, bench.py collects them in a pass of their own)
    // host
    std::vector<uint32_t> h_list_n;
    std::vector<uint64_t> nblk_desc_prefix; // prefix sums of per-list block counts sorted descending
    std::mutex mu;
    std::vector<Workspace*> pool;
    std::map<hipStream_t, Workspace*> stream_ws; // rbq_search_batch_device: one workspace per caller stream
    // profiling
    bool profiling = false;
    uint32_t prof_mask = 0xf; // stages that are timed while `profiling` (bit s = stage s)
    uint32_t prof_every = 1;  // time every n-th launch of a stage
    uint32_t prof_seq[4] = {0, 0, 0, 0};
    StageProf stage_prof[4]; // prep, rank, select, scan
    EventPool ev_pool;
    uint64_t prof_counters[kProfSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
    Replica() = default;
    Replica(const Replica&) = delete;
};

} // namespace

namespace {
// Persistent host threads per replica beyond the first (rbq_search_batch on N replicas): the shard of replica r is
// enqueued and awaited by a worker of replica r while the caller's own thread serves replica 0.  Started on the first
// multi-replica call, joined when the index is destroyed — no thread is created per call (8 replicas: seven thread start-ups
// of 30-50 us each per call were as long as a 1024-query shard itself).  Two threads per replica, so that the shards of two
// concurrent callers overlap on it; a worker that has just finished a job polls for the next one for ~100 us before it
// blocks (a caller that issues calls back to back does not pay a wake-up per call).
struct ReplicaWorker {
    static constexpr int kThreadsPerReplica = 2;
    std::thread th[kThreadsPerReplica];
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> jobs;
    std::atomic<uint32_t> pending{0};
    bool stop = false;
    void run() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return stop || !jobs.empty(); });
                if (jobs.empty()) return; // stop requested and nothing left
                job = std::move(jobs.front());
                jobs.pop_front();
            }
            job();
            pending.fetch_sub(1, std::memory_order_release);
            const auto t0 = std::chrono::steady_clock::now(); // short poll before blocking again
            while (pending.load(std::memory_order_acquire) == 0 &&
                   std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(100)) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
        }
    }
    // false: a thread could not be created (the ones that did start are shut down again; the caller stages inline)
    bool start() {
        try {
            for (auto& t : th) t = std::thread([this] { run(); });
        } catch (...) {
            shutdown();
            return false;
        }
        return true;
    }
    void post(std::function<void()> job) {
        pending.fetch_add(1, std::memory_order_release);
        { std::lock_guard<std::mutex> lk(mu); jobs.push_back(std::move(job)); }
        cv.notify_one();
    }
    void shutdown() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto& t : th) if (t.joinable()) t.join();
    }
};
} // namespace

struct rbq_index {
    std::vector<Replica*> reps; // reps[r] lives on device reps[r]->device; all hold the same index
    int debug_replica = 0;      // which replica the rbq_debug_copy_* calls read
    std::mutex worker_mu;
    std::vector<std::unique_ptr<ReplicaWorker>> workers; // workers[r - 1] serves replica r (created on first use)
};

namespace {

int alloc_arr(Arr& a, size_t bytes) {
    HIP_TRY(hipMalloc(&a.p, bytes ? bytes : 16));
    a.bytes = bytes;
    return RBQ_OK;
}
int upload_arr(Arr& a, const void* src, size_t bytes) {
    int rc = alloc_arr(a, bytes);
    if (rc) return rc;
    if (bytes) HIP_TRY(hipMemcpy(a.p, src, bytes, hipMemcpyHostToDevice));
    return RBQ_OK;
}

void free_replica(Replica* ix) {
    if (!ix) return;
    if (ix->stagers) ix->stagers->shutdown();
    DeviceGuard g(ix->device);
    (void)hipDeviceSynchronize();
    for (Arr* a : ix->arrays)
        if (a->p && !(a == &ix->raw && ix->raw_borrowed)) (void)hipFree(a->p);
    for (Workspace* w : ix->pool) { w->release(); delete w; }
    for (auto& kv : ix->stream_ws) { kv.second->release(); delete kv.second; }
    for (auto& sp : ix->stage_prof)
        for (auto& e : sp.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    ix->ev_pool.destroy();
    delete ix;
}
void free_index(rbq_index* h) {
    if (!h) return;
    for (auto& w : h->workers) w->shutdown();
    for (Replica* r : h->reps) free_replica(r);
    delete h;
}

int validate_header(const rbq_header* h) {
    if (!h) return fail(RBQ_INVALID_CONFIG, "null header");
    if (h->dim == 0) return fail(RBQ_INVALID_CONFIG, "dimension must be positive");
    if (h->padded_dim < h->dim) return fail(RBQ_INVALID_CONFIG, "padded_dim must be >= dim");
    if (h->metric > 1) return fail(RBQ_INVALID_CONFIG, "unknown metric tag");
    if (h->rotator > RBQ_ROTATOR_NONE) return fail(RBQ_INVALID_CONFIG, "unknown rotator type tag");
    if (h->ex_bits != 0 && h->ex_bits != 2 && h->ex_bits != 6)
        return fail(RBQ_INVALID_CONFIG, "Unsupported ex_bits: only 0 (1-bit total), 2 (3-bit total), and 6 (7-bit total) are supported");
    if (h->padded_dim % 16 != 0) return fail(RBQ_INVALID_CONFIG, "Dimension must be multiple of 16 for SIMD");
    if (h->padded_dim > 2048)
        return fail(RBQ_INVALID_CONFIG, "padded_dim > 2048 (high-accuracy i32 LUT mode) is not supported");
    if (h->rotator_len != 0 && !h->rotator_blob) return fail(RBQ_INVALID_CONFIG, "null rotator blob");
    if (h->rotator == RBQ_ROTATOR_NONE) {
        if (h->padded_dim != h->dim) return fail(RBQ_INVALID_CONFIG, "rotator NONE requires padded_dim == dim");
        if (h->rotator_len != 0) return fail(RBQ_INVALID_CONFIG, "rotator NONE takes no rotator blob");
    } else if (h->rotator == RBQ_ROTATOR_FHT_KAC) {
        if (h->padded_dim % 64 != 0) return fail(RBQ_INVALID_CONFIG, "FHT rotator requires dimension to be multiple of 64");
        if (h->rotator_len != (uint64_t)4 * h->padded_dim / 8) return fail(RBQ_INVALID_PERSISTENCE, "FHT rotator flip bits length mismatch");
    } else {
        if (h->rotator_len != (uint64_t)h->padded_dim * h->padded_dim * 4) return fail(RBQ_INVALID_PERSISTENCE, "rotator matrix length mismatch");
    }
    if (h->n_lists == 0) return fail(RBQ_INVALID_CONFIG, "nlist must be positive");
    if (h->n_lists > 0xffffffffull) return fail(RBQ_INVALID_CONFIG, "too many lists");
    return RBQ_OK;
}

// device list: n_devices ordinals, or the current device
int resolve_devices(int n_devices, const int* devices, std::vector<int>& out) {
    if (n_devices < 1 || n_devices > 16) return fail(RBQ_INVALID_CONFIG, "n_devices must be in 1..16");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    out.clear();
    for (int i = 0; i < n_devices; ++i) {
        int d = 0;
        if (devices) d = devices[i];
        else if (n_devices == 1) HIP_TRY(hipGetDevice(&d));
        else d = i;
        if (d < 0 || d >= count) return fail(RBQ_INVALID_CONFIG, "no such device: " + std::to_string(d));
        out.push_back(d);
    }
    return RBQ_OK;
}

Replica* new_replica(const rbq_header* hdr, int dev) {
    Replica* ix = new Replica();
    ix->device = dev;
    ix->dim = hdr->dim; ix->D = hdr->padded_dim; ix->Dc = (hdr->padded_dim + 63u) / 64u * 64u;
    ix->metric = hdr->metric; ix->rotator = hdr->rotator; ix->ex_bits = hdr->ex_bits;
    ix->n_lists = hdr->n_lists;
    ix->trunc = 1u << floor_log2_u32(hdr->dim);
    ix->fac = 1.0f / std::sqrt((float)ix->trunc);
    return ix;
}

// what every construction path shares once list sizes and rotated centroids are on the device: block geometry on
// the host side, centroid-derived arrays, counters, environment switches
int finish_replica(Replica* ix, const std::vector<uint32_t>& ln) {
    const uint32_t nlist = (uint32_t)ix->n_lists, D = ix->D;
    ix->h_list_n = ln;
    {
        std::vector<uint64_t> nblk(nlist);
        for (uint32_t c = 0; c < nlist; ++c) nblk[c] = (ln[c] + 31u) / 32u;
        std::sort(nblk.begin(), nblk.end(), std::greater<uint64_t>());
        ix->nblk_desc_prefix.assign((size_t)nlist + 1, 0);
        for (uint32_t c = 0; c < nlist; ++c) ix->nblk_desc_prefix[c + 1] = ix->nblk_desc_prefix[c] + nblk[c];
    }
    int rc;
    if ((rc = alloc_arr(ix->cnorm2, (size_t)nlist * 4))) return rc;
    if ((rc = alloc_arr(ix->cent_hi, (size_t)nlist * D * 2))) return rc;
    if ((rc = alloc_arr(ix->cent_lo, (size_t)nlist * D * 2))) return rc;
    HIP_TRY(launch_centroid_arrays((const float*)ix->centroids.p, nlist, D, (float*)ix->cnorm2.p, (uint16_t*)ix->cent_hi.p,
                                   (uint16_t*)ix->cent_lo.p, 0));
    std::vector<float> cn(nlist);
    HIP_TRY(hipMemcpy(cn.data(), ix->cnorm2.p, (size_t)nlist * 4, hipMemcpyDeviceToHost));
    double mx = 0;
    for (float v : cn) if (std::isfinite(v)) mx = std::max(mx, (double)v);
    ix->cnorm2_max = (float)(mx * 1.000001); // rounded up
    if ((rc = alloc_arr(ix->lsum, (size_t)nlist * sizeof(BlockSummary)))) return rc;
    if ((rc = alloc_arr(ix->bsumx, ix->n_blocks * sizeof(BlockSummaryEx)))) return rc;
    HIP_TRY(launch_list_summaries((const uint8_t*)ix->blocks.p, (const uint8_t*)ix->ex.p, (const float*)ix->fadd_ex.p,
                                  (const float*)ix->fres_ex.p, (const float*)ix->centroids.p, (const BlockSummary*)ix->bsum.p,
                                  (const uint32_t*)ix->list_gb0.p, (const uint32_t*)ix->list_n.p, nlist, D, ix->Dc, ix->ex_bits,
                                  (BlockSummaryEx*)ix->bsumx.p, (BlockSummary*)ix->lsum.p, 0));
    if ((rc = alloc_arr(ix->fallbacks, 32))) return rc;   // [0] rank fallbacks, [1] heap restarts, [2] exact-head guard trips, [3] exact-head evaluations,
                                                          // [4..7] tie log: replays, entries replayed, real heap operations, overflowed logs
    HIP_TRY(hipMemset(ix->fallbacks.p, 0, 32));
    if ((rc = alloc_arr(ix->prof, (size_t)kProfStripes * kProfSlots * 8))) return rc;
    HIP_TRY(hipMemset(ix->prof.p, 0, (size_t)kProfStripes * kProfSlots * 8));
    const char* e = std::getenv("RBQ_EXACT_RANK");
    ix->exact_rank = e && e[0] == '1';
    const char* f = std::getenv("RBQ_FORCE_RANK_FALLBACK");
    ix->force_rank_fallback = f && f[0] == '1';
    const char* lz = std::getenv("RBQ_LAZY_SELECT");
    ix->lazy_select = !(lz && lz[0] == '0');
    return RBQ_OK;
}

// Device-to-device copy between two replicas' devices: peer copy over xGMI when the runtime can (peer access is enabled
// when the devices report it; hipMemcpyPeer itself stages through the host otherwise); if that fails, an explicit bounce
// through a page-locked host buffer, 64 MB at a time.
// RBQ_FORCE_NO_PEER=1 in the environment (read when a handle replicates): every replica copy takes the bounce path, also
// between two replicas on ONE device — the only way the path can run on a one-GPU box (tests/test_gpu_round4.py).
std::atomic<uint64_t> g_bounce_copies{0}; // replica arrays copied through the page-locked bounce buffer (rbq_debug_bounce_copies)
hipError_t copy_cross_device(void* dst, int ddev, const void* src, int sdev, size_t bytes) {
    if (!bytes) return hipSuccess;
    const char* fnp = std::getenv("RBQ_FORCE_NO_PEER");
    const bool no_peer = fnp && fnp[0] == '1';
    hipError_t e = hipSuccess;
    if (!no_peer) {
        if (ddev == sdev) return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice);
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, ddev, sdev) == hipSuccess && can) {
            const hipError_t pe = hipDeviceEnablePeerAccess(sdev, 0); // (current device = ddev)
            if (pe != hipSuccess) (void)hipGetLastError();            // already enabled, or refused: the copy below decides
        } else {
            (void)hipGetLastError();
        }
        e = hipMemcpyPeer(dst, ddev, src, sdev, bytes);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
    }
    g_bounce_copies.fetch_add(1, std::memory_order_relaxed);
    const size_t CH = (size_t)64 << 20;
    void* bounce = nullptr;
    e = hipHostMalloc(&bounce, std::min(bytes, CH), hipHostMallocPortable);
    if (e != hipSuccess) return e;
    for (size_t off = 0; off < bytes && e == hipSuccess; off += CH) {
        const size_t n = std::min(CH, bytes - off);
        e = hipSetDevice(sdev);
        if (e == hipSuccess) e = hipMemcpy(bounce, (const uint8_t*)src + off, n, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipSetDevice(ddev);
        if (e == hipSuccess) e = hipMemcpy((uint8_t*)dst + off, bounce, n, hipMemcpyHostToDevice);
    }
    (void)hipSetDevice(ddev);
    (void)hipHostFree(bounce);
    return e;
}

// A second replica of `src` on device `dev` (may be the same device: exercised by tests on one GPU).
int clone_replica(const Replica* src, int dev, Replica** out) {
    DeviceGuard g(dev);
    if (!g.ok) return fail(RBQ_DEVICE, "hipSetDevice failed");
    Replica* ix = new Replica();
    ix->device = dev;
    ix->dim = src->dim; ix->D = src->D; ix->Dc = src->Dc; ix->metric = src->metric; ix->rotator = src->rotator; ix->ex_bits = src->ex_bits;
    ix->n_vectors = src->n_vectors; ix->n_lists = src->n_lists; ix->n_blocks = src->n_blocks; ix->trunc = src->trunc; ix->fac = src->fac;
    ix->cnorm2_max = src->cnorm2_max; ix->h_list_n = src->h_list_n; ix->nblk_desc_prefix = src->nblk_desc_prefix;
    ix->exact_rank = src->exact_rank; ix->force_rank_fallback = src->force_rank_fallback; ix->lazy_select = src->lazy_select; ix->profile_counters = src->profile_counters; ix->head_exact = src->head_exact; ix->lazy_filter = src->lazy_filter;
    for (size_t i = 0; i < sizeof(ix->arrays) / sizeof(ix->arrays[0]); ++i) {
        const Arr* s = src->arrays[i];
        Arr* d = ix->arrays[i];
        if (!s->p || s == &src->raw) continue;
        hipError_t e = hipMalloc(&d->p, s->bytes ? s->bytes : 16);
        if (e == hipSuccess && s->bytes) e = copy_cross_device(d->p, dev, s->p, src->device, s->bytes);
        if (e != hipSuccess) { free_replica(ix); return fail(RBQ_DEVICE, std::string("replicating the index: ") + hipGetErrorString(e)); }
        d->bytes = s->bytes;
    }
    hipError_t e = hipMemset(ix->fallbacks.p, 0, 32);
    if (e == hipSuccess) e = hipMemset(ix->prof.p, 0, (size_t)kProfStripes * kProfSlots * 8);
    if (e != hipSuccess) { free_replica(ix); return fail(RBQ_DEVICE, hipGetErrorString(e)); }
    *out = ix;
    return RBQ_OK;
}

int wrap_and_replicate(Replica* first, const std::vector<int>& devs, rbq_index** out) {
    rbq_index* h = new rbq_index();
    h->reps.push_back(first);
    for (size_t i = 1; i < devs.size(); ++i) {
        Replica* r = nullptr;
        int rc = clone_replica(first, devs[i], &r);
        if (rc) { free_index(h); return rc; }
        h->reps.push_back(r);
    }
    *out = h;
    return RBQ_OK;
}

// ---- create / load: reference layout in, device layout out --------------------------------------------------------
int create_from_sources(const rbq_header* hdr, const std::vector<ListSrc>& lists, int dev, Replica** out) {
    DeviceGuard g(dev);
    if (!g.ok) return fail(RBQ_DEVICE, "hipSetDevice failed");
    Replica* ix = new_replica(hdr, dev);
    struct Cleanup { Replica*& ix; std::vector<void*> dev_tmp; std::vector<void*> pin_tmp; std::vector<hipEvent_t> evs; hipStream_t st = nullptr;
        ~Cleanup() { for (void* p : dev_tmp) (void)hipFree(p); for (void* p : pin_tmp) (void)hipHostFree(p);
                     for (hipEvent_t e : evs) (void)hipEventDestroy(e); if (st) (void)hipStreamDestroy(st); if (ix) free_replica(ix); } } cl{ix};
    const uint32_t D = ix->D, Dc = ix->Dc, nlist = (uint32_t)ix->n_lists, exbits = ix->ex_bits;
    const size_t ref_stride = (size_t)D * 4 + 384, dev_stride = (size_t)Dc * 4 + 384, exb = (size_t)D * exbits / 8;
    const size_t exd = ex_bytes_dev(D, exbits);
    std::vector<uint32_t> gb0(nlist), ln(nlist);
    uint64_t nblocks = 0, nvec = 0;
    for (uint32_t c = 0; c < nlist; ++c) {
        const ListSrc& L = lists[c];
        if (L.n > 0xffffffffull) return fail(RBQ_INVALID_CONFIG, "list too large");
        gb0[c] = (uint32_t)nblocks; ln[c] = (uint32_t)L.n;
        nblocks += (L.n + 31) / 32; nvec += L.n;
        if (nblocks * 32 > 0xffffffffull) return fail(RBQ_INVALID_CONFIG, "index too large for 32-bit vector slots");
    }
    ix->n_blocks = nblocks; ix->n_vectors = nvec;
    const uint64_t nslots = nblocks * 32;
    int rc;
    if ((rc = upload_arr(ix->rot_blob, hdr->rotator_blob, hdr->rotator_len))) return rc;
    if ((rc = upload_arr(ix->list_gb0, gb0.data(), (size_t)nlist * 4))) return rc;
    if ((rc = upload_arr(ix->list_n, ln.data(), (size_t)nlist * 4))) return rc;
    {
        std::vector<float> cent((size_t)nlist * D);
        for (uint32_t c = 0; c < nlist; ++c) std::memcpy(&cent[(size_t)c * D], lists[c].centroid, (size_t)D * 4);
        if ((rc = upload_arr(ix->centroids, cent.data(), cent.size() * 4))) return rc;
    }
    if ((rc = alloc_arr(ix->blocks, nblocks * dev_stride))) return rc;
    if ((rc = alloc_arr(ix->ids, nslots * 8))) return rc;
    if ((rc = alloc_arr(ix->ex, exd ? nslots * exd + 256 : 0))) return rc;
    if ((rc = alloc_arr(ix->fadd_ex, exbits ? nslots * 4 : 0))) return rc;
    if ((rc = alloc_arr(ix->fres_ex, exbits ? nslots * 4 : 0))) return rc;
    if ((rc = alloc_arr(ix->bsum, nblocks * sizeof(BlockSummary)))) return rc;

    // Chunks of whole blocks (a long list may span several), staged through two pinned buffers: the host fills
    // one while the GPU converts the other.  Staging layout (16-byte aligned sections):
    //   recs [nb][ref_stride] | ex [nv][exb] | ids [nv] u64 | fadd [nv] f32 | fres [nv] f32 | dense0 [nb] u64 | nvb [nb] u32
    const size_t per_block = ref_stride + 32 * (exb + 16) + 12 + 64;
    uint64_t chunk_blocks = std::max<uint64_t>(1, ((size_t)64 << 20) / per_block);
    chunk_blocks = std::min<uint64_t>(chunk_blocks, std::max<uint64_t>(nblocks, 1));
    const size_t cap = align_up(chunk_blocks * ref_stride, 16) + align_up(chunk_blocks * 32 * exb, 16) + chunk_blocks * 32 * 16 +
                       chunk_blocks * 12 + 256;
    uint8_t* pin[2] = {nullptr, nullptr};
    uint8_t* dst[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    HIP_TRY(hipStreamCreateWithFlags(&cl.st, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipHostMalloc((void**)&pin[i], cap, hipHostMallocDefault)); cl.pin_tmp.push_back(pin[i]);
        HIP_TRY(hipMalloc((void**)&dst[i], cap)); cl.dev_tmp.push_back(dst[i]);
        HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); cl.evs.push_back(ev[i]);
    }
    uint32_t c = 0;      // list cursor
    uint64_t lb = 0;     // next block inside list c
    uint64_t b0 = 0;     // first global block of the chunk
    int turn = 0;
    while (b0 < nblocks) {
        const uint64_t nb = std::min<uint64_t>(chunk_blocks, nblocks - b0);
        const int i = turn & 1;
        if (turn >= 2) HIP_TRY(hipEventSynchronize(ev[i]));
        uint8_t* base = pin[i];
        const size_t o_recs = 0, o_ex = align_up(nb * ref_stride, 16);
        // the number of vectors of the chunk is known only after the walk: lay the dense arrays out for the maximum
        const size_t o_ids = o_ex + align_up(nb * 32 * exb, 16), o_fadd = o_ids + nb * 32 * 8, o_fres = o_fadd + nb * 32 * 4,
                     o_d0 = o_fres + nb * 32 * 4, o_nv = o_d0 + nb * 8, total = o_nv + nb * 4;
        uint64_t* dense0 = reinterpret_cast<uint64_t*>(base + o_d0);
        uint32_t* nvb = reinterpret_cast<uint32_t*>(base + o_nv);
        uint64_t filled = 0, dense = 0;
        while (filled < nb) {
            while (c < nlist && lb >= ((uint64_t)ln[c] + 31) / 32) { ++c; lb = 0; }
            const ListSrc& L = lists[c];
            const uint64_t lnb = ((uint64_t)ln[c] + 31) / 32, take = std::min<uint64_t>(lnb - lb, nb - filled);
            const uint64_t v0 = lb * 32, v1 = std::min<uint64_t>(L.n, (lb + take) * 32), nv = v1 - v0;
            std::memcpy(base + o_recs + filled * ref_stride, L.batch_data + lb * ref_stride, take * ref_stride);
            if (nv) std::memcpy(base + o_ids + dense * 8, L.ids + v0 * 8, nv * 8);
            if (exbits && nv) {
                if (L.ex_stride == exb) std::memcpy(base + o_ex + dense * exb, L.ex + v0 * exb, nv * exb);
                else for (uint64_t v = 0; v < nv; ++v) std::memcpy(base + o_ex + (dense + v) * exb, L.ex + (v0 + v) * L.ex_stride, exb);
                std::memcpy(base + o_fadd + dense * 4, L.fadd + v0 * 4, nv * 4);
                std::memcpy(base + o_fres + dense * 4, L.fres + v0 * 4, nv * 4);
            }
            for (uint64_t b = 0; b < take; ++b) {
                dense0[filled + b] = dense + b * 32;
                nvb[filled + b] = (uint32_t)std::min<uint64_t>(32, L.n - (lb + b) * 32);
            }
            filled += take; dense += nv; lb += take;
        }
        uint8_t* d = dst[i];
        HIP_TRY(hipMemcpyAsync(d, base, total, hipMemcpyHostToDevice, cl.st));
        const uint64_t* d_d0 = reinterpret_cast<const uint64_t*>(d + o_d0);
        const uint32_t* d_nv = reinterpret_cast<const uint32_t*>(d + o_nv);
        uint8_t* oblocks = (uint8_t*)ix->blocks.p + b0 * dev_stride;
        HIP_TRY(launch_relayout_blocks(d + o_recs, (uint32_t)nb, D, Dc, oblocks, cl.st));
        HIP_TRY(launch_block_summary(oblocks, d_nv, (uint32_t)nb, Dc, (BlockSummary*)ix->bsum.p + b0, cl.st));
        HIP_TRY(launch_spread_u64(reinterpret_cast<const uint64_t*>(d + o_ids), d_d0, d_nv, (uint32_t)nb, ~0ull,
                                  (uint64_t*)ix->ids.p + b0 * 32, cl.st));
        if (exbits) {
            HIP_TRY(launch_relayout_ex(d + o_ex, d_d0, d_nv, (uint32_t)nb, D, exbits, (uint8_t*)ix->ex.p + b0 * 32 * exd, cl.st));
            HIP_TRY(launch_spread_f32(reinterpret_cast<const float*>(d + o_fadd), d_d0, d_nv, (uint32_t)nb, 0.0f,
                                      (float*)ix->fadd_ex.p + b0 * 32, cl.st));
            HIP_TRY(launch_spread_f32(reinterpret_cast<const float*>(d + o_fres), d_d0, d_nv, (uint32_t)nb, 0.0f,
                                      (float*)ix->fres_ex.p + b0 * 32, cl.st));
        }
        HIP_TRY(hipEventRecord(ev[i], cl.st));
        b0 += nb;
        ++turn;
    }
    if (exd) HIP_TRY(hipMemsetAsync((uint8_t*)ix->ex.p + nslots * exd, 0, 256, cl.st)); // read-ahead pad of the refine loads
    HIP_TRY(hipStreamSynchronize(cl.st));
    if ((rc = finish_replica(ix, ln))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    *out = ix;
    cl.ix = nullptr; // keep
    // (cl's reference member now refers to a null local copy: nothing freed)
    return RBQ_OK;
}

int create_impl(const rbq_header* hdr, const rbq_list_view* lists, int n_devices, const int* devices, rbq_index** out) {
    if (!out) return fail(RBQ_INVALID_CONFIG, "null out pointer");
    *out = nullptr;
    int rc = validate_header(hdr);
    if (rc) return rc;
    if (!lists) return fail(RBQ_INVALID_CONFIG, "null lists");
    std::vector<int> devs;
    if ((rc = resolve_devices(n_devices, devices, devs))) return rc;
    const size_t ref_stride = (size_t)hdr->padded_dim * 4 + 384, exb = (size_t)hdr->padded_dim * hdr->ex_bits / 8;
    std::vector<ListSrc> src(hdr->n_lists);
    for (uint64_t c = 0; c < hdr->n_lists; ++c) {
        const rbq_list_view& L = lists[c];
        if (L.n > 0xffffffffull) return fail(RBQ_INVALID_CONFIG, "list too large");
        const uint64_t nb = (L.n + 31) / 32;
        if (L.batch_len != nb * ref_stride)
            return fail(RBQ_INVALID_PERSISTENCE, "batch_data length mismatch - possible corruption or version incompatibility");
        if (L.n && (!L.centroid || !L.ids || !L.batch_data || (hdr->ex_bits && (!L.ex_codes || !L.f_add_ex || !L.f_rescale_ex))))
            return fail(RBQ_INVALID_CONFIG, "null list array");
        if (!L.centroid) return fail(RBQ_INVALID_CONFIG, "null centroid");
        ListSrc& S = src[c];
        S.centroid = (const uint8_t*)L.centroid; S.n = L.n; S.ids = (const uint8_t*)L.ids; S.batch_data = L.batch_data;
        S.ex = L.ex_codes; S.ex_stride = exb; S.fadd = (const uint8_t*)L.f_add_ex; S.fres = (const uint8_t*)L.f_rescale_ex;
    }
    Replica* first = nullptr;
    if ((rc = create_from_sources(hdr, src, devs[0], &first))) return rc;
    return wrap_and_replicate(first, devs, out);
}

// ---- GPU-side encoder: the device analogue of train_with_clusters' quantisation loop ------------------------------
struct TempDev { // device scratch freed on scope exit
    std::vector<void*> ptrs;
    hipError_t alloc(void** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes ? bytes : 16); if (e == hipSuccess) ptrs.push_back(*p); return e; }
    ~TempDev() { for (void* p : ptrs) if (p) (void)hipFree(p); }
};
struct ReplicaOwner { // frees a half-built replica on an early return
    Replica* ix;
    ~ReplicaOwner() { if (ix) free_replica(ix); }
    Replica* release() { Replica* r = ix; ix = nullptr; return r; }
};

// rotated centroids + list geometry + final arrays of an index that the encoder fills (shared by the one-shot and the
// streamed build).  `counts` = vectors per list.
int encoder_prepare(Replica* ix, const rbq_header* hdr, const float* centroids, const std::vector<uint32_t>& ln,
                    std::vector<uint32_t>& gb0, bool zero_fill) {
    const uint32_t D = ix->D, Dc = ix->Dc, dim = ix->dim, nlist = (uint32_t)ix->n_lists;
    const size_t dev_stride = (size_t)Dc * 4 + 384, exd = ex_bytes_dev(D, ix->ex_bits);
    int rc;
    if ((rc = upload_arr(ix->rot_blob, hdr->rotator_blob, hdr->rotator_len))) return rc;
    {
        TempDev t;
        float* d_craw = nullptr;
        HIP_TRY(t.alloc((void**)&d_craw, (size_t)nlist * dim * 4));
        HIP_TRY(hipMemcpy(d_craw, centroids, (size_t)nlist * dim * 4, hipMemcpyHostToDevice));
        if ((rc = alloc_arr(ix->centroids, (size_t)nlist * D * 4))) return rc;
        HIP_TRY(launch_rotate_rows(d_craw, nullptr, nlist, dim, D, (int)ix->rotator, (const uint8_t*)ix->rot_blob.p, ix->trunc, ix->fac,
                                   (float*)ix->centroids.p, 0));
        HIP_TRY(hipDeviceSynchronize());
    }
    uint64_t nblocks = 0, nvec = 0;
    gb0.resize(nlist);
    for (uint32_t c = 0; c < nlist; ++c) {
        gb0[c] = (uint32_t)nblocks; nblocks += (ln[c] + 31u) / 32u; nvec += ln[c];
        if (nblocks * 32 > 0xffffffffull) return fail(RBQ_INVALID_CONFIG, "index too large for 32-bit vector slots");
    }
    ix->n_blocks = nblocks; ix->n_vectors = nvec;
    const uint64_t nslots = nblocks * 32;
    if ((rc = upload_arr(ix->list_gb0, gb0.data(), (size_t)nlist * 4))) return rc;
    if ((rc = upload_arr(ix->list_n, ln.data(), (size_t)nlist * 4))) return rc;
    if ((rc = alloc_arr(ix->blocks, nblocks * dev_stride))) return rc;
    HIP_TRY(hipMemset(ix->blocks.p, 0, nblocks * dev_stride));
    if ((rc = alloc_arr(ix->ids, nslots * 8))) return rc;
    if ((rc = alloc_arr(ix->ex, exd ? nslots * exd + 256 : 0))) return rc;
    if ((rc = alloc_arr(ix->fadd_ex, ix->ex_bits ? nslots * 4 : 0))) return rc;
    if ((rc = alloc_arr(ix->fres_ex, ix->ex_bits ? nslots * 4 : 0))) return rc;
    if ((rc = alloc_arr(ix->bsum, nblocks * sizeof(BlockSummary)))) return rc;
    if (exd) HIP_TRY(hipMemset((uint8_t*)ix->ex.p + nslots * exd, 0, 256));
    if (zero_fill) { // streamed build: padding slots are never visited by the scatter kernels
        HIP_TRY(hipMemset(ix->ids.p, 0xff, nslots * 8));
        if (exd) HIP_TRY(hipMemset(ix->ex.p, 0, nslots * exd));
        if (ix->ex_bits) { HIP_TRY(hipMemset(ix->fadd_ex.p, 0, nslots * 4)); HIP_TRY(hipMemset(ix->fres_ex.p, 0, nslots * 4)); }
    }
    return RBQ_OK;
}

// block -> list and block -> number of real vectors, on the device
int upload_block_tables(const std::vector<uint32_t>& ln, const std::vector<uint32_t>& gb0, uint64_t nblocks, TempDev& t,
                        uint32_t** d_block_list, uint32_t** d_block_nv) {
    std::vector<uint32_t> bl(nblocks), bn(nblocks);
    for (size_t c = 0; c < ln.size(); ++c) {
        const uint32_t nb = (ln[c] + 31u) / 32u;
        for (uint32_t b = 0; b < nb; ++b) { bl[gb0[c] + b] = (uint32_t)c; bn[gb0[c] + b] = std::min<uint32_t>(32u, ln[c] - b * 32u); }
    }
    HIP_TRY(t.alloc((void**)d_block_list, nblocks * 4)); HIP_TRY(t.alloc((void**)d_block_nv, nblocks * 4));
    HIP_TRY(hipMemcpy(*d_block_list, bl.data(), nblocks * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(*d_block_nv, bn.data(), nblocks * 4, hipMemcpyHostToDevice));
    return RBQ_OK;
}

int build_device_impl(const rbq_header* hdr, const float* centroids, const float* d_data, const uint32_t* d_assign,
                      uint64_t n, float t_const, int dev, rbq_index** out) {
    if (!out) return fail(RBQ_INVALID_CONFIG, "null out pointer");
    *out = nullptr;
    int rc = validate_header(hdr);
    if (rc) return rc;
    if (!centroids || !d_data || !d_assign) return fail(RBQ_INVALID_CONFIG, "null buffer");
    if (n == 0) return fail(RBQ_INVALID_CONFIG, "no vectors");
    if (n > 0xfffffff0ull) return fail(RBQ_INVALID_CONFIG, "too many vectors for 32-bit slots");
    if (hdr->ex_bits > 0 && !(t_const > 0.0f)) return fail(RBQ_INVALID_CONFIG, "the device encoder needs the constant rescale factor (faster config)");
    std::vector<int> devs;
    if ((rc = resolve_devices(1, &dev, devs))) return rc;
    DeviceGuard g(dev);
    if (!g.ok) return fail(RBQ_DEVICE, "hipSetDevice failed");

    ReplicaOwner own{new_replica(hdr, dev)};
    Replica* ix = own.ix;
    const uint32_t D = ix->D, Dc = ix->Dc, dim = ix->dim, nlist = (uint32_t)ix->n_lists;
    const size_t dev_stride = (size_t)Dc * 4 + 384, exd = ex_bytes_dev(D, ix->ex_bits);
    TempDev t;

    // list sizes
    std::vector<uint32_t> ln(nlist), gb0;
    {
        uint32_t* d_counts = nullptr;
        HIP_TRY(t.alloc((void**)&d_counts, (size_t)(nlist + 1) * 4));
        HIP_TRY(hipMemset(d_counts, 0, (size_t)(nlist + 1) * 4));
        HIP_TRY(launch_count_assign(d_assign, n, nlist, d_counts, d_counts + nlist, 0));
        std::vector<uint32_t> hc((size_t)nlist + 1);
        HIP_TRY(hipMemcpy(hc.data(), d_counts, hc.size() * 4, hipMemcpyDeviceToHost));
        if (hc[nlist]) return fail(RBQ_INVALID_CONFIG, "assignment out of range");
        std::copy(hc.begin(), hc.begin() + nlist, ln.begin());
    }
    if ((rc = encoder_prepare(ix, hdr, centroids, ln, gb0, /*zero_fill=*/false))) return rc;
    const uint64_t nblocks = ix->n_blocks, nslots = nblocks * 32;
    std::vector<uint64_t> vstart(nlist);
    { uint64_t run = 0; for (uint32_t c = 0; c < nlist; ++c) { vstart[c] = run; run += ln[c]; } }

    // stable grouping by list (ascending vector index inside a list, src/ivf.rs:1141-1149): radix sort on the list id
    uint32_t* d_slot_src = nullptr;
    {
        uint32_t *d_ko = nullptr, *d_vi = nullptr, *d_vo = nullptr;
        uint64_t* d_vstart = nullptr;
        HIP_TRY(t.alloc((void**)&d_ko, n * 4)); HIP_TRY(t.alloc((void**)&d_vi, n * 4)); HIP_TRY(t.alloc((void**)&d_vo, n * 4));
        HIP_TRY(t.alloc((void**)&d_vstart, (size_t)nlist * 8));
        HIP_TRY(hipMemcpy(d_vstart, vstart.data(), (size_t)nlist * 8, hipMemcpyHostToDevice));
        HIP_TRY(launch_iota(d_vi, n, 0));
        unsigned bits = 1;
        while ((1ull << bits) < nlist) ++bits;
        size_t tb = 0;
        HIP_TRY(sort_pairs_u32(nullptr, &tb, d_assign, d_ko, d_vi, d_vo, (size_t)n, bits, 0));
        void* d_tmp = nullptr;
        HIP_TRY(t.alloc(&d_tmp, tb));
        HIP_TRY(sort_pairs_u32(d_tmp, &tb, d_assign, d_ko, d_vi, d_vo, (size_t)n, bits, 0));
        HIP_TRY(t.alloc((void**)&d_slot_src, nslots * 4));
        HIP_TRY(hipMemset(d_slot_src, 0xff, nslots * 4));
        HIP_TRY(launch_scatter_slots(d_ko, d_vo, n, (const uint32_t*)ix->list_gb0.p, d_vstart, d_slot_src, 0));
    }
    uint32_t *d_block_list = nullptr, *d_block_nv = nullptr;
    if ((rc = upload_block_tables(ln, gb0, nblocks, t, &d_block_list, &d_block_nv))) return rc;

    // encode, a chunk of blocks at a time (scratch: rotated rows + raw ex codes of the chunk)
    {
        uint64_t chunk_blocks = std::max<uint64_t>(2, ((512ull << 20) / ((size_t)D * 4) / 32) & ~1ull);
        chunk_blocks = std::min<uint64_t>(chunk_blocks, (nblocks + 1) & ~1ull);
        const uint64_t chunk_slots = chunk_blocks * 32;
        float* d_rows = nullptr;
        uint8_t* d_raw = nullptr;
        HIP_TRY(t.alloc((void**)&d_rows, chunk_slots * D * 4));
        HIP_TRY(t.alloc((void**)&d_raw, ix->ex_bits ? chunk_slots * D : 16));
        for (uint64_t b0 = 0; b0 < nblocks; b0 += chunk_blocks) {
            const uint64_t nb = std::min<uint64_t>(chunk_blocks, nblocks - b0), ns = nb * 32, s0 = b0 * 32;
            HIP_TRY(launch_rotate_rows(d_data, d_slot_src + s0, (uint32_t)ns, dim, D, (int)ix->rotator, (const uint8_t*)ix->rot_blob.p,
                                       ix->trunc, ix->fac, d_rows, 0));
            EncodeParams P;
            P.rows = d_rows; P.centroids = (const float*)ix->centroids.p; P.slot_src = d_slot_src + s0; P.block_list = d_block_list + b0;
            P.row_slot = nullptr;
            P.blocks = (uint8_t*)ix->blocks.p + b0 * dev_stride; P.raw_ex = d_raw;
            P.f_add_ex = (float*)ix->fadd_ex.p + s0; P.f_rescale_ex = (float*)ix->fres_ex.p + s0; P.ids = (uint64_t*)ix->ids.p + s0;
            P.src_base = 0; P.nslots = (uint32_t)ns; P.D = D; P.Dc = Dc; P.ex_bits = ix->ex_bits; P.metric = ix->metric; P.t_const = t_const;
            HIP_TRY(launch_encode(P, 0));
            if (ix->ex_bits)
                HIP_TRY(launch_pack_ex(d_raw, d_slot_src + s0, nullptr, (uint32_t)ns, D, (uint32_t)ix->ex_bits, (uint8_t*)ix->ex.p + s0 * exd, 0));
        }
        HIP_TRY(launch_block_summary((const uint8_t*)ix->blocks.p, d_block_nv, (uint32_t)nblocks, Dc, (BlockSummary*)ix->bsum.p, 0));
        HIP_TRY(hipDeviceSynchronize());
    }
    if ((rc = finish_replica(ix, ln))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return wrap_and_replicate(own.release(), devs, out);
}

} // namespace

// ---- streamed build: the same encoder fed chunk by chunk (rbq_build_stream_*) -----------------------------------
struct rbq_builder {
    Replica* ix = nullptr;
    int device = 0;
    float t_const = 0.0f;
    std::vector<uint32_t> ln, gb0;
    uint64_t n_total = 0, pushed = 0, next_id = 0;
    uint32_t *d_cursor = nullptr, *d_chunk_first = nullptr, *d_block_list = nullptr, *d_block_nv = nullptr, *d_counts = nullptr;
    DevBuf vec, assign, ko, vi, vo, tmp, row_src, row_slot, rows, raw;
    TempDev tables;
    ~rbq_builder() {
        DeviceGuard g(device);
        for (DevBuf* b : {&vec, &assign, &ko, &vi, &vo, &tmp, &row_src, &row_slot, &rows, &raw}) b->release();
        if (ix) free_replica(ix);
    }
};

namespace {

bool is_device_pointer(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice;
}
// The WHOLE range [p, p + bytes) is page-locked host memory of one allocation / registration: the kernels write (and the
// DMA engines read) every byte of it, so a registration that covers only the first pages must not pass.
bool is_pinned_host_range(const void* p, size_t bytes) {
    if (!p) return false;
    hipPointerAttribute_t a, b;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a.type != hipMemoryTypeHost) return false;
    if (bytes <= 1) return true;
    const uint8_t* last = (const uint8_t*)p + bytes - 1;
    if (hipPointerGetAttributes(&b, last) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (b.type != hipMemoryTypeHost) return false;
    // same mapping: the device-side addresses of the two ends are as far apart as the host-side ones
    if (a.devicePointer && b.devicePointer && (const uint8_t*)b.devicePointer - (const uint8_t*)a.devicePointer != (ptrdiff_t)(bytes - 1)) return false;
    if (a.hostPointer && b.hostPointer && (const uint8_t*)b.hostPointer - (const uint8_t*)a.hostPointer != (ptrdiff_t)(bytes - 1)) return false;
    return true;
}

int stream_begin_impl(const rbq_header* hdr, const float* centroids, const uint32_t* list_sizes, float t_const, int dev,
                      rbq_builder** out) {
    if (!out) return fail(RBQ_INVALID_CONFIG, "null out pointer");
    *out = nullptr;
    int rc = validate_header(hdr);
    if (rc) return rc;
    if (!centroids || !list_sizes) return fail(RBQ_INVALID_CONFIG, "null buffer");
    if (hdr->ex_bits > 0 && !(t_const > 0.0f)) return fail(RBQ_INVALID_CONFIG, "the device encoder needs the constant rescale factor (faster config)");
    std::vector<int> devs;
    if ((rc = resolve_devices(1, &dev, devs))) return rc;
    DeviceGuard g(dev);
    if (!g.ok) return fail(RBQ_DEVICE, "hipSetDevice failed");
    std::unique_ptr<rbq_builder> b(new rbq_builder());
    b->device = dev; b->t_const = t_const;
    b->ix = new_replica(hdr, dev);
    const uint32_t nlist = (uint32_t)hdr->n_lists;
    b->ln.assign(list_sizes, list_sizes + nlist);
    for (uint32_t c = 0; c < nlist; ++c) b->n_total += b->ln[c];
    if (b->n_total == 0) return fail(RBQ_INVALID_CONFIG, "no vectors");
    if ((rc = encoder_prepare(b->ix, hdr, centroids, b->ln, b->gb0, /*zero_fill=*/true))) return rc;
    if ((rc = upload_block_tables(b->ln, b->gb0, b->ix->n_blocks, b->tables, &b->d_block_list, &b->d_block_nv))) return rc;
    HIP_TRY(b->tables.alloc((void**)&b->d_cursor, (size_t)nlist * 4));
    HIP_TRY(b->tables.alloc((void**)&b->d_chunk_first, (size_t)nlist * 4));
    HIP_TRY(b->tables.alloc((void**)&b->d_counts, (size_t)(nlist + 1) * 4));
    HIP_TRY(hipMemset(b->d_cursor, 0, (size_t)nlist * 4));
    HIP_TRY(hipMemset(b->d_counts, 0, (size_t)(nlist + 1) * 4));
    *out = b.release();
    return RBQ_OK;
}

int stream_push_impl(rbq_builder* b, const float* vectors, const uint32_t* assign, uint64_t first_id, uint64_t count) {
    if (!b || !b->ix) return fail(RBQ_INVALID_CONFIG, "null builder");
    if (count == 0) return RBQ_OK;
    if (!vectors || !assign) return fail(RBQ_INVALID_CONFIG, "null buffer");
    if (first_id < b->next_id) return fail(RBQ_INVALID_CONFIG, "chunks must be pushed in ascending id order (list membership order, src/ivf.rs:1141-1149)");
    if (b->pushed + count > b->n_total) return fail(RBQ_INVALID_CONFIG, "more vectors pushed than the list sizes announced");
    DeviceGuard g(b->device);
    if (!g.ok) return fail(RBQ_DEVICE, "hipSetDevice failed");
    Replica* ix = b->ix;
    const uint32_t D = ix->D, Dc = ix->Dc, dim = ix->dim, nlist = (uint32_t)ix->n_lists;
    const size_t exd = ex_bytes_dev(D, ix->ex_bits);
    const bool vec_dev = is_device_pointer(vectors), asg_dev = is_device_pointer(assign);
    // sub-chunks bounded by the scratch for the rotated rows (512 MB)
    const uint64_t SUB = std::max<uint64_t>(1024, ((512ull << 20) / ((size_t)D * 4)) & ~63ull);
    unsigned bits = 1;
    while ((1ull << bits) < nlist) ++bits;
    int rc;
    for (uint64_t s0 = 0; s0 < count; s0 += SUB) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(SUB, count - s0);
        const float* d_vec = vectors + s0 * dim;
        const uint32_t* d_asg = assign + s0;
        if (!vec_dev) {
            if ((rc = b->vec.ensure((size_t)n * dim * 4))) return rc;
            HIP_TRY(hipMemcpy(b->vec.p, vectors + s0 * dim, (size_t)n * dim * 4, hipMemcpyHostToDevice));
            d_vec = (const float*)b->vec.p;
        }
        if (!asg_dev) {
            if ((rc = b->assign.ensure((size_t)n * 4))) return rc;
            HIP_TRY(hipMemcpy(b->assign.p, assign + s0, (size_t)n * 4, hipMemcpyHostToDevice));
            d_asg = (const uint32_t*)b->assign.p;
        }
        // counts so far incl. this sub-chunk: no list may outgrow its announced size (its slots are fixed)
        HIP_TRY(launch_count_assign(d_asg, n, nlist, b->d_counts, b->d_counts + nlist, 0));
        {
            std::vector<uint32_t> hc((size_t)nlist + 1);
            HIP_TRY(hipMemcpy(hc.data(), b->d_counts, hc.size() * 4, hipMemcpyDeviceToHost));
            if (hc[nlist]) return fail(RBQ_INVALID_CONFIG, "assignment out of range");
            for (uint32_t c = 0; c < nlist; ++c)
                if (hc[c] > b->ln[c]) return fail(RBQ_INVALID_CONFIG, "list " + std::to_string(c) + " received more vectors than announced");
        }
        if ((rc = b->ko.ensure((size_t)n * 4)) || (rc = b->vi.ensure((size_t)n * 4)) || (rc = b->vo.ensure((size_t)n * 4)) ||
            (rc = b->row_src.ensure((size_t)n * 4)) || (rc = b->row_slot.ensure((size_t)n * 4)) ||
            (rc = b->rows.ensure((size_t)n * D * 4)) || (rc = b->raw.ensure(ix->ex_bits ? (size_t)n * D : 16)))
            return rc;
        HIP_TRY(launch_iota((uint32_t*)b->vi.p, n, 0));
        size_t tb = 0;
        HIP_TRY(sort_pairs_u32(nullptr, &tb, d_asg, (uint32_t*)b->ko.p, (const uint32_t*)b->vi.p, (uint32_t*)b->vo.p, n, bits, 0));
        if ((rc = b->tmp.ensure(tb))) return rc;
        HIP_TRY(sort_pairs_u32(b->tmp.p, &tb, d_asg, (uint32_t*)b->ko.p, (const uint32_t*)b->vi.p, (uint32_t*)b->vo.p, n, bits, 0));
        HIP_TRY(launch_chunk_first((const uint32_t*)b->ko.p, n, b->d_chunk_first, 0));
        HIP_TRY(launch_chunk_slots((const uint32_t*)b->ko.p, (const uint32_t*)b->vo.p, n, (const uint32_t*)ix->list_gb0.p, b->d_cursor,
                                   b->d_chunk_first, (uint32_t*)b->row_src.p, (uint32_t*)b->row_slot.p, 0));
        HIP_TRY(launch_rotate_rows(d_vec, (const uint32_t*)b->row_src.p, n, dim, D, (int)ix->rotator, (const uint8_t*)ix->rot_blob.p,
                                   ix->trunc, ix->fac, (float*)b->rows.p, 0));
        EncodeParams P;
        P.rows = (const float*)b->rows.p; P.centroids = (const float*)ix->centroids.p; P.slot_src = (const uint32_t*)b->row_src.p;
        P.block_list = b->d_block_list; P.row_slot = (const uint32_t*)b->row_slot.p;
        P.blocks = (uint8_t*)ix->blocks.p; P.raw_ex = (uint8_t*)b->raw.p;
        P.f_add_ex = (float*)ix->fadd_ex.p; P.f_rescale_ex = (float*)ix->fres_ex.p; P.ids = (uint64_t*)ix->ids.p;
        P.src_base = first_id + s0; P.nslots = n; P.D = D; P.Dc = Dc; P.ex_bits = ix->ex_bits; P.metric = ix->metric; P.t_const = b->t_const;
        HIP_TRY(launch_encode(P, 0));
        if (ix->ex_bits)
            HIP_TRY(launch_pack_ex((const uint8_t*)b->raw.p, (const uint32_t*)b->row_src.p, (const uint32_t*)b->row_slot.p, n, D,
                                   (uint32_t)ix->ex_bits, (uint8_t*)ix->ex.p, 0));
        (void)exd;
        HIP_TRY(launch_chunk_advance((const uint32_t*)b->ko.p, n, b->d_chunk_first, b->d_cursor, 0));
        HIP_TRY(hipDeviceSynchronize()); // the scratch (and a host caller's buffers) are reused by the next sub-chunk
    }
    b->pushed += count;
    b->next_id = first_id + count;
    return RBQ_OK;
}

int stream_finish_impl(rbq_builder* b, int n_devices, const int* devices, rbq_index** out) {
    if (!out) return fail(RBQ_INVALID_CONFIG, "null out pointer");
    *out = nullptr;
    if (!b || !b->ix) return fail(RBQ_INVALID_CONFIG, "null builder");
    if (b->pushed != b->n_total) return fail(RBQ_INVALID_CONFIG, "list sizes do not match the pushed vectors");
    std::vector<int> devs;
    int rc;
    if (n_devices <= 1 && !devices) devs.push_back(b->device);
    else if ((rc = resolve_devices(n_devices, devices, devs))) return rc;
    if (devs[0] != b->device) return fail(RBQ_INVALID_CONFIG, "devices[0] must be the device the builder was opened on");
    DeviceGuard g(b->device);
    if (!g.ok) return fail(RBQ_DEVICE, "hipSetDevice failed");
    Replica* ix = b->ix;
    HIP_TRY(launch_block_summary((const uint8_t*)ix->blocks.p, b->d_block_nv, (uint32_t)ix->n_blocks, ix->Dc, (BlockSummary*)ix->bsum.p, 0));
    HIP_TRY(hipDeviceSynchronize());
    if ((rc = finish_replica(ix, b->ln))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    b->ix = nullptr;
    return wrap_and_replicate(ix, devs, out);
}

// ---- search ---------------------------------------------------------------------------------------------------------
Workspace* take_ws(Replica* ix) {
    {
        std::lock_guard<std::mutex> g(ix->mu);
        if (!ix->pool.empty()) { Workspace* w = ix->pool.back(); ix->pool.pop_back(); return w; }
    }
    Workspace* w = new Workspace();
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) { delete w; return nullptr; }
    if (hipEventCreateWithFlags(&w->done, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(w->stream); delete w; return nullptr; }
    return w;
}
void give_ws(Replica* ix, Workspace* w) {
    std::lock_guard<std::mutex> g(ix->mu);
    ix->pool.push_back(w);
}

struct ProfScope {
    Replica* ix; int stage; hipStream_t s; std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    bool on = false, ext = false;
    // ext: the launch itself carries the event pair (hipExtLaunchKernelGGL: start/stop come from the dispatch
    // packet, no separate marker packets in the queue); otherwise the pair is recorded around the scope
    ProfScope(Replica* ix_, int st, hipStream_t s_, bool ext_ = false) : ix(ix_), stage(st), s(s_), ext(ext_) {
        if (ix->profiling && !stage_probes() && ((ix->prof_mask >> st) & 1u)) { // (a probed call launches nothing: no event pair for it)
            {
                std::lock_guard<std::mutex> g(ix->mu);
                if (ix->prof_seq[st]++ % ix->prof_every == 0) on = ix->ev_pool.take(ev);
            }
            if (on && !ext) (void)hipEventRecord(ev.first, s);
        }
    }
    hipEvent_t start() const { return on && ext ? ev.first : nullptr; }
    hipEvent_t stop() const { return on && ext ? ev.second : nullptr; }
    ~ProfScope() {
        if (on) {
            if (!ext) (void)hipEventRecord(ev.second, s);
            std::lock_guard<std::mutex> g(ix->mu);
            ix->stage_prof[stage].ev.push_back(ev);
        }
    }
};

// k_scan launch shared by the IVF search and the MSTG posting-list scan
int scan_stage(Replica* ix, Workspace* w, uint64_t nq, uint32_t probe_stride, uint32_t top_k, uint64_t wl_stride,
               const uint32_t* d_filter, uint64_t filter_nbits, uint64_t* d_ids, float* d_scores, uint32_t* d_counts,
               rbq_diag* d_diag, bool mstg, const uint32_t* d_dead_skipped, hipStream_t stream) {
    ProfScope ps(ix, 3, stream, /*ext=*/true);
    ScanParams P;
    P.blocks = (const uint8_t*)ix->blocks.p; P.ids = (const uint64_t*)ix->ids.p; P.ex_codes = (const uint8_t*)ix->ex.p;
    P.f_add_ex = (const float*)ix->fadd_ex.p; P.f_rescale_ex = (const float*)ix->fres_ex.p;
    P.lut = (const uint8_t*)w->lut.p; P.rot = (const float*)w->rot.p; P.consts = (const QueryConsts*)w->consts.p;
    P.probe = (const ProbeInfo*)w->probe.p; P.wl = (const StreamItem*)w->wl.p; P.nstream = (const uint32_t*)w->nstream.p;
    P.filter = d_filter; P.filter_nbits = filter_nbits; P.wl_stride = wl_stride;
    P.out_ids = d_ids; P.out_scores = d_scores; P.out_counts = d_counts; P.diag = (unsigned long long*)d_diag;
    P.D = ix->D; P.Dc = ix->Dc; P.nprobe = probe_stride; P.top_k = top_k; P.metric = ix->metric;
    P.ex_bits = mstg ? 0u : ix->ex_bits; // MSTG search never evaluates the ex codes (src/mstg/index.rs:216-330)
    P.no_block_bound = ix->no_block_bound ? 1u : 0u;
    P.exact_heap = ix->exact_heap ? 1u : 0u;
    P.heap_restarts = (unsigned int*)ix->fallbacks.p + 1;
    P.tie_stats = (unsigned int*)ix->fallbacks.p + 4;
    P.mstg = mstg ? 1u : 0u;
    P.prof = (ix->profiling && ix->profile_counters) ? (unsigned long long*)ix->prof.p : nullptr;
    P.dead_skipped = d_dead_skipped;
    // scan_wave = 2 (default): the wave-per-query kernel serves the pruned regime of batches large enough to fill the chip with
    // waves; with the block bound off (every probed block streamed: the roofline configuration) the four-wave kernel streams
    // at twice its rate and serves the call.  Results are identical either way.
    // A host call (rbq_search_batch) that waits for ONE chain of kernels is served by the kernel with the shorter chain (k_scan's
    // four waves finish a 1024-query launch in 0.10 ms, k_scanw's single wave in 0.14 ms) until the call is large enough for the rate
    // to matter more than the last chain's latency.
    P.wave_kernel = (ix->scan_wave == 1 || (ix->scan_wave == 2 && nq >= kScanWaveMinQueries && !ix->no_block_bound && !w->latency_first)) ? 1u : 0u;
    // tie log (k_scan, no diagnostics, top-k in registers): a tied query replays its logged candidates instead of its lists
    P.tie_log = nullptr; P.tie_log_cap = 0;
    if (ix->tie_log && !P.wave_kernel && !d_diag && !mstg && top_k >= 1 && top_k <= kTopKRegMax && !ix->exact_heap && !stage_probes()) {
        // entries per query: ~80 candidates are refined per query at top_k = 10 and ~400 at top_k = 100 on the bench data; a query that
        // logs more than the capacity is scanned again, as before (option tie_log_cap: tests of that path)
        const uint32_t cap = ix->tie_log_cap ? ix->tie_log_cap : (top_k < 64u ? 2048u : 8192u);
        if ((uint64_t)nq * cap * 12u <= (1ull << 30)) {
            if (w->tie_log.ensure((size_t)nq * cap * 12u) == RBQ_OK) { P.tie_log = (uint32_t*)w->tie_log.p; P.tie_log_cap = cap; }
            else { (void)hipGetLastError(); g_err.clear(); } // (no memory for the log: the call is served without it)
        }
    }
    P.heap_ws = nullptr;
    if (top_k > kTopKMax || scan_lds_bytes(ix->Dc, ix->D, P.ex_bits, top_k) > kLdsPerWorkgroupMax) { // the heap does not fit the LDS
        int rc = w->heap_ws.ensure((size_t)nq * 2 * ((size_t)top_k + 1) * 4);
        if (rc) return rc;
        P.heap_ws = (uint32_t*)w->heap_ws.p;
    }
    HIP_TRY(launch_scan(P, (uint32_t)nq, ix->device, stream, ps.start(), ps.stop()));
    return RBQ_OK;
}

// Core: everything on device pointers, enqueued on `stream`. Workspace buffers come from `w`.
int search_device(Replica* ix, Workspace* w, const float* d_queries, uint64_t nq, uint32_t top_k, uint32_t nprobe_in,
                  const uint32_t* d_filter, uint64_t filter_nbits, uint64_t* d_ids, float* d_scores, uint32_t* d_counts,
                  rbq_diag* d_diag, hipStream_t stream) {
    const uint32_t D = ix->D, Dc = ix->Dc;
    const uint32_t nlist = (uint32_t)ix->n_lists;
    uint32_t nprobe = nprobe_in < 1 ? 1 : nprobe_in;
    if (nprobe > nlist) nprobe = nlist;
    // nprobe > kNprobeMax: the exact all-pairs ranking with its key window in global memory; top_k > kTopKMax: the heap in
    // global memory — both slow, both served (the reference clamps nprobe to n_lists and accepts any top_k)
    const bool big_nprobe = nprobe > kNprobeMax;
    // the 8 GiB bound is the global-memory heap's (top_k beyond what the LDS holds); calls whose heap lives in LDS need no workspace
    const bool heap_in_global = top_k > kTopKMax || scan_lds_bytes(Dc, D, ix->ex_bits, top_k) > kLdsPerWorkgroupMax;
    if (top_k > kTopKHardMax || (heap_in_global && (uint64_t)nq * ((uint64_t)top_k + 1) * 8 > (8ull << 30)))
        return fail(RBQ_INVALID_CONFIG, "top_k too large for one call (top_k <= 2^20 and nq * top_k * 8 bytes of heap workspace <= 8 GiB)");
    if (ix->rerank && top_k > 1024) return fail(RBQ_INVALID_CONFIG, "rerank supports top_k <= 1024");
    const uint64_t wl_stride = std::max<uint64_t>(ix->nblk_desc_prefix[nprobe], 1);

    int rc;
    if ((rc = w->rot.ensure(nq * D * 4))) return rc;
    if ((rc = w->lut.ensure(nq * (size_t)Dc * 4))) return rc;
    if ((rc = w->consts.ensure(nq * sizeof(QueryConsts)))) return rc;
    if ((rc = w->scores.ensure(nq * (size_t)nlist * 4))) return rc;
    if ((rc = w->probe.ensure(nq * (size_t)nprobe * sizeof(ProbeInfo)))) return rc;
    if ((rc = w->wl.ensure(nq * wl_stride * sizeof(StreamItem)))) return rc;
    if ((rc = w->nstream.ensure(nq * 4))) return rc;
    if ((rc = w->nvec.ensure(nq * 8))) return rc;
    if ((rc = w->dead_skipped.ensure(nq * 16))) return rc;
    const bool split_rank = !ix->exact_rank && !ix->f32_rank && D % 64 == 0; // k_rank_bf16_db (else k_rank_mfma)
    if (split_rank) {
        if ((rc = w->rot_hi.ensure(nq * D * 2))) return rc;
        if ((rc = w->rot_lo.ensure(nq * D * 2))) return rc;
    }
    unsigned long long* prof = (ix->profiling && ix->profile_counters) ? (unsigned long long*)ix->prof.p : nullptr;
    const uint32_t smask = ix->stage_mask;
    // small calls: prep + exact scores of every list in one launch (the ranking GEMM and its launch boundary are skipped)
    // split-K of the ranking GEMM for batches whose tiles leave the chip idle (the rows are cleared by the workgroup-per-query preparation)
    uint32_t ksplit = 1;
    // (only when the whole CALL is small: the sub-batches of a 1024-query host call overlap each other, and the extra workgroups and
    // atomics of a split GEMM then cost more than its shorter K loop saves — 277 -> 295 us per call, measured)
    if (ix->rank_ksplit && ix->latency_path && nq <= kPrepWgMaxQueries && (w->call_nq ? w->call_nq : nq) <= kSplitKMaxCallQueries && ix->rotator != 0 &&
        !ix->wg_prep && D % 16 == 0 && split_rank && !ix->exact_rank && !big_nprobe && smask == 0xfu) {
        const uint32_t T = (!ix->small_rank_tiles && (uint64_t)((nlist + 127) / 128) * ((nq + 127) / 128) >= 192 && ix->rank_tile != 64) || ix->rank_tile == 128 ? 128u : 64u;
        const uint64_t tiles = (uint64_t)((nlist + T - 1) / T) * ((nq + T - 1) / T);
        ksplit = tiles >= 512 ? 1u : (tiles >= 256 ? 2u : 4u);
        while (ksplit > 1 && (D / 32) / ksplit < 6) ksplit >>= 1; // (at least six slabs per part)
        if (ix->rank_ksplit > 1) ksplit = (uint32_t)ix->rank_ksplit; // (option: forced)
    }
    const bool lat_front = ix->latency_path && nq <= kLatMaxQueries && ix->rotator != 0 && !ix->wg_prep && !ix->exact_rank && !big_nprobe &&
                           !ix->f32_rank && (uint64_t)nlist * D * 4 * nq <= kLatMaxBytes && D % 16 == 0 && smask == 0xfu;
    if (lat_front) {
        ProfScope ps(ix, 0, stream);
        PrepParams p;
        p.queries = d_queries; p.nq = (uint32_t)nq; p.dim = ix->dim; p.D = D; p.Dc = Dc; p.rotator = (int)ix->rotator;
        p.rot_blob = (const uint8_t*)ix->rot_blob.p; p.trunc = ix->trunc; p.fac = ix->fac; p.ex_bits = ix->ex_bits;
        p.rot = (float*)w->rot.p; p.lut = (uint8_t*)w->lut.p; p.consts = (QueryConsts*)w->consts.p;
        p.rot_hi = nullptr; p.rot_lo = nullptr; p.wg_prep = false;
        RankParams r;
        r.metric = ix->metric; r.cent = (const float*)ix->centroids.p; r.nlist = nlist; r.D = D; r.nq = (uint32_t)nq;
        r.scores = (float*)w->scores.p;
        HIP_TRY(launch_lat_front(p, r, ix->device, stream));
    } else if (smask & 1u) {
        ProfScope ps(ix, 0, stream);
        PrepParams p;
        p.queries = d_queries; p.nq = (uint32_t)nq; p.dim = ix->dim; p.D = D; p.Dc = Dc; p.rotator = (int)ix->rotator;
        p.rot_blob = (const uint8_t*)ix->rot_blob.p; p.trunc = ix->trunc; p.fac = ix->fac; p.ex_bits = ix->ex_bits;
        p.rot = (float*)w->rot.p; p.lut = (uint8_t*)w->lut.p; p.consts = (QueryConsts*)w->consts.p;
        p.rot_hi = split_rank ? (uint16_t*)w->rot_hi.p : nullptr; p.rot_lo = split_rank ? (uint16_t*)w->rot_lo.p : nullptr;
        p.wg_prep = ix->wg_prep;
        // batches a caller waits for (a few hundred queries: the chip is not full): a workgroup per query — the serial sums on one
        // wave while the others build the LUT (latency.hpp, preparation only); full batches keep one wave per query (k_prep_wave)
        if (ix->latency_path && nq <= kPrepWgMaxQueries && ix->rotator != 0 && !ix->wg_prep && D % 16 == 0) {
            RankParams r;
            r.metric = ix->metric; r.cent = nullptr; r.nlist = nlist; r.D = D; r.nq = (uint32_t)nq;
            r.scores = ksplit > 1 ? (float*)w->scores.p : nullptr; r.ksplit = ksplit; // (split-K GEMM: this kernel clears the score rows)
            HIP_TRY(launch_lat_front(p, r, ix->device, stream));
        } else {
            HIP_TRY(launch_prep(p, ix->device, stream));
        }
    }
    RankParams rp;
    rp.metric = ix->metric; rp.rot = (const float*)w->rot.p; rp.rot_hi = (const uint16_t*)w->rot_hi.p; rp.rot_lo = (const uint16_t*)w->rot_lo.p;
    rp.cent = (const float*)ix->centroids.p; rp.cent_hi = (const uint16_t*)ix->cent_hi.p; rp.cent_lo = (const uint16_t*)ix->cent_lo.p;
    rp.consts = (const QueryConsts*)w->consts.p; rp.cnorm2 = (const float*)ix->cnorm2.p; rp.nq = (uint32_t)nq; rp.nlist = nlist; rp.D = D;
    rp.scores = (float*)w->scores.p; rp.split = split_rank; rp.ksplit = ksplit;
    rp.big = !ix->small_rank_tiles && (uint64_t)((nlist + 127) / 128) * ((nq + 127) / 128) >= 192; // enough 128x128 tiles to fill the chip
    // option rank_tile: 0 = by problem size, 64 / 128 / 256 = forced (tests; A/B)
    rp.wide = ix->rank_tile == 256 || (ix->rank_tile == 0 && !ix->small_rank_tiles && (uint64_t)((nlist + 255) / 256) * ((nq + 127) / 128) >= 2048);
    if (ix->rank_tile == 64) rp.big = false;
    if (ix->rank_tile == 128) rp.big = true;
    SelectParams sp;
    sp.scores = (float*)w->scores.p; sp.nq = (uint32_t)nq; sp.nlist = nlist; sp.nprobe = nprobe; sp.metric = ix->metric;
    sp.rot = (const float*)w->rot.p; sp.cent = (const float*)ix->centroids.p; sp.D = D; sp.consts = (const QueryConsts*)w->consts.p;
    sp.cnorm2_max = ix->cnorm2_max; sp.list_gb0 = (const uint32_t*)ix->list_gb0.p; sp.list_n = (const uint32_t*)ix->list_n.p;
    sp.probe = (ProbeInfo*)w->probe.p; sp.wl = (StreamItem*)w->wl.p; sp.wl_stride = wl_stride; sp.nstream = (uint32_t*)w->nstream.p;
    sp.nvec = (unsigned long long*)w->nvec.p; sp.prof_total = prof; sp.fallback_count = (unsigned int*)ix->fallbacks.p;
    sp.force_fallback = ix->force_rank_fallback ? 1 : 0; sp.bsum = (const BlockSummary*)ix->bsum.p;
    sp.cnorm2 = (const float*)ix->cnorm2.p; sp.lsum = (const BlockSummary*)ix->lsum.p; sp.bsumx = (const BlockSummaryEx*)ix->bsumx.p;
    sp.dead_skipped = (uint32_t*)w->dead_skipped.p; sp.top_k = top_k; sp.ex_bits = ix->ex_bits;
    sp.slack = ix->slack;
    sp.audit_dead = nullptr;
    if (ix->lazy_audit) { // diagnostic: the lists the selection drops as a whole are exported (rbq_debug_copy_workspace "audit_dead")
        if ((rc = w->audit_dead.ensure(nq * (size_t)(kAuditCap + 1) * 4))) return rc;
        HIP_TRY(hipMemsetAsync(w->audit_dead.p, 0, nq * (size_t)(kAuditCap + 1) * 4, stream));
        sp.audit_dead = (uint32_t*)w->audit_dead.p;
    }
    // lazy selection: not with a filter (filtered vectors are never pushed, so no select-time bound of the k-th distance
    // exists) and not when every probed block is to be streamed
    // (round 4: under a filter the exact head evaluation can still bound the k-th distance — it looks at real vectors and counts
    // the ones that pass; with diagnostics the skip counter of a dropped list would need every id's filter bit: eager then)
    const bool lazy_with_filter = ix->lazy_filter && ix->head_exact && !d_diag;
    sp.lazy = (ix->lazy_select && (!d_filter || lazy_with_filter) && !ix->no_block_bound) ? 1 : 0;
    sp.filter = d_filter; sp.filter_nbits = filter_nbits; sp.ids = (const uint64_t*)ix->ids.p;
    sp.exact_members = d_diag ? 1 : 0;
    sp.fault_dead_all = ix->lazy_fault_inject ? 1 : 0;
    sp.lut = (const uint8_t*)w->lut.p; sp.blocks = (const uint8_t*)ix->blocks.p; sp.ex_codes = (const uint8_t*)ix->ex.p;
    sp.f_add_ex = (const float*)ix->fadd_ex.p; sp.f_rescale_ex = (const float*)ix->fres_ex.p; sp.Dc = Dc;
    sp.head_exact = ix->head_exact ? 1 : 0;
    sp.n_blocks = (uint32_t)ix->n_blocks;
    if (ix->exact_rank || big_nprobe) {
        uint64_t* kw = nullptr;
        if (big_nprobe) {
            if ((rc = w->key_window.ensure((size_t)nq * select_exact_np2(nprobe) * 8))) return rc;
            kw = (uint64_t*)w->key_window.p;
        }
        if (smask & 2u) { ProfScope ps(ix, 1, stream); HIP_TRY(launch_rank_exact(rp, stream)); }
        if (smask & 4u) { ProfScope ps(ix, 2, stream); HIP_TRY(launch_select_exact(sp, ix->device, stream, kw)); }
    } else {
        if ((smask & 2u) && !lat_front) { ProfScope ps(ix, 1, stream); HIP_TRY(launch_rank_gemm(rp, ix->device, stream)); } // approximate scores: one MFMA GEMM
        if (smask & 4u) { ProfScope ps(ix, 2, stream); HIP_TRY(launch_select_mfma(sp, ix->device, stream)); }         // shortlist + exact canonical scores + exact select
    }
    if ((smask & 8u) && (rc = scan_stage(ix, w, nq, nprobe, top_k, wl_stride, d_filter, filter_nbits, d_ids, d_scores, d_counts, d_diag,
                         /*mstg=*/false, (ix->exact_rank || big_nprobe) ? nullptr : (const uint32_t*)w->dead_skipped.p, stream)))
        return rc;
    if (ix->rerank && !stage_probes()) // optional, default off: exact re-scoring of the returned ids against the attached raw vectors
        HIP_TRY(launch_rerank(d_queries, (uint32_t)nq, ix->dim, (const float*)ix->raw.p, ix->n_raw, ix->metric, top_k, d_ids, d_scores,
                              d_counts, stream));
    return RBQ_OK;
}

int check_query_args(const rbq_index* h, uint32_t query_dim) {
    if (!h || h->reps.empty()) return fail(RBQ_INVALID_CONFIG, "null index");
    const Replica* ix = h->reps[0];
    if (ix->n_vectors == 0) return fail(RBQ_EMPTY_INDEX, "index is empty");
    if (query_dim != ix->dim) {
        char b[96];
        std::snprintf(b, sizeof b, "expected %u, got %u", ix->dim, query_dim);
        return fail(RBQ_DIMENSION_MISMATCH, b);
    }
    return RBQ_OK;
}

// rbq_search_batch on ONE replica.  The batch is cut into sub-batches of (by default) 1024 queries that travel through
// up to six lanes (stream + workspace + pinned staging each), so the stages of neighbouring sub-batches overlap on the
// GPU exactly like bench.py's device-resident batches do, and the host stages sub-batch j+1 while j runs.
//  * results: k_scan writes ids / scores / counts / diagnostics STRAIGHT into page-locked host memory (the caller's
//    buffers when they are page-locked — rbq_host_alloc / hipHostMalloc / hipHostRegister — else the lane's pinned
//    staging, handed over with one memcpy): 123 KB per 1024 queries, no D2H copy command in the queue;
//  * queries: page-locked caller buffers are DMA-ed as they are, pageable ones are staged through the lane's pinned
//    buffer in pieces (the DMA of piece i runs under the memcpy of piece i+1);
//  * one event per sub-batch, no stream synchronisation.
int search_host(Replica* ix, const float* queries, uint64_t nq, uint32_t query_dim, uint32_t top_k, uint32_t nprobe,
                const uint32_t* filter_words, uint64_t filter_nbits, uint64_t* out_ids, float* out_scores, uint32_t* out_counts,
                rbq_diag* diag) {
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(RBQ_DEVICE, "hipSetDevice failed");
    using clk = std::chrono::steady_clock;
    const bool trace = ix->host_trace != 0;
    double t_attr = 0, t_ws = 0, t_stage = 0, t_enq = 0, t_wait = 0, t_out = 0;
    auto tick = [&](clk::time_point& t0, double& acc) { if (trace) { const auto t1 = clk::now(); acc += std::chrono::duration<double, std::micro>(t1 - t0).count(); t0 = t1; } };
    clk::time_point tp = clk::now();
    // sub-batches shrink towards the end of the call (rbq_host_logic.hpp): the last one's kernel chain is what the caller waits for
    const std::vector<std::pair<uint64_t, uint64_t>> plan = rbq_host::subbatch_plan(nq, ix->host_subbatch, ix->host_taper);
    const uint64_t nsub = plan.size();
    const uint32_t nlanes = (uint32_t)std::min<uint64_t>(nsub, ix->host_lanes ? ix->host_lanes : 6u);
    const bool in_pinned = is_pinned_host_range(queries, nq * query_dim * 4);
    bool out_pinned = !ix->rerank && is_pinned_host_range(out_ids, nq * top_k * 8) && is_pinned_host_range(out_scores, nq * top_k * 4) &&
                      is_pinned_host_range(out_counts, nq * 4) && (!diag || is_pinned_host_range(diag, nq * sizeof(rbq_diag)));
    // device-side addresses of page-locked caller buffers (identical under unified addressing; asked for, not assumed)
    uint64_t* c_ids = nullptr; float* c_scores = nullptr; uint32_t* c_counts = nullptr; rbq_diag* c_diag = nullptr;
    if (out_pinned) {
        if (hipHostGetDevicePointer((void**)&c_ids, out_ids, 0) != hipSuccess || hipHostGetDevicePointer((void**)&c_scores, out_scores, 0) != hipSuccess ||
            hipHostGetDevicePointer((void**)&c_counts, out_counts, 0) != hipSuccess ||
            (diag && hipHostGetDevicePointer((void**)&c_diag, diag, 0) != hipSuccess)) {
            (void)hipGetLastError();
            out_pinned = false;
        }
    }
    // page-locked queries are read by k_prep where they lie (device-side address of the caller's buffer)
    // (measured, GIST-1M shape: 1024 queries per call 297 -> 280 us, 1536: 388 -> 376, 256: 212 -> 198 us; but one query 133 -> 147 us,
    // 2048 per call 445 -> 457 and from 4096 the copy engine clearly beats the waves' own PCIe reads: 678 -> 778 us — hence the window)
    // (pageable queries, read from the lane's pinned staging buffer: 1024 per call 381 -> 373 us, but 2048 per call 524 -> 591 us:
    // there the staged DMA pieces overlap the host copy better than one big copy followed by in-place reads — window <= 1024)
    // (round 5: the preparation kernels now request a query's elements eight at a time instead of one per round trip — 15 PCIe round
    // trips per query at D = 960 became 2 — and the window opens at 5 queries: 8 per call 149 -> 142 us, 16: 156 -> 149, 24: 161 -> 153)
    const bool zero_copy = ix->host_zero_copy && nq >= ix->host_zero_copy_min && nq <= (in_pinned ? 1536u : 1024u);
    const float* c_queries = nullptr;
    if (zero_copy && in_pinned && hipHostGetDevicePointer((void**)&c_queries, const_cast<float*>(queries), 0) != hipSuccess) {
        (void)hipGetLastError();
        c_queries = nullptr;
    }
    tick(tp, t_attr);
    std::vector<Workspace*> lanes;
    struct Give { Replica* ix; std::vector<Workspace*>& l; bool drained = false;
                  ~Give() { for (Workspace* w : l) { if (!drained) (void)hipStreamSynchronize(w->stream); give_ws(ix, w); } } } give{ix, lanes};
    for (uint32_t i = 0; i < nlanes; ++i) {
        Workspace* w = take_ws(ix);
        if (!w) return fail(RBQ_DEVICE, "cannot create workspace stream");
        lanes.push_back(w);
    }
    tick(tp, t_ws);
    int rc;
    const uint32_t* d_filter = nullptr;
    if (filter_words) {
        const size_t fb = (size_t)((filter_nbits + 31) / 32) * 4;
        Workspace* w0 = lanes[0];
        if ((rc = w0->filter.ensure(fb ? fb : 4))) return rc;
        if (fb) HIP_TRY(hipMemcpyAsync(w0->filter.p, filter_words, fb, hipMemcpyHostToDevice, w0->stream));
        HIP_TRY(hipEventRecord(w0->done, w0->stream));
        for (uint32_t i = 1; i < nlanes; ++i) HIP_TRY(hipStreamWaitEvent(lanes[i]->stream, w0->done, 0));
        d_filter = (const uint32_t*)w0->filter.p;
    }
    // hand the finished sub-batch j (in the lane's pinned buffer) to the caller's arrays
    auto deliver = [&](Workspace* w, uint64_t j) -> int {
        clk::time_point td = clk::now();
        HIP_TRY(hipEventSynchronize(w->done));
        tick(td, t_wait);
        if (out_pinned) return RBQ_OK;
        const uint64_t q0 = plan[j].first, n = plan[j].second;
        const OutPack op(n, top_k, diag != nullptr);
        const uint8_t* src = (const uint8_t*)w->h_out.p;
        std::memcpy(out_ids + q0 * top_k, src + op.o_ids, n * top_k * 8);
        std::memcpy(out_scores + q0 * top_k, src + op.o_scores, n * top_k * 4);
        std::memcpy(out_counts + q0, src + op.o_counts, n * 4);
        if (diag) std::memcpy(diag + q0, src + op.o_diag, n * sizeof(rbq_diag));
        tick(td, t_out);
        return RBQ_OK;
    };
    // pageable queries, every sub-batch on its own lane (nsub <= nlanes): the helpers stage sub-batches 1.. into their lanes' pinned
    // buffers while this thread stages and launches sub-batch 0
    constexpr uint64_t kMaxHelped = 8;
    std::atomic<int> staged[kMaxHelped];
    uint64_t n_helped = 0; // sub-batches 1 .. n_helped are staged by helpers
    struct WaitHelpers { std::atomic<int>* f; uint64_t& n; // (no job may outlive this frame: it writes the flags)
        ~WaitHelpers() { for (uint64_t j = 1; j <= n; ++j) while (!f[j].load(std::memory_order_acquire)) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        } } } wait_helpers{staged, n_helped};
    bool helpers = !in_pinned && ix->host_stage_helpers && nsub >= 2 && nsub <= nlanes && nsub <= kMaxHelped && nq * query_dim * 4 >= (512u << 10);
    if (helpers) {
        // the helpers are published only once every thread of theirs runs (a half-started set used to stay installed, and
        // jobs posted to it could wait forever); if a thread cannot be created the call stages inline, like a small one does
        std::lock_guard<std::mutex> lk(ix->mu);
        if (!ix->stagers && !ix->stagers_failed) {
            std::unique_ptr<StageHelpers> sh(new (std::nothrow) StageHelpers());
            if (sh && sh->start()) ix->stagers = std::move(sh);
            else ix->stagers_failed = true;
        }
        helpers = ix->stagers != nullptr;
    }
    if (helpers) {
        for (uint64_t j = 1; j < nsub; ++j) {
            Workspace* wj = lanes[j];
            if ((rc = wj->h_in.ensure(plan[j].second * query_dim * 4))) return rc; // (jobs posted so far are awaited by wait_helpers)
            staged[j].store(0, std::memory_order_relaxed);
            ix->stagers->post({wj->h_in.p, queries + plan[j].first * query_dim, (size_t)(plan[j].second * query_dim * 4), &staged[j]});
            n_helped = j;
        }
    }
    for (uint64_t j = 0; j < nsub; ++j) {
        Workspace* w = lanes[j % nlanes];
        if (j >= nlanes && (rc = deliver(w, j - nlanes))) return rc;
        const uint64_t q0 = plan[j].first, n = plan[j].second;
        const OutPack op(n, top_k, diag != nullptr);
        const bool staged_in_hbm = !(c_queries || (zero_copy && !in_pinned)); // an H2D copy into the lane's device buffer
        if (staged_in_hbm && (rc = w->queries.ensure(n * query_dim * 4))) return rc;
        const float* src = queries + q0 * query_dim;
        tp = clk::now();
        const float* d_q = (const float*)w->queries.p; // where k_prep reads this sub-batch
        if (c_queries) {
            d_q = c_queries + q0 * query_dim;
        } else if (zero_copy && !in_pinned) { // pageable: one host copy into the lane's pinned buffer, read from there
            if (j >= 1 && j <= n_helped) { // staged by a helper: wait for its copy
                while (!staged[j].load(std::memory_order_acquire)) {
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                }
            } else {
                if ((rc = w->h_in.ensure(n * query_dim * 4))) return rc;
                std::memcpy(w->h_in.p, src, n * query_dim * 4);
            }
            float* hp = nullptr;
            HIP_TRY(hipHostGetDevicePointer((void**)&hp, w->h_in.p, 0));
            d_q = hp;
        } else if (in_pinned) {
            HIP_TRY(hipMemcpyAsync(w->queries.p, src, n * query_dim * 4, hipMemcpyHostToDevice, w->stream));
        } else if (j >= 1 && j <= n_helped) { // pageable, staged by a helper: wait for its copy, one DMA
            while (!staged[j].load(std::memory_order_acquire)) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            HIP_TRY(hipMemcpyAsync(w->queries.p, w->h_in.p, n * query_dim * 4, hipMemcpyHostToDevice, w->stream));
        } else {
            if ((rc = w->h_in.ensure(n * query_dim * 4))) return rc;
            const uint64_t piece = std::max<uint64_t>(64, (n + 3) / 4); // queries per piece
            for (uint64_t p0 = 0; p0 < n; p0 += piece) {
                const size_t off = p0 * query_dim * 4, len = std::min(piece, n - p0) * query_dim * 4;
                std::memcpy((uint8_t*)w->h_in.p + off, (const uint8_t*)src + off, len);
                HIP_TRY(hipMemcpyAsync((uint8_t*)w->queries.p + off, (const uint8_t*)w->h_in.p + off, len, hipMemcpyHostToDevice, w->stream));
            }
        }
        tick(tp, t_stage);
        uint64_t* k_ids; float* k_scores; uint32_t* k_counts; rbq_diag* k_diag; // where the kernels write
        if (out_pinned) {
            k_ids = c_ids + q0 * top_k; k_scores = c_scores + q0 * top_k; k_counts = c_counts + q0; k_diag = diag ? c_diag + q0 : nullptr;
        } else if (!ix->rerank) {
            if ((rc = w->h_out.ensure(op.total))) return rc;
            uint8_t* hp = nullptr;
            HIP_TRY(hipHostGetDevicePointer((void**)&hp, w->h_out.p, 0));
            k_ids = (uint64_t*)(hp + op.o_ids); k_scores = (float*)(hp + op.o_scores); k_counts = (uint32_t*)(hp + op.o_counts);
            k_diag = diag ? (rbq_diag*)(hp + op.o_diag) : nullptr;
        } else { // the rerank kernel re-reads the ids: keep them in HBM, one D2H copy afterwards
            if ((rc = w->out_pack.ensure(op.total))) return rc;
            uint8_t* dp = (uint8_t*)w->out_pack.p;
            k_ids = (uint64_t*)(dp + op.o_ids); k_scores = (float*)(dp + op.o_scores); k_counts = (uint32_t*)(dp + op.o_counts);
            k_diag = diag ? (rbq_diag*)(dp + op.o_diag) : nullptr;
        }
        // which sub-batches of a call below kHostWaveMinQueries take the short-chain kernel: all (policy 0), only the LAST one — the chain
        // the caller actually waits for; the earlier ones overlap it — (1), none (2)
        w->latency_first = nq < kHostWaveMinQueries && (ix->host_wave_policy == 0 || (ix->host_wave_policy == 1 && j + 1 == nsub));
        w->call_nq = nq;
        rc = search_device(ix, w, d_q, n, top_k, nprobe, d_filter, filter_nbits, k_ids, k_scores, k_counts, k_diag, w->stream);
        w->latency_first = false;
        w->call_nq = 0;
        if (rc) return rc;
        if (ix->rerank) {
            if ((rc = w->h_out.ensure(op.total))) return rc;
            HIP_TRY(hipMemcpyAsync(w->h_out.p, w->out_pack.p, op.total, hipMemcpyDeviceToHost, w->stream));
        }
        HIP_TRY(hipEventRecord(w->done, w->stream));
        tick(tp, t_enq);
    }
    for (uint64_t j = nsub > nlanes ? nsub - nlanes : 0; j < nsub; ++j)
        if ((rc = deliver(lanes[j % nlanes], j))) return rc;
    give.drained = true; // every lane's last event has been waited for
    if (trace)
        std::fprintf(stderr, "[rbq host] nq=%llu sub=%llu lanes=%u pinned(in,out)=%d,%d us: attr %.1f ws %.1f stage %.1f enqueue %.1f wait %.1f copy-out %.1f\n",
                     (unsigned long long)nq, (unsigned long long)plan[0].second, nlanes, (int)in_pinned, (int)out_pinned, t_attr, t_ws, t_stage, t_enq, t_wait, t_out);
    return RBQ_OK;
}

Replica* replica_of_pointer(rbq_index* h, const void* dptr) {
    if (h->reps.size() == 1) return h->reps[0];
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, dptr) == hipSuccess) {
        for (Replica* r : h->reps) if (r->device == a.device) return r;
    } else {
        (void)hipGetLastError();
    }
    return h->reps[0];
}

} // namespace

extern "C" {

uint32_t rbq_abi_version(void) { return (2u << 16) | 1u; } // (minor 1: rbq_debug_tie_log_stats; options latency_path, tie_log)

const char* rbq_strerror(int code) {
    switch (code) {
        case RBQ_OK: return "ok";
        case RBQ_DIMENSION_MISMATCH: return "dimension mismatch";
        case RBQ_INVALID_CONFIG: return "invalid configuration";
        case RBQ_EMPTY_INDEX: return "index is empty";
        case RBQ_IO: return "io error";
        case RBQ_INVALID_PERSISTENCE: return "invalid persisted index";
        case RBQ_DEVICE: return "device error";
        default: return "unknown error";
    }
}

int rbq_last_error_detail(char* buf, size_t n) {
    if (buf && n) {
        size_t c = std::min(n - 1, g_err.size());
        std::memcpy(buf, g_err.data(), c);
        buf[c] = 0;
    }
    return (int)g_err.size();
}

int rbq_index_create(const rbq_header* hdr, const rbq_list_view* lists, int n_devices, const int* devices, rbq_index** out) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    return create_impl(hdr, lists, n_devices, devices, out);
    RBQ_GUARD_END
}

void rbq_index_destroy(rbq_index* h) { try { free_index(h); } catch (...) {} }

uint64_t rbq_index_len(const rbq_index* h) { return h && !h->reps.empty() ? h->reps[0]->n_vectors : 0; }
uint64_t rbq_index_cluster_count(const rbq_index* h) { return h && !h->reps.empty() ? h->reps[0]->n_lists : 0; }
uint32_t rbq_index_dim(const rbq_index* h) { return h && !h->reps.empty() ? h->reps[0]->dim : 0; }
uint32_t rbq_index_padded_dim(const rbq_index* h) { return h && !h->reps.empty() ? h->reps[0]->D : 0; }
uint32_t rbq_index_device_count(const rbq_index* h) { return h ? (uint32_t)h->reps.size() : 0; }

// ---- RBQ1 v3 reader: load_from_reader, src/ivf.rs:1484-1702 --------------------------------------
} // extern "C"
namespace {
int load_rbq1_impl(const void* bytes, size_t len, int n_devices, const int* devices, rbq_index** out) {
    if (!out) return fail(RBQ_INVALID_CONFIG, "null out pointer");
    *out = nullptr;
    rbq_header h;
    std::vector<ListSrc> lists; // byte ranges of the stream: nothing is copied on the host but the chunk staging
    std::string detail;
    int rc = rbq_host::rbq1_parse(bytes, len, &h, &lists, &detail); // load_from_reader's validation, CRC included
    if (rc) return fail(rc, detail);
    rc = validate_header(&h); // what this build cannot serve (ex_bits outside {0,2,6}, padded_dim > 2048 ...)
    if (rc) return rc;
    std::vector<int> devs;
    if ((rc = resolve_devices(n_devices, devices, devs))) return rc;
    Replica* first = nullptr;
    if ((rc = create_from_sources(&h, lists, devs[0], &first))) return rc;
    return wrap_and_replicate(first, devs, out);
}
} // namespace
extern "C" {

int rbq_index_load_rbq1(const void* bytes, size_t len, int n_devices, const int* devices, rbq_index** out) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    return load_rbq1_impl(bytes, len, n_devices, devices, out);
    RBQ_GUARD_END
}

int rbq_index_build_device(const rbq_header* hdr, const float* centroids, const float* d_data, const uint32_t* d_assign,
                           uint64_t n, float t_const, int device, rbq_index** out) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    return build_device_impl(hdr, centroids, d_data, d_assign, n, t_const, device, out);
    RBQ_GUARD_END
}

int rbq_build_stream_begin(const rbq_header* hdr, const float* centroids, const uint32_t* list_sizes, float t_const, int device,
                           rbq_builder** out) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    return stream_begin_impl(hdr, centroids, list_sizes, t_const, device, out);
    RBQ_GUARD_END
}
int rbq_build_stream_push(rbq_builder* b, const float* vectors, const uint32_t* assign, uint64_t first_id, uint64_t count) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    return stream_push_impl(b, vectors, assign, first_id, count);
    RBQ_GUARD_END
}
int rbq_build_stream_finish(rbq_builder* b, int n_devices, const int* devices, rbq_index** out) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    int rc = stream_finish_impl(b, n_devices, devices, out);
    if (rc == RBQ_OK) delete b;
    return rc;
    RBQ_GUARD_END
}
void rbq_build_stream_abort(rbq_builder* b) { try { delete b; } catch (...) {} }

// ---- search ---------------------------------------------------------------------------------------
int rbq_search_batch_device(const rbq_index* ch, const float* d_queries, uint64_t nq, uint32_t query_dim, uint32_t top_k,
                            uint32_t nprobe, const uint32_t* d_filter_words, uint64_t filter_nbits, uint64_t* d_out_ids,
                            float* d_out_scores, uint32_t* d_out_counts, rbq_diag* d_diag, void* hip_stream) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    rbq_index* h = const_cast<rbq_index*>(ch);
    int rc = check_query_args(h, query_dim);
    if (rc) return rc;
    if (nq == 0) return RBQ_OK;
    if (nq > 0x7fffffffull) return fail(RBQ_INVALID_CONFIG, "batch too large");
    Replica* ix = replica_of_pointer(h, d_queries);
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(RBQ_DEVICE, "hipSetDevice failed");
    hipStream_t s = (hipStream_t)hip_stream;
    if (top_k == 0) { // Ok(vec![]) for every query, src/ivf.rs:1792-1794
        HIP_TRY(hipMemsetAsync(d_out_counts, 0, nq * 4, s));
        return RBQ_OK;
    }
    // One workspace per caller stream, never handed to anyone else: successive calls on the same stream are
    // stream-ordered, so their kernels may share the scratch buffers without any host synchronisation.
    Workspace* w;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        Workspace*& slot = ix->stream_ws[s];
        if (!slot) slot = new Workspace();
        w = slot;
    }
    return search_device(ix, w, d_queries, nq, top_k, nprobe, d_filter_words, filter_nbits, d_out_ids, d_out_scores,
                         d_out_counts, d_diag, s);
    RBQ_GUARD_END
}

int rbq_release_stream(rbq_index* h, void* hip_stream) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    if (!h) return fail(RBQ_INVALID_CONFIG, "null index");
    for (Replica* ix : h->reps) {
        Workspace* w = nullptr;
        {
            std::lock_guard<std::mutex> lk(ix->mu);
            auto it = ix->stream_ws.find((hipStream_t)hip_stream);
            if (it != ix->stream_ws.end()) { w = it->second; ix->stream_ws.erase(it); }
        }
        if (w) { DeviceGuard g(ix->device); w->release(); delete w; }
    }
    return RBQ_OK;
    RBQ_GUARD_END
}

int rbq_search_batch(const rbq_index* ch, const float* queries, uint64_t nq, uint32_t query_dim, uint32_t top_k,
                     uint32_t nprobe, const uint32_t* filter_words, uint64_t filter_nbits, uint64_t* out_ids,
                     float* out_scores, uint32_t* out_counts, rbq_diag* diag) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    rbq_index* h = const_cast<rbq_index*>(ch);
    int rc = check_query_args(h, query_dim);
    if (rc) return rc;
    if (nq == 0) return RBQ_OK;
    if (top_k == 0) {
        if (out_counts) std::memset(out_counts, 0, nq * 4);
        if (diag) std::memset(diag, 0, nq * sizeof(rbq_diag));
        return RBQ_OK;
    }
    if (!queries || !out_ids || !out_scores || !out_counts) return fail(RBQ_INVALID_CONFIG, "null buffer");
    const size_t R = h->reps.size();
    if (R == 1 || nq < 2 * R) return search_host(h->reps[0], queries, nq, query_dim, top_k, nprobe, filter_words, filter_nbits, out_ids,
                                                  out_scores, out_counts, diag);
    // N replicas: contiguous shards [r*nq/R, (r+1)*nq/R) (batch_search is a par_iter over queries, src/ivf.rs:1743-1752);
    // replica 0 is served by the calling thread, replica r > 0 by its persistent worker thread; results land straight in
    // the caller's arrays, no exchange between devices
    {
        std::lock_guard<std::mutex> lk(h->worker_mu);
        while (h->workers.size() + 1 < R) {
            h->workers.emplace_back(new ReplicaWorker());
            h->workers.back()->start();
        }
    }
    // Everything the posted jobs touch lives in one shared state they hold by value: a job that is still running when this
    // frame unwinds (post() threw after earlier replicas were posted) writes into memory it co-owns, never into a dead stack.
    struct ShardState {
        std::mutex mu; std::condition_variable cv; size_t left = 0;
        std::vector<int> rcs; std::vector<std::string> details;
    };
    auto st = std::make_shared<ShardState>();
    st->rcs.assign(R, RBQ_OK);
    st->details.resize(R);
    st->left = R - 1;
    auto shard = [st, h, R, nq, queries, query_dim, top_k, nprobe, filter_words, filter_nbits, out_ids, out_scores, out_counts, diag](size_t r) {
        uint64_t q0, q1;
        rbq_host::shard_range(r, R, nq, &q0, &q1);
        try {
            st->rcs[r] = search_host(h->reps[r], queries + q0 * query_dim, q1 - q0, query_dim, top_k, nprobe, filter_words, filter_nbits,
                                     out_ids + q0 * top_k, out_scores + q0 * top_k, out_counts + q0, diag ? diag + q0 : nullptr);
        } catch (...) { st->rcs[r] = RBQ_IO; g_err = "internal error"; }
        st->details[r] = g_err; // thread-local in the worker
    };
    size_t posted = 0;
    bool post_failed = false;
    try {
        for (size_t r = 1; r < R; ++r) {
            h->workers[r - 1]->post([st, shard, r] {
                shard(r);
                std::lock_guard<std::mutex> lk(st->mu);
                if (--st->left == 0) st->cv.notify_one();
            });
            ++posted;
        }
    } catch (...) { post_failed = true; }
    if (post_failed) { // the caller's buffers must stay valid until the shards already under way have finished
        std::unique_lock<std::mutex> lk(st->mu);
        st->left -= (R - 1 - posted);
        st->cv.wait(lk, [&] { return st->left == 0; });
        return fail(RBQ_IO, "out of host memory");
    }
    shard(0);
    {
        // the other shards finish about when this one does: poll briefly before blocking on the latch
        const auto t0 = std::chrono::steady_clock::now();
        std::unique_lock<std::mutex> lk(st->mu);
        while (st->left != 0) {
            if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(200)) { lk.unlock(); std::this_thread::yield(); lk.lock(); }
            else st->cv.wait(lk, [&] { return st->left == 0; });
        }
    }
    for (size_t r = 0; r < R; ++r) if (st->rcs[r]) return fail(st->rcs[r], st->details[r]);
    return RBQ_OK;
    RBQ_GUARD_END
}

// ---- MSTG posting-list scan ------------------------------------------------------------
int rbq_posting_scan_batch(const rbq_index* ch, const float* queries, uint64_t nq, uint32_t query_dim, uint32_t top_k,
                           const uint32_t* list_ids, const uint32_t* list_counts, uint32_t max_lists,
                           uint64_t* out_ids, float* out_scores, uint32_t* out_counts) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    rbq_index* h = const_cast<rbq_index*>(ch);
    int rc = check_query_args(h, query_dim);
    if (rc) return rc;
    Replica* ix = h->reps[0];
    if (ix->rotator != RBQ_ROTATOR_NONE) return fail(RBQ_INVALID_CONFIG, "posting-list scan needs an index created with rotator NONE");
    if (nq == 0) return RBQ_OK;
    if (!queries || !list_ids || !list_counts || !out_ids || !out_scores || !out_counts) return fail(RBQ_INVALID_CONFIG, "null buffer");
    if (top_k == 0) { std::memset(out_counts, 0, nq * 4); return RBQ_OK; }
    if (top_k > kTopKHardMax || (uint64_t)std::min<uint64_t>(nq, 16384) * ((uint64_t)top_k + 1) * 8 > (8ull << 30))
        return fail(RBQ_INVALID_CONFIG, "top_k too large for one call (top_k <= 2^20)");
    if (max_lists == 0) {
        std::memset(out_counts, 0, nq * 4);
        for (uint64_t i = 0; i < nq * top_k; ++i) { out_ids[i] = ~0ull; out_scores[i] = NAN; }
        return RBQ_OK;
    }
    if (max_lists > (1u << 26)) return fail(RBQ_INVALID_CONFIG, "too many lists per query");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(RBQ_DEVICE, "hipSetDevice failed");
    // exact work-list bound from the host copy of the list sizes (a list may legally repeat)
    uint64_t wl_stride = 1;
    for (uint64_t q = 0; q < nq; ++q) {
        uint64_t tot = 0;
        const uint32_t n = std::min(list_counts[q], max_lists);
        for (uint32_t r = 0; r < n; ++r) {
            const uint32_t cid = list_ids[q * max_lists + r];
            if (cid < ix->n_lists) tot += (ix->h_list_n[cid] + 31u) / 32u;
        }
        wl_stride = std::max(wl_stride, tot);
    }
    if (wl_stride > 0xffffffffull) return fail(RBQ_INVALID_CONFIG, "posting lists too long for one query");
    Workspace* w = take_ws(ix);
    if (!w) return fail(RBQ_DEVICE, "cannot create workspace stream");
    auto run = [&]() -> int {
        int r2;
        const uint32_t D = ix->D, Dc = ix->Dc;
        const uint64_t CH = 16384;
        DevBuf& d_lists = w->scores; // reuse: [n][max_lists] u32
        DevBuf& d_cnts = w->nvec;    // reuse: [n] u32
        for (uint64_t q0 = 0; q0 < nq; q0 += CH) {
            const uint64_t n = std::min(CH, nq - q0);
            const OutPack op(n, top_k, false);
            if ((r2 = w->queries.ensure(n * query_dim * 4))) return r2;
            if ((r2 = w->rot.ensure(n * D * 4))) return r2;
            if ((r2 = w->lut.ensure(n * (size_t)Dc * 4))) return r2;
            if ((r2 = w->consts.ensure(n * sizeof(QueryConsts)))) return r2;
            if ((r2 = d_lists.ensure(n * (size_t)max_lists * 4))) return r2;
            if ((r2 = d_cnts.ensure(n * 8))) return r2;
            if ((r2 = w->probe.ensure(n * (size_t)max_lists * sizeof(ProbeInfo)))) return r2;
            if ((r2 = w->wl.ensure(n * wl_stride * sizeof(StreamItem)))) return r2;
            if ((r2 = w->nstream.ensure(n * 4))) return r2;
            if ((r2 = w->out_pack.ensure(op.total))) return r2;
            hipStream_t st = w->stream;
            uint8_t* dp = (uint8_t*)w->out_pack.p;
            HIP_TRY(hipMemcpyAsync(w->queries.p, queries + q0 * query_dim, n * query_dim * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d_lists.p, list_ids + q0 * max_lists, n * (size_t)max_lists * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d_cnts.p, list_counts + q0, n * 4, hipMemcpyHostToDevice, st));
            {
                ProfScope ps(ix, 0, st);
                PrepParams p;
                p.queries = (const float*)w->queries.p; p.nq = (uint32_t)n; p.dim = ix->dim; p.D = D; p.Dc = Dc; p.rotator = (int)ix->rotator;
                p.rot_blob = (const uint8_t*)ix->rot_blob.p; p.trunc = ix->trunc; p.fac = ix->fac; p.ex_bits = 0u;
                p.rot = (float*)w->rot.p; p.lut = (uint8_t*)w->lut.p; p.consts = (QueryConsts*)w->consts.p;
                p.rot_hi = nullptr; p.rot_lo = nullptr; p.wg_prep = false;
                HIP_TRY(launch_prep(p, ix->device, st));
            }
            {
                ProfScope ps(ix, 2, st);
                ProbesGivenParams p;
                p.list_ids = (const uint32_t*)d_lists.p; p.list_counts = (const uint32_t*)d_cnts.p; p.max_lists = max_lists;
                p.nq = (uint32_t)n; p.nlist = (uint32_t)ix->n_lists; p.metric = (int)ix->metric; p.rot = (const float*)w->rot.p;
                p.cent = (const float*)ix->centroids.p; p.D = D; p.list_gb0 = (const uint32_t*)ix->list_gb0.p;
                p.list_n = (const uint32_t*)ix->list_n.p; p.probe = (ProbeInfo*)w->probe.p; p.wl = (StreamItem*)w->wl.p;
                p.wl_stride = wl_stride; p.nstream = (uint32_t*)w->nstream.p; p.consts = (const QueryConsts*)w->consts.p;
                p.bsum = (const BlockSummary*)ix->bsum.p;
                HIP_TRY(launch_probes_given(p, st));
            }
            if ((r2 = scan_stage(ix, w, n, max_lists, top_k, wl_stride, nullptr, 0, (uint64_t*)(dp + op.o_ids), (float*)(dp + op.o_scores),
                                 (uint32_t*)(dp + op.o_counts), nullptr, /*mstg=*/true, nullptr, st)))
                return r2;
            HIP_TRY(hipMemcpyAsync(out_ids + q0 * top_k, dp + op.o_ids, n * top_k * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out_scores + q0 * top_k, dp + op.o_scores, n * top_k * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out_counts + q0, dp + op.o_counts, n * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        return RBQ_OK;
    };
    rc = run();
    if (rc) (void)hipStreamSynchronize(w->stream);
    give_ws(ix, w);
    return rc;
    RBQ_GUARD_END
}

// ---- optional full-precision rerank (NOT in the reference: its index stores no raw vectors, src/ivf.rs:207-242) ----
int rbq_index_set_rerank_vectors(rbq_index* h, const float* vectors, uint64_t n) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    if (!h || h->reps.empty()) return fail(RBQ_INVALID_CONFIG, "null index");
    // all replicas or none: the new copies are made first; only when every one of them exists are the old ones released
    // and the new ones attached (a failure half-way leaves every replica exactly as it was)
    const size_t R = h->reps.size();
    std::vector<Arr> fresh(R);
    std::vector<char> borrowed(R, 0);
    const bool attach = vectors && n;
    if (attach) {
        hipPointerAttribute_t a;
        const bool on_dev = hipPointerGetAttributes(&a, vectors) == hipSuccess && a.type == hipMemoryTypeDevice;
        if (!on_dev) (void)hipGetLastError();
        int rc = RBQ_OK;
        for (size_t r = 0; r < R && rc == RBQ_OK; ++r) {
            Replica* ix = h->reps[r];
            DeviceGuard g(ix->device);
            if (!g.ok) { rc = fail(RBQ_DEVICE, "hipSetDevice failed"); break; }
            const size_t bytes = (size_t)n * ix->dim * 4;
            if (on_dev && a.device == ix->device) { // borrowed: stays owned by the caller, must outlive the index
                fresh[r].p = const_cast<float*>(vectors); fresh[r].bytes = bytes; borrowed[r] = 1;
                continue;
            }
            rc = alloc_arr(fresh[r], bytes);
            if (rc == RBQ_OK) {
                const hipError_t e = on_dev ? copy_cross_device(fresh[r].p, ix->device, vectors, a.device, bytes)
                                            : hipMemcpy(fresh[r].p, vectors, bytes, hipMemcpyHostToDevice);
                if (e != hipSuccess) rc = fail(RBQ_DEVICE, std::string("attaching the rerank vectors: ") + hipGetErrorString(e));
            }
        }
        if (rc != RBQ_OK) {
            const std::string keep = g_err;
            for (size_t r = 0; r < R; ++r)
                if (fresh[r].p && !borrowed[r]) { DeviceGuard g(h->reps[r]->device); (void)hipFree(fresh[r].p); }
            return fail(rc, keep);
        }
    }
    for (size_t r = 0; r < R; ++r) {
        Replica* ix = h->reps[r];
        DeviceGuard g(ix->device);
        if (!g.ok) return fail(RBQ_DEVICE, "hipSetDevice failed");
        HIP_TRY(hipDeviceSynchronize());
        if (ix->raw.p && !ix->raw_borrowed) (void)hipFree(ix->raw.p);
        ix->raw = Arr(); ix->n_raw = 0; ix->rerank = false; ix->raw_borrowed = false;
        if (!attach) continue; // detach
        ix->raw = fresh[r]; ix->raw_borrowed = borrowed[r] != 0;
        ix->n_raw = n; ix->rerank = true;
    }
    return RBQ_OK;
    RBQ_GUARD_END
}

// ---- pinned host memory for callers that want zero-copy DMA of queries and results ---------------------
void* rbq_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void rbq_host_free(void* p) { if (p) (void)hipHostFree(p); }

// ---- profiling taps ---------------------------------------------------------------------------------
void rbq_profile_begin(rbq_index* h) {
    if (!h) return;
    for (Replica* ix : h->reps) {
        DeviceGuard g(ix->device);
        std::lock_guard<std::mutex> lk(ix->mu);
        for (auto& sp : ix->stage_prof) {
            for (auto& e : sp.ev) ix->ev_pool.give(e);
            sp.ev.clear(); sp.ms = 0; sp.launches = 0; sp.samples.clear();
        }
        for (auto& c : ix->prof_counters) c = 0;
        (void)hipMemset(ix->prof.p, 0, (size_t)kProfStripes * kProfSlots * 8);
        ix->profiling = true;
    }
}
void rbq_profile_end(rbq_index* h) {
    if (!h) return;
    for (Replica* ix : h->reps) {
        DeviceGuard g(ix->device);
        (void)hipDeviceSynchronize();
        std::lock_guard<std::mutex> lk(ix->mu);
        ix->profiling = false;
        std::vector<unsigned long long> c((size_t)kProfStripes * kProfSlots, 0ull);
        (void)hipMemcpy(c.data(), ix->prof.p, c.size() * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < kProfSlots; ++i) {
            ix->prof_counters[i] = 0;
            for (uint32_t st = 0; st < kProfStripes; ++st) ix->prof_counters[i] += c[(size_t)st * kProfSlots + i];
        }
        for (auto& sp : ix->stage_prof) {
            for (auto& e : sp.ev) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { sp.ms += ms; sp.launches++; sp.samples.push_back(ms); }
                ix->ev_pool.give(e);
            }
            sp.ev.clear();
        }
    }
}
double rbq_profile_stage_ms(const rbq_index* h, const char* stage, uint64_t* launches) {
    if (!h || h->reps.empty() || !stage) return -1;
    int s = !std::strcmp(stage, "prep") ? 0 : !std::strcmp(stage, "rank") ? 1 : !std::strcmp(stage, "select") ? 2 : !std::strcmp(stage, "scan") ? 3 : -1;
    if (s < 0) return -1;
    double ms = 0; uint64_t n = 0;
    for (const Replica* ix : h->reps) { ms += ix->stage_prof[s].ms; n += ix->stage_prof[s].launches; }
    if (launches) *launches = n;
    return n ? ms / (double)n : 0.0;
}
uint64_t rbq_profile_stage_samples(const rbq_index* h, const char* stage, float* out, uint64_t cap) {
    if (!h || h->reps.empty() || !stage) return 0;
    int s = !std::strcmp(stage, "prep") ? 0 : !std::strcmp(stage, "rank") ? 1 : !std::strcmp(stage, "select") ? 2 : !std::strcmp(stage, "scan") ? 3 : -1;
    if (s < 0) return 0;
    uint64_t n = 0;
    for (const Replica* ix : h->reps)
        for (float v : ix->stage_prof[s].samples) { if (out && n < cap) out[n] = v; ++n; }
    return n;
}
uint64_t rbq_profile_scan_bytes(const rbq_index* h) {
    if (!h || h->reps.empty()) return 0;
    uint64_t v = 0;
    for (const Replica* ix : h->reps) v += ix->prof_counters[kProfVectorsProbed];
    return v * (uint64_t)(h->reps[0]->D / 8 + 12); // sum_q sum_{c in probe(q)} n_c * (D/8 + 12)
}
int rbq_profile_counters(const rbq_index* h, uint64_t* out, uint32_t n) {
    if (!h || h->reps.empty() || !out) return RBQ_INVALID_CONFIG;
    for (uint32_t i = 0; i < n; ++i) {
        out[i] = 0;
        if (i < (uint32_t)kProfSlots) for (const Replica* ix : h->reps) out[i] += ix->prof_counters[i];
    }
    return RBQ_OK;
}
void rbq_profile_select_stages(rbq_index* h, uint32_t mask) { if (h) for (Replica* ix : h->reps) ix->prof_mask = mask & 0xfu; }
void rbq_profile_set_sampling(rbq_index* h, uint32_t every) { if (h) for (Replica* ix : h->reps) ix->prof_every = every ? every : 1u; }

int rbq_debug_set_option(rbq_index* h, const char* name, int value) {
    if (!h || h->reps.empty() || !name) return RBQ_INVALID_CONFIG;
    if (!std::strcmp(name, "debug_replica")) {
        if (value < 0 || (size_t)value >= h->reps.size()) return fail(RBQ_INVALID_CONFIG, "no such replica");
        h->debug_replica = value;
        return RBQ_OK;
    }
    for (Replica* ix : h->reps) {
        if (!std::strcmp(name, "block_bound")) ix->no_block_bound = value == 0;
        else if (!std::strcmp(name, "exact_rank")) ix->exact_rank = value != 0;
        else if (!std::strcmp(name, "exact_heap")) ix->exact_heap = value != 0;
        else if (!std::strcmp(name, "lazy_select")) ix->lazy_select = value != 0;
        else if (!std::strcmp(name, "rank_tile")) ix->rank_tile = value;
        else if (!std::strcmp(name, "host_wave_policy")) ix->host_wave_policy = value;
        else if (!std::strcmp(name, "latency_path")) ix->latency_path = value;
        else if (!std::strcmp(name, "tie_log")) ix->tie_log = value != 0;
        else if (!std::strcmp(name, "tie_log_cap")) ix->tie_log_cap = value > 0 ? (uint32_t)value : 0u;
        else if (!std::strcmp(name, "stage_mask")) ix->stage_mask = (uint32_t)value & 0xfu;
        else if (!std::strcmp(name, "scan_wave")) ix->scan_wave = value < 0 ? scan_wave_default() : (value > 2 ? 2 : value);
        else if (!std::strcmp(name, "profile_counters")) ix->profile_counters = value != 0;
        else if (!std::strcmp(name, "f32_rank")) ix->f32_rank = value != 0;
        else if (!std::strcmp(name, "wg_prep")) ix->wg_prep = value != 0;
        else if (!std::strcmp(name, "small_rank_tiles")) ix->small_rank_tiles = value != 0;
        else if (!std::strcmp(name, "force_rank_fallback")) ix->force_rank_fallback = value != 0;
        else if (!std::strcmp(name, "host_lanes")) ix->host_lanes = value > 0 ? (uint32_t)value : 0u;
        else if (!std::strcmp(name, "host_subbatch")) ix->host_subbatch = value > 0 ? (uint32_t)value : 0u;
        else if (!std::strcmp(name, "host_trace")) ix->host_trace = value != 0;
        else if (!std::strcmp(name, "lazy_fault_inject")) ix->lazy_fault_inject = value != 0;
        else if (!std::strcmp(name, "lazy_audit")) ix->lazy_audit = value != 0;
        else if (!std::strcmp(name, "slack_term")) ix->slack_term = value;
        else if (!std::strcmp(name, "slack_milli")) { // TEST ONLY: term `slack_term` of block_ub()'s slack times value / 1000 (1000 = the product)
            float* f[6] = {&ix->slack.ge, &ix->slack.eip, &ix->slack.est, &ix->slack.lb, &ix->slack.et, &ix->slack.dist};
            if (ix->slack_term < 0 || ix->slack_term > 5) return fail(RBQ_INVALID_CONFIG, "slack_term is 0..5");
            *f[ix->slack_term] = (float)value * 1e-3f;
        }
        else if (!std::strcmp(name, "head_exact")) ix->head_exact = value != 0;
        else if (!std::strcmp(name, "lazy_filter")) ix->lazy_filter = value != 0;
        else if (!std::strcmp(name, "host_zero_copy")) ix->host_zero_copy = value != 0;
        else if (!std::strcmp(name, "host_taper")) ix->host_taper = value;
        else if (!std::strcmp(name, "rank_ksplit")) ix->rank_ksplit = value < 0 ? 0 : value;
        else if (!std::strcmp(name, "host_zero_copy_min")) ix->host_zero_copy_min = value > 0 ? (uint32_t)value : 1u;
        else if (!std::strcmp(name, "host_stage_helpers")) ix->host_stage_helpers = value != 0;
        else if (!std::strcmp(name, "rerank")) {
            if (value && !ix->raw.p) return fail(RBQ_INVALID_CONFIG, "no raw vectors attached (rbq_index_set_rerank_vectors)");
            ix->rerank = value != 0;
        } else return fail(RBQ_INVALID_CONFIG, std::string("unknown option ") + name);
    }
    return RBQ_OK;
}
static uint64_t read_counter(const rbq_index* h, int slot) {
    if (!h) return 0;
    uint64_t tot = 0;
    for (const Replica* ix : h->reps) {
        unsigned int v = 0;
        DeviceGuard g(ix->device);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&v, (const unsigned int*)ix->fallbacks.p + slot, 4, hipMemcpyDeviceToHost);
        tot += v;
    }
    return tot;
}
uint64_t rbq_debug_rank_fallbacks(const rbq_index* h) { return read_counter(h, 0); }
void rbq_debug_tie_log_stats(const rbq_index* h, uint64_t* out4) { if (out4) for (int i = 0; i < 4; ++i) out4[i] = read_counter(h, 4 + i); }
uint64_t rbq_debug_head_exact_guard_trips(const rbq_index* h) { return read_counter(h, 2); }
uint64_t rbq_debug_head_exact_evaluations(const rbq_index* h) { return read_counter(h, 3); }
// Which kernel instantiation each of the four stages launches for a call of this shape, and what it occupies.
// out[stage][6] = workgroups, threads per workgroup, VGPRs per lane, LDS bytes per workgroup (static + dynamic), scratch bytes per
// lane, 0; stages in the order prep, rank, select, scan.  Nothing is launched.
int rbq_debug_stage_resources(rbq_index* h, uint64_t nq, uint32_t top_k, uint32_t nprobe, uint32_t* out) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    if (!h || h->reps.empty() || !out || nq == 0 || nq > 0x7fffffffull || top_k == 0) return fail(RBQ_INVALID_CONFIG, "bad argument");
    Replica* ix = h->reps[0];
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(RBQ_DEVICE, "hipSetDevice failed");
    Workspace w; // scratch buffers are sized like a real call's (and freed on return); no kernel runs
    StageProbes probes;
    int rc;
    {
        // the probe comes off this thread on EVERY exit path (an exception out of search_device — bad_alloc from a
        // workspace — used to leave it installed, and every later search of the thread then launched nothing and returned RBQ_OK)
        struct ProbeGuard { ProbeGuard(StageProbes* p) { stage_probes() = p; } ~ProbeGuard() { stage_probes() = nullptr; } } guard(&probes);
        struct WsGuard { Workspace& w; ~WsGuard() { w.release(); } } wguard{w};
        rc = search_device(ix, &w, nullptr, nq, top_k, nprobe, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
    }
    if (rc) return rc;
    for (int st = 0; st < 4; ++st) {
        const KernelProbe& k = probes.k[st];
        hipFuncAttributes fa;
        uint32_t* o = out + 6 * st;
        if (!k.fn) { // a stage this call shape does not launch (the latency-first front of a small call covers prep AND rank): zeros
            if (st != 1) return fail(RBQ_DEVICE, "stage was not probed");
            for (int j = 0; j < 6; ++j) o[j] = 0;
            continue;
        }
        HIP_TRY(hipFuncGetAttributes(&fa, k.fn));
        o[0] = k.grid_x * (k.grid_y ? k.grid_y : 1u); o[1] = k.block; o[2] = (uint32_t)fa.numRegs;
        o[3] = (uint32_t)(fa.sharedSizeBytes + k.dyn_lds); o[4] = (uint32_t)fa.localSizeBytes; o[5] = 0;
    }
    return RBQ_OK;
    RBQ_GUARD_END
}
uint64_t rbq_debug_bounce_copies(void) { return g_bounce_copies.load(std::memory_order_relaxed); }
uint64_t rbq_debug_heap_restarts(const rbq_index* h) { return read_counter(h, 1); }

/* Diagnostic: copy an intermediate buffer of the workspace bound to `hip_stream` (after the caller synchronised). */
int rbq_debug_copy_workspace(rbq_index* h, void* hip_stream, const char* name, void* dst, uint64_t bytes) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    if (!h || h->reps.empty() || !name || !dst) return RBQ_INVALID_CONFIG;
    Replica* ix = h->reps[h->debug_replica];
    Workspace* w = nullptr;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        auto it = ix->stream_ws.find((hipStream_t)hip_stream);
        if (it != ix->stream_ws.end()) w = it->second;
    }
    if (!w) return fail(RBQ_INVALID_CONFIG, "no workspace for this stream");
    DevBuf* b = nullptr;
    if (!std::strcmp(name, "rot")) b = &w->rot;
    else if (!std::strcmp(name, "lut")) b = &w->lut;
    else if (!std::strcmp(name, "consts")) b = &w->consts;
    else if (!std::strcmp(name, "scores")) b = &w->scores;
    else if (!std::strcmp(name, "probe")) b = &w->probe;
    else if (!std::strcmp(name, "nstream")) b = &w->nstream;
    else if (!std::strcmp(name, "wl")) b = &w->wl;
    else if (!std::strcmp(name, "nvec")) b = &w->nvec;
    else if (!std::strcmp(name, "dead_skipped")) b = &w->dead_skipped;
    else if (!std::strcmp(name, "audit_dead")) b = &w->audit_dead;
    if (!b || !b->p || bytes > b->cap) return fail(RBQ_INVALID_CONFIG, "unknown buffer or size");
    DeviceGuard g(ix->device);
    HIP_TRY(hipMemcpy(dst, b->p, bytes, hipMemcpyDeviceToHost));
    return RBQ_OK;
    RBQ_GUARD_END
}

/* Diagnostic: copy one of the index's device arrays to the host. */
int rbq_debug_copy_index(rbq_index* h, const char* name, void* dst, uint64_t bytes) {
    g_err.clear();
    RBQ_GUARD_BEGIN
    if (!h || h->reps.empty() || !name || !dst) return RBQ_INVALID_CONFIG;
    Replica* ix = h->reps[h->debug_replica];
    const size_t stride = (size_t)ix->Dc * 4 + 384, exd = ex_bytes_dev(ix->D, ix->ex_bits), slots = ix->n_blocks * 32;
    const void* p = nullptr;
    size_t have = 0;
    if (!std::strcmp(name, "blocks")) { p = ix->blocks.p; have = ix->n_blocks * stride; }
    else if (!std::strcmp(name, "ids")) { p = ix->ids.p; have = slots * 8; }
    else if (!std::strcmp(name, "ex")) { p = ix->ex.p; have = slots * exd; }
    else if (!std::strcmp(name, "fadd_ex")) { p = ix->fadd_ex.p; have = ix->ex_bits ? slots * 4 : 0; }
    else if (!std::strcmp(name, "fres_ex")) { p = ix->fres_ex.p; have = ix->ex_bits ? slots * 4 : 0; }
    else if (!std::strcmp(name, "bsum")) { p = ix->bsum.p; have = ix->n_blocks * sizeof(BlockSummary); }
    else if (!std::strcmp(name, "lsum")) { p = ix->lsum.p; have = ix->n_lists * sizeof(BlockSummary); }
    else if (!std::strcmp(name, "bsumx")) { p = ix->bsumx.p; have = ix->n_blocks * sizeof(BlockSummaryEx); }
    else if (!std::strcmp(name, "centroids")) { p = ix->centroids.p; have = ix->n_lists * ix->D * 4; }
    else if (!std::strcmp(name, "list_gb0")) { p = ix->list_gb0.p; have = ix->n_lists * 4; }
    else if (!std::strcmp(name, "list_n")) { p = ix->list_n.p; have = ix->n_lists * 4; }
    if (!p || bytes != have) return fail(RBQ_INVALID_CONFIG, "unknown array or size (have " + std::to_string(have) + " bytes)");
    DeviceGuard g(ix->device);
    HIP_TRY(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
    return RBQ_OK;
    RBQ_GUARD_END
}

} // extern "C"
