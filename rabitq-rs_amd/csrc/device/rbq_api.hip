// rbq_api.hip — C ABI (include/rbq.h) over the HIP kernels of kernels.hpp.  gfx950 only.
//
// Host responsibilities: validate like the reference (src/ivf.rs:1754-1769,1484-1702), re-lay the
// reference's ClusterData bytes into the device layout (one-time, at create/load), own HBM, and
// enqueue prep -> rank -> select -> scan for each query batch.  There is no CPU compute path:
// every failure to reach the GPU surfaces as RBQ_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rbq.h"
#include "kernels.hpp"
#include "scan.hpp"
#include "rank_mfma.hpp"
#include "encode.hpp"
#include <rocprim/rocprim.hpp>

using namespace rbq;

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

uint32_t floor_log2_u32(uint32_t x) { uint32_t r = 0; while (x >>= 1) ++r; return r; }
uint32_t next_pow2(uint32_t x) { uint32_t p = 1; while (p < x) p <<= 1; return p; }

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

struct Workspace {
    hipStream_t stream = nullptr;
    DevBuf queries, rot, lut, consts, scores, probe, wl, nstream, nvec, out_ids, out_scores, out_counts, diag, filter, rot_hi, rot_lo;
    void release() {
        for (DevBuf* b : {&queries, &rot, &lut, &consts, &scores, &probe, &wl, &nstream, &nvec, &rot_hi, &rot_lo, &out_ids, &out_scores,
                          &out_counts, &diag, &filter})
            b->release();
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
};

struct StageProf {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; // pairs recorded since rbq_profile_begin
    double ms = 0;
    uint64_t launches = 0;
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

} // namespace

struct rbq_index {
    int device = 0;
    uint32_t dim = 0, D = 0, Dc = 0;
    uint8_t metric = 0, rotator = 0, ex_bits = 0;
    uint64_t n_vectors = 0, n_lists = 0, n_blocks = 0;
    uint32_t trunc = 0;
    float fac = 1.0f;
    // device arrays
    void *d_rot_blob = nullptr, *d_centroids = nullptr, *d_blocks = nullptr, *d_ids = nullptr, *d_ex = nullptr,
         *d_fadd_ex = nullptr, *d_fres_ex = nullptr, *d_list_gb0 = nullptr, *d_list_n = nullptr, *d_prof_total = nullptr, *d_bsum = nullptr, *d_cnorm2 = nullptr, *d_fallbacks = nullptr, *d_cent_hi = nullptr, *d_cent_lo = nullptr;
    float cnorm2_max = 0.0f;
    bool no_block_bound = false; // rbq_debug_set_option("block_bound", 0)
    bool f32_rank = false;       // rbq_debug_set_option("f32_rank", 1): f32 MFMA GEMM instead of the split-bf16 one
    bool small_rank_tiles = false; // rbq_debug_set_option("small_rank_tiles", 1): 64x64 GEMM tiles whatever the problem size
    bool wg_prep = false;        // rbq_debug_set_option("wg_prep", 1): workgroup-per-query k_prep for every rotator
    bool exact_heap = false;     // rbq_debug_set_option("exact_heap", 1): BinaryHeap emulation from the first candidate
    bool force_rank_fallback = false; // RBQ_FORCE_RANK_FALLBACK=1: exercise the all-lists canonical fallback
    bool exact_rank = false; // RBQ_EXACT_RANK=1: rank all pairs in canonical order (A/B and debugging)
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
    StageProf prof[4]; // prep, rank, select, scan
    EventPool ev_pool;
    uint64_t prof_scan_bytes = 0;
};

namespace {

void free_index(rbq_index* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    for (void* p : {ix->d_rot_blob, ix->d_centroids, ix->d_blocks, ix->d_ids, ix->d_ex, ix->d_fadd_ex, ix->d_fres_ex,
                    ix->d_list_gb0, ix->d_list_n, ix->d_prof_total, ix->d_bsum, ix->d_cnorm2, ix->d_fallbacks, ix->d_cent_hi, ix->d_cent_lo})
        if (p) (void)hipFree(p);
    for (Workspace* w : ix->pool) { w->release(); delete w; }
    for (auto& kv : ix->stream_ws) { kv.second->release(); delete kv.second; }
    for (auto& sp : ix->prof)
        for (auto& e : sp.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    ix->ev_pool.destroy();
    delete ix;
}

// inverse of pack_codes (src/simd.rs:864-904): KPERM0[j] = (j>>1) + 8*(j&1)  =>  j = 2*(u&7) + (u>>3)
inline uint8_t fastscan_byte(const uint8_t* packed, size_t col, uint32_t v) {
    const uint32_t u = v & 15u, j = 2u * (u & 7u) + (u >> 3);
    const uint8_t a = packed[col * 32 + j], b = packed[col * 32 + 16 + j];
    const uint8_t hi = v < 16 ? (a & 15) : (a >> 4);
    const uint8_t lo = v < 16 ? (b & 15) : (b >> 4);
    return (uint8_t)((hi << 4) | lo);
}

// Reference packed ex code of one vector (src/simd.rs:2478-2541,2601-2695) -> device lane-major record:
// lane l = dim % 16 holds codes of dims 16t+l as a little-endian bit string (ex bits each), stored
// [j][lane][16 B] so that 16 lanes read 256 contiguous bytes per 16-byte load.
void relayout_ex(const uint8_t* src, uint32_t D, uint32_t ex_bits, uint8_t* dst) {
    const uint32_t w4 = ex_w4(D, ex_bits);
    std::memset(dst, 0, (size_t)w4 * 256);
    uint32_t* out = reinterpret_cast<uint32_t*>(dst);
    for (uint32_t t = 0; t < D / 16; ++t) {
        uint32_t codes[16];
        if (ex_bits == 2) {
            uint32_t w;
            std::memcpy(&w, src + t * 4, 4);
            for (uint32_t l = 0; l < 16; ++l) codes[l] = (w >> (8 * (l & 3) + 2 * (l >> 2))) & 3u;
        } else { // 6
            uint64_t lo;
            uint32_t hi;
            std::memcpy(&lo, src + t * 12, 8);
            std::memcpy(&hi, src + t * 12 + 8, 4);
            for (uint32_t l = 0; l < 16; ++l) {
                const uint32_t low4 = l < 8 ? (uint32_t)((lo >> (8 * l)) & 15u) : (uint32_t)((lo >> (8 * (l - 8) + 4)) & 15u);
                const uint32_t top2 = (hi >> (8 * (l & 3) + 2 * (l >> 2))) & 3u;
                codes[l] = low4 | (top2 << 4);
            }
        }
        const uint32_t cpu = ex_cpu(ex_bits), unit = t / cpu, k = t % cpu;
        const uint32_t bit = k * ex_bits, idx = bit >> 5, sh = bit & 31u; // inside the 128-bit unit
        for (uint32_t l = 0; l < 16; ++l) {
            uint32_t* u = out + (unit * 16 + l) * 4;
            u[idx] |= codes[l] << sh;
            if (sh + ex_bits > 32) u[idx + 1] |= codes[l] >> (32 - sh);
        }
    }
}

// One reference batch record -> one device block: lane-major code granules + the three factor rows.
void relayout_block(const uint8_t* rec, uint32_t D, uint32_t Dc, uint8_t* dst) {
    const size_t dim_bytes = D / 8, G16 = Dc >> 7;
    std::memset(dst, 0, (size_t)Dc * 4);
    for (uint32_t v = 0; v < 32; ++v)
        for (size_t col = 0; col < dim_bytes; ++col) {
            const uint8_t b = fastscan_byte(rec, col, v);
            const size_t g = col >> 4;
            if (g < G16) dst[g * 512 + v * 16 + (col & 15)] = b;
            else dst[G16 * 512 + v * 8 + (col & 7)] = b;
        }
    std::memcpy(dst + (size_t)Dc * 4, rec + (size_t)D * 4, 384);
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

template <typename T>
int upload(void** dptr, const std::vector<T>& v) {
    size_t bytes = v.size() * sizeof(T);
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 16));
    if (bytes) HIP_TRY(hipMemcpy(*dptr, v.data(), bytes, hipMemcpyHostToDevice));
    return RBQ_OK;
}

// arrays derived from the rotated centroids + the diagnostic counters (shared by create_impl and the device build)
int finish_centroid_arrays(rbq_index* ix, const std::vector<float>& cent) {
    const uint32_t D = ix->D;
    std::vector<float> cn(ix->n_lists);
    double mx = 0;
    for (uint64_t c = 0; c < ix->n_lists; ++c) {
        double a = 0;
        for (uint32_t i = 0; i < D; ++i) a += (double)cent[c * D + i] * (double)cent[c * D + i];
        cn[c] = (float)a;
        if (std::isfinite(a)) mx = std::max(mx, a);
    }
    ix->cnorm2_max = (float)(mx * 1.0000002); // rounded up
    int rc;
    if ((rc = upload(&ix->d_cnorm2, cn))) return rc;
    std::vector<uint16_t> ch(cent.size()), cl(cent.size()); // split-bf16 image of the centroids (k_rank_bf16_db)
    for (size_t i = 0; i < cent.size(); ++i) bf16_split(cent[i], ch[i], cl[i]);
    if ((rc = upload(&ix->d_cent_hi, ch))) return rc;
    if ((rc = upload(&ix->d_cent_lo, cl))) return rc;
    std::vector<unsigned int> z(2, 0); // [0] rank fallbacks, [1] heap restarts
    if ((rc = upload(&ix->d_fallbacks, z))) return rc;
    std::vector<unsigned long long> z1(1, 0);
    if ((rc = upload(&ix->d_prof_total, z1))) return rc;
    const char* e = std::getenv("RBQ_EXACT_RANK");
    ix->exact_rank = e && e[0] == '1';
    const char* f = std::getenv("RBQ_FORCE_RANK_FALLBACK");
    ix->force_rank_fallback = f && f[0] == '1';
    return RBQ_OK;
}

int create_impl(const rbq_header* hdr, const rbq_list_view* lists, int n_devices, const int* devices, rbq_index** out) {
    if (!out) return fail(RBQ_INVALID_CONFIG, "null out pointer");
    *out = nullptr;
    int rc = validate_header(hdr);
    if (rc) return rc;
    if (n_devices != 1) return fail(RBQ_INVALID_CONFIG, "n_devices must be 1 (one handle per GPU; shard queries across handles)");
    if (!lists) return fail(RBQ_INVALID_CONFIG, "null lists");

    int dev = 0;
    if (devices) dev = devices[0];
    else HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipSetDevice(dev));

    rbq_index* ix = new rbq_index();
    ix->device = dev;
    ix->dim = hdr->dim; ix->D = hdr->padded_dim; ix->Dc = (hdr->padded_dim + 63u) / 64u * 64u;
    ix->metric = hdr->metric; ix->rotator = hdr->rotator; ix->ex_bits = hdr->ex_bits;
    ix->n_lists = hdr->n_lists;
    ix->trunc = 1u << floor_log2_u32(hdr->dim);
    ix->fac = 1.0f / std::sqrt((float)ix->trunc);

    const uint32_t D = ix->D, Dc = ix->Dc;
    const size_t ref_stride = (size_t)D * 4 + 384, dev_stride = (size_t)Dc * 4 + 384, exb = (size_t)D * ix->ex_bits / 8;
    const size_t exd = ex_bytes_dev(D, ix->ex_bits);
    std::vector<uint32_t> gb0(ix->n_lists), ln(ix->n_lists);
    uint64_t nblocks = 0, nvec = 0;
    for (uint64_t c = 0; c < ix->n_lists; ++c) {
        const rbq_list_view& L = lists[c];
        if (L.n > 0xffffffffull) { free_index(ix); return fail(RBQ_INVALID_CONFIG, "list too large"); }
        const uint64_t nb = (L.n + 31) / 32;
        if (L.batch_len != nb * ref_stride) { free_index(ix); return fail(RBQ_INVALID_PERSISTENCE, "batch_data length mismatch - possible corruption or version incompatibility"); }
        if (L.n && (!L.centroid || !L.ids || !L.batch_data || (ix->ex_bits && (!L.ex_codes || !L.f_add_ex || !L.f_rescale_ex)))) {
            free_index(ix); return fail(RBQ_INVALID_CONFIG, "null list array");
        }
        if (!L.centroid) { free_index(ix); return fail(RBQ_INVALID_CONFIG, "null centroid"); }
        gb0[c] = (uint32_t)nblocks; ln[c] = (uint32_t)L.n;
        nblocks += nb; nvec += L.n;
    }
    if (nblocks * 32 > 0xffffffffull) { free_index(ix); return fail(RBQ_INVALID_CONFIG, "index too large for 32-bit vector slots"); }
    ix->n_blocks = nblocks; ix->n_vectors = nvec; ix->h_list_n = ln;

    // host staging in device layout
    std::vector<float> cent((size_t)ix->n_lists * D);
    std::vector<uint8_t> blocks(nblocks * dev_stride);
    std::vector<uint64_t> ids(nblocks * 32, ~0ull);
    std::vector<uint8_t> ex(exd ? nblocks * 32 * exd + 256 : 0);
    std::vector<float> fa(ix->ex_bits ? nblocks * 32 : 0), fr(ix->ex_bits ? nblocks * 32 : 0);
    std::vector<BlockSummary> bsum(nblocks);
    for (uint64_t c = 0; c < ix->n_lists; ++c) {
        const rbq_list_view& L = lists[c];
        std::memcpy(&cent[c * D], L.centroid, sizeof(float) * D);
        const uint64_t nb = (L.n + 31) / 32;
        for (uint64_t b = 0; b < nb; ++b) {
            relayout_block(L.batch_data + b * ref_stride, D, Dc, &blocks[(gb0[c] + b) * dev_stride]);
            // factor ranges over the block's real vectors (block-level lower bound of k_scan)
            float fac[96];
            std::memcpy(fac, L.batch_data + b * ref_stride + (size_t)D * 4, 384);
            const uint32_t nv = (uint32_t)std::min<uint64_t>(32, L.n - b * 32);
            BlockSummary bs;
            bs.fadd_min = bs.fres_min = bs.ferr_min = INFINITY;
            bs.fadd_max = bs.fres_max = bs.ferr_max = -INFINITY;
            bs.usable = 1; bs.pad = 0;
            for (uint32_t v = 0; v < nv; ++v) {
                const float a = fac[v], r = fac[32 + v], e = fac[64 + v];
                if (!std::isfinite(a) || !std::isfinite(r) || !std::isfinite(e)) bs.usable = 0;
                bs.fadd_min = std::min(bs.fadd_min, a); bs.fadd_max = std::max(bs.fadd_max, a);
                bs.fres_min = std::min(bs.fres_min, r); bs.fres_max = std::max(bs.fres_max, r);
                bs.ferr_min = std::min(bs.ferr_min, e); bs.ferr_max = std::max(bs.ferr_max, e);
            }
            if (!bs.usable) { bs.fadd_min = bs.fadd_max = bs.fres_min = bs.fres_max = bs.ferr_min = bs.ferr_max = 0.0f; }
            bsum[gb0[c] + b] = bs;
        }
        const size_t s0 = (size_t)gb0[c] * 32;
        if (L.n) {
            std::memcpy(&ids[s0], L.ids, L.n * 8);
            if (ix->ex_bits) {
                for (uint64_t v = 0; v < L.n; ++v) relayout_ex(L.ex_codes + v * exb, D, ix->ex_bits, &ex[(s0 + v) * exd]);
                std::memcpy(&fa[s0], L.f_add_ex, L.n * 4);
                std::memcpy(&fr[s0], L.f_rescale_ex, L.n * 4);
            }
        }
    }
    std::vector<uint8_t> blob;
    if (hdr->rotator_len) blob.assign(hdr->rotator_blob, hdr->rotator_blob + hdr->rotator_len);
    std::vector<uint64_t> nblk(ix->n_lists);
    for (uint64_t c = 0; c < ix->n_lists; ++c) nblk[c] = (ln[c] + 31u) / 32u;
    std::sort(nblk.begin(), nblk.end(), std::greater<uint64_t>());
    ix->nblk_desc_prefix.assign(ix->n_lists + 1, 0);
    for (uint64_t c = 0; c < ix->n_lists; ++c) ix->nblk_desc_prefix[c + 1] = ix->nblk_desc_prefix[c] + nblk[c];

#define UP(dst, vec)                                             \
    do {                                                         \
        int _rc = upload(&ix->dst, vec);                         \
        if (_rc) { free_index(ix); return _rc; }                 \
    } while (0)
    UP(d_rot_blob, blob); UP(d_centroids, cent); UP(d_blocks, blocks); UP(d_ids, ids); UP(d_ex, ex);
    UP(d_fadd_ex, fa); UP(d_fres_ex, fr); UP(d_list_gb0, gb0); UP(d_list_n, ln);
    UP(d_bsum, bsum);
    if ((rc = finish_centroid_arrays(ix, cent))) { free_index(ix); return rc; }
#undef UP
    *out = ix;
    return RBQ_OK;
}

// ---- GPU-side encoder (encode.hpp): the device analogue of train_with_clusters' quantisation loop ------------
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
    HIP_TRY(hipSetDevice(dev));

    rbq_index* ix = new rbq_index();
    ix->device = dev;
    ix->dim = hdr->dim; ix->D = hdr->padded_dim; ix->Dc = (hdr->padded_dim + 63u) / 64u * 64u;
    ix->metric = hdr->metric; ix->rotator = hdr->rotator; ix->ex_bits = hdr->ex_bits;
    ix->n_lists = hdr->n_lists;
    ix->trunc = 1u << floor_log2_u32(hdr->dim);
    ix->fac = 1.0f / std::sqrt((float)ix->trunc);
    const uint32_t D = ix->D, Dc = ix->Dc, dim = ix->dim, nlist = (uint32_t)ix->n_lists;
    const size_t dev_stride = (size_t)Dc * 4 + 384, exd = ex_bytes_dev(D, ix->ex_bits);

    std::vector<void*> temps;
    auto cleanup = [&](int code) { for (void* p : temps) if (p) (void)hipFree(p); if (code) free_index(ix); return code; };
#define TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { (void)fail(RBQ_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); return cleanup(RBQ_DEVICE); } } while (0)
    auto tmp_alloc = [&](void** p, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(p, bytes ? bytes : 16); if (e == hipSuccess) temps.push_back(*p); return e; };

    std::vector<uint8_t> blob;
    if (hdr->rotator_len) blob.assign(hdr->rotator_blob, hdr->rotator_blob + hdr->rotator_len);
    if ((rc = upload(&ix->d_rot_blob, blob))) return cleanup(rc);

    // rotated centroids
    std::vector<float> cent((size_t)nlist * D);
    {
        float* d_craw = nullptr;
        TRY(tmp_alloc((void**)&d_craw, (size_t)nlist * dim * 4));
        TRY(hipMemcpy(d_craw, centroids, (size_t)nlist * dim * 4, hipMemcpyHostToDevice));
        TRY(hipMalloc(&ix->d_centroids, (size_t)nlist * D * 4));
        hipLaunchKernelGGL(k_rotate_rows, dim3(nlist), dim3(kThreads), (size_t)D * 4 * 2, 0, (const float*)d_craw, (const uint32_t*)nullptr,
                           dim, D, (int)ix->rotator, (const uint8_t*)ix->d_rot_blob, ix->trunc, ix->fac, (float*)ix->d_centroids);
        TRY(hipGetLastError());
        TRY(hipMemcpy(cent.data(), ix->d_centroids, cent.size() * 4, hipMemcpyDeviceToHost));
    }

    // list sizes
    std::vector<uint32_t> ln(nlist), gb0(nlist);
    {
        uint32_t* d_counts = nullptr;
        TRY(tmp_alloc((void**)&d_counts, (size_t)(nlist + 1) * 4));
        TRY(hipMemset(d_counts, 0, (size_t)(nlist + 1) * 4));
        hipLaunchKernelGGL(k_count_assign, dim3(1024), dim3(256), 0, 0, d_assign, n, nlist, d_counts, d_counts + nlist);
        TRY(hipGetLastError());
        std::vector<uint32_t> hc(nlist + 1);
        TRY(hipMemcpy(hc.data(), d_counts, hc.size() * 4, hipMemcpyDeviceToHost));
        if (hc[nlist]) { (void)fail(RBQ_INVALID_CONFIG, "assignment out of range"); return cleanup(RBQ_INVALID_CONFIG); }
        std::copy(hc.begin(), hc.begin() + nlist, ln.begin());
    }
    uint64_t nblocks = 0;
    std::vector<uint64_t> vstart(nlist);
    {
        uint64_t run = 0;
        for (uint32_t c = 0; c < nlist; ++c) { gb0[c] = (uint32_t)nblocks; vstart[c] = run; nblocks += (ln[c] + 31u) / 32u; run += ln[c]; }
    }
    if (nblocks * 32 > 0xffffffffull) { (void)fail(RBQ_INVALID_CONFIG, "index too large for 32-bit vector slots"); return cleanup(RBQ_INVALID_CONFIG); }
    ix->n_blocks = nblocks; ix->n_vectors = n; ix->h_list_n = ln;
    {
        std::vector<uint64_t> nblk(nlist);
        for (uint32_t c = 0; c < nlist; ++c) nblk[c] = (ln[c] + 31u) / 32u;
        std::sort(nblk.begin(), nblk.end(), std::greater<uint64_t>());
        ix->nblk_desc_prefix.assign(nlist + 1, 0);
        for (uint32_t c = 0; c < nlist; ++c) ix->nblk_desc_prefix[c + 1] = ix->nblk_desc_prefix[c] + nblk[c];
    }
    if ((rc = upload(&ix->d_list_gb0, gb0))) return cleanup(rc);
    if ((rc = upload(&ix->d_list_n, ln))) return cleanup(rc);
    const uint64_t nslots = nblocks * 32;

    // stable grouping by list (ascending vector index inside a list, src/ivf.rs:1141-1149): radix sort on the list id
    uint32_t* d_slot_src = nullptr;
    {
        uint32_t *d_ko = nullptr, *d_vi = nullptr, *d_vo = nullptr;
        uint64_t* d_vstart = nullptr;
        TRY(tmp_alloc((void**)&d_ko, n * 4)); TRY(tmp_alloc((void**)&d_vi, n * 4)); TRY(tmp_alloc((void**)&d_vo, n * 4));
        TRY(tmp_alloc((void**)&d_vstart, (size_t)nlist * 8));
        TRY(hipMemcpy(d_vstart, vstart.data(), (size_t)nlist * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_iota, dim3(1024), dim3(256), 0, 0, d_vi, n);
        TRY(hipGetLastError());
        unsigned bits = 1;
        while ((1ull << bits) < nlist) ++bits;
        size_t tb = 0;
        TRY(rocprim::radix_sort_pairs(nullptr, tb, d_assign, d_ko, d_vi, d_vo, (size_t)n, 0u, bits, (hipStream_t)0));
        void* d_tmp = nullptr;
        TRY(tmp_alloc(&d_tmp, tb));
        TRY(rocprim::radix_sort_pairs(d_tmp, tb, d_assign, d_ko, d_vi, d_vo, (size_t)n, 0u, bits, (hipStream_t)0));
        TRY(tmp_alloc((void**)&d_slot_src, nslots * 4));
        TRY(hipMemset(d_slot_src, 0xff, nslots * 4));
        hipLaunchKernelGGL(k_scatter_slots, dim3(1024), dim3(256), 0, 0, (const uint32_t*)d_ko, (const uint32_t*)d_vo, n,
                           (const uint32_t*)ix->d_list_gb0, (const uint64_t*)d_vstart, d_slot_src);
        TRY(hipGetLastError());
    }
    // block -> list, block -> number of real vectors
    uint32_t *d_block_list = nullptr, *d_block_nv = nullptr;
    {
        std::vector<uint32_t> bl(nblocks), bn(nblocks);
        for (uint32_t c = 0; c < nlist; ++c) {
            const uint32_t nb = (ln[c] + 31u) / 32u;
            for (uint32_t b = 0; b < nb; ++b) { bl[gb0[c] + b] = c; bn[gb0[c] + b] = std::min<uint32_t>(32u, ln[c] - b * 32u); }
        }
        TRY(tmp_alloc((void**)&d_block_list, nblocks * 4)); TRY(tmp_alloc((void**)&d_block_nv, nblocks * 4));
        TRY(hipMemcpy(d_block_list, bl.data(), nblocks * 4, hipMemcpyHostToDevice));
        TRY(hipMemcpy(d_block_nv, bn.data(), nblocks * 4, hipMemcpyHostToDevice));
    }

    // final arrays
    TRY(hipMalloc(&ix->d_blocks, nblocks * dev_stride)); TRY(hipMemset(ix->d_blocks, 0, nblocks * dev_stride));
    TRY(hipMalloc(&ix->d_ids, nslots * 8));
    TRY(hipMalloc(&ix->d_ex, exd ? nslots * exd + 256 : 16));
    TRY(hipMalloc(&ix->d_fadd_ex, ix->ex_bits ? nslots * 4 : 16)); TRY(hipMalloc(&ix->d_fres_ex, ix->ex_bits ? nslots * 4 : 16));
    TRY(hipMalloc(&ix->d_bsum, nblocks * sizeof(BlockSummary)));

    // encode, a chunk of blocks at a time (scratch: rotated rows + raw ex codes of the chunk)
    {
        uint64_t chunk_blocks = std::max<uint64_t>(2, ((512ull << 20) / ((size_t)D * 4) / 32) & ~1ull);
        chunk_blocks = std::min<uint64_t>(chunk_blocks, (nblocks + 1) & ~1ull);
        const uint64_t chunk_slots = chunk_blocks * 32;
        float* d_rows = nullptr;
        uint8_t* d_raw = nullptr;
        TRY(tmp_alloc((void**)&d_rows, chunk_slots * D * 4));
        TRY(tmp_alloc((void**)&d_raw, ix->ex_bits ? chunk_slots * D : 16));
        for (uint64_t b0 = 0; b0 < nblocks; b0 += chunk_blocks) {
            const uint64_t nb = std::min<uint64_t>(chunk_blocks, nblocks - b0), ns = nb * 32, s0 = b0 * 32;
            hipLaunchKernelGGL(k_rotate_rows, dim3((uint32_t)ns), dim3(kThreads), (size_t)D * 4 * 2, 0, d_data, (const uint32_t*)(d_slot_src + s0),
                               dim, D, (int)ix->rotator, (const uint8_t*)ix->d_rot_blob, ix->trunc, ix->fac, d_rows);
            TRY(hipGetLastError());
            EncodeParams P;
            P.rows = d_rows; P.centroids = (const float*)ix->d_centroids; P.slot_src = d_slot_src + s0; P.block_list = d_block_list + b0;
            P.blocks = (uint8_t*)ix->d_blocks + b0 * dev_stride; P.raw_ex = d_raw;
            P.f_add_ex = (float*)ix->d_fadd_ex + s0; P.f_rescale_ex = (float*)ix->d_fres_ex + s0; P.ids = (uint64_t*)ix->d_ids + s0;
            P.src_base = 0; P.nslots = (uint32_t)ns; P.D = D; P.Dc = Dc; P.ex_bits = ix->ex_bits; P.metric = ix->metric; P.t_const = t_const;
            hipLaunchKernelGGL(k_encode, dim3((uint32_t)((ns + kEncThreads - 1) / kEncThreads)), dim3(kEncThreads), 0, 0, P);
            TRY(hipGetLastError());
            if (ix->ex_bits) {
                hipLaunchKernelGGL(k_pack_ex, dim3((uint32_t)((ns + 15) / 16)), dim3(256), 0, 0, (const uint8_t*)d_raw, (const uint32_t*)(d_slot_src + s0),
                                   (uint32_t)ns, D, (uint32_t)ix->ex_bits, (uint8_t*)ix->d_ex + s0 * exd);
                TRY(hipGetLastError());
            }
        }
        hipLaunchKernelGGL(k_block_summary, dim3((uint32_t)((nblocks + 7) / 8)), dim3(256), 0, 0, (const uint8_t*)ix->d_blocks,
                           (const uint32_t*)d_block_nv, (uint32_t)nblocks, Dc, (BlockSummary*)ix->d_bsum);
        TRY(hipGetLastError());
        TRY(hipDeviceSynchronize());
    }
#undef TRY
    if ((rc = finish_centroid_arrays(ix, cent))) return cleanup(rc);
    cleanup(0);
    *out = ix;
    return RBQ_OK;
}

Workspace* take_ws(rbq_index* ix) {
    {
        std::lock_guard<std::mutex> g(ix->mu);
        if (!ix->pool.empty()) { Workspace* w = ix->pool.back(); ix->pool.pop_back(); return w; }
    }
    Workspace* w = new Workspace();
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) { delete w; return nullptr; }
    return w;
}
void give_ws(rbq_index* ix, Workspace* w) {
    std::lock_guard<std::mutex> g(ix->mu);
    ix->pool.push_back(w);
}

struct ProfScope {
    rbq_index* ix; int stage; hipStream_t s; std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    bool on = false, ext = false;
    // ext: the launch itself carries the event pair (hipExtLaunchKernelGGL: start/stop come from the dispatch
    // packet, no separate marker packets in the queue); otherwise the pair is recorded around the scope
    ProfScope(rbq_index* ix_, int st, hipStream_t s_, bool ext_ = false) : ix(ix_), stage(st), s(s_), ext(ext_) {
        if (ix->profiling && ((ix->prof_mask >> st) & 1u)) {
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
            ix->prof[stage].ev.push_back(ev);
        }
    }
};

template <int DT, int EX>
hipError_t launch_scan_t(const ScanParams& P, uint32_t nq, size_t lds, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (lds > 48 * 1024) { // default dynamic-LDS limit covers the common case; raising it is a driver call
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan<DT, EX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (ev0) hipExtLaunchKernelGGL((k_scan<DT, EX>), dim3(nq), dim3(kScanThreads), lds, s, ev0, ev1, 0, P);
    else hipLaunchKernelGGL((k_scan<DT, EX>), dim3(nq), dim3(kScanThreads), lds, s, P);
    return hipGetLastError();
}
template <int DT>
hipError_t launch_scan(const ScanParams& P, uint32_t nq, size_t lds, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (DT == 0) return launch_scan_t<0, 0>(P, nq, lds, s, ev0, ev1);
    switch (P.ex_bits) {
        case 0: return launch_scan_t<DT, 0>(P, nq, lds, s, ev0, ev1);
        case 2: return launch_scan_t<DT, 2>(P, nq, lds, s, ev0, ev1);
        default: return launch_scan_t<DT, 6>(P, nq, lds, s, ev0, ev1);
    }
}

// k_scan launch shared by the IVF search and the MSTG posting-list scan
int scan_stage(rbq_index* ix, Workspace* w, uint64_t nq, uint32_t probe_stride, uint32_t top_k, uint64_t wl_stride,
               const uint32_t* d_filter, uint64_t filter_nbits, uint64_t* d_ids, float* d_scores, uint32_t* d_counts,
               rbq_diag* d_diag, bool mstg, hipStream_t stream) {
    const uint32_t D = ix->D, Dc = ix->Dc;
    ProfScope ps(ix, 3, stream, /*ext=*/true);
    ScanParams P;
    P.blocks = (const uint8_t*)ix->d_blocks; P.ids = (const uint64_t*)ix->d_ids; P.ex_codes = (const uint8_t*)ix->d_ex;
    P.f_add_ex = (const float*)ix->d_fadd_ex; P.f_rescale_ex = (const float*)ix->d_fres_ex;
    P.lut = (const uint8_t*)w->lut.p; P.rot = (const float*)w->rot.p; P.consts = (const QueryConsts*)w->consts.p;
    P.probe = (const ProbeInfo*)w->probe.p; P.wl = (const StreamItem*)w->wl.p; P.nstream = (const uint32_t*)w->nstream.p;
    P.filter = d_filter; P.filter_nbits = filter_nbits; P.wl_stride = wl_stride;
    P.out_ids = d_ids; P.out_scores = d_scores; P.out_counts = d_counts; P.diag = (unsigned long long*)d_diag;
    P.D = D; P.Dc = Dc; P.nprobe = probe_stride; P.top_k = top_k; P.metric = ix->metric;
    P.ex_bits = mstg ? 0u : ix->ex_bits; // MSTG search never evaluates the ex codes (src/mstg/index.rs:216-330)
    P.no_block_bound = ix->no_block_bound ? 1u : 0u;
    P.exact_heap = ix->exact_heap ? 1u : 0u;
    P.heap_restarts = (unsigned int*)ix->d_fallbacks + 1;
    P.mstg = mstg ? 1u : 0u;
    const size_t lds = scan_lds_bytes(Dc, D, P.ex_bits, top_k);
    hipError_t e;
    if (D == Dc && D == 960) e = launch_scan<960>(P, (uint32_t)nq, lds, stream, ps.start(), ps.stop());
    else if (D == Dc && D == 768) e = launch_scan<768>(P, (uint32_t)nq, lds, stream, ps.start(), ps.stop());
    else if (D == Dc && D == 128) e = launch_scan<128>(P, (uint32_t)nq, lds, stream, ps.start(), ps.stop());
    else e = launch_scan<0>(P, (uint32_t)nq, lds, stream, ps.start(), ps.stop());
    HIP_TRY(e);
    return RBQ_OK;
}

// Core: everything on device pointers, enqueued on `stream`. Workspace buffers come from `w`.
int search_device(rbq_index* ix, Workspace* w, const float* d_queries, uint64_t nq, uint32_t top_k, uint32_t nprobe_in,
                  const uint32_t* d_filter, uint64_t filter_nbits, uint64_t* d_ids, float* d_scores, uint32_t* d_counts,
                  rbq_diag* d_diag, hipStream_t stream) {
    const uint32_t D = ix->D, Dc = ix->Dc;
    const uint32_t nlist = (uint32_t)ix->n_lists;
    uint32_t nprobe = nprobe_in < 1 ? 1 : nprobe_in;
    if (nprobe > nlist) nprobe = nlist;
    if (nprobe > 4096) return fail(RBQ_INVALID_CONFIG, "nprobe > 4096 is not supported by the GPU probe selector");
    if (top_k > 4096) return fail(RBQ_INVALID_CONFIG, "top_k > 4096 is not supported by the GPU top-k stage");
    const uint32_t np2 = next_pow2(nprobe);
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
    const bool split_rank = !ix->exact_rank && !ix->f32_rank && D % 64 == 0; // k_rank_bf16_db (else k_rank_mfma)
    if (split_rank) {
        if ((rc = w->rot_hi.ensure(nq * D * 2))) return rc;
        if ((rc = w->rot_lo.ensure(nq * D * 2))) return rc;
    }

    {
        ProfScope ps(ix, 0, stream);
        if (ix->rotator == RBQ_ROTATOR_MATRIX || ix->wg_prep) { // O(D^2) matrix rotation: one workgroup per query
            const size_t lds = (size_t)D * 4 * 2;
            hipLaunchKernelGGL(k_prep, dim3((uint32_t)nq), dim3(kThreads), lds, stream, d_queries, ix->dim, D, Dc, (int)ix->rotator,
                               (const uint8_t*)ix->d_rot_blob, ix->trunc, ix->fac, (uint32_t)ix->ex_bits, (float*)w->rot.p,
                               (uint8_t*)w->lut.p, (QueryConsts*)w->consts.p, split_rank ? (uint16_t*)w->rot_hi.p : nullptr,
                               split_rank ? (uint16_t*)w->rot_lo.p : nullptr);
        } else { // FHT-Kac / identity: one wave per query
            const uint32_t qpw = kThreads / 64;
            hipLaunchKernelGGL(k_prep_wave, dim3((uint32_t)((nq + qpw - 1) / qpw)), dim3(kThreads), (size_t)D * 4 * 2 * qpw + D / 2, stream,
                               d_queries, (uint32_t)nq, ix->dim, D, Dc, (int)ix->rotator, (const uint8_t*)ix->d_rot_blob, ix->trunc,
                               ix->fac, (uint32_t)ix->ex_bits, (float*)w->rot.p, (uint8_t*)w->lut.p, (QueryConsts*)w->consts.p,
                               split_rank ? (uint16_t*)w->rot_hi.p : nullptr, split_rank ? (uint16_t*)w->rot_lo.p : nullptr);
        }
        HIP_TRY(hipGetLastError());
    }
    if (ix->exact_rank) {
        {
            ProfScope ps(ix, 1, stream);
            dim3 grid((nlist + 31) / 32, (uint32_t)((nq + 31) / 32));
            if (ix->metric == 0)
                hipLaunchKernelGGL(k_rank_scores<0>, grid, dim3(kThreads), 0, stream, (const float*)w->rot.p,
                                   (const float*)ix->d_centroids, (uint32_t)nq, nlist, D, (float*)w->scores.p);
            else
                hipLaunchKernelGGL(k_rank_scores<1>, grid, dim3(kThreads), 0, stream, (const float*)w->rot.p,
                                   (const float*)ix->d_centroids, (uint32_t)nq, nlist, D, (float*)w->scores.p);
            HIP_TRY(hipGetLastError());
        }
        {
            ProfScope ps(ix, 2, stream);
            const size_t lds = (size_t)np2 * 8 + (size_t)D * 4 + kThreads * 4;
            hipLaunchKernelGGL(k_select, dim3((uint32_t)nq), dim3(kThreads), lds, stream, (const float*)w->scores.p, nlist, nprobe,
                               np2, (int)ix->metric, (const float*)w->rot.p, (const float*)ix->d_centroids, D,
                               (const uint32_t*)ix->d_list_gb0, (const uint32_t*)ix->d_list_n, (ProbeInfo*)w->probe.p,
                               (StreamItem*)w->wl.p, wl_stride, (uint32_t*)w->nstream.p, (unsigned long long*)w->nvec.p,
                               ix->profiling ? (unsigned long long*)ix->d_prof_total : nullptr,
                               (const QueryConsts*)w->consts.p, (const BlockSummary*)ix->d_bsum);
            HIP_TRY(hipGetLastError());
        }
    } else {
        {
            ProfScope ps(ix, 1, stream); // approximate scores: one MFMA GEMM
            const bool big = !ix->small_rank_tiles && (uint64_t)((nlist + 127) / 128) * ((nq + 127) / 128) >= 192; // enough 128x128 tiles to fill the chip
            const uint32_t T = big ? 128u : 64u;
            dim3 grid((nlist + T - 1) / T, (uint32_t)((nq + T - 1) / T));
            if (split_rank) {
                // big problems: 128x128 tiles, 8 waves (each 64x32) — the tile traffic of the 4-wave form with twice
                // the waves to hide the staging behind the MFMAs; small ones: 64x64 tiles, 4 waves
#define RBQ_RANK_KERNEL k_rank_bf16_db
#define RBQ_RANK_LDS(TM, TN, WM, WN) ((size_t)(2 * 32 * TM * WM + 2 * 32 * TN * WN) * 80 * 2) /* two slabs of 32, rows of 80 B */
#define RBQ_LAUNCH_RANKB(M, TM, TN, WM, WN)                                                                            \
    do {                                                                                                               \
        const size_t lds = RBQ_RANK_LDS(TM, TN, WM, WN);                                                               \
        static std::atomic<int> attr_dev_mask{0}; /* once per device: the call is slow and serialises launches */      \
        if (lds > 48 * 1024 && !(attr_dev_mask.load() & (1 << ix->device))) {                                          \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&RBQ_RANK_KERNEL<M, TM, TN, WM, WN>),            \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
            attr_dev_mask.fetch_or(1 << ix->device);                                                                   \
        }                                                                                                              \
        hipLaunchKernelGGL((RBQ_RANK_KERNEL<M, TM, TN, WM, WN>), grid, dim3(64 * WM * WN), lds, stream,                \
                           (const uint16_t*)w->rot_hi.p, (const uint16_t*)w->rot_lo.p, (const uint16_t*)ix->d_cent_hi, \
                           (const uint16_t*)ix->d_cent_lo, (const QueryConsts*)w->consts.p, (const float*)ix->d_cnorm2, \
                           (uint32_t)nq, nlist, D, (float*)w->scores.p);                                               \
    } while (0)
                if (ix->metric == 0) { if (big) RBQ_LAUNCH_RANKB(0, 2, 1, 2, 4); else RBQ_LAUNCH_RANKB(0, 1, 1, 2, 2); }
                else { if (big) RBQ_LAUNCH_RANKB(1, 2, 1, 2, 4); else RBQ_LAUNCH_RANKB(1, 1, 1, 2, 2); }
#undef RBQ_LAUNCH_RANKB
            } else {
#define RBQ_LAUNCH_RANK(M, TW)                                                                                         \
    hipLaunchKernelGGL((k_rank_mfma<M, TW>), grid, dim3(256), 0, stream, (const float*)w->rot.p,                       \
                       (const float*)ix->d_centroids, (const QueryConsts*)w->consts.p, (const float*)ix->d_cnorm2,   \
                       (uint32_t)nq, nlist, D, (float*)w->scores.p)
            if (ix->metric == 0) { if (big) RBQ_LAUNCH_RANK(0, 2); else RBQ_LAUNCH_RANK(0, 1); }
            else { if (big) RBQ_LAUNCH_RANK(1, 2); else RBQ_LAUNCH_RANK(1, 1); }
            }
#undef RBQ_LAUNCH_RANK
            HIP_TRY(hipGetLastError());
        }
        {
            ProfScope ps(ix, 2, stream); // shortlist + exact canonical scores + exact select
            const uint32_t cap2 = std::max<uint32_t>(64u, next_pow2(2 * nprobe));
            const int row_in_lds = (nlist > 4096 && (size_t)nlist * 4 <= 65536) ? 1 : 0; // <= 4096: registers
            size_t lds = (size_t)cap2 * 8 + (size_t)D * 4 + kThreads * 4 + (row_in_lds ? (size_t)nlist * 4 : 0);
            const int stage = lds + (size_t)nprobe * 16 <= 48 * 1024 ? 1 : 0; // per-probe geometry staged in LDS
            if (stage) lds += (size_t)nprobe * 16;
#define RBQ_LAUNCH_SELECT(RM)                                                                                         \
    do {                                                                                                               \
        if (lds > 48 * 1024)                                                                                           \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select_mfma<RM>),                             \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
        hipLaunchKernelGGL(k_select_mfma<RM>, dim3((uint32_t)nq), dim3(kThreads), lds, stream, (float*)w->scores.p,    \
                           nlist, nprobe, cap2, row_in_lds, (int)ix->metric, (const float*)w->rot.p,                   \
                           (const float*)ix->d_centroids, D, (const QueryConsts*)w->consts.p, ix->cnorm2_max,          \
                           (const uint32_t*)ix->d_list_gb0, (const uint32_t*)ix->d_list_n, (ProbeInfo*)w->probe.p,     \
                           (StreamItem*)w->wl.p, wl_stride, (uint32_t*)w->nstream.p, (unsigned long long*)w->nvec.p,   \
                           ix->profiling ? (unsigned long long*)ix->d_prof_total : nullptr,                            \
                           (unsigned int*)ix->d_fallbacks, ix->force_rank_fallback ? 1 : 0,                            \
                           (const BlockSummary*)ix->d_bsum, stage);                                                    \
    } while (0)
            if (nlist <= 4096) RBQ_LAUNCH_SELECT(2);
            else if (row_in_lds) RBQ_LAUNCH_SELECT(1);
            else RBQ_LAUNCH_SELECT(0);
#undef RBQ_LAUNCH_SELECT
            HIP_TRY(hipGetLastError());
        }
    }
    return scan_stage(ix, w, nq, nprobe, top_k, wl_stride, d_filter, filter_nbits, d_ids, d_scores, d_counts, d_diag,
                      /*mstg=*/false, stream);
    return RBQ_OK;
}

} // namespace

extern "C" {

uint32_t rbq_abi_version(void) { return (1u << 16) | 0u; }

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
    return create_impl(hdr, lists, n_devices, devices, out);
}

void rbq_index_destroy(rbq_index* ix) { free_index(ix); }

uint64_t rbq_index_len(const rbq_index* ix) { return ix ? ix->n_vectors : 0; }
uint64_t rbq_index_cluster_count(const rbq_index* ix) { return ix ? ix->n_lists : 0; }
uint32_t rbq_index_dim(const rbq_index* ix) { return ix ? ix->dim : 0; }
uint32_t rbq_index_padded_dim(const rbq_index* ix) { return ix ? ix->D : 0; }

// ---- RBQ1 v3 reader: load_from_reader, src/ivf.rs:1484-1702 --------------------------------------
namespace {
struct Reader {
    const uint8_t* p; size_t len, off = 0;
    bool take(void* dst, size_t n) { if (off + n > len || off + n < off) return false; std::memcpy(dst, p + off, n); off += n; return true; }
    const uint8_t* view(size_t n) { if (off + n > len || off + n < off) return nullptr; const uint8_t* r = p + off; off += n; return r; }
};
uint32_t crc32_ieee(const uint8_t* p, size_t n) {
    static uint32_t table[8][256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xff];
    });
    uint32_t crc = ~0u;
    while (n >= 8) {
        uint32_t a, b;
        std::memcpy(&a, p, 4); std::memcpy(&b, p + 4, 4);
        a ^= crc;
        crc = table[7][a & 0xff] ^ table[6][(a >> 8) & 0xff] ^ table[5][(a >> 16) & 0xff] ^ table[4][a >> 24] ^
              table[3][b & 0xff] ^ table[2][(b >> 8) & 0xff] ^ table[1][(b >> 16) & 0xff] ^ table[0][b >> 24];
        p += 8; n -= 8;
    }
    while (n--) crc = table[0][(crc ^ *p++) & 0xff] ^ (crc >> 8);
    return ~crc;
}
} // namespace

int rbq_index_load_rbq1(const void* bytes, size_t len, int n_devices, const int* devices, rbq_index** out) {
    g_err.clear();
    if (out) *out = nullptr;
    if (!bytes) return fail(RBQ_IO, "null buffer");
    Reader r{(const uint8_t*)bytes, len};
    auto eof = [] { return fail(RBQ_IO, "failed to fill whole buffer"); };
    char magic[4];
    if (!r.take(magic, 4)) return eof();
    if (std::memcmp(magic, "RBQ1", 4) != 0) return fail(RBQ_INVALID_PERSISTENCE, "unrecognized file header");
    uint32_t version;
    if (!r.take(&version, 4)) return eof();
    if (version != 3) return fail(RBQ_INVALID_PERSISTENCE, "unsupported index format version (expected V3 with unified memory layout)");
    rbq_header h;
    std::memset(&h, 0, sizeof h);
    uint8_t tags[4];
    if (!r.take(&h.dim, 4)) return eof();
    if (h.dim == 0) return fail(RBQ_INVALID_PERSISTENCE, "dimension must be positive");
    if (!r.take(&h.padded_dim, 4)) return eof();
    if (h.padded_dim < h.dim) return fail(RBQ_INVALID_PERSISTENCE, "padded_dim must be >= dim");
    if (!r.take(tags, 4)) return eof();
    if (tags[0] > 1) return fail(RBQ_INVALID_PERSISTENCE, "unknown metric tag");
    if (tags[1] > 1) return fail(RBQ_INVALID_PERSISTENCE, "unknown rotator type tag");
    if (tags[2] > 16) return fail(RBQ_INVALID_PERSISTENCE, "ex_bits out of range");
    if (tags[3] == 0 || tags[3] > 16) return fail(RBQ_INVALID_PERSISTENCE, "total_bits out of range");
    if ((uint8_t)(tags[3] - 1) != tags[2]) return fail(RBQ_INVALID_PERSISTENCE, "total_bits does not match ex_bits");
    h.metric = tags[0]; h.rotator = tags[1]; h.ex_bits = tags[2];
    uint64_t expected_vectors, cluster_count, rot_len;
    if (!r.take(&expected_vectors, 8) || !r.take(&cluster_count, 8) || !r.take(&rot_len, 8)) return eof();
    const uint8_t* blob = r.view(rot_len);
    if (!blob) return eof();
    h.rotator_blob = blob; h.rotator_len = rot_len; h.n_lists = cluster_count; h.n_vectors = expected_vectors;
    { // DynamicRotator::deserialize length checks (src/rotation.rs:213-219,491-497)
        const uint64_t want = h.rotator == RBQ_ROTATOR_FHT_KAC ? (uint64_t)4 * h.padded_dim / 8 : (uint64_t)h.padded_dim * h.padded_dim * 4;
        if (rot_len != want)
            return fail(RBQ_INVALID_PERSISTENCE, h.rotator == RBQ_ROTATOR_FHT_KAC ? "FHT rotator flip bits length mismatch" : "rotator matrix length mismatch");
    }
    if (cluster_count > (len / 8)) return eof(); // every cluster costs >= 8 bytes; guards the allocation below
    const size_t D = h.padded_dim, stride = D * 4 + 384;
    const size_t exb_expected = h.ex_bits ? D * h.ex_bits / 8 : 0;
    std::vector<rbq_list_view> lists(cluster_count);
    std::vector<std::vector<uint8_t>> ex_flat(cluster_count);
    std::vector<std::vector<float>> fl(cluster_count); // centroid | f_add_ex | f_rescale_ex (alignment-safe copies)
    std::vector<std::vector<uint64_t>> idv(cluster_count);
    uint64_t actual = 0;
    for (uint64_t c = 0; c < cluster_count; ++c) {
        rbq_list_view& L = lists[c];
        const uint8_t* cen = r.view(D * 4);
        if (!cen) return eof();
        uint64_t n;
        if (!r.take(&n, 8)) return eof();
        if (n > 1000000) return fail(RBQ_INVALID_PERSISTENCE, "cluster size exceeds reasonable limits - possible corruption");
        const uint8_t* idp = r.view(n * 8);
        if (!idp) return eof();
        uint64_t blen;
        if (!r.take(&blen, 8)) return eof();
        if (blen != ((n + 31) / 32) * stride)
            return fail(RBQ_INVALID_PERSISTENCE, "batch_data length mismatch - possible corruption or version incompatibility");
        const uint8_t* bd = r.view(blen);
        if (!bd) return eof();
        ex_flat[c].resize(n * exb_expected);
        for (uint64_t v = 0; v < n; ++v) {
            uint64_t el;
            if (!r.take(&el, 8)) return eof();
            if (el != exb_expected)
                return fail(RBQ_INVALID_PERSISTENCE, "ex_code_packed length mismatch - possible corruption or version incompatibility");
            if (el && !r.take(&ex_flat[c][v * exb_expected], el)) return eof();
        }
        fl[c].resize(D + 2 * n);
        std::memcpy(fl[c].data(), cen, D * 4);
        if (!r.take(fl[c].data() + D, n * 4) || !r.take(fl[c].data() + D + n, n * 4)) return eof();
        if (!r.view(n * 4) || !r.view(n * 4)) return eof(); // delta, vl: reconstruction only
        idv[c].resize(n);
        std::memcpy(idv[c].data(), idp, n * 8);
        L.centroid = fl[c].data(); L.n = n; L.ids = idv[c].data(); L.batch_data = bd; L.batch_len = blen;
        L.ex_codes = ex_flat[c].empty() ? nullptr : ex_flat[c].data();
        L.f_add_ex = fl[c].data() + D; L.f_rescale_ex = fl[c].data() + D + n;
        actual += n;
    }
    if (actual != expected_vectors) return fail(RBQ_INVALID_PERSISTENCE, "vector count metadata mismatch");
    const size_t body_end = r.off;
    uint32_t stored;
    if (!r.take(&stored, 4)) return eof();
    if (crc32_ieee((const uint8_t*)bytes + 8, body_end - 8) != stored) return fail(RBQ_INVALID_PERSISTENCE, "checksum mismatch");
    // batch_data records are only byte-addressed by relayout_block (memcpy for the factor rows)
    return create_impl(&h, lists.data(), n_devices, devices, out);
}

// ---- search ---------------------------------------------------------------------------------------
static int check_query_args(const rbq_index* ix, uint32_t query_dim) {
    if (!ix) return fail(RBQ_INVALID_CONFIG, "null index");
    if (ix->n_vectors == 0) return fail(RBQ_EMPTY_INDEX, "index is empty");
    if (query_dim != ix->dim) {
        char b[96];
        std::snprintf(b, sizeof b, "expected %u, got %u", ix->dim, query_dim);
        return fail(RBQ_DIMENSION_MISMATCH, b);
    }
    return RBQ_OK;
}

int rbq_search_batch_device(const rbq_index* cix, const float* d_queries, uint64_t nq, uint32_t query_dim, uint32_t top_k,
                            uint32_t nprobe, const uint32_t* d_filter_words, uint64_t filter_nbits, uint64_t* d_out_ids,
                            float* d_out_scores, uint32_t* d_out_counts, rbq_diag* d_diag, void* hip_stream) {
    g_err.clear();
    rbq_index* ix = const_cast<rbq_index*>(cix);
    int rc = check_query_args(ix, query_dim);
    if (rc) return rc;
    if (nq == 0) return RBQ_OK;
    if (nq > 0x7fffffffull) return fail(RBQ_INVALID_CONFIG, "batch too large");
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = (hipStream_t)hip_stream;
    if (top_k == 0) { // Ok(vec![]) for every query, src/ivf.rs:1792-1794
        HIP_TRY(hipMemsetAsync(d_out_counts, 0, nq * 4, s));
        return RBQ_OK;
    }
    // One workspace per caller stream, never handed to anyone else: successive calls on the same stream are
    // stream-ordered, so their kernels may share the scratch buffers without any host synchronisation.
    Workspace* w;
    {
        std::lock_guard<std::mutex> g(ix->mu);
        Workspace*& slot = ix->stream_ws[s];
        if (!slot) slot = new Workspace();
        w = slot;
    }
    return search_device(ix, w, d_queries, nq, top_k, nprobe, d_filter_words, filter_nbits, d_out_ids, d_out_scores,
                         d_out_counts, d_diag, s);
}

int rbq_search_batch(const rbq_index* cix, const float* queries, uint64_t nq, uint32_t query_dim, uint32_t top_k,
                     uint32_t nprobe, const uint32_t* filter_words, uint64_t filter_nbits, uint64_t* out_ids,
                     float* out_scores, uint32_t* out_counts, rbq_diag* diag) {
    g_err.clear();
    rbq_index* ix = const_cast<rbq_index*>(cix);
    int rc = check_query_args(ix, query_dim);
    if (rc) return rc;
    if (nq == 0) return RBQ_OK;
    if (top_k == 0) {
        if (out_counts) std::memset(out_counts, 0, nq * 4);
        if (diag) std::memset(diag, 0, nq * sizeof(rbq_diag));
        return RBQ_OK;
    }
    if (!queries || !out_ids || !out_scores || !out_counts) return fail(RBQ_INVALID_CONFIG, "null buffer");
    HIP_TRY(hipSetDevice(ix->device));
    Workspace* w = take_ws(ix);
    if (!w) return fail(RBQ_DEVICE, "cannot create workspace stream");
    auto run = [&]() -> int {
        int r2;
        const uint64_t CH = 16384; // queries per device pass (bounds the nq x nlist score matrix)
        if (filter_words) {
            const size_t fb = (size_t)((filter_nbits + 31) / 32) * 4;
            if ((r2 = w->filter.ensure(fb ? fb : 4))) return r2;
            if (fb) HIP_TRY(hipMemcpyAsync(w->filter.p, filter_words, fb, hipMemcpyHostToDevice, w->stream));
        }
        for (uint64_t q0 = 0; q0 < nq; q0 += CH) {
            const uint64_t n = std::min(CH, nq - q0);
            if ((r2 = w->queries.ensure(n * query_dim * 4))) return r2;
            if ((r2 = w->out_ids.ensure(n * top_k * 8))) return r2;
            if ((r2 = w->out_scores.ensure(n * top_k * 4))) return r2;
            if ((r2 = w->out_counts.ensure(n * 4))) return r2;
            if (diag && (r2 = w->diag.ensure(n * sizeof(rbq_diag)))) return r2;
            HIP_TRY(hipMemcpyAsync(w->queries.p, queries + q0 * query_dim, n * query_dim * 4, hipMemcpyHostToDevice, w->stream));
            r2 = search_device(ix, w, (const float*)w->queries.p, n, top_k, nprobe,
                               filter_words ? (const uint32_t*)w->filter.p : nullptr, filter_nbits, (uint64_t*)w->out_ids.p,
                               (float*)w->out_scores.p, (uint32_t*)w->out_counts.p, diag ? (rbq_diag*)w->diag.p : nullptr,
                               w->stream);
            if (r2) return r2;
            HIP_TRY(hipMemcpyAsync(out_ids + q0 * top_k, w->out_ids.p, n * top_k * 8, hipMemcpyDeviceToHost, w->stream));
            HIP_TRY(hipMemcpyAsync(out_scores + q0 * top_k, w->out_scores.p, n * top_k * 4, hipMemcpyDeviceToHost, w->stream));
            HIP_TRY(hipMemcpyAsync(out_counts + q0, w->out_counts.p, n * 4, hipMemcpyDeviceToHost, w->stream));
            if (diag) HIP_TRY(hipMemcpyAsync(diag + q0, w->diag.p, n * sizeof(rbq_diag), hipMemcpyDeviceToHost, w->stream));
            HIP_TRY(hipStreamSynchronize(w->stream));
        }
        return RBQ_OK;
    };
    rc = run();
    if (rc) (void)hipStreamSynchronize(w->stream);
    give_ws(ix, w);
    return rc;
}


// ---- MSTG posting-list scan ------------------------------------------------------------
int rbq_posting_scan_batch(const rbq_index* cix, const float* queries, uint64_t nq, uint32_t query_dim, uint32_t top_k,
                           const uint32_t* list_ids, const uint32_t* list_counts, uint32_t max_lists,
                           uint64_t* out_ids, float* out_scores, uint32_t* out_counts) {
    g_err.clear();
    rbq_index* ix = const_cast<rbq_index*>(cix);
    int rc = check_query_args(ix, query_dim);
    if (rc) return rc;
    if (ix->rotator != RBQ_ROTATOR_NONE) return fail(RBQ_INVALID_CONFIG, "posting-list scan needs an index created with rotator NONE");
    if (nq == 0) return RBQ_OK;
    if (!queries || !list_ids || !list_counts || !out_ids || !out_scores || !out_counts) return fail(RBQ_INVALID_CONFIG, "null buffer");
    if (top_k == 0) { std::memset(out_counts, 0, nq * 4); return RBQ_OK; }
    if (top_k > 4096) return fail(RBQ_INVALID_CONFIG, "top_k > 4096 is not supported by the GPU top-k stage");
    if (max_lists == 0) {
        std::memset(out_counts, 0, nq * 4);
        for (uint64_t i = 0; i < nq * top_k; ++i) { out_ids[i] = ~0ull; out_scores[i] = NAN; }
        return RBQ_OK;
    }
    if (max_lists > (1u << 26)) return fail(RBQ_INVALID_CONFIG, "too many lists per query");
    HIP_TRY(hipSetDevice(ix->device));
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
            if ((r2 = w->queries.ensure(n * query_dim * 4))) return r2;
            if ((r2 = w->rot.ensure(n * D * 4))) return r2;
            if ((r2 = w->lut.ensure(n * (size_t)Dc * 4))) return r2;
            if ((r2 = w->consts.ensure(n * sizeof(QueryConsts)))) return r2;
            if ((r2 = d_lists.ensure(n * (size_t)max_lists * 4))) return r2;
            if ((r2 = d_cnts.ensure(n * 8))) return r2;
            if ((r2 = w->probe.ensure(n * (size_t)max_lists * sizeof(ProbeInfo)))) return r2;
            if ((r2 = w->wl.ensure(n * wl_stride * sizeof(StreamItem)))) return r2;
            if ((r2 = w->nstream.ensure(n * 4))) return r2;
            if ((r2 = w->out_ids.ensure(n * top_k * 8))) return r2;
            if ((r2 = w->out_scores.ensure(n * top_k * 4))) return r2;
            if ((r2 = w->out_counts.ensure(n * 4))) return r2;
            hipStream_t st = w->stream;
            HIP_TRY(hipMemcpyAsync(w->queries.p, queries + q0 * query_dim, n * query_dim * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d_lists.p, list_ids + q0 * max_lists, n * (size_t)max_lists * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d_cnts.p, list_counts + q0, n * 4, hipMemcpyHostToDevice, st));
            {
                ProfScope ps(ix, 0, st);
                hipLaunchKernelGGL(k_prep_wave, dim3((uint32_t)((n + 3) / 4)), dim3(kThreads), (size_t)D * 4 * 2 * 4 + D / 2, st,
                                   (const float*)w->queries.p, (uint32_t)n, ix->dim, D, Dc, (int)ix->rotator,
                                   (const uint8_t*)ix->d_rot_blob, ix->trunc, ix->fac, 0u, (float*)w->rot.p, (uint8_t*)w->lut.p,
                                   (QueryConsts*)w->consts.p, (uint16_t*)nullptr, (uint16_t*)nullptr);
                HIP_TRY(hipGetLastError());
            }
            {
                ProfScope ps(ix, 2, st);
                hipLaunchKernelGGL(k_probes_given, dim3((uint32_t)n), dim3(kThreads), (size_t)D * 4 + kThreads * 4, st,
                                   (const uint32_t*)d_lists.p, (const uint32_t*)d_cnts.p, max_lists, (uint32_t)ix->n_lists,
                                   (int)ix->metric, (const float*)w->rot.p, (const float*)ix->d_centroids, D,
                                   (const uint32_t*)ix->d_list_gb0, (const uint32_t*)ix->d_list_n, (ProbeInfo*)w->probe.p,
                                   (StreamItem*)w->wl.p, wl_stride, (uint32_t*)w->nstream.p, (const QueryConsts*)w->consts.p,
                                   (const BlockSummary*)ix->d_bsum);
                HIP_TRY(hipGetLastError());
            }
            if ((r2 = scan_stage(ix, w, n, max_lists, top_k, wl_stride, nullptr, 0, (uint64_t*)w->out_ids.p,
                                 (float*)w->out_scores.p, (uint32_t*)w->out_counts.p, nullptr, /*mstg=*/true, st)))
                return r2;
            HIP_TRY(hipMemcpyAsync(out_ids + q0 * top_k, w->out_ids.p, n * top_k * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out_scores + q0 * top_k, w->out_scores.p, n * top_k * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out_counts + q0, w->out_counts.p, n * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        return RBQ_OK;
    };
    rc = run();
    if (rc) (void)hipStreamSynchronize(w->stream);
    give_ws(ix, w);
    return rc;
}

// ---- profiling taps ---------------------------------------------------------------------------------
void rbq_profile_begin(rbq_index* ix) {
    if (!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    for (auto& sp : ix->prof) {
        for (auto& e : sp.ev) ix->ev_pool.give(e);
        sp.ev.clear(); sp.ms = 0; sp.launches = 0;
    }
    ix->prof_scan_bytes = 0;
    (void)hipSetDevice(ix->device);
    (void)hipMemset(ix->d_prof_total, 0, 8);
    ix->profiling = true;
}
void rbq_profile_end(rbq_index* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> g(ix->mu);
    ix->profiling = false;
    {
        unsigned long long tot = 0;
        (void)hipMemcpy(&tot, ix->d_prof_total, 8, hipMemcpyDeviceToHost);
        ix->prof_scan_bytes = tot * (uint64_t)(ix->D / 8 + 12); // sum_q sum_{c in probe(q)} n_c * (D/8 + 12)
    }
    for (auto& sp : ix->prof) {
        for (auto& e : sp.ev) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { sp.ms += ms; sp.launches++; }
            ix->ev_pool.give(e);
        }
        sp.ev.clear();
    }
}
double rbq_profile_stage_ms(const rbq_index* ix, const char* stage, uint64_t* launches) {
    if (!ix || !stage) return -1;
    int s = !std::strcmp(stage, "prep") ? 0 : !std::strcmp(stage, "rank") ? 1 : !std::strcmp(stage, "select") ? 2 : !std::strcmp(stage, "scan") ? 3 : -1;
    if (s < 0) return -1;
    if (launches) *launches = ix->prof[s].launches;
    return ix->prof[s].launches ? ix->prof[s].ms / (double)ix->prof[s].launches : 0.0;
}
uint64_t rbq_profile_scan_bytes(const rbq_index* ix) { return ix ? ix->prof_scan_bytes : 0; }
void rbq_profile_select_stages(rbq_index* ix, uint32_t mask) { if (ix) ix->prof_mask = mask & 0xfu; }
void rbq_profile_set_sampling(rbq_index* ix, uint32_t every) { if (ix) ix->prof_every = every ? every : 1u; }
int rbq_debug_set_option(rbq_index* ix, const char* name, int value) {
    if (!ix || !name) return RBQ_INVALID_CONFIG;
    if (!std::strcmp(name, "block_bound")) { ix->no_block_bound = value == 0; return RBQ_OK; }
    if (!std::strcmp(name, "exact_rank")) { ix->exact_rank = value != 0; return RBQ_OK; }
    if (!std::strcmp(name, "exact_heap")) { ix->exact_heap = value != 0; return RBQ_OK; }
    if (!std::strcmp(name, "f32_rank")) { ix->f32_rank = value != 0; return RBQ_OK; }
    if (!std::strcmp(name, "wg_prep")) { ix->wg_prep = value != 0; return RBQ_OK; }
    if (!std::strcmp(name, "small_rank_tiles")) { ix->small_rank_tiles = value != 0; return RBQ_OK; }
    if (!std::strcmp(name, "force_rank_fallback")) { ix->force_rank_fallback = value != 0; return RBQ_OK; }
    return fail(RBQ_INVALID_CONFIG, std::string("unknown option ") + name);
}
uint64_t rbq_debug_rank_fallbacks(const rbq_index* ix) {
    if (!ix) return 0;
    unsigned int v = 0;
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&v, ix->d_fallbacks, 4, hipMemcpyDeviceToHost);
    return v;
}

/* Diagnostic: copy an intermediate buffer of the workspace bound to `hip_stream` (after the caller synchronised). */
int rbq_debug_copy_workspace(rbq_index* ix, void* hip_stream, const char* name, void* dst, uint64_t bytes) {
    if (!ix || !name || !dst) return RBQ_INVALID_CONFIG;
    Workspace* w = nullptr;
    {
        std::lock_guard<std::mutex> g(ix->mu);
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
    if (!b || !b->p || bytes > b->cap) return fail(RBQ_INVALID_CONFIG, "unknown buffer or size");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipMemcpy(dst, b->p, bytes, hipMemcpyDeviceToHost));
    return RBQ_OK;
}

int rbq_index_build_device(const rbq_header* hdr, const float* centroids, const float* d_data, const uint32_t* d_assign,
                           uint64_t n, float t_const, int device, rbq_index** out) {
    g_err.clear();
    return build_device_impl(hdr, centroids, d_data, d_assign, n, t_const, device, out);
}

/* Diagnostic: copy one of the index's device arrays to the host. */
int rbq_debug_copy_index(rbq_index* ix, const char* name, void* dst, uint64_t bytes) {
    if (!ix || !name || !dst) return RBQ_INVALID_CONFIG;
    const size_t stride = (size_t)ix->Dc * 4 + 384, exd = ex_bytes_dev(ix->D, ix->ex_bits), slots = ix->n_blocks * 32;
    const void* p = nullptr;
    size_t have = 0;
    if (!std::strcmp(name, "blocks")) { p = ix->d_blocks; have = ix->n_blocks * stride; }
    else if (!std::strcmp(name, "ids")) { p = ix->d_ids; have = slots * 8; }
    else if (!std::strcmp(name, "ex")) { p = ix->d_ex; have = slots * exd; }
    else if (!std::strcmp(name, "fadd_ex")) { p = ix->d_fadd_ex; have = ix->ex_bits ? slots * 4 : 0; }
    else if (!std::strcmp(name, "fres_ex")) { p = ix->d_fres_ex; have = ix->ex_bits ? slots * 4 : 0; }
    else if (!std::strcmp(name, "bsum")) { p = ix->d_bsum; have = ix->n_blocks * sizeof(BlockSummary); }
    else if (!std::strcmp(name, "centroids")) { p = ix->d_centroids; have = ix->n_lists * ix->D * 4; }
    else if (!std::strcmp(name, "list_gb0")) { p = ix->d_list_gb0; have = ix->n_lists * 4; }
    else if (!std::strcmp(name, "list_n")) { p = ix->d_list_n; have = ix->n_lists * 4; }
    if (!p || bytes != have) return fail(RBQ_INVALID_CONFIG, "unknown array or size (have " + std::to_string(have) + " bytes)");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
    return RBQ_OK;
}

uint64_t rbq_debug_heap_restarts(const rbq_index* ix) {
    if (!ix) return 0;
    unsigned int v = 0;
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&v, (const unsigned int*)ix->d_fallbacks + 1, 4, hipMemcpyDeviceToHost);
    return v;
}

} // extern "C"
