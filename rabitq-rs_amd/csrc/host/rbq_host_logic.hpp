// rbq_host_logic.hpp — the host-side logic of the C ABI that touches no GPU: the RBQ1-v3 stream parser / validator
// (load_from_reader, reference src/ivf.rs:1484-1702), CRC-32/IEEE, and the sub-batch / shard / result-packing arithmetic
// of rbq_search_batch.  Pure C++ (no HIP): librbq.so includes it, and tests/test_sanitizers.py builds the same code with
// -fsanitize=address,undefined (csrc/host/rbq_hostcheck.cpp) and fuzzes it on the CPU.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "rbq.h"

namespace rbq_host {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// One ClusterData as byte ranges (an RBQ1 stream has no alignment; rbq_list_view arrays are viewed the same way).
struct ListSrc {
    const uint8_t* centroid = nullptr; // D f32
    uint64_t n = 0;
    const uint8_t* ids = nullptr;        // n u64
    const uint8_t* batch_data = nullptr; // ceil(n/32) records of D*4 + 384 bytes
    const uint8_t* ex = nullptr;         // n packed ex codes, `ex_stride` bytes apart
    size_t ex_stride = 0;
    const uint8_t* fadd = nullptr;       // n f32
    const uint8_t* fres = nullptr;       // n f32
};

struct Reader {
    const uint8_t* p; size_t len, off = 0;
    bool take(void* dst, size_t n) { if (off + n > len || off + n < off) return false; std::memcpy(dst, p + off, n); off += n; return true; }
    const uint8_t* view(size_t n) { if (off + n > len || off + n < off) return nullptr; const uint8_t* r = p + off; off += n; return r; }
};

inline uint32_t crc32_ieee(const uint8_t* p, size_t n) {
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

// Parses and validates an RBQ1 v3 stream in place (the lists are byte ranges of `bytes`: nothing is copied).  Returns
// RBQ_OK or the error code with the reference's message in *detail — the checks, their order and their strings are
// load_from_reader's (src/ivf.rs:1484-1702; DynamicRotator::deserialize src/rotation.rs:213-219,491-497).  The header's
// rotator_blob points into `bytes`.  What THIS build cannot serve (ex_bits outside {0,2,6}, padded_dim > 2048 ...) is
// left to the caller's validate_header.
inline int rbq1_parse(const void* bytes, size_t len, rbq_header* hout, std::vector<ListSrc>* lists_out, std::string* detail) {
    auto fail = [&](int code, const char* msg) { if (detail) *detail = msg; return code; };
    auto eof = [&] { return fail(RBQ_IO, "failed to fill whole buffer"); };
    if (!bytes) return fail(RBQ_IO, "null buffer");
    Reader r{(const uint8_t*)bytes, len};
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
    { // DynamicRotator::deserialize length checks
        const uint64_t want = h.rotator == RBQ_ROTATOR_FHT_KAC ? (uint64_t)4 * h.padded_dim / 8 : (uint64_t)h.padded_dim * h.padded_dim * 4;
        if (rot_len != want)
            return fail(RBQ_INVALID_PERSISTENCE, h.rotator == RBQ_ROTATOR_FHT_KAC ? "FHT rotator flip bits length mismatch" : "rotator matrix length mismatch");
    }
    if (cluster_count > (len / 8)) return eof(); // every cluster costs >= 8 bytes; guards the allocation below
    const size_t D = h.padded_dim, stride = D * 4 + 384;
    const size_t exb_expected = h.ex_bits ? D * h.ex_bits / 8 : 0;
    std::vector<ListSrc> lists(cluster_count);
    uint64_t actual = 0;
    for (uint64_t c = 0; c < cluster_count; ++c) {
        ListSrc& L = lists[c];
        L.centroid = r.view(D * 4);
        if (!L.centroid) return eof();
        uint64_t n;
        if (!r.take(&n, 8)) return eof();
        if (n > 1000000) return fail(RBQ_INVALID_PERSISTENCE, "cluster size exceeds reasonable limits - possible corruption");
        L.n = n;
        L.ids = r.view(n * 8);
        if (!L.ids) return eof();
        uint64_t blen;
        if (!r.take(&blen, 8)) return eof();
        if (blen != ((n + 31) / 32) * stride)
            return fail(RBQ_INVALID_PERSISTENCE, "batch_data length mismatch - possible corruption or version incompatibility");
        L.batch_data = r.view(blen);
        if (!L.batch_data) return eof();
        L.ex_stride = exb_expected + 8; // every packed code carries a u64 length prefix
        for (uint64_t v = 0; v < n; ++v) {
            uint64_t el;
            if (!r.take(&el, 8)) return eof();
            if (el != exb_expected)
                return fail(RBQ_INVALID_PERSISTENCE, "ex_code_packed length mismatch - possible corruption or version incompatibility");
            const uint8_t* e = r.view(el);
            if (!e) return eof();
            if (v == 0) L.ex = e;
        }
        L.fadd = r.view(n * 4);
        L.fres = r.view(n * 4);
        if (!L.fadd || !L.fres) return eof();
        if (!r.view(n * 4) || !r.view(n * 4)) return eof(); // delta, vl: reconstruction only
        actual += n;
    }
    if (actual != expected_vectors) return fail(RBQ_INVALID_PERSISTENCE, "vector count metadata mismatch");
    const size_t body_end = r.off;
    uint32_t stored;
    if (!r.take(&stored, 4)) return eof();
    if (crc32_ieee((const uint8_t*)bytes + 8, body_end - 8) != stored) return fail(RBQ_INVALID_PERSISTENCE, "checksum mismatch");
    *hout = h;
    *lists_out = std::move(lists);
    return RBQ_OK;
}

// layout of one sub-batch's results in the packed device / pinned buffers
struct OutPack {
    size_t o_ids, o_scores, o_counts, o_diag, total;
    OutPack(uint64_t n, uint32_t top_k, bool diag) {
        o_ids = 0; o_scores = align_up(n * top_k * 8, 16); o_counts = o_scores + align_up(n * top_k * 4, 16);
        o_diag = o_counts + align_up(n * 4, 16); total = o_diag + (diag ? n * sizeof(rbq_diag) : 0);
    }
};

// replica r of R serves the contiguous query shard [q0, q1) of a batch of nq (batch_search is a par_iter over queries)
inline void shard_range(uint64_t r, uint64_t R, uint64_t nq, uint64_t* q0, uint64_t* q1) { *q0 = r * nq / R; *q1 = (r + 1) * nq / R; }

// Sub-batches of one rbq_search_batch call (or replica shard) of nq queries, as (first query, count) in order.
//   nq < 256            one sub-batch
//   nq <= 2048          four sub-batches in the proportion 4:3:2:1 (multiples of 32 queries) over four lanes: the large ones
//                       go first and run under the preparation of the later ones, the last one — whose kernel chain the caller
//                       waits for — is small.  (Round 4, GIST-1M shape, page-locked buffers: 2048 queries per call 532 -> 462 us
//                       against equal quarters, 562 against halves; 1024 per call: 296 either way — such a call is the time the
//                       GPU needs for 1024 queries at its pipelined rate plus the latency of one four-kernel chain.)
//   above               sub-batches of 1024 (which also bounds the nq x n_lists score matrix per lane); a tapered tail was
//                       measured slower there (4096 per call: 828 against 688 us — the chip is saturated, smaller kernels only
//                       add launches)
// `forced` (diagnostic option host_subbatch) gives equal sub-batches of that size.
inline std::vector<std::pair<uint64_t, uint64_t>> subbatch_plan(uint64_t nq, uint64_t forced, int taper_kind = 0) {
    std::vector<std::pair<uint64_t, uint64_t>> plan;
    if (nq == 0) return plan;
    if (forced) {
        for (uint64_t q0 = 0; q0 < nq; q0 += forced) plan.emplace_back(q0, std::min<uint64_t>(forced, nq - q0));
        return plan;
    }
    auto taper = [&](uint64_t q0, uint64_t n, std::initializer_list<uint64_t> weights) {
        uint64_t wsum = 0;
        for (uint64_t w : weights) wsum += w;
        uint64_t left = n, pos = q0;
        size_t i = 0;
        for (uint64_t w : weights) {
            ++i;
            uint64_t take = i == weights.size() ? left : std::min<uint64_t>(left, align_up((size_t)(n * w / wsum), 32));
            if (take) plan.emplace_back(pos, take);
            pos += take; left -= take;
        }
    };
    if (nq < 256) { plan.emplace_back(0, nq); return plan; }
    if (nq <= 2048) { // (taper_kind: option host_taper — A/B of the weights)
        if (taper_kind == 1) taper(0, nq, {1, 1, 1, 1});
        else if (taper_kind == 2) taper(0, nq, {3, 3, 2, 2});
        else if (taper_kind == 3) taper(0, nq, {4, 4, 3, 2});
        else if (taper_kind == 4) taper(0, nq, {2, 2, 2, 1});
        else taper(0, nq, {4, 3, 2, 1});
        return plan;
    }
    for (uint64_t q0 = 0; q0 < nq; q0 += 1024) plan.emplace_back(q0, std::min<uint64_t>(1024, nq - q0));
    return plan;
}

} // namespace rbq_host
