// rbq_build.cpp — CPU index builder (train-time harness), C ABI `rbq_build_*`.
//
// Training stays on the CPU by design (BASELINE.json north_star): in a real
// deployment the Rust crate trains and hands its `ClusterData` arrays (or an
// RBQ1-v3 file) to rbq_index_create / rbq_index_load_rbq1.  There is no Rust
// toolchain in this pipeline, so this file restates the train-time side in C++
// purely so tests and bench.py can produce indexes with the reference's exact
// byte layout.  It is NOT on the measured query path and never runs on the GPU.
//
// Follows (reference lqhl/rabitq-rs v0.7.0):
//   IvfRabitqIndex::train_with_clusters / build_from_rotated  src/ivf.rs:1025-1215
//   ClusterData::from_quantized_vectors / pack_batch_*         src/ivf.rs:409-696
//   quantize_with_centroid and helpers                         src/quantizer.rs:140-592
//   pack_binary_code / pack_codes / *_cpp_compat packers       src/simd.rs:141-150,864-904,2406-2695
//   FhtKacRotator / MatrixRotator ctor + rotate                src/rotation.rs:81-173,248-401
//   save_to_writer (RBQ1 v3 + CRC32)                           src/ivf.rs:1317-1474
// RNG streams (flip bits, t_const samples, Gram-Schmidt) cannot reproduce the
// reference's ChaCha12 `StdRng`; their outputs are carried inside the index, so
// query parity does not depend on them.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <queue>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../../include/rbq.h"
#include "rbq_build.h"

namespace {

// ---------------------------------------------------------------- RNG
struct Rng {
    uint64_t s[4];
    explicit Rng(uint64_t seed) {
        uint64_t z = seed;
        for (int i = 0; i < 4; ++i) {
            z += 0x9e3779b97f4a7c15ULL;
            uint64_t x = z;
            x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
            x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
            s[i] = x ^ (x >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    bool have = false; double spare = 0;
    double normal() {
        if (have) { have = false; return spare; }
        double u, v, r;
        do { u = 2 * uniform() - 1; v = 2 * uniform() - 1; r = u * u + v * v; } while (r >= 1 || r == 0);
        double f = std::sqrt(-2 * std::log(r) / r);
        spare = v * f; have = true;
        return u * f;
    }
};

// ---------------------------------------------------------------- math.rs (AVX2 lane order)
float dot8(const float* a, const float* b, size_t len) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t chunks = len / 8, i = 0;
    for (; i < chunks * 8; i += 8)
        for (int l = 0; l < 8; ++l) { float p = a[i + l] * b[i + l]; acc[l] = acc[l] + p; }
    float sum = 0.0f;
    if (chunks) { sum = -0.0f; for (int l = 0; l < 8; ++l) sum = sum + acc[l]; }
    for (; i < len; ++i) { float p = a[i] * b[i]; sum = sum + p; }
    return sum;
}
inline float l2_norm_sqr(const float* v, size_t n) { return dot8(v, v, n); }

// ---------------------------------------------------------------- rotation.rs
uint32_t floor_log2(uint64_t x) { uint32_t r = 0; while (x >>= 1) ++r; return r; }

void fht(float* d, size_t n) {
    for (size_t h = 1; h < n; h *= 2)
        for (size_t i = 0; i < n; i += 2 * h)
            for (size_t j = i; j < i + h; ++j) { float x = d[j], y = d[j + h]; d[j] = x + y; d[j + h] = x - y; }
}
void flip_sign(float* d, size_t n, const uint8_t* f) {
    for (size_t i = 0; i < n; ++i) if ((f[i / 8] >> (i % 8)) & 1) d[i] = -d[i];
}
void kacs(float* d, size_t n) {
    size_t half = n / 2;
    for (size_t i = 0; i < half; ++i) { float x = d[i], y = d[i + half]; d[i] = x + y; d[i + half] = x - y; }
}
void rescale(float* d, size_t n, float f) { for (size_t i = 0; i < n; ++i) d[i] *= f; }

void fht_kac_rotate(uint32_t dim, uint32_t D, const uint8_t* flip, const float* in, float* out) {
    std::memcpy(out, in, sizeof(float) * dim);
    for (uint32_t i = dim; i < D; ++i) out[i] = 0.0f;
    size_t fo = D / 8;
    uint32_t trunc = 1u << floor_log2(dim);
    float fac = 1.0f / std::sqrt((float)trunc);
    if (trunc == D) {
        for (int r = 0; r < 4; ++r) { flip_sign(out, D, flip + r * fo); fht(out, D); rescale(out, D, fac); }
    } else {
        uint32_t start = D - trunc;
        for (int r = 0; r < 4; ++r) {
            flip_sign(out, D, flip + r * fo);
            float* part = (r % 2 == 0) ? out : out + start;
            fht(part, trunc); rescale(part, trunc, fac); kacs(out, D);
        }
        rescale(out, D, 0.25f);
    }
}

void matrix_rotate(uint32_t dim, uint32_t D, const float* m, const float* in, float* out) {
    for (uint32_t r = 0; r < D; ++r) {
        const float* row = m + (size_t)r * D;
        float acc = 0.0f;
        for (uint32_t c = 0; c < D; ++c) { float v = c < dim ? in[c] : 0.0f; float p = v * row[c]; acc = acc + p; }
        out[r] = acc;
    }
}

// MatrixRotator::with_rng, src/rotation.rs:87-139 (Gram-Schmidt over Gaussian rows)
std::vector<float> make_matrix_rotator(uint32_t D, uint64_t seed) {
    Rng rng(seed);
    std::vector<float> m((size_t)D * D);
    for (uint32_t r = 0; r < D; ++r) {
        float* v = &m[(size_t)r * D];
        for (int attempt = 0;; ++attempt) {
            for (uint32_t c = 0; c < D; ++c) v[c] = (float)rng.normal();
            for (uint32_t p = 0; p < r; ++p) {
                const float* pv = &m[(size_t)p * D];
                float proj = dot8(v, pv, D);
                for (uint32_t c = 0; c < D; ++c) v[c] -= proj * pv[c];
            }
            float norm = std::sqrt(dot8(v, v, D));
            if (norm > std::numeric_limits<float>::epsilon()) { for (uint32_t c = 0; c < D; ++c) v[c] /= norm; break; }
            if (attempt > 8) { std::fill(v, v + D, 0.0f); v[attempt % D] = 1.0f; break; }
        }
    }
    return m;
}

// ---------------------------------------------------------------- quantizer.rs
const double K_TIGHT_START[9] = {0.0, 0.15, 0.20, 0.52, 0.59, 0.71, 0.75, 0.77, 0.81};
const double K_EPS = 1e-5, K_NENUM = 10.0;
const float K_CONST_EPSILON = 1.9f;
const float F32_EPS = std::numeric_limits<float>::epsilon();

// best_rescale_factor, src/quantizer.rs:337-427
double best_rescale_factor(const float* o_abs, size_t dim, uint32_t ex_bits) {
    float mx = 0.0f;
    for (size_t i = 0; i < dim; ++i) mx = std::max(mx, o_abs[i]);
    double max_o = mx;
    if (max_o <= std::numeric_limits<double>::epsilon()) return 1.0;
    size_t ti = std::min<size_t>(ex_bits, 8);
    double t_end = ((double)((1 << ex_bits) - 1) + K_NENUM) / max_o;
    double t_start = t_end * K_TIGHT_START[ti];
    std::vector<int32_t> cur(dim);
    double sqr_den = (double)dim * 0.25, num = 0.0;
    for (size_t i = 0; i < dim; ++i) {
        int32_t c = (int32_t)(t_start * (double)o_abs[i] + K_EPS);
        cur[i] = c;
        sqr_den += (double)(c * c + c);
        num += ((double)c + 0.5) * (double)o_abs[i];
    }
    typedef std::pair<double, size_t> ent; // min-heap on (t, idx); t > 0 so < == total_cmp
    std::priority_queue<ent, std::vector<ent>, std::greater<ent>> heap;
    for (size_t i = 0; i < dim; ++i)
        if (o_abs[i] > 0.0f) heap.push(ent((double)(cur[i] + 1) / (double)o_abs[i], i));
    double max_ip = 0.0, best_t = t_start;
    while (!heap.empty()) {
        ent e = heap.top(); heap.pop();
        double cur_t = e.first; size_t idx = e.second;
        if (cur_t >= t_end) continue;
        cur[idx] += 1;
        int32_t upd = cur[idx];
        sqr_den += 2.0 * (double)upd;
        num += (double)o_abs[idx];
        double ip = num / std::sqrt(sqr_den);
        if (ip > max_ip) { max_ip = ip; best_t = cur_t; }
        if (upd < (1 << ex_bits) - 1 && o_abs[idx] > 0.0f) {
            double tn = (double)(upd + 1) / (double)o_abs[idx];
            if (tn < t_end) heap.push(ent(tn, idx));
        }
    }
    if (best_t <= 0.0) return std::max(t_start, std::numeric_limits<double>::epsilon());
    return best_t;
}

// compute_const_scaling_factor, src/quantizer.rs:563-592
float const_scaling_factor(size_t dim, uint32_t ex_bits, uint64_t seed) {
    Rng rng(seed);
    double sum_t = 0.0;
    std::vector<float> v(dim), na(dim);
    for (int s = 0; s < 100; ++s) {
        for (size_t i = 0; i < dim; ++i) v[i] = (float)rng.normal();
        float n2 = -0.0f;
        for (size_t i = 0; i < dim; ++i) { float p = v[i] * v[i]; n2 = n2 + p; }
        float norm = std::sqrt(n2);
        if (norm <= F32_EPS) continue;
        for (size_t i = 0; i < dim; ++i) na[i] = std::fabs(v[i] / norm);
        sum_t += best_rescale_factor(na.data(), dim, ex_bits);
    }
    return (float)(sum_t / 100.0);
}

struct QV {
    std::vector<uint8_t> bin_packed, ex_packed;
    float f_add, f_rescale, f_error, f_add_ex, f_rescale_ex, delta, vl;
};

void pack_binary_code(const uint8_t* bits, uint8_t* packed, size_t dim) {
    std::memset(packed, 0, (dim + 7) / 8);
    for (size_t i = 0; i < dim; ++i) if (bits[i]) packed[i / 8] |= (uint8_t)(1u << (7 - (i % 8)));
}
void pack_ex2(const uint16_t* c, uint8_t* packed, size_t dim) {
    for (size_t b = 0; b < dim; b += 16) {
        uint32_t w = 0;
        for (int g = 0; g < 4; ++g)
            for (int i = 0; i < 4; ++i) w |= (uint32_t)(c[b + 4 * g + i] & 3u) << (8 * i + 2 * g);
        std::memcpy(packed + b / 16 * 4, &w, 4);
    }
}
void pack_ex6(const uint16_t* c, uint8_t* packed, size_t dim) {
    for (size_t b = 0; b < dim; b += 16) {
        uint64_t lo = 0; uint32_t hi = 0;
        for (int i = 0; i < 8; ++i) {
            lo |= (uint64_t)(c[b + i] & 15u) << (8 * i);
            lo |= (uint64_t)(c[b + 8 + i] & 15u) << (8 * i + 4);
        }
        for (int g = 0; g < 4; ++g)
            for (int i = 0; i < 4; ++i) hi |= (uint32_t)((c[b + 4 * g + i] >> 4) & 3u) << (8 * i + 2 * g);
        std::memcpy(packed + b / 16 * 12, &lo, 8);
        std::memcpy(packed + b / 16 * 12 + 8, &hi, 4);
    }
}
void pack_ex1(const uint16_t* c, uint8_t* packed, size_t dim) {
    for (size_t b = 0; b < dim; b += 16) {
        uint16_t w = 0;
        for (int i = 0; i < 16; ++i) w |= (uint16_t)((c[b + i] & 1u) << i);
        std::memcpy(packed + b / 16 * 2, &w, 2);
    }
}

// quantize_with_centroid, src/quantizer.rs:140-262 (+ :264-308, :310-335, :429-535)
void quantize_with_centroid(const float* data, const float* centroid, size_t dim, uint32_t total_bits,
                            bool has_t_const, float t_const, int metric, QV& out) {
    uint32_t ex_bits = total_bits - 1;
    std::vector<float> residual(dim), tmp(dim);
    std::vector<uint8_t> bits(dim);
    std::vector<uint16_t> ex_code(dim, 0);
    for (size_t i = 0; i < dim; ++i) { residual[i] = data[i] - centroid[i]; bits[i] = residual[i] >= 0.0f ? 1 : 0; }

    float ipnorm_inv = 1.0f;
    if (ex_bits > 0) {
        std::vector<float> na(dim);
        float n2 = -0.0f;
        for (size_t i = 0; i < dim; ++i) { na[i] = std::fabs(residual[i]); float p = na[i] * na[i]; n2 = n2 + p; }
        float norm = std::sqrt(n2);
        if (norm > F32_EPS) {
            for (size_t i = 0; i < dim; ++i) na[i] /= norm;
            double t = has_t_const ? (double)t_const : best_rescale_factor(na.data(), dim, ex_bits);
            int32_t max_val = (1 << ex_bits) - 1;
            double ipnorm = 0.0;
            for (size_t i = 0; i < dim; ++i) {
                int32_t cur = (int32_t)(t * (double)na[i] + K_EPS);
                if (cur > max_val) cur = max_val;
                ex_code[i] = (uint16_t)cur;
                ipnorm += ((double)cur + 0.5) * (double)na[i];
            }
            ipnorm_inv = (std::isfinite(ipnorm) && ipnorm > 0.0) ? (float)(1.0 / ipnorm) : 1.0f;
            for (size_t i = 0; i < dim; ++i)
                if (residual[i] < 0.0f) ex_code[i] = (uint16_t)((~ex_code[i]) & (uint16_t)max_val);
            if (!std::isfinite(ipnorm_inv)) ipnorm_inv = 1.0f;
        }
    }

    // compute_one_bit_factors
    float l2_sqr = l2_norm_sqr(residual.data(), dim), l2_norm = std::sqrt(l2_sqr);
    for (size_t i = 0; i < dim; ++i) tmp[i] = (float)bits[i] - 0.5f;
    float xu_norm_sqr = l2_norm_sqr(tmp.data(), dim);
    float ip_resi_xucb = dot8(residual.data(), tmp.data(), dim);
    float ip_cent_xucb = dot8(centroid, tmp.data(), dim);
    float dot_res_cent = dot8(residual.data(), centroid, dim);
    {
        float denom = ip_resi_xucb;
        if (std::fabs(denom) <= F32_EPS) denom = INFINITY;
        float tmp_error = 0.0f;
        if (dim > 1) {
            float ratio = ((l2_sqr * xu_norm_sqr) / (denom * denom)) - 1.0f;
            if (std::isfinite(ratio) && ratio > 0.0f)
                tmp_error = l2_norm * K_CONST_EPSILON * std::sqrt(std::max(ratio / (float)(dim - 1), 0.0f));
        }
        if (metric == RBQ_METRIC_L2) {
            out.f_add = l2_sqr + 2.0f * l2_sqr * ip_cent_xucb / denom;
            out.f_rescale = -2.0f * l2_sqr / denom;
            out.f_error = 2.0f * tmp_error;
        } else {
            out.f_add = 1.0f - dot_res_cent + l2_sqr * ip_cent_xucb / denom;
            out.f_rescale = -l2_sqr / denom;
            out.f_error = tmp_error;
        }
    }
    // delta / vl (reconstruction params, unused by search)
    float cb = -((float)(1 << ex_bits) - 0.5f);
    for (size_t i = 0; i < dim; ++i) tmp[i] = (float)(uint16_t)(ex_code[i] + ((uint16_t)bits[i] << ex_bits)) + cb;
    {
        float nq2 = l2_norm_sqr(tmp.data(), dim), drq = dot8(residual.data(), tmp.data(), dim);
        float nq = std::sqrt(nq2);
        float denom = std::max(l2_norm * nq, F32_EPS);
        float cosv = std::min(std::max(drq / denom, -1.0f), 1.0f);
        out.delta = nq <= F32_EPS ? 0.0f : (l2_norm / nq) * cosv;
        out.vl = out.delta * cb;
    }
    out.f_add_ex = 0.0f; out.f_rescale_ex = 0.0f;
    if (ex_bits > 0) {
        // compute_extended_factors: xu_cb is the same vector as tmp above
        float ip_r = dot8(residual.data(), tmp.data(), dim);
        float ip_c = dot8(centroid, tmp.data(), dim);
        float safe = std::fabs(ip_r) <= F32_EPS ? INFINITY : ip_r;
        if (metric == RBQ_METRIC_L2) {
            out.f_add_ex = l2_sqr + 2.0f * l2_sqr * ip_c / safe;
            out.f_rescale_ex = -2.0f * l2_norm * ipnorm_inv;
        } else {
            out.f_add_ex = 1.0f - dot_res_cent + l2_sqr * ip_c / safe;
            out.f_rescale_ex = -l2_norm * ipnorm_inv;
        }
    }
    out.bin_packed.assign((dim + 7) / 8, 0);
    pack_binary_code(bits.data(), out.bin_packed.data(), dim);
    size_t exb = ex_bits == 0 ? 0 : dim * ex_bits / 8;
    out.ex_packed.assign(exb, 0);
    if (ex_bits == 2) pack_ex2(ex_code.data(), out.ex_packed.data(), dim);
    else if (ex_bits == 6) pack_ex6(ex_code.data(), out.ex_packed.data(), dim);
}

// pack_codes, src/simd.rs:864-904 (one 32-vector batch)
const int KPERM0[16] = {0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6, 14, 7, 15};
void pack_codes_batch(const uint8_t* const* vec_bytes /*[32] or null*/, size_t dim_bytes, uint8_t* packed) {
    for (size_t col = 0; col < dim_bytes; ++col) {
        uint8_t c0[32], c1[32];
        for (int j = 0; j < 32; ++j) {
            uint8_t b = vec_bytes[j] ? vec_bytes[j][col] : 0;
            c0[j] = b >> 4; c1[j] = b & 15;
        }
        for (int j = 0; j < 16; ++j) {
            packed[col * 32 + j] = (uint8_t)(c0[KPERM0[j]] | (c0[KPERM0[j] + 16] << 4));
            packed[col * 32 + 16 + j] = (uint8_t)(c1[KPERM0[j]] | (c1[KPERM0[j] + 16] << 4));
        }
    }
}

// CRC-32/IEEE (crc32fast), used by save_to_writer
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}

} // namespace

struct rbq_built {
    rbq_header hdr;
    std::vector<uint8_t> rotator_blob;
    struct List {
        std::vector<float> centroid;
        std::vector<uint64_t> ids;
        std::vector<uint8_t> batch_data, ex_codes;
        std::vector<float> f_add_ex, f_rescale_ex, delta, vl;
    };
    std::vector<List> lists;
    std::vector<rbq_list_view> views;
    float t_const = 0.0f;
    void finish() {
        views.resize(lists.size());
        uint64_t total = 0;
        for (size_t c = 0; c < lists.size(); ++c) {
            List& l = lists[c];
            rbq_list_view& v = views[c];
            v.centroid = l.centroid.data(); v.n = l.ids.size(); v.ids = l.ids.data();
            v.batch_data = l.batch_data.data(); v.batch_len = l.batch_data.size();
            v.ex_codes = l.ex_codes.empty() ? nullptr : l.ex_codes.data();
            v.f_add_ex = l.f_add_ex.data(); v.f_rescale_ex = l.f_rescale_ex.data();
            total += v.n;
        }
        hdr.n_vectors = total; hdr.n_lists = lists.size();
        hdr.rotator_blob = rotator_blob.data(); hdr.rotator_len = rotator_blob.size();
    }
};

extern "C" {

int rbq_build_train_with_clusters(const float* data, uint64_t n, uint32_t dim,
                                  const float* centroids, uint64_t nlist, const uint32_t* assignments,
                                  uint32_t total_bits, uint8_t metric, uint8_t rotator_type,
                                  uint64_t seed, int use_faster_config, rbq_built** out) {
    *out = nullptr;
    if (n == 0 || nlist == 0 || nlist > n || dim == 0) return RBQ_INVALID_CONFIG;
    if (total_bits == 0 || total_bits > 16) return RBQ_INVALID_CONFIG;
    uint32_t ex_bits = total_bits - 1;
    if (ex_bits != 0 && ex_bits != 2 && ex_bits != 6) return RBQ_INVALID_CONFIG; // select_excode_ipfunc panics
    for (uint64_t i = 0; i < n; ++i) if (assignments[i] >= nlist) return RBQ_INVALID_CONFIG;
    uint32_t D = rotator_type == RBQ_ROTATOR_FHT_KAC ? (dim + 63) / 64 * 64 : dim;

    rbq_built* b = new rbq_built();
    std::memset(&b->hdr, 0, sizeof b->hdr);
    b->hdr.dim = dim; b->hdr.padded_dim = D; b->hdr.metric = metric; b->hdr.rotator = rotator_type; b->hdr.ex_bits = (uint8_t)ex_bits;
    if (rotator_type > RBQ_ROTATOR_NONE) { delete b; return RBQ_INVALID_CONFIG; }
    if (rotator_type == RBQ_ROTATOR_NONE) {
        if (dim % 16 != 0) { delete b; return RBQ_INVALID_CONFIG; } // FastScan asserts dim % 16 == 0
    } else if (rotator_type == RBQ_ROTATOR_FHT_KAC) {
        Rng rng(seed);
        b->rotator_blob.resize(4 * D / 8);
        for (auto& x : b->rotator_blob) x = (uint8_t)(rng.next() >> 56);
    } else {
        std::vector<float> m = make_matrix_rotator(D, seed);
        b->rotator_blob.resize(m.size() * 4);
        std::memcpy(b->rotator_blob.data(), m.data(), m.size() * 4);
    }
    auto rotate = [&](const float* in, float* o) {
        if (rotator_type == RBQ_ROTATOR_FHT_KAC) fht_kac_rotate(dim, D, b->rotator_blob.data(), in, o);
        else if (rotator_type == RBQ_ROTATOR_NONE) std::memcpy(o, in, sizeof(float) * dim);
        else matrix_rotate(dim, D, (const float*)b->rotator_blob.data(), in, o);
    };
    bool has_t = use_faster_config && ex_bits > 0;
    float t_const = has_t ? const_scaling_factor(D, ex_bits, seed) : 0.0f;
    b->t_const = t_const;

    // group vector indices by cluster in ascending index order (src/ivf.rs:1141-1149)
    std::vector<std::vector<uint64_t>> members(nlist);
    for (uint64_t i = 0; i < n; ++i) members[assignments[i]].push_back(i);

    b->lists.resize(nlist);
    size_t dim_bytes = D / 8, stride = (size_t)D * 4 + 384, exb = (size_t)D * ex_bits / 8;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t c = 0; c < (int64_t)nlist; ++c) {
        rbq_built::List& L = b->lists[c];
        L.centroid.resize(D);
        rotate(centroids + (size_t)c * dim, L.centroid.data());
        const std::vector<uint64_t>& mem = members[c];
        size_t m = mem.size(), nb = (m + 31) / 32;
        L.ids = mem;
        L.batch_data.assign(nb * stride, 0);
        L.ex_codes.assign(m * exb, 0);
        L.f_add_ex.assign(m, 0.0f); L.f_rescale_ex.assign(m, 0.0f); L.delta.assign(m, 0.0f); L.vl.assign(m, 0.0f);
        std::vector<float> rv(D);
        std::vector<QV> qv(32);
        for (size_t bi = 0; bi < nb; ++bi) {
            size_t cnt = std::min<size_t>(32, m - bi * 32);
            const uint8_t* ptrs[32];
            uint8_t* rec = L.batch_data.data() + bi * stride;
            float* f_add = (float*)(rec + (size_t)D * 4);
            float* f_rescale = f_add + 32; float* f_error = f_rescale + 32;
            for (size_t j = 0; j < 32; ++j) {
                if (j < cnt) {
                    size_t v = bi * 32 + j;
                    rotate(data + (size_t)mem[v] * dim, rv.data());
                    quantize_with_centroid(rv.data(), L.centroid.data(), D, total_bits, has_t, t_const, metric, qv[j]);
                    ptrs[j] = qv[j].bin_packed.data();
                    f_add[j] = qv[j].f_add; f_rescale[j] = qv[j].f_rescale; f_error[j] = qv[j].f_error;
                    if (exb) std::memcpy(L.ex_codes.data() + v * exb, qv[j].ex_packed.data(), exb);
                    if (ex_bits > 0) { L.f_add_ex[v] = qv[j].f_add_ex; L.f_rescale_ex[v] = qv[j].f_rescale_ex; }
                    L.delta[v] = qv[j].delta; L.vl[v] = qv[j].vl;
                } else {
                    ptrs[j] = nullptr; // zero-padded tail: codes 0, factors 0 (src/ivf.rs:466-491)
                    f_add[j] = 0.0f; f_rescale[j] = 0.0f; f_error[j] = 0.0f;
                }
            }
            pack_codes_batch(ptrs, dim_bytes, rec);
        }
    }
    b->finish();
    *out = b;
    return RBQ_OK;
}

const rbq_header* rbq_built_header(const rbq_built* b) { return &b->hdr; }
const rbq_list_view* rbq_built_lists(const rbq_built* b) { return b->views.data(); }
float rbq_built_t_const(const rbq_built* b) { return b->t_const; }
// ClusterData.delta / .vl of list c (reconstruction parameters: persisted by save_to_writer, src/ivf.rs:1455-1463, unused by
// search and therefore not part of rbq_list_view)
int rbq_built_list_recon(const rbq_built* b, uint64_t c, const float** delta, const float** vl) {
    if (!b || c >= b->lists.size() || !delta || !vl) return RBQ_INVALID_CONFIG;
    *delta = b->lists[c].delta.data(); *vl = b->lists[c].vl.data();
    return RBQ_OK;
}
void rbq_built_free(rbq_built* b) { delete b; }

// save_to_writer, src/ivf.rs:1317-1474. Caller frees *bytes with rbq_build_free_bytes.
int rbq_built_save_rbq1(const rbq_built* b, uint8_t** bytes, uint64_t* len) {
    std::vector<uint8_t> o;
    auto put = [&](const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; o.insert(o.end(), q, q + n); };
    auto u32 = [&](uint32_t v) { put(&v, 4); };
    auto u64 = [&](uint64_t v) { put(&v, 8); };
    put("RBQ1", 4); u32(3);
    u32(b->hdr.dim); u32(b->hdr.padded_dim);
    uint8_t tags[4] = {b->hdr.metric, b->hdr.rotator, b->hdr.ex_bits, (uint8_t)(b->hdr.ex_bits + 1)};
    put(tags, 4);
    u64(b->hdr.n_vectors); u64(b->hdr.n_lists);
    u64(b->rotator_blob.size()); put(b->rotator_blob.data(), b->rotator_blob.size());
    size_t exb = (size_t)b->hdr.padded_dim * b->hdr.ex_bits / 8;
    for (const auto& L : b->lists) {
        put(L.centroid.data(), L.centroid.size() * 4);
        u64(L.ids.size());
        put(L.ids.data(), L.ids.size() * 8);
        u64(L.batch_data.size()); put(L.batch_data.data(), L.batch_data.size());
        for (size_t v = 0; v < L.ids.size(); ++v) { u64(exb); if (exb) put(L.ex_codes.data() + v * exb, exb); }
        put(L.f_add_ex.data(), L.f_add_ex.size() * 4);
        put(L.f_rescale_ex.data(), L.f_rescale_ex.size() * 4);
        put(L.delta.data(), L.delta.size() * 4);
        put(L.vl.data(), L.vl.size() * 4);
    }
    uint32_t crc = crc32_update(0, o.data() + 8, o.size() - 8);
    u32(crc);
    *bytes = (uint8_t*)std::malloc(o.size());
    std::memcpy(*bytes, o.data(), o.size());
    *len = o.size();
    return RBQ_OK;
}
void rbq_build_free_bytes(uint8_t* p) { std::free(p); }

// --- primitives exported for the reference's literal known-answer tests -------
void rbq_build_pack_binary_code(const uint8_t* bits, uint8_t* packed, uint64_t dim) { pack_binary_code(bits, packed, dim); }
void rbq_build_pack_ex_code_1bit(const uint16_t* c, uint8_t* p, uint64_t dim) { pack_ex1(c, p, dim); }
void rbq_build_pack_ex_code_2bit(const uint16_t* c, uint8_t* p, uint64_t dim) { pack_ex2(c, p, dim); }
void rbq_build_pack_ex_code_6bit(const uint16_t* c, uint8_t* p, uint64_t dim) { pack_ex6(c, p, dim); }
void rbq_build_pack_codes(const uint8_t* codes, uint64_t num_vectors, uint64_t dim_bytes, uint8_t* packed) {
    uint64_t nb = (num_vectors + 31) / 32;
    for (uint64_t b = 0; b < nb; ++b) {
        const uint8_t* ptrs[32];
        for (uint64_t j = 0; j < 32; ++j) ptrs[j] = b * 32 + j < num_vectors ? codes + (b * 32 + j) * dim_bytes : nullptr;
        pack_codes_batch(ptrs, dim_bytes, packed + b * 32 * dim_bytes);
    }
}
uint32_t rbq_build_crc32(const uint8_t* p, uint64_t n) { return crc32_update(0, p, n); }
void rbq_build_rotate(const rbq_header* h, const float* in, float* out) {
    if (h->rotator == RBQ_ROTATOR_FHT_KAC) fht_kac_rotate(h->dim, h->padded_dim, h->rotator_blob, in, out);
    else if (h->rotator == RBQ_ROTATOR_NONE) std::memcpy(out, in, sizeof(float) * h->dim);
    else matrix_rotate(h->dim, h->padded_dim, (const float*)h->rotator_blob, in, out);
}

// Lloyd k-means (harness; the reference's Faiss-style run_kmeans, src/kmeans.rs:49, is
// out of scope and `train_with_clusters` accepts any clustering, src/ivf.rs:1025-1034).
int rbq_build_kmeans(const float* data, uint64_t n, uint32_t dim, uint64_t k, int iters, uint64_t seed,
                     float* centroids, uint32_t* assignments) {
    if (k == 0 || k > n) return RBQ_INVALID_CONFIG;
    Rng rng(seed);
    std::vector<uint64_t> perm(n);
    for (uint64_t i = 0; i < n; ++i) perm[i] = i;
    for (uint64_t i = 0; i < k; ++i) { uint64_t j = i + rng.next() % (n - i); std::swap(perm[i], perm[j]); }
    for (uint64_t c = 0; c < k; ++c) std::memcpy(centroids + c * dim, data + perm[c] * dim, sizeof(float) * dim);
    std::vector<double> sums; std::vector<uint64_t> counts;
    for (int it = 0; it <= iters; ++it) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            const float* x = data + (size_t)i * dim;
            float best = INFINITY; uint32_t bc = 0;
            for (uint64_t c = 0; c < k; ++c) {
                const float* m = centroids + c * dim;
                float d = 0.0f;
                for (uint32_t j = 0; j < dim; ++j) { float t = x[j] - m[j]; d += t * t; }
                if (d < best) { best = d; bc = (uint32_t)c; }
            }
            assignments[i] = bc;
        }
        if (it == iters) break;
        sums.assign((size_t)k * dim, 0.0); counts.assign(k, 0);
        for (uint64_t i = 0; i < n; ++i) {
            uint32_t c = assignments[i]; counts[c]++;
            for (uint32_t j = 0; j < dim; ++j) sums[(size_t)c * dim + j] += data[i * dim + j];
        }
        for (uint64_t c = 0; c < k; ++c) {
            if (!counts[c]) { uint64_t p = rng.next() % n; std::memcpy(centroids + c * dim, data + p * dim, sizeof(float) * dim); continue; }
            for (uint32_t j = 0; j < dim; ++j) centroids[c * dim + j] = (float)(sums[(size_t)c * dim + j] / (double)counts[c]);
        }
    }
    return RBQ_OK;
}

} // extern "C"
