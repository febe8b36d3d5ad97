/*
 * rbq_ref.c — CPU ORACLE for the IVF+RaBitQ query path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference algorithm (lqhl/rabitq-rs v0.7.0), used
 * as the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  Nothing in the product path (rabitq-rs_amd/) links, loads or calls it.
 *
 * Parity pin status: the Rust crate cannot be built or imported in this
 * pipeline (no cargo/rustc, un-vendored deps), and its tests hold
 * no golden search outputs.  This restatement is therefore pinned by
 *   (1) the reference's own literal known-answer tests (tests/test_oracle_kat.py
 *       cites each one), and
 *   (2) agreement of independent formulations of the same arithmetic
 *       (KPERM-scalar vs pshufb-emulation vs AVX2/AVX-512 intrinsics accumulate).
 * End-to-end `search` outputs are "parity unpinned" by the reference itself.
 *
 * Numeric variant restated: the `target-cpu=native` build on an AVX-512 host
 * (what .cargo/config.toml selects): AVX2 bodies of math::dot/l2_distance_sqr
 * (runtime-detected, src/math.rs:8-11,41-44), AVX2 body of
 * compute_batch_distances_u16 (compile-time cfg, src/simd.rs:1946), AVX-512
 * bodies of ip_packed_ex{2,6}_f32 (src/simd.rs:1551,1588).  All lane orders are
 * reproduced in scalar C so the oracle runs on any x86-64; compile with
 * -ffp-contract=off so that only the explicit fmaf() calls fuse.
 *
 * Third-party behaviour restated from its published source (absent from
 * /root/reference): Rust std `alloc::collections::BinaryHeap` (push / pop /
 * into_sorted_vec sift order; toolchain pinned nightly-2025-12-13 by
 * rust-toolchain.toml:2) and `core::arch::x86_64::_mm512_reduce_add_ps`
 * (stdarch: 16->8->4->2->1 halving tree).  They only matter for exact ties and
 * for the last-bit rounding of the ex-code dot product.  Since round 4 both are
 * pinned by a second definition (tests/test_oracle_kat.py): the reduce tree and
 * the 16-lane FMA order against real AVX-512 instructions (gcc's own
 * _mm512_reduce_add_ps), the heap's push / pop order against CPython's heapq.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/rbq.h"
#include "rbq_ref.h"

/* ------------------------------------------------------------------------- */
/* helpers                                                                   */
/* ------------------------------------------------------------------------- */

/* Numeric variants of the reference (ref_set_variant; 0 = the default restated above).  The crate picks its arithmetic at    */
/* COMPILE time (cfg(target_feature), .cargo/config.toml:11-14), so a build for another host returns other last bits; these   */
/* switches restate the other bodies so that tests/test_oracle_variants.py can measure what each is worth.  Bits:             */
/*   REF_VAR_EX_AVX2     ip_packed_ex{2,6}_f32_avx2 (src/simd.rs:1722-1825): ONE 8-lane accumulator, two FMAs per 16 dims      */
/*                       (dims 16t+l, then 16t+8+l), hsum lo128+hi128 / movehl / shuffle 0x55 — an AVX2-only host             */
/*   REF_VAR_EX_SCALAR   ip_packed_ex{2,6}_f32_scalar (:1615-1713): one running sum, unfused `sum += c * q` in the scalar      */
/*                       bodies' own order — a host without AVX2 (what CI's RUSTFLAGS="" builds, .github/workflows/ci.yml)     */
/*   REF_VAR_EPI_SCALAR  compute_batch_distances_u16_scalar (:2039-2061): `delta * accu + sum_vl` NOT fused                   */
/*   REF_VAR_CONTRACT    what `-C llvm-args=--ffast-math` may do IF the pinned LLVM honours it: every `a * b + c` of the path  */
/*                       that the source leaves unfused becomes one fused multiply-add (math::dot / l2_distance_sqr AVX2       */
/*                       bodies, the rest of the epilogue, the refine formula).  Reassociation and reciprocal division, which  */
/*                       the same flag also permits, are compiler-version specific and are NOT restated.                       */
static int g_variant = 0;
void ref_set_variant(int v) { g_variant = v; }
int ref_get_variant(void) { return g_variant; }

static inline int32_t f32_bits(float x) { int32_t i; memcpy(&i, &x, 4); return i; }

/* f32::total_cmp key (core::f32::total_cmp): flip the magnitude bits of negatives. */
static inline int32_t total_key(float x) {
    int32_t i = f32_bits(x);
    i ^= (int32_t)(((uint32_t)(i >> 31)) >> 1);
    return i;
}
static inline int total_cmp(float a, float b) {
    int32_t ka = total_key(a), kb = total_key(b);
    return (ka > kb) - (ka < kb);
}

/* ------------------------------------------------------------------------- */
/* src/math.rs: dot / l2_distance_sqr, AVX2 bodies (:154-181, :216-245)      */
/* 8 strided lanes, unfused mul then add, lanes summed sequentially 0..7     */
/* (iter().sum() folds from -0.0), scalar tail.                              */
/* ------------------------------------------------------------------------- */

float ref_dot(const float* a, const float* b, size_t len) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t chunks = len / 8, i = 0;
    for (; i < chunks * 8; i += 8)
        for (int l = 0; l < 8; ++l) {
            if (g_variant & REF_VAR_CONTRACT) { acc[l] = fmaf(a[i + l], b[i + l], acc[l]); continue; }
            float p = a[i + l] * b[i + l];
            acc[l] = acc[l] + p;
        }
    float sum = 0.0f;
    if (chunks > 0) {
        sum = -0.0f;
        for (int l = 0; l < 8; ++l) sum = sum + acc[l];
    }
    for (; i < len; ++i) {
        float p = a[i] * b[i];
        sum = sum + p;
    }
    return sum;
}

float ref_l2_distance_sqr(const float* a, const float* b, size_t len) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t chunks = len / 8, i = 0;
    for (; i < chunks * 8; i += 8)
        for (int l = 0; l < 8; ++l) {
            float d = a[i + l] - b[i + l];
            if (g_variant & REF_VAR_CONTRACT) { acc[l] = fmaf(d, d, acc[l]); continue; }
            float p = d * d;
            acc[l] = acc[l] + p;
        }
    float sum = 0.0f;
    if (chunks > 0) {
        sum = -0.0f;
        for (int l = 0; l < 8; ++l) sum = sum + acc[l];
    }
    for (; i < len; ++i) {
        float d = a[i] - b[i];
        float p = d * d;
        sum = sum + p;
    }
    return sum;
}

/* ------------------------------------------------------------------------- */
/* src/rotation.rs                                                           */
/* ------------------------------------------------------------------------- */

/* floor_log2, src/rotation.rs:514-517 */
uint32_t ref_floor_log2(uint64_t x) {
    uint32_t r = 0;
    while (x >>= 1) ++r;
    return r;
}

/* padding_requirement, src/rotation.rs:27-32 */
uint32_t ref_padded_dim(uint32_t dim, int rotator) {
    if (rotator == RBQ_ROTATOR_MATRIX) return dim;
    return (dim + 63u) / 64u * 64u;
}

/* flip_sign, src/rotation.rs:278-289 (LSB-first bit per element) */
static void flip_sign(float* data, size_t n, const uint8_t* flip, size_t flip_len) {
    for (size_t i = 0; i < n; ++i) {
        size_t byte = i / 8;
        if (byte < flip_len && ((flip[byte] >> (i % 8)) & 1)) data[i] = -data[i];
    }
}

/* fht, src/rotation.rs:292-312 */
void ref_fht(float* data, size_t n) {
    for (size_t h = 1; h < n; h *= 2)
        for (size_t i = 0; i < n; i += 2 * h)
            for (size_t j = i; j < i + h; ++j) {
                float x = data[j], y = data[j + h];
                data[j] = x + y;
                data[j + h] = x - y;
            }
}

/* kacs_walk, src/rotation.rs:315-324 */
static void kacs_walk(float* data, size_t len) {
    size_t half = len / 2;
    for (size_t i = 0; i < half; ++i) {
        float x = data[i], y = data[i + half];
        data[i] = x + y;
        data[i + half] = x - y;
    }
}

static void rescale(float* data, size_t n, float f) {
    for (size_t i = 0; i < n; ++i) data[i] *= f;
}

/* FhtKacRotator::rotate_into, src/rotation.rs:350-401; ctor consts :263-266 */
void ref_fht_kac_rotate(uint32_t dim, uint32_t D, const uint8_t* flip, const float* in, float* out) {
    memcpy(out, in, sizeof(float) * dim);
    for (uint32_t i = dim; i < D; ++i) out[i] = 0.0f;
    size_t fo = D / 8;
    uint32_t trunc = 1u << ref_floor_log2(dim);
    float fac = 1.0f / sqrtf((float)trunc);
    if (trunc == D) {
        for (int r = 0; r < 4; ++r) {
            flip_sign(out, D, flip + r * fo, fo);
            ref_fht(out, D);
            rescale(out, D, fac);
        }
    } else {
        uint32_t start = D - trunc;
        flip_sign(out, D, flip, fo);
        ref_fht(out, trunc);
        rescale(out, trunc, fac);
        kacs_walk(out, D);

        flip_sign(out, D, flip + fo, fo);
        ref_fht(out + start, trunc);
        rescale(out + start, trunc, fac);
        kacs_walk(out, D);

        flip_sign(out, D, flip + 2 * fo, fo);
        ref_fht(out, trunc);
        rescale(out, trunc, fac);
        kacs_walk(out, D);

        flip_sign(out, D, flip + 3 * fo, fo);
        ref_fht(out + start, trunc);
        rescale(out + start, trunc, fac);
        kacs_walk(out, D);

        rescale(out, D, 0.25f);
    }
}

/* MatrixRotator::rotate_into, src/rotation.rs:158-173: sequential unfused acc */
void ref_matrix_rotate(uint32_t dim, uint32_t D, const float* matrix, const float* in, float* out) {
    for (uint32_t r = 0; r < D; ++r) {
        const float* row = matrix + (size_t)r * D;
        float acc = 0.0f;
        for (uint32_t c = 0; c < D; ++c) {
            float v = c < dim ? in[c] : 0.0f;
            float p = v * row[c];
            acc = acc + p;
        }
        out[r] = acc;
    }
}

void ref_rotate(const rbq_header* h, const float* in, float* out) {
    if (h->rotator == RBQ_ROTATOR_FHT_KAC)
        ref_fht_kac_rotate(h->dim, h->padded_dim, h->rotator_blob, in, out);
    else if (h->rotator == RBQ_ROTATOR_NONE)
        memcpy(out, in, sizeof(float) * h->dim);
    else
        ref_matrix_rotate(h->dim, h->padded_dim, (const float*)h->rotator_blob, in, out);
}

/* ------------------------------------------------------------------------- */
/* src/simd.rs:771-840 pack_lut_f32 ; src/ivf.rs:798-845 QueryLut::new       */
/* ------------------------------------------------------------------------- */

static const int KPOS[16] = {3, 3, 2, 3, 1, 3, 2, 3, 0, 3, 2, 3, 1, 3, 2, 3};
static const int KPERM0[16] = {0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6, 14, 7, 15};

/* Per-thread scratch reused across queries (slots never shrink): the timed CPU baseline then measures the search
 * itself, not five malloc/free pairs per query. */
static __thread void* g_scr[8];
static __thread size_t g_scr_cap[8];
static void* scratch(int slot, size_t bytes) {
    if (bytes > g_scr_cap[slot]) {
        free(g_scr[slot]);
        g_scr[slot] = malloc(bytes ? bytes : 1);
        g_scr_cap[slot] = g_scr[slot] ? bytes : 0;
    }
    return g_scr[slot];
}

void ref_pack_lut_f32(const float* q, size_t D, float* lut) {
    size_t ncb = D / 4;
    for (size_t i = 0; i < ncb; ++i) {
        float* l = lut + i * 16;
        l[0] = 0.0f;
        for (int j = 1; j < 16; ++j) {
            int lowbit = j & (-j);
            l[j] = l[j - lowbit] + q[i * 4 + KPOS[j]];
        }
    }
}

void ref_query_lut(const float* q, size_t D, uint8_t* lut8, float* delta_out, float* sum_vl_out) {
    size_t T = D * 4;
    float* lf = (float*)scratch(0, sizeof(float) * T);
    ref_pack_lut_f32(q, D, lf);
    float vl = lf[0], vr = lf[0];
    for (size_t i = 1; i < T; ++i) {
        if (total_cmp(lf[i], vl) < 0) vl = lf[i];
        if (total_cmp(lf[i], vr) >= 0) vr = lf[i]; /* max_by keeps the last maximum */
    }
    float delta = (vr - vl) / 255.0f;
    memset(lut8, 0, T);
    if (delta > 0.0f) {
        for (size_t i = 0; i < T; ++i) {
            float qv = roundf((lf[i] - vl) / delta);
            if (!(qv >= 0.0f)) qv = 0.0f; /* clamp; NaN -> `as u8` = 0 */
            if (qv > 255.0f) qv = 255.0f;
            lut8[i] = (uint8_t)qv;
        }
    }
    *delta_out = delta;
    *sum_vl_out = vl * (float)(T / 16);
}

/* QueryPrecomputed::new, src/ivf.rs:862-878 */
void ref_query_precompute(const float* q, size_t D, uint32_t ex_bits, ref_query_consts* c) {
    float s = -0.0f, n2 = -0.0f;
    for (size_t i = 0; i < D; ++i) s = s + q[i];
    for (size_t i = 0; i < D; ++i) {
        float p = q[i] * q[i];
        n2 = n2 + p;
    }
    c->sum_q = s;
    c->query_norm = sqrtf(n2);
    c->k1x_sum_q = -0.5f * s;
    float cb = -((float)(1 << ex_bits) - 0.5f);
    c->kbx_sum_q = cb * s;
    c->binary_scale = (float)(1 << ex_bits);
}

/* ------------------------------------------------------------------------- */
/* accumulate_batch: three independent formulations                          */
/* ------------------------------------------------------------------------- */

/* (1) KPERM scalar, src/simd.rs:1462-1525 */
void ref_accumulate_batch_scalar(const uint8_t* codes, const uint8_t* lut, size_t D, uint16_t* res) {
    int32_t sums[32];
    memset(sums, 0, sizeof sums);
    size_t dim_bytes = D / 8;
    for (size_t col = 0; col < dim_bytes; ++col) {
        const uint8_t* p = codes + col * 32;
        const uint8_t* lhi = lut + (col * 2) * 16;
        const uint8_t* llo = lut + (col * 2 + 1) * 16;
        for (int j = 0; j < 16; ++j) {
            sums[KPERM0[j]] += lhi[p[j] & 15];
            sums[KPERM0[j] + 16] += lhi[p[j] >> 4];
        }
        for (int j = 0; j < 16; ++j) {
            sums[KPERM0[j]] += llo[p[16 + j] & 15];
            sums[KPERM0[j] + 16] += llo[p[16 + j] >> 4];
        }
    }
    for (int i = 0; i < 32; ++i) res[i] = (uint16_t)sums[i];
}

/* (2) portable emulation of the AVX2 pshufb algorithm incl. its wrapping-u16
 * lane tricks, src/simd.rs:1016-1110.  A 256-bit register = 32 bytes = 16 u16. */
void ref_accumulate_batch_shuffle_emul(const uint8_t* codes, const uint8_t* lut, size_t D, uint16_t* res) {
    uint16_t a0[16], a1[16], a2[16], a3[16];
    memset(a0, 0, 32); memset(a1, 0, 32); memset(a2, 0, 32); memset(a3, 0, 32);
    size_t code_length = D * 4;
    for (size_t i = 0; i < code_length; i += 32) {
        uint8_t rlo[32], rhi[32];
        for (int b = 0; b < 32; ++b) {
            uint8_t c = codes[i + b];
            int lane = b & 16; /* _mm256_shuffle_epi8 shuffles within each 128-bit lane */
            rlo[b] = lut[i + lane + (c & 15)];
            rhi[b] = lut[i + lane + (c >> 4)];
        }
        for (int w = 0; w < 16; ++w) {
            uint16_t wl = (uint16_t)(rlo[2 * w] | (rlo[2 * w + 1] << 8));
            uint16_t wh = (uint16_t)(rhi[2 * w] | (rhi[2 * w + 1] << 8));
            a0[w] = (uint16_t)(a0[w] + wl);
            a1[w] = (uint16_t)(a1[w] + (wl >> 8));
            a2[w] = (uint16_t)(a2[w] + wh);
            a3[w] = (uint16_t)(a3[w] + (wh >> 8));
        }
    }
    for (int w = 0; w < 16; ++w) {
        a0[w] = (uint16_t)(a0[w] - (uint16_t)(a1[w] << 8));
        a2[w] = (uint16_t)(a2[w] - (uint16_t)(a3[w] << 8));
    }
    /* dis0 = permute2f128(a0,a1,0x21) + blend_epi32(a0,a1,0xF0)
     *      = [a0.hi128, a1.lo128] + [a0.lo128, a1.hi128]           */
    for (int w = 0; w < 8; ++w) {
        res[w] = (uint16_t)(a0[8 + w] + a0[w]);
        res[8 + w] = (uint16_t)(a1[w] + a1[8 + w]);
        res[16 + w] = (uint16_t)(a2[8 + w] + a2[w]);
        res[24 + w] = (uint16_t)(a3[w] + a3[8 + w]);
    }
}

/* (3) real intrinsics (the path the timed CPU baseline uses) */
__attribute__((target("avx2")))
static void accumulate_batch_avx2(const uint8_t* codes, const uint8_t* lut, size_t D, uint16_t* res) {
    const __m256i low_mask = _mm256_set1_epi8(0x0f);
    __m256i a0 = _mm256_setzero_si256(), a1 = a0, a2 = a0, a3 = a0;
    size_t code_length = D * 4;
    for (size_t i = 0; i < code_length; i += 32) {
        __m256i c = _mm256_loadu_si256((const __m256i*)(codes + i));
        __m256i lv = _mm256_loadu_si256((const __m256i*)(lut + i));
        __m256i lo = _mm256_and_si256(c, low_mask);
        __m256i hi = _mm256_and_si256(_mm256_srli_epi16(c, 4), low_mask);
        __m256i rl = _mm256_shuffle_epi8(lv, lo);
        __m256i rh = _mm256_shuffle_epi8(lv, hi);
        a0 = _mm256_add_epi16(a0, rl);
        a1 = _mm256_add_epi16(a1, _mm256_srli_epi16(rl, 8));
        a2 = _mm256_add_epi16(a2, rh);
        a3 = _mm256_add_epi16(a3, _mm256_srli_epi16(rh, 8));
    }
    a0 = _mm256_sub_epi16(a0, _mm256_slli_epi16(a1, 8));
    a2 = _mm256_sub_epi16(a2, _mm256_slli_epi16(a3, 8));
    __m256i d0 = _mm256_add_epi16(_mm256_permute2f128_si256(a0, a1, 0x21), _mm256_blend_epi32(a0, a1, 0xF0));
    __m256i d1 = _mm256_add_epi16(_mm256_permute2f128_si256(a2, a3, 0x21), _mm256_blend_epi32(a2, a3, 0xF0));
    _mm256_storeu_si256((__m256i*)res, d0);
    _mm256_storeu_si256((__m256i*)(res + 16), d1);
}

__attribute__((target("avx512f,avx512bw")))
static void accumulate_batch_avx512(const uint8_t* codes, const uint8_t* lut, size_t D, uint16_t* res) {
    const __m512i low_mask = _mm512_set1_epi8(0x0f);
    __m512i a0 = _mm512_setzero_si512(), a1 = a0, a2 = a0, a3 = a0;
    size_t code_length = D * 4;
    for (size_t i = 0; i < code_length; i += 64) {
        __m512i c = _mm512_loadu_si512((const void*)(codes + i));
        __m512i lv = _mm512_loadu_si512((const void*)(lut + i));
        __m512i lo = _mm512_and_si512(c, low_mask);
        __m512i hi = _mm512_and_si512(_mm512_srli_epi16(c, 4), low_mask);
        __m512i rl = _mm512_shuffle_epi8(lv, lo);
        __m512i rh = _mm512_shuffle_epi8(lv, hi);
        a0 = _mm512_add_epi16(a0, rl);
        a1 = _mm512_add_epi16(a1, _mm512_srli_epi16(rl, 8));
        a2 = _mm512_add_epi16(a2, rh);
        a3 = _mm512_add_epi16(a3, _mm512_srli_epi16(rh, 8));
    }
    a0 = _mm512_sub_epi16(a0, _mm512_slli_epi16(a1, 8));
    a2 = _mm512_sub_epi16(a2, _mm512_slli_epi16(a3, 8));
    __m512i r1 = _mm512_add_epi16(_mm512_mask_blend_epi64(0xF0, a0, a1), _mm512_shuffle_i64x2(a0, a1, 0x4E));
    __m512i r2 = _mm512_add_epi16(_mm512_mask_blend_epi64(0xF0, a2, a3), _mm512_shuffle_i64x2(a2, a3, 0x4E));
    __m512i ret = _mm512_add_epi16(_mm512_shuffle_i64x2(r1, r2, 0x88), _mm512_shuffle_i64x2(r1, r2, 0xDD));
    _mm512_storeu_si512((void*)res, ret);
}

static int g_simd_level = -1; /* 0 scalar, 1 avx2, 2 avx512 */
int ref_simd_level(void) {
    if (g_simd_level < 0) {
        __builtin_cpu_init();
        if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw")) g_simd_level = 2;
        else if (__builtin_cpu_supports("avx2")) g_simd_level = 1;
        else g_simd_level = 0;
    }
    return g_simd_level;
}
void ref_force_simd_level(int level) { g_simd_level = level; }

/* dispatch chain of accumulate_batch_avx2, src/simd.rs:972-1014 */
void ref_accumulate_batch(const uint8_t* codes, const uint8_t* lut, size_t D, uint16_t* res) {
    int lv = ref_simd_level();
    if (lv == 2 && (D * 4) % 64 == 0) accumulate_batch_avx512(codes, lut, D, res);
    else if (lv >= 1) accumulate_batch_avx2(codes, lut, D, res);
    else ref_accumulate_batch_scalar(codes, lut, D, res);
}

/* unpack_single_vector, src/simd.rs:915-960: FastScan block -> per-vector bytes */
void ref_unpack_single_vector_bytes(const uint8_t* packed, int vec_idx, size_t dim_bytes, uint8_t* out_bytes) {
    for (size_t col = 0; col < dim_bytes; ++col) {
        const uint8_t* p = packed + col * 32;
        uint8_t hi = 0, lo = 0;
        for (int j = 0; j < 16; ++j) {
            if (KPERM0[j] == vec_idx) { hi = p[j] & 15; lo = p[16 + j] & 15; }
            if (KPERM0[j] + 16 == vec_idx) { hi = p[j] >> 4; lo = p[16 + j] >> 4; }
        }
        out_bytes[col] = (uint8_t)((hi << 4) | lo);
    }
}

/* ------------------------------------------------------------------------- */
/* compute_batch_distances_u16, AVX2 body src/simd.rs:2090-2140              */
/* ------------------------------------------------------------------------- */
void ref_compute_batch_distances(const uint16_t* accu, float delta, float sum_vl,
                                 const float* f_add, const float* f_rescale, const float* f_error,
                                 float g_add, float g_error, float k1x,
                                 float* ip, float* est, float* lb) {
    for (int i = 0; i < 32; ++i) {
        float a = (float)(int32_t)accu[i];
        float ipv = fmaf(delta, a, sum_vl);
        if ((g_variant & REF_VAR_EPI_SCALAR) && !(g_variant & REF_VAR_CONTRACT)) { /* src/simd.rs:2056: `lut_delta * accu + lut_sum_vl` */
            float pm = delta * a;
            ipv = pm + sum_vl;
        }
        ip[i] = ipv;
        float t = ipv + k1x;
        if (g_variant & REF_VAR_CONTRACT) {
            float e0 = f_add[i] + g_add;
            float e1 = fmaf(f_rescale[i], t, e0);
            est[i] = e1;
            lb[i] = fmaf(-f_error[i], g_error, e1);
            continue;
        }
        float r = f_rescale[i] * t;
        float e = f_add[i] + g_add;
        e = e + r;
        est[i] = e;
        float er = f_error[i] * g_error;
        lb[i] = e - er;
    }
}

/* ------------------------------------------------------------------------- */
/* ex-code dot products, AVX-512 bodies src/simd.rs:1835-1915                */
/* lane l accumulates dims 16t+l with one fused multiply-add per t; final    */
/* _mm512_reduce_add_ps = halving tree 16->8->4->2->1 (stdarch).             */
/* ------------------------------------------------------------------------- */
static inline float reduce_add_16(const float* s) {
    float a[8], b[4], c[2];
    for (int i = 0; i < 8; ++i) a[i] = s[i] + s[i + 8];
    for (int i = 0; i < 4; ++i) b[i] = a[i] + a[i + 4];
    for (int i = 0; i < 2; ++i) c[i] = b[i] + b[i + 2];
    return c[0] + c[1];
}

float ref_ip_packed_ex2(const float* q, const uint8_t* code, size_t D) {
    float s[16];
    for (int l = 0; l < 16; ++l) s[l] = 0.0f;
    for (size_t t = 0; t < D / 16; ++t) {
        uint32_t w;
        memcpy(&w, code + t * 4, 4);
        for (int i = 0; i < 4; ++i)
            for (int g = 0; g < 4; ++g) {
                int l = i + 4 * g;
                float cf = (float)((w >> (8 * i + 2 * g)) & 3u);
                s[l] = fmaf(cf, q[t * 16 + l], s[l]);
            }
    }
    return reduce_add_16(s);
}

float ref_ip_packed_ex6(const float* q, const uint8_t* code, size_t D) {
    float s[16];
    for (int l = 0; l < 16; ++l) s[l] = 0.0f;
    for (size_t t = 0; t < D / 16; ++t) {
        uint64_t lo;
        uint32_t hi;
        memcpy(&lo, code + t * 12, 8);
        memcpy(&hi, code + t * 12 + 8, 4);
        for (int l = 0; l < 16; ++l) {
            uint32_t low4 = l < 8 ? (uint32_t)((lo >> (8 * l)) & 15u) : (uint32_t)((lo >> (8 * (l - 8) + 4)) & 15u);
            uint32_t top2 = (hi >> (8 * (l & 3) + 2 * (l >> 2))) & 3u;
            float cf = (float)(low4 | (top2 << 4));
            s[l] = fmaf(cf, q[t * 16 + l], s[l]);
        }
    }
    return reduce_add_16(s);
}

/* The same dot products on REAL AVX-512 instructions: 16-lane _mm512_fmadd_ps per 16-dim step and the compiler's own     */
/* _mm512_reduce_add_ps (gcc avx512fintrin.h: extract 256 + add, extract 128 + add, shuffle {2,3,0,1} + add, lane 0 + lane 1 */
/* — Intel's sequence, the halving tree above).  The codes are unpacked by the scalar emulation's own expressions.  They pin */
/* reduce_add_16 and the lane order to a second definition (round-3 VERDICT, weak 1).  Return 0 when the host lacks AVX-512. */
__attribute__((target("avx512f")))
static float ip_packed_ex_avx512(const float* q, const uint8_t* code, size_t D, uint32_t ex_bits) {
    __m512 acc = _mm512_setzero_ps();
    for (size_t t = 0; t < D / 16; ++t) {
        float cf[16];
        if (ex_bits == 2) {
            uint32_t w;
            memcpy(&w, code + t * 4, 4);
            for (int i = 0; i < 4; ++i)
                for (int g = 0; g < 4; ++g) cf[i + 4 * g] = (float)((w >> (8 * i + 2 * g)) & 3u);
        } else {
            uint64_t lo;
            uint32_t hi;
            memcpy(&lo, code + t * 12, 8);
            memcpy(&hi, code + t * 12 + 8, 4);
            for (int l = 0; l < 16; ++l) {
                uint32_t low4 = l < 8 ? (uint32_t)((lo >> (8 * l)) & 15u) : (uint32_t)((lo >> (8 * (l - 8) + 4)) & 15u);
                uint32_t top2 = (hi >> (8 * (l & 3) + 2 * (l >> 2))) & 3u;
                cf[l] = (float)(low4 | (top2 << 4));
            }
        }
        acc = _mm512_fmadd_ps(_mm512_loadu_ps(cf), _mm512_loadu_ps(q + t * 16), acc);
    }
    return _mm512_reduce_add_ps(acc);
}
__attribute__((target("avx512f")))
static float reduce_add_16_avx512(const float* s) { return _mm512_reduce_add_ps(_mm512_loadu_ps(s)); }
int ref_have_avx512(void) { return __builtin_cpu_supports("avx512f") ? 1 : 0; }
float ref_reduce_add_16(const float* s) { return reduce_add_16(s); }
float ref_reduce_add_16_avx512(const float* s) { return ref_have_avx512() ? reduce_add_16_avx512(s) : 0.0f; }
float ref_ip_packed_ex_avx512(const float* q, const uint8_t* code, size_t D, uint32_t ex_bits) {
    return ref_have_avx512() ? ip_packed_ex_avx512(q, code, D, ex_bits) : 0.0f;
}

/* fast bodies with identical numerics (two 8-lane FMA accumulators = lanes 0-7, 8-15) */
__attribute__((target("avx2,fma")))
static float ip_packed_ex6_fast(const float* q, const uint8_t* code, size_t D) {
    __m256 s0 = _mm256_setzero_ps(), s1 = _mm256_setzero_ps();
    const int64_t MASK4 = 0x0f0f0f0f0f0f0f0fLL;
    const __m128i mask2 = _mm_set1_epi8(0x30);
    for (size_t t = 0; t < D / 16; ++t) {
        int64_t c4;
        int32_t c2v;
        memcpy(&c4, code + t * 12, 8);
        memcpy(&c2v, code + t * 12 + 8, 4);
        __m128i v4 = _mm_set_epi64x((c4 >> 4) & MASK4, c4 & MASK4);
        __m128i v2 = _mm_and_si128(_mm_set_epi32(c2v >> 2, c2v, (int32_t)((uint32_t)c2v << 2), (int32_t)((uint32_t)c2v << 4)), mask2);
        __m128i c6 = _mm_or_si128(v2, v4);
        __m256 f0 = _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(c6));
        __m256 f1 = _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_unpackhi_epi64(c6, c6)));
        s0 = _mm256_fmadd_ps(f0, _mm256_loadu_ps(q + t * 16), s0);
        s1 = _mm256_fmadd_ps(f1, _mm256_loadu_ps(q + t * 16 + 8), s1);
    }
    __m256 a = _mm256_add_ps(s0, s1);                                   /* i + (i+8) */
    __m128 b = _mm_add_ps(_mm256_castps256_ps128(a), _mm256_extractf128_ps(a, 1)); /* i + (i+4) */
    __m128 c = _mm_add_ps(b, _mm_movehl_ps(b, b));                      /* [0]+[2], [1]+[3] */
    return _mm_cvtss_f32(c) + _mm_cvtss_f32(_mm_shuffle_ps(c, c, 0x55));
}

__attribute__((target("avx2,fma")))
static float ip_packed_ex2_fast(const float* q, const uint8_t* code, size_t D) {
    __m256 s0 = _mm256_setzero_ps(), s1 = _mm256_setzero_ps();
    const __m128i mask = _mm_set1_epi8(3);
    for (size_t t = 0; t < D / 16; ++t) {
        int32_t w;
        memcpy(&w, code + t * 4, 4);
        __m128i c = _mm_and_si128(_mm_set_epi32(w >> 6, w >> 4, w >> 2, w), mask);
        __m256 f0 = _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(c));
        __m256 f1 = _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_unpackhi_epi64(c, c)));
        s0 = _mm256_fmadd_ps(f0, _mm256_loadu_ps(q + t * 16), s0);
        s1 = _mm256_fmadd_ps(f1, _mm256_loadu_ps(q + t * 16 + 8), s1);
    }
    __m256 a = _mm256_add_ps(s0, s1);
    __m128 b = _mm_add_ps(_mm256_castps256_ps128(a), _mm256_extractf128_ps(a, 1));
    __m128 c = _mm_add_ps(b, _mm_movehl_ps(b, b));
    return _mm_cvtss_f32(c) + _mm_cvtss_f32(_mm_shuffle_ps(c, c, 0x55));
}

/* ip_packed_ex{2,6}_f32_avx2, src/simd.rs:1722-1825, in scalar C: lane l of ONE 8-lane accumulator takes dims 16t+l and then   */
/* 16t+8+l (two FMAs per step); hsum: lo128 + hi128, movehl, shuffle 0x55                                                     */
static void ex_codes16(const uint8_t* code, size_t t, uint32_t ex_bits, float* cf) {
    if (ex_bits == 2) {
        uint32_t w;
        memcpy(&w, code + t * 4, 4);
        for (int i = 0; i < 4; ++i)
            for (int g = 0; g < 4; ++g) cf[i + 4 * g] = (float)((w >> (8 * i + 2 * g)) & 3u);
    } else {
        uint64_t lo;
        uint32_t hi;
        memcpy(&lo, code + t * 12, 8);
        memcpy(&hi, code + t * 12 + 8, 4);
        for (int l = 0; l < 16; ++l) {
            uint32_t low4 = l < 8 ? (uint32_t)((lo >> (8 * l)) & 15u) : (uint32_t)((lo >> (8 * (l - 8) + 4)) & 15u);
            uint32_t top2 = (hi >> (8 * (l & 3) + 2 * (l >> 2))) & 3u;
            cf[l] = (float)(low4 | (top2 << 4));
        }
    }
}
float ref_ip_packed_ex_avx2_order(const float* q, const uint8_t* code, size_t D, uint32_t ex_bits) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cf[16];
    for (size_t t = 0; t < D / 16; ++t) {
        ex_codes16(code, t, ex_bits, cf);
        for (int l = 0; l < 8; ++l) s[l] = fmaf(cf[l], q[t * 16 + l], s[l]);
        for (int l = 0; l < 8; ++l) s[l] = fmaf(cf[8 + l], q[t * 16 + 8 + l], s[l]);
    }
    float b[4];
    for (int i = 0; i < 4; ++i) b[i] = s[i] + s[i + 4]; /* sum_lo + sum_hi */
    float c0 = b[0] + b[2], c1 = b[1] + b[3];           /* + movehl */
    return c0 + c1;                                     /* + shuffle 0x55 */
}
/* the same on REAL AVX2 instructions, as src/simd.rs:1722-1825 issues them (pins the scalar restatement above) */
__attribute__((target("avx2,fma")))
float ref_ip_packed_ex_avx2_real(const float* q, const uint8_t* code, size_t D, uint32_t ex_bits) {
    __m256 sum = _mm256_setzero_ps();
    const int64_t MASK4 = 0x0f0f0f0f0f0f0f0fLL;
    for (size_t t = 0; t < D / 16; ++t) {
        __m128i c;
        if (ex_bits == 2) {
            int32_t w;
            memcpy(&w, code + t * 4, 4);
            c = _mm_and_si128(_mm_set_epi32(w >> 6, w >> 4, w >> 2, w), _mm_set1_epi8(3));
        } else {
            int64_t c4;
            int32_t c2v;
            memcpy(&c4, code + t * 12, 8);
            memcpy(&c2v, code + t * 12 + 8, 4);
            __m128i v4 = _mm_set_epi64x((c4 >> 4) & MASK4, c4 & MASK4);
            __m128i v2 = _mm_and_si128(_mm_set_epi32(c2v >> 2, c2v, (int32_t)((uint32_t)c2v << 2), (int32_t)((uint32_t)c2v << 4)), _mm_set1_epi8(0x30));
            c = _mm_or_si128(v2, v4);
        }
        __m256 f0 = _mm256_cvtepi32_ps(_mm256_cvtepi8_epi32(c));
        sum = _mm256_fmadd_ps(f0, _mm256_loadu_ps(q + t * 16), sum);
        __m256 f1 = _mm256_cvtepi32_ps(_mm256_cvtepi8_epi32(_mm_unpackhi_epi64(c, c)));
        sum = _mm256_fmadd_ps(f1, _mm256_loadu_ps(q + t * 16 + 8), sum);
    }
    __m128 s128 = _mm_add_ps(_mm256_castps256_ps128(sum), _mm256_extractf128_ps(sum, 1));
    __m128 s64 = _mm_add_ps(s128, _mm_movehl_ps(s128, s128));
    __m128 s32 = _mm_add_ss(s64, _mm_shuffle_ps(s64, s64, 0x55));
    return _mm_cvtss_f32(s32);
}
/* ip_packed_ex{2,6}_f32_scalar, src/simd.rs:1615-1713: one running sum, `sum += c * q` (mul, then add) */
float ref_ip_packed_ex_scalar_order(const float* q, const uint8_t* code, size_t D, uint32_t ex_bits) {
    float sum = 0.0f, cf[16];
    for (size_t t = 0; t < D / 16; ++t) {
        ex_codes16(code, t, ex_bits, cf);
        if (ex_bits == 2) { /* i = 0..3: codes i, i+4, i+8, i+12 (:1636-1647) */
            for (int i = 0; i < 4; ++i)
                for (int g = 0; g < 4; ++g) {
                    float p = cf[i + 4 * g] * q[t * 16 + i + 4 * g];
                    sum = sum + p;
                }
        } else { /* i = 0..15 (:1703-1706) */
            for (int i = 0; i < 16; ++i) {
                float p = cf[i] * q[t * 16 + i];
                sum = sum + p;
            }
        }
    }
    return sum;
}

/* select_excode_ipfunc, src/simd.rs:3205-3215 */
float ref_ex_dot(const float* q, const uint8_t* code, size_t D, uint32_t ex_bits) {
    if (ex_bits == 0) return 0.0f;
    if (g_variant & REF_VAR_EX_SCALAR) return ref_ip_packed_ex_scalar_order(q, code, D, ex_bits);
    if (g_variant & REF_VAR_EX_AVX2) return ref_ip_packed_ex_avx2_order(q, code, D, ex_bits);
    if (ref_simd_level() >= 1) return ex_bits == 2 ? ip_packed_ex2_fast(q, code, D) : ip_packed_ex6_fast(q, code, D);
    return ex_bits == 2 ? ref_ip_packed_ex2(q, code, D) : ref_ip_packed_ex6(q, code, D);
}

/* ------------------------------------------------------------------------- */
/* Rust std BinaryHeap<HeapEntry> (max-heap on distance via total_cmp,        */
/* src/ivf.rs:904-931) — sift order restated from alloc::collections.        */
/* ------------------------------------------------------------------------- */
typedef struct { uint64_t id; float distance; } hent;
typedef struct { hent* d; size_t len; } heap_t;

static inline int h_le(const hent* a, const hent* b) { return total_cmp(a->distance, b->distance) <= 0; }
static inline int h_lt(const hent* a, const hent* b) { return total_cmp(a->distance, b->distance) < 0; }

static size_t sift_up(heap_t* h, size_t start, size_t pos) {
    hent e = h->d[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (h_le(&e, &h->d[parent])) break;
        h->d[pos] = h->d[parent];
        pos = parent;
    }
    h->d[pos] = e;
    return pos;
}
static void heap_push(heap_t* h, hent e) {
    size_t old = h->len;
    h->d[h->len++] = e;
    sift_up(h, 0, old);
}
static void sift_down_to_bottom(heap_t* h, size_t pos) {
    size_t end = h->len, start = pos;
    hent e = h->d[pos];
    size_t child = 2 * pos + 1;
    size_t lim = end >= 2 ? end - 2 : 0;
    while (child <= lim && end >= 2) {
        child += h_le(&h->d[child], &h->d[child + 1]);
        h->d[pos] = h->d[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1) {
        h->d[pos] = h->d[child];
        pos = child;
    }
    h->d[pos] = e;
    sift_up(h, start, pos);
}
static void heap_pop(heap_t* h) {
    if (h->len == 0) return;
    hent item = h->d[--h->len];
    if (h->len > 0) {
        hent top = h->d[0];
        h->d[0] = item;
        (void)top;
        sift_down_to_bottom(h, 0);
    }
}
static void sift_down_range(heap_t* h, size_t pos, size_t end) {
    hent e = h->d[pos];
    size_t child = 2 * pos + 1;
    size_t lim = end >= 2 ? end - 2 : 0;
    while (child <= lim && end >= 2) {
        child += h_le(&h->d[child], &h->d[child + 1]);
        if (!h_lt(&e, &h->d[child])) { h->d[pos] = e; return; }
        h->d[pos] = h->d[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1 && h_lt(&e, &h->d[child])) {
        h->d[pos] = h->d[child];
        pos = child;
    }
    h->d[pos] = e;
}
static void heap_into_sorted(heap_t* h) {
    size_t end = h->len;
    while (end > 1) {
        --end;
        hent t = h->d[0]; h->d[0] = h->d[end]; h->d[end] = t;
        sift_down_range(h, 0, end);
    }
}

/* The top-k bookkeeping of search_cluster_v2_batched alone (src/ivf.rs:2116-2126: push, pop while len > top_k; then            */
/* into_sorted_vec, :1874-1878) over a given sequence of (distance, id) pushes — exported so that tests/test_oracle_kat.py can   */
/* pin the sift order (which decides exact ties) against CPython's heapq, an independent implementation of the same published  */
/* algorithm (push = sift up from the end; pop = move the last element to the root, sift it to the bottom, sift it back up).    */
int ref_heap_trace(const float* dist, const uint64_t* ids, size_t n, uint32_t top_k, uint64_t* out_ids, float* out_dist,
                   uint32_t* out_len) {
    heap_t heap;
    heap.d = (hent*)malloc(((size_t)top_k + 2) * sizeof(hent));
    heap.len = 0;
    if (!heap.d) return RBQ_IO;
    for (size_t i = 0; i < n; ++i) {
        hent e; e.id = ids[i]; e.distance = dist[i];
        heap_push(&heap, e);
        if (heap.len > top_k) heap_pop(&heap);
    }
    heap_into_sorted(&heap);
    for (size_t i = 0; i < heap.len; ++i) { out_ids[i] = heap.d[i].id; out_dist[i] = heap.d[i].distance; }
    *out_len = (uint32_t)heap.len;
    free(heap.d);
    return RBQ_OK;
}

/* ------------------------------------------------------------------------- */
/* probe selection, src/ivf.rs:1782-1835                                      */
/* ------------------------------------------------------------------------- */
typedef struct { int64_t key; } probe_key; /* (ordered score << 32) | cid ; ascending */

static int cmp_i64(const void* a, const void* b) {
    int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return (x > y) - (x < y);
}

/* after the call a[k] is the k-th smallest, everything before it is smaller, everything after it larger (quickselect with a
 * median-of-three pivot on distinct keys) */
static void select_nth_i64(int64_t* a, size_t n, size_t k) {
    size_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        int64_t x = a[lo], y = a[mid], z = a[hi];
        const int64_t pivot = x < y ? (y < z ? y : (x < z ? z : x)) : (x < z ? x : (y < z ? z : y));
        size_t i = lo, j = hi;
        while (i <= j) {
            while (a[i] < pivot) ++i;
            while (a[j] > pivot) --j;
            if (i <= j) { const int64_t t = a[i]; a[i] = a[j]; a[j] = t; ++i; if (j == 0) break; --j; }
        }
        if (k <= j) hi = j;
        else if (k >= i) lo = i;
        else return;
    }
}

/* returns nprobe list ids in probe order */
size_t ref_select_probes(const rbq_header* h, const rbq_list_view* lists, const float* rq,
                         uint32_t nprobe_in, uint32_t* out_cids) {
    size_t nl = h->n_lists, D = h->padded_dim;
    int64_t* keys = (int64_t*)scratch(1, sizeof(int64_t) * nl);
    for (size_t c = 0; c < nl; ++c) {
        float s = h->metric == RBQ_METRIC_L2 ? ref_l2_distance_sqr(rq, lists[c].centroid, D)
                                             : ref_dot(rq, lists[c].centroid, D);
        int32_t k = total_key(s);
        if (h->metric == RBQ_METRIC_IP) k = ~k; /* descending score: b.total_cmp(a) */
        keys[c] = (int64_t)(((uint64_t)(uint32_t)k << 32) | (uint64_t)(uint32_t)c); /* (the bit pattern of k * 2^32 + c; no shift of a negative value) */
    }
    size_t nprobe = nprobe_in < 1 ? 1 : nprobe_in;
    if (nprobe > nl) nprobe = nl;
    /* select_nth_unstable_by(nprobe) (src/ivf.rs:1808: index nprobe, only when nprobe < len) then sort of the first nprobe
     * (:1808-1823).  Selecting at index nprobe - 1 here leaves the same prefix SET, and the keys are distinct (the list id is
     * part of them), so the sorted prefix is the full sort's either way. */
    if (nprobe < nl) select_nth_i64(keys, nl, nprobe - 1);
    qsort(keys, nprobe, sizeof(int64_t), cmp_i64);
    for (size_t i = 0; i < nprobe; ++i) out_cids[i] = (uint32_t)(keys[i] & 0xffffffff);
    return nprobe;
}

/* ------------------------------------------------------------------------- */
/* search_fastscan + search_cluster_v2_batched, src/ivf.rs:1754-2129          */
/* ------------------------------------------------------------------------- */
static inline int filter_contains(const uint32_t* words, uint64_t nbits, uint64_t id) {
    uint32_t i = (uint32_t)id; /* `vector_id as u32`, src/ivf.rs:2019 */
    if ((uint64_t)i >= nbits) return 0;
    return (words[i >> 5] >> (i & 31)) & 1;
}

/* ref_search_lists: the same search; additionally, per probed list in probe order, its id and the number of its vectors that were NOT */
/* skipped by the lower bound (evaluated: `lower_bound < distk` at their moment, src/ivf.rs:2045-2058) — what the GPU's lazy selection   */
/* must find to be 0 for every list it drops as a whole (tests: the `lazy_audit` shadow check).                                        */
static __thread uint32_t* g_trace_cids;
static __thread uint32_t* g_trace_eval;
static __thread uint32_t* g_trace_n;
int ref_search_lists(const rbq_header* h, const rbq_list_view* lists, const float* query, uint32_t query_dim,
                     uint32_t top_k, uint32_t nprobe_in, const uint32_t* filter_words, uint64_t filter_nbits,
                     uint64_t* out_ids, float* out_scores, uint32_t* out_count,
                     uint32_t* probe_cids, uint32_t* probe_evaluated, uint32_t* n_probed) {
    g_trace_cids = probe_cids; g_trace_eval = probe_evaluated; g_trace_n = n_probed;
    int rc = ref_search(h, lists, query, query_dim, top_k, nprobe_in, filter_words, filter_nbits, out_ids, out_scores, out_count, NULL);
    g_trace_cids = NULL; g_trace_eval = NULL; g_trace_n = NULL;
    return rc;
}

int ref_search(const rbq_header* h, const rbq_list_view* lists, const float* query, uint32_t query_dim,
               uint32_t top_k, uint32_t nprobe_in, const uint32_t* filter_words, uint64_t filter_nbits,
               uint64_t* out_ids, float* out_scores, uint32_t* out_count, rbq_diag* diag) {
    if (out_count) *out_count = 0;
    if (diag) memset(diag, 0, sizeof *diag);
    uint64_t total = 0;
    for (size_t c = 0; c < h->n_lists; ++c) total += lists[c].n;
    if (total == 0) return RBQ_EMPTY_INDEX;
    if (query_dim != h->dim) return RBQ_DIMENSION_MISMATCH;
    size_t D = h->padded_dim;
    if (D > 2048 || (h->ex_bits != 0 && h->ex_bits != 2 && h->ex_bits != 6)) return RBQ_INVALID_CONFIG;
    for (uint32_t i = 0; i < top_k; ++i) {
        if (out_ids) out_ids[i] = UINT64_MAX;
        if (out_scores) out_scores[i] = NAN;
    }

    float* rq = (float*)scratch(2, sizeof(float) * D);
    uint8_t* lut8 = (uint8_t*)scratch(3, D * 4);
    uint32_t* cids = (uint32_t*)scratch(4, sizeof(uint32_t) * h->n_lists);
    ref_rotate(h, query, rq);
    ref_query_consts qc;
    ref_query_precompute(rq, D, h->ex_bits, &qc);
    float lut_delta, lut_sum_vl;
    ref_query_lut(rq, D, lut8, &lut_delta, &lut_sum_vl);
    size_t nprobe = ref_select_probes(h, lists, rq, nprobe_in, cids);
    if (g_trace_n) *g_trace_n = (uint32_t)nprobe;

    if (top_k == 0) return RBQ_OK;

    heap_t heap;
    heap.d = (hent*)scratch(5, sizeof(hent) * ((size_t)top_k + 1));
    heap.len = 0;
    size_t stride = D * 4 + 384, ex_bytes = D * h->ex_bits / 8;

    for (size_t r = 0; r < nprobe; ++r) {
        const rbq_list_view* cl = &lists[cids[r]];
        float centroid_dist = ref_l2_distance_sqr(rq, cl->centroid, D);
        float dot_qc = ref_dot(rq, cl->centroid, D);
        float g_add = h->metric == RBQ_METRIC_L2 ? centroid_dist : -dot_qc;
        float g_error = sqrtf(centroid_dist);
        size_t nb = (cl->n + 31) / 32;
        if (g_trace_cids) { g_trace_cids[r] = cids[r]; g_trace_eval[r] = 0; }
        for (size_t b = 0; b < nb; ++b) {
            const uint8_t* rec = cl->batch_data + b * stride;
            const float* f_add = (const float*)(rec + D * 4);
            const float* f_rescale = f_add + 32;
            const float* f_error = f_rescale + 32;
            uint16_t accu[32];
            float ip[32], est[32], lb[32];
            ref_accumulate_batch(rec, lut8, D, accu);
            ref_compute_batch_distances(accu, lut_delta, lut_sum_vl, f_add, f_rescale, f_error,
                                        g_add, g_error, qc.k1x_sum_q, ip, est, lb);
            size_t start = b * 32, end = start + 32 < cl->n ? start + 32 : cl->n;
            for (size_t gi = start; gi < end; ++gi) {
                size_t i = gi - start;
                uint64_t vid = cl->ids[gi];
                if (filter_words && !filter_contains(filter_words, filter_nbits, vid)) continue;
                float lower = lb[i];
                if (!isfinite(lower))
                    lower = h->metric == RBQ_METRIC_L2 ? 0.0f : -(dot_qc + qc.query_norm);
                float distk = heap.len < top_k ? INFINITY : heap.d[0].distance;
                if (lower >= distk) {
                    if (diag) diag->skipped_by_lower_bound++;
                    continue;
                }
                if (g_trace_eval) g_trace_eval[r]++;
                float distance = est[i];
                if (h->ex_bits > 0) {
                    if (diag) diag->extended_evaluations++;
                    float ex_dot = ref_ex_dot(rq, cl->ex_codes + gi * ex_bytes, D, h->ex_bits);
                    float t = qc.binary_scale * ip[i];
                    t = t + ex_dot;
                    t = t + qc.kbx_sum_q;
                    float a = cl->f_add_ex[gi] + g_add;
                    float m = cl->f_rescale_ex[gi] * t;
                    distance = a + m;
                    if (g_variant & REF_VAR_CONTRACT) { /* `binary_scale * ip + ex_dot` and `f_rescale_ex * t + a` fused */
                        float t2 = fmaf(qc.binary_scale, ip[i], ex_dot);
                        t2 = t2 + qc.kbx_sum_q;
                        distance = fmaf(cl->f_rescale_ex[gi], t2, a);
                    }
                }
                if (!isfinite(distance)) continue;
                if (diag) diag->estimated++;
                hent e = {vid, distance};
                heap_push(&heap, e);
                if (heap.len > top_k) heap_pop(&heap);
            }
        }
    }
    heap_into_sorted(&heap);
    /* stable re-sort (src/ivf.rs:1880-1883) is the identity on an ascending-
     * distance vector: L2 key = distance; IP key = -distance descending. */
    for (size_t i = 0; i < heap.len; ++i) {
        if (out_ids) out_ids[i] = heap.d[i].id;
        if (out_scores) out_scores[i] = h->metric == RBQ_METRIC_L2 ? heap.d[i].distance : -heap.d[i].distance;
    }
    if (out_count) *out_count = (uint32_t)heap.len;
    return RBQ_OK;
}

/* batch_search, src/ivf.rs:1743-1752: one query per worker (Rayon par_iter) */
int ref_search_batch(const rbq_header* h, const rbq_list_view* lists, const float* queries, uint64_t nq,
                     uint32_t query_dim, uint32_t top_k, uint32_t nprobe,
                     const uint32_t* filter_words, uint64_t filter_nbits,
                     uint64_t* out_ids, float* out_scores, uint32_t* out_counts, rbq_diag* diag,
                     int nthreads) {
    int rc_all = RBQ_OK;
    ref_simd_level();
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t q = 0; q < (int64_t)nq; ++q) {
        int rc = ref_search(h, lists, queries + (size_t)q * query_dim, query_dim, top_k, nprobe,
                            filter_words, filter_nbits, out_ids + (size_t)q * top_k,
                            out_scores + (size_t)q * top_k, out_counts + q, diag ? diag + q : NULL);
        if (rc != RBQ_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
            rc_all = rc;
        }
    }
    return rc_all;
}

int ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* search_naive (the reference's own test oracle), src/ivf.rs:2143-2240:      */
/* exact-float binary dot, no LUT, no pruning, full sort.                     */
/* ------------------------------------------------------------------------- */
typedef struct { uint64_t id; float score; size_t seq; } naive_cand;
static int g_naive_desc;
static int cmp_naive(const void* a, const void* b) {
    const naive_cand* x = (const naive_cand*)a; const naive_cand* y = (const naive_cand*)b;
    int c = g_naive_desc ? total_cmp(y->score, x->score) : total_cmp(x->score, y->score);
    if (c) return c;
    return (x->seq > y->seq) - (x->seq < y->seq); /* stable */
}

int ref_search_naive(const rbq_header* h, const rbq_list_view* lists, const float* query, uint32_t query_dim,
                     uint32_t top_k, uint32_t nprobe_in, uint64_t* out_ids, float* out_scores, uint32_t* out_count) {
    *out_count = 0;
    uint64_t total = 0;
    for (size_t c = 0; c < h->n_lists; ++c) total += lists[c].n;
    if (total == 0) return RBQ_EMPTY_INDEX;
    if (query_dim != h->dim) return RBQ_DIMENSION_MISMATCH;
    size_t D = h->padded_dim, dim_bytes = D / 8;
    float* rq = (float*)malloc(sizeof(float) * D);
    uint32_t* cids = (uint32_t*)malloc(sizeof(uint32_t) * h->n_lists);
    uint8_t* vb = (uint8_t*)malloc(dim_bytes);
    ref_rotate(h, query, rq);
    /* naive uses a stable sort on score only (sort_by without cid key): ties keep cid order,
     * identical to the (score,cid) key. */
    size_t nprobe = ref_select_probes(h, lists, rq, nprobe_in, cids);
    float sum_q = -0.0f;
    for (size_t i = 0; i < D; ++i) sum_q = sum_q + rq[i];
    float c1 = -0.5f, scale = (float)(1 << h->ex_bits), cb = -((float)(1 << h->ex_bits) - 0.5f);
    size_t cap = 0;
    for (size_t r = 0; r < nprobe; ++r) cap += lists[cids[r]].n;
    naive_cand* cand = (naive_cand*)malloc(sizeof(naive_cand) * (cap ? cap : 1));
    size_t nc = 0, stride = D * 4 + 384, ex_bytes = D * h->ex_bits / 8;
    for (size_t r = 0; r < nprobe; ++r) {
        const rbq_list_view* cl = &lists[cids[r]];
        float centroid_dist = ref_l2_distance_sqr(rq, cl->centroid, D);
        float dot_qc = ref_dot(rq, cl->centroid, D);
        float g_add = h->metric == RBQ_METRIC_L2 ? centroid_dist : -dot_qc;
        for (size_t v = 0; v < cl->n; ++v) {
            const uint8_t* rec = cl->batch_data + (v / 32) * stride;
            const float* f_add = (const float*)(rec + D * 4);
            const float* f_rescale = f_add + 32;
            ref_unpack_single_vector_bytes(rec, (int)(v % 32), dim_bytes, vb);
            float bdot = 0.0f;
            for (size_t i = 0; i < D; ++i) {
                float bit = (float)((vb[i / 8] >> (7 - (i % 8))) & 1);
                float p = bit * rq[i];
                bdot = bdot + p;
            }
            float bt = bdot + c1 * sum_q;
            float distance = f_add[v % 32] + g_add + f_rescale[v % 32] * bt;
            if (h->ex_bits > 0) {
                float ex_dot = ref_ex_dot(rq, cl->ex_codes + v * ex_bytes, D, h->ex_bits);
                float tt = scale * bdot + ex_dot + cb * sum_q;
                distance = cl->f_add_ex[v] + g_add + cl->f_rescale_ex[v] * tt;
            }
            if (!isfinite(distance)) continue;
            cand[nc].id = cl->ids[v];
            cand[nc].score = h->metric == RBQ_METRIC_L2 ? distance : -distance;
            cand[nc].seq = nc;
            ++nc;
        }
    }
    g_naive_desc = h->metric == RBQ_METRIC_IP;
    qsort(cand, nc, sizeof(naive_cand), cmp_naive);
    size_t k = top_k < nc ? top_k : nc;
    for (size_t i = 0; i < k; ++i) { out_ids[i] = cand[i].id; out_scores[i] = cand[i].score; }
    *out_count = (uint32_t)k;
    free(cand); free(vb); free(cids); free(rq);
    return RBQ_OK;
}

/* ------------------------------------------------------------------------- */
/* MSTG posting-list scan: search_posting_list_fastscan,        */
/* src/mstg/index.rs:216-330, + the top-k partial sort of MstgIndex::search,  */
/* :185-205.  No query rotation (QueryContext::new takes the raw query,       */
/* src/fastscan.rs:159-176); f_error row and g_error are zero; non-finite     */
/* estimates are dropped; L2 estimates are clamped to >= 0.  The reference's  */
/* select_nth_unstable/sort_unstable leave ties unordered; this restatement   */
/* breaks them by (list order, vector order).                                  */
/* ------------------------------------------------------------------------- */
int ref_posting_scan(const rbq_header* h, const rbq_list_view* lists, const float* query, uint32_t query_dim,
                     uint32_t top_k, const uint32_t* list_ids, uint32_t n_sel,
                     uint64_t* out_ids, float* out_scores, uint32_t* out_count) {
    *out_count = 0;
    if (query_dim != h->dim || h->dim != h->padded_dim) return RBQ_DIMENSION_MISMATCH;
    size_t D = h->padded_dim, stride = D * 4 + 384;
    if (D > 2048 || D % 16) return RBQ_INVALID_CONFIG;
    for (uint32_t i = 0; i < top_k; ++i) { out_ids[i] = UINT64_MAX; out_scores[i] = NAN; }
    uint8_t* lut8 = (uint8_t*)malloc(D * 4);
    float delta, sum_vl;
    ref_query_lut(query, D, lut8, &delta, &sum_vl);
    float sum_q = -0.0f;
    for (size_t i = 0; i < D; ++i) sum_q = sum_q + query[i];
    const float k1x = -0.5f * sum_q;
    size_t cap = 0;
    for (uint32_t r = 0; r < n_sel; ++r) if (list_ids[r] < h->n_lists) cap += lists[list_ids[r]].n;
    naive_cand* cand = (naive_cand*)malloc(sizeof(naive_cand) * (cap ? cap : 1));
    size_t nc = 0;
    float zeros[32];
    memset(zeros, 0, sizeof zeros);
    for (uint32_t r = 0; r < n_sel; ++r) {
        if (list_ids[r] >= h->n_lists) continue;
        const rbq_list_view* pl = &lists[list_ids[r]];
        if (pl->n == 0) continue;
        float g_add = h->metric == RBQ_METRIC_L2 ? ref_l2_distance_sqr(query, pl->centroid, D) : -ref_dot(query, pl->centroid, D);
        size_t nb = (pl->n + 31) / 32;
        for (size_t b = 0; b < nb; ++b) {
            const uint8_t* rec = pl->batch_data + b * stride;
            const float* f_add = (const float*)(rec + D * 4);
            const float* f_rescale = f_add + 32;
            uint16_t accu[32];
            float ip[32], est[32], lb[32];
            ref_accumulate_batch(rec, lut8, D, accu);
            ref_compute_batch_distances(accu, delta, sum_vl, f_add, f_rescale, zeros, g_add, 0.0f, k1x, ip, est, lb);
            size_t start = b * 32, end = start + 32 < pl->n ? start + 32 : pl->n;
            for (size_t gi = start; gi < end; ++gi) {
                float d = est[gi - start];
                if (!isfinite(d)) continue;
                if (h->metric == RBQ_METRIC_L2 && !(d > 0.0f)) d = 0.0f; /* distance.max(0.0) */
                cand[nc].id = pl->ids[gi];
                cand[nc].score = d;
                cand[nc].seq = nc;
                ++nc;
            }
        }
    }
    g_naive_desc = 0;
    qsort(cand, nc, sizeof(naive_cand), cmp_naive);
    size_t k = top_k < nc ? top_k : nc;
    for (size_t i = 0; i < k; ++i) { out_ids[i] = cand[i].id; out_scores[i] = cand[i].score; }
    *out_count = (uint32_t)k;
    free(cand); free(lut8);
    return RBQ_OK;
}

int ref_posting_scan_batch(const rbq_header* h, const rbq_list_view* lists, const float* queries, uint64_t nq,
                           uint32_t query_dim, uint32_t top_k, const uint32_t* list_ids, const uint32_t* list_counts,
                           uint32_t max_lists, uint64_t* out_ids, float* out_scores, uint32_t* out_counts, int nthreads) {
    int rc_all = RBQ_OK;
    ref_simd_level();
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t q = 0; q < (int64_t)nq; ++q) {
        uint32_t n = list_counts[q] < max_lists ? list_counts[q] : max_lists;
        int rc = ref_posting_scan(h, lists, queries + (size_t)q * query_dim, query_dim, top_k,
                                  list_ids + (size_t)q * max_lists, n, out_ids + (size_t)q * top_k,
                                  out_scores + (size_t)q * top_k, out_counts + q);
        if (rc != RBQ_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
            rc_all = rc;
        }
    }
    return rc_all;
}
