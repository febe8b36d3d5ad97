// rank_mfma.hpp — probe selection through an MFMA shortlist (gfx950).
//
// Reference: search_fastscan's centroid ranking (src/ivf.rs:1782-1835) with math::l2_distance_sqr / dot in
// their AVX2 lane order (src/math.rs:154-245).  Ranking all nq x nlist pairs in that exact order costs
// 3*nq*nlist*D unfused VALU ops; instead
//   k_rank_bf16_db one split-bf16 MFMA GEMM gives APPROXIMATE scores
//                  A(q,c) = |q|^2 + |c|^2 - 2 q.c   (L2)   or   q.c   (IP)
//                  with x = hi + lo + r (hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-16 |x|) and
//                  q.c ~ qh.ch + qh.cl + ql.ch on v_mfma_f32_32x32x16_bf16 (products exact, f32 accumulate);
//                  the dropped terms are <= 3.01 * 2^-16 * sum|q_i||c_i| <= 1.51 * 2^-16 (|q|^2 + |c|^2)
//   k_rank_mfma    the same scores from one f32 MFMA GEMM (v_mfma_f32_32x32x2_f32), used when D % 64 != 0
//   k_select_mfma  per query: nprobe-th approximate score tau, shortlist {c : A(c) within 2*eps of tau},
//                  EXACT canonical-order scores for the shortlist only, exact (score, cid) sort.
// eps bounds |A - canonical| rigorously (standard rounding-error model, n = D, u = 2^-24):
//   |canonical - s*| <= gamma_n s*  (positive terms),  |A - s*| <= 2 gamma_{n+2} (|q|^2 + |c|^2)
//   =>  |A - canonical| <= 4 gamma_{n+2} (|q|^2 + |c|^2)   (s* <= 2(|q|^2+|c|^2));  eps uses 6 n u (..), > that.
// The split-bf16 GEMM accumulates 3n exact products: 2 gamma_{3n+2} * (sum|q_i||c_i| <= (|q|^2+|c|^2)/2) plus the
// norm terms stays below 4 n u (|q|^2+|c|^2) even if the MFMA adder truncates (u -> 2u), and the split itself
// adds 2 * 1.51 * 2^-16 (|q|^2+|c|^2) to the L2 score: eps = (6 n u + 4 * 2^-16)(|q|^2 + max|c|^2).
// Any true top-nprobe member has canonical <= (largest canonical among the approximate top-nprobe) <= tau+eps,
// hence A <= tau + 2 eps: it is in the shortlist.  If the shortlist overflows its LDS capacity (many
// near-equal scores) or a score is not finite, the query falls back to canonical scores for ALL lists.
#pragma once
#include "kernels.hpp"

namespace rbq {

#ifndef RBQ_CANON_UNROLL
#define RBQ_CANON_UNROLL 4 // 16-byte loads a lane keeps in flight in the canonical rescoring pass
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));

// TW = 2: 128x128 tile (64x64 per wave, 4 MFMA per k-pair); TW = 1: 64x64 tile (32x32 per wave) for small
// problems where 128x128 tiles would leave most CUs idle.
template <int METRIC, int TW>
__global__ __launch_bounds__(256) void k_rank_mfma(const float* __restrict__ rot, const float* __restrict__ cent,
                                                   const QueryConsts* __restrict__ consts,
                                                   const float* __restrict__ cnorm2, uint32_t nq, uint32_t nlist,
                                                   uint32_t D, float* __restrict__ scores) {
    constexpr int BM = 64 * TW, BN = 64 * TW, BK = 32, LD = BK + 1; // +1: 32 rows of one k column hit 32 distinct banks
    __shared__ float As[BM][LD];
    __shared__ float Bs[BN][LD];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, wm = w >> 1, wn = w & 1u;
    const uint32_t q0 = blockIdx.y * BM, c0 = blockIdx.x * BN;
    f32x16 acc[TW][TW];
#pragma unroll
    for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int b = 0; b < TW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    for (uint32_t k0 = 0; k0 < D; k0 += BK) {
#pragma unroll
        for (int i = 0; i < 2 * TW; ++i) { // BM rows x 32 floats = BM*8 float4 per operand
            const uint32_t idx = tid + 256u * i, row = idx >> 3, c4 = (idx & 7u) * 4u, k = k0 + c4;
            float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
            if (q0 + row < nq && k < D) av = *reinterpret_cast<const float4*>(rot + (size_t)(q0 + row) * D + k);
            if (c0 + row < nlist && k < D) bv = *reinterpret_cast<const float4*>(cent + (size_t)(c0 + row) * D + k);
            As[row][c4] = av.x; As[row][c4 + 1] = av.y; As[row][c4 + 2] = av.z; As[row][c4 + 3] = av.w;
            Bs[row][c4] = bv.x; Bs[row][c4 + 1] = bv.y; Bs[row][c4 + 2] = bv.z; Bs[row][c4 + 3] = bv.w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const uint32_t r = lane & 31u, kx = kk + (lane >> 5);
            float av[TW], bv[TW];
#pragma unroll
            for (int a = 0; a < TW; ++a) {
                av[a] = As[wm * 32 * TW + a * 32 + r][kx];
                bv[a] = Bs[wn * 32 * TW + a * 32 + r][kx];
            }
#pragma unroll
            for (int a = 0; a < TW; ++a)
#pragma unroll
                for (int b = 0; b < TW; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int b = 0; b < TW; ++b) {
            const uint32_t c = c0 + wn * 32 * TW + b * 32 + (lane & 31u);
            const float cn = (METRIC == 0 && c < nlist) ? cnorm2[c] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t qi = q0 + wm * 32 * TW + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (qi < nq && c < nlist) {
                    const float dot = acc[a][b][r];
                    float v = dot;
                    if (METRIC == 0) {
                        const float qn = consts[qi].qnorm2;
                        v = fmaf(-2.0f, dot, qn + cn);
                    }
                    scores[(size_t)qi * nlist + c] = v;
                }
            }
        }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma_x8(bf16x8 a, bf16x8 b, f32x16 c) { // one K = 16 step: c += a(32x16) * b(16x32)
#ifndef RBQ_MFMA_X8
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); // gfx950's double-rate form
#else
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 as = __builtin_bit_cast(s16x8, a), bs = __builtin_bit_cast(s16x8, b);
    const s16x4 a0 = {as[0], as[1], as[2], as[3]}, a1 = {as[4], as[5], as[6], as[7]};
    const s16x4 b0 = {bs[0], bs[1], bs[2], bs[3]}, b1 = {bs[4], bs[5], bs[6], bs[7]};
    c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a0, b0, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a1, b1, c, 0, 0, 0);
#endif
}

// Split-bf16 GEMM (32*TM*WM x 32*TN*WN tile per workgroup of WM*WN wavefronts, each wavefront a 32*TM x 32*TN
// sub-tile of 32x32 MFMA blocks) with K slabs of 32 double-buffered in LDS: slab s+1 is written (from the registers
// the previous iteration filled) and slab s+2 requested BEFORE the MFMAs of slab s, and one barrier per slab closes
// the iteration — the staging of the next slab runs under the matrix work of this one instead of between two
// barriers.
// Rows are padded to 80 bytes (16 lanes of a ds_read_b128 group: 16 distinct 4-bank sets).  D % 32 == 0.
// dynamic LDS: 2 x { Ah | Al [BM][80 B] | Bh | Bl [BN][80 B] }
template <int METRIC, int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void k_rank_bf16_db(const uint16_t* __restrict__ rot_hi, const uint16_t* __restrict__ rot_lo,
                                                      const uint16_t* __restrict__ cent_hi, const uint16_t* __restrict__ cent_lo,
                                                      const QueryConsts* __restrict__ consts,
                                                      const float* __restrict__ cnorm2, uint32_t nq, uint32_t nlist,
                                                      uint32_t D, float* __restrict__ scores) {
    constexpr int NT = 64 * WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN, BK = 32, LDB = BK * 2 + 16, SEG = BK / 8;
    constexpr int NLA = BM * SEG / NT, NLB = BN * SEG / NT; // 16-byte loads per thread and array
    constexpr int BUF = (2 * BM + 2 * BN) * LDB;
    static_assert(BM * SEG % NT == 0 && BN * SEG % NT == 0, "tile rows must divide over the threads");
    extern __shared__ __align__(16) unsigned char smraw[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, wm = w / WN, wn = w % WN;
    const uint32_t q0 = blockIdx.y * BM, c0 = blockIdx.x * BN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const unsigned char* ga_h[NLA];
    const unsigned char* ga_l[NLA];
    const unsigned char* gb_h[NLB];
    const unsigned char* gb_l[NLB];
    uint32_t soa[NLA], sob[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const uint32_t idx = tid + (uint32_t)NT * i, row = idx / SEG, seg = idx % SEG;
        const uint32_t ra = q0 + row < nq ? q0 + row : nq - 1u;
        ga_h[i] = reinterpret_cast<const unsigned char*>(rot_hi) + (size_t)ra * D * 2 + seg * 16;
        ga_l[i] = reinterpret_cast<const unsigned char*>(rot_lo) + (size_t)ra * D * 2 + seg * 16;
        soa[i] = row * LDB + seg * 16;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const uint32_t idx = tid + (uint32_t)NT * i, row = idx / SEG, seg = idx % SEG;
        const uint32_t rb = c0 + row < nlist ? c0 + row : nlist - 1u;
        gb_h[i] = reinterpret_cast<const unsigned char*>(cent_hi) + (size_t)rb * D * 2 + seg * 16;
        gb_l[i] = reinterpret_cast<const unsigned char*>(cent_lo) + (size_t)rb * D * 2 + seg * 16;
        sob[i] = 2 * BM * LDB + row * LDB + seg * 16;
    }
    u32x4 pah[NLA], pal[NLA], pbh[NLB], pbl[NLB];
    // SPLIT-K (gridDim.z > 1; small batches, whose few tiles leave most of the chip idle while each walks all D / 32 slabs): workgroup z
    // takes the slabs [z * per, (z + 1) * per) and ADDS its part of the score atomically to a row the preparation kernel zeroed
    // (z = 0 also adds the norms).  The order of those additions is not fixed; the approximate score only feeds the shortlist, whose
    // eps covers it (k_select_mfma), and every result is decided on exact scores.
    const uint32_t nslab_all = D / BK, per_z = (nslab_all + gridDim.z - 1u) / gridDim.z, s_first = blockIdx.z * per_z;
    if (s_first >= nslab_all) return; // (whole workgroup)
    const uint32_t nslab = nslab_all - s_first < per_z ? nslab_all - s_first : per_z;
    {
        const size_t k_first = (size_t)s_first * BK * 2;
#pragma unroll
        for (int i = 0; i < NLA; ++i) { ga_h[i] += k_first; ga_l[i] += k_first; }
#pragma unroll
        for (int i = 0; i < NLB; ++i) { gb_h[i] += k_first; gb_l[i] += k_first; }
    }
#define RBQ_RANK_FETCH(S)                                                                                              \
    do {                                                                                                               \
        const size_t ko = (size_t)(S) * BK * 2;                                                                        \
        _Pragma("unroll") for (int i = 0; i < NLA; ++i) {                                                              \
            pah[i] = *reinterpret_cast<const u32x4*>(ga_h[i] + ko);                                                    \
            pal[i] = *reinterpret_cast<const u32x4*>(ga_l[i] + ko);                                                    \
        }                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NLB; ++i) {                                                              \
            pbh[i] = *reinterpret_cast<const u32x4*>(gb_h[i] + ko);                                                    \
            pbl[i] = *reinterpret_cast<const u32x4*>(gb_l[i] + ko);                                                    \
        }                                                                                                              \
    } while (0)
#define RBQ_RANK_STORE(B)                                                                                              \
    do {                                                                                                               \
        unsigned char* base = smraw + (B) * BUF;                                                                       \
        _Pragma("unroll") for (int i = 0; i < NLA; ++i) {                                                              \
            *reinterpret_cast<u32x4*>(base + soa[i]) = pah[i];                                                         \
            *reinterpret_cast<u32x4*>(base + BM * LDB + soa[i]) = pal[i];                                              \
        }                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NLB; ++i) {                                                              \
            *reinterpret_cast<u32x4*>(base + sob[i]) = pbh[i];                                                         \
            *reinterpret_cast<u32x4*>(base + BN * LDB + sob[i]) = pbl[i];                                              \
        }                                                                                                              \
    } while (0)
    // fragment sets of the two 16-wide K steps of a slab; A/B operand of the MFMA: lane l holds row (l & 31),
    // k = 8 * (l >> 5) .. + 7
    bf16x8 f0ah[TM], f0al[TM], f0bh[TN], f0bl[TN], f1ah[TM], f1al[TM], f1bh[TN], f1bl[TN];
    const uint32_t fo = (lane & 31u) * LDB + (lane >> 5) * 16;
#define RBQ_RANK_FRAGS(B, KC, AH, AL, BH, BL)                                                                          \
    do {                                                                                                               \
        const unsigned char* sAh = smraw + (B) * BUF;                                                                  \
        const unsigned char* sBh = sAh + 2 * BM * LDB;                                                                 \
        _Pragma("unroll") for (int a = 0; a < TM; ++a) {                                                               \
            const uint32_t ra = (wm * 32 * TM + a * 32) * LDB + fo + (KC) * 32;                                        \
            AH[a] = *reinterpret_cast<const bf16x8*>(sAh + ra);                                                        \
            AL[a] = *reinterpret_cast<const bf16x8*>(sAh + BM * LDB + ra);                                             \
        }                                                                                                              \
        _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                               \
            const uint32_t rb = (wn * 32 * TN + b * 32) * LDB + fo + (KC) * 32;                                        \
            BH[b] = *reinterpret_cast<const bf16x8*>(sBh + rb);                                                        \
            BL[b] = *reinterpret_cast<const bf16x8*>(sBh + BN * LDB + rb);                                             \
        }                                                                                                              \
    } while (0)
    // The MFMA is gfx950's double-rate v_mfma_f32_32x32x16_bf16 (mfma_x8 above).  Round 1 had seen wrong LUT bytes from
    // the workgroup-per-query k_prep of a NEIGHBOURING stream while an x16 MFMA GEMM ran — in the single-buffered GEMM
    // kernel of that time, which no longer exists.  With this kernel the combination was re-examined in round 2
    // (tests/diag/x16_probe.py, x16_tests.py: 3000 rounds of a k_prep victim stream beside three noise streams compared
    // byte for byte with a quiet run, 12 rounds of the two multi-stream parity tests with k_prep forced): no difference,
    // so the interaction went with the kernel it was seen in.  test_concurrent_streams_match_oracle (both preparation
    // kernels) and test_matrix_rotator_concurrent_streams guard the multi-stream path; -DRBQ_MFMA_X8 restores the K=8 form.
#define RBQ_RANK_MMA(AH, AL, BH, BL)                                                                                   \
    _Pragma("unroll") for (int a = 0; a < TM; ++a) _Pragma("unroll") for (int b = 0; b < TN; ++b) {                    \
        acc[a][b] = mfma_x8(AL[a], BH[b], acc[a][b]);                                                                  \
        acc[a][b] = mfma_x8(AH[a], BL[b], acc[a][b]);                                                                  \
        acc[a][b] = mfma_x8(AH[a], BH[b], acc[a][b]);                                                                  \
    }
    RBQ_RANK_FETCH(0);
    RBQ_RANK_STORE(0);
    if (nslab > 1) RBQ_RANK_FETCH(1);
    __syncthreads();
    RBQ_RANK_FRAGS(0, 0, f0ah, f0al, f0bh, f0bl);
    for (uint32_t s = 0; s < nslab; ++s) {
        // the LDS reads of one K step run under the MFMAs of the other: step 1 of this slab is requested before
        // the MFMAs of step 0, step 0 of the NEXT slab (complete in LDS once the barrier is passed) before those
        // of step 1
        const uint32_t cur = s & 1u;
        RBQ_RANK_FRAGS(cur, 1, f1ah, f1al, f1bh, f1bl);
        if (s + 1 < nslab) RBQ_RANK_STORE(cur ^ 1u);
        if (s + 2 < nslab) RBQ_RANK_FETCH(s + 2);
        RBQ_RANK_MMA(f0ah, f0al, f0bh, f0bl);
        __builtin_amdgcn_sched_barrier(0); // keep these MFMAs in front of the barrier: they cover the reads of step 1
        __syncthreads();
        if (s + 1 < nslab) RBQ_RANK_FRAGS(cur ^ 1u, 0, f0ah, f0al, f0bh, f0bl);
        RBQ_RANK_MMA(f1ah, f1al, f1bh, f1bl);
    }
#undef RBQ_RANK_FRAGS
#undef RBQ_RANK_MMA
#undef RBQ_RANK_FETCH
#undef RBQ_RANK_STORE
    // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  The query norms of
    // all rows are fetched up front (one wait instead of one per row).
    float qn[TM][16];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t qi = q0 + wm * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            qn[a][r] = METRIC == 0 ? consts[qi < nq ? qi : nq - 1u].qnorm2 : 0.0f;
        }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const uint32_t c = c0 + wn * 32 * TN + b * 32 + (lane & 31u);
            const float cn = (METRIC == 0 && c < nlist) ? cnorm2[c] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t qi = q0 + wm * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (qi < nq && c < nlist) {
                    const float dot = acc[a][b][r];
                    if (gridDim.z == 1) scores[(size_t)qi * nlist + c] = METRIC == 0 ? fmaf(-2.0f, dot, qn[a][r] + cn) : dot;
                    else atomicAdd(&scores[(size_t)qi * nlist + c], METRIC == 0 ? (blockIdx.z == 0 ? fmaf(-2.0f, dot, qn[a][r] + cn) : -2.0f * dot) : dot);
                }
            }
        }
}

// Canonical-order score of one (query, list) pair on 2 lanes: lane h owns the reference's accumulators
// 4h..4h+3 (elements 8t+4h..8t+4h+3, one 16-byte load per step); the final sum adds the eight accumulators
// in order 0..7 starting from -0.0 (Rust iter().sum()), then the scalar tail.  D % 4 == 0.
template <int METRIC>
__device__ __forceinline__ float canon_pair2(const float* qrot, const float* __restrict__ c, uint32_t D, uint32_t h) {
    const uint32_t Dmain = D & ~7u;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll RBQ_CANON_UNROLL
    for (uint32_t i = 4 * h; i < Dmain; i += 8) {
        const float4 cv = *reinterpret_cast<const float4*>(c + i);
        const float4 qv = *reinterpret_cast<const float4*>(qrot + i);
        float p0, p1, p2, p3;
        if (METRIC == 0) {
            const float d0 = qv.x - cv.x, d1 = qv.y - cv.y, d2 = qv.z - cv.z, d3 = qv.w - cv.w;
            p0 = d0 * d0; p1 = d1 * d1; p2 = d2 * d2; p3 = d3 * d3;
        } else {
            p0 = qv.x * cv.x; p1 = qv.y * cv.y; p2 = qv.z * cv.z; p3 = qv.w * cv.w;
        }
        a0 = a0 + p0; a1 = a1 + p1; a2 = a2 + p2; a3 = a3 + p3;
    }
    float sum = 0.0f;
    if (Dmain) {
        sum = -0.0f;
        sum = sum + __shfl(a0, 0, 2); sum = sum + __shfl(a1, 0, 2); sum = sum + __shfl(a2, 0, 2); sum = sum + __shfl(a3, 0, 2);
        sum = sum + __shfl(a0, 1, 2); sum = sum + __shfl(a1, 1, 2); sum = sum + __shfl(a2, 1, 2); sum = sum + __shfl(a3, 1, 2);
    }
    for (uint32_t i = Dmain; i < D; ++i) {
        float p;
        if (METRIC == 0) {
            const float d = qrot[i] - c[i];
            p = d * d;
        } else {
            p = qrot[i] * c[i];
        }
        sum = sum + p;
    }
    return sum;
}

// Inner-product metric: the canonical dot (the score) and the canonical squared distance (g_error needs it for
// every probed list) of one (query, list) pair in ONE pass over the centroid row; same accumulators, same order
// as canon_pair2<1> and canon_pair2<0>.
__device__ __forceinline__ void canon_pair2_ip_l2(const float* qrot, const float* __restrict__ c, uint32_t D, uint32_t h,
                                                  float& dot, float& l2) {
    const uint32_t Dmain = D & ~7u;
    float a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (uint32_t i = 4 * h; i < Dmain; i += 8) {
        const float4 cv = *reinterpret_cast<const float4*>(c + i);
        const float4 qv = *reinterpret_cast<const float4*>(qrot + i);
        const float q4[4] = {qv.x, qv.y, qv.z, qv.w}, c4[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float p = q4[k] * c4[k];
            const float d = q4[k] - c4[k];
            const float r = d * d;
            a[k] = a[k] + p;
            b[k] = b[k] + r;
        }
    }
    float s0 = 0.0f, s1 = 0.0f;
    if (Dmain) {
        s0 = -0.0f; s1 = -0.0f;
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s0 = s0 + __shfl(a[k], l, 2);
                s1 = s1 + __shfl(b[k], l, 2);
            }
    }
    for (uint32_t i = Dmain; i < D; ++i) {
        const float p = qrot[i] * c[i];
        const float d = qrot[i] - c[i];
        const float r = d * d;
        s0 = s0 + p;
        s1 = s1 + r;
    }
    dot = s0; l2 = s1;
}

#ifdef RBQ_SEL_WAVES // experiment: waves per SIMD as register target of k_select_mfma
#define RBQ_SEL_BOUNDS __launch_bounds__(kThreads, RBQ_SEL_WAVES)
#else
#define RBQ_SEL_BOUNDS __launch_bounds__(kThreads)
#endif

// launch geometry of k_select_mfma that the host derives from (nlist, nprobe, D): see launch_select_mfma
struct SelectGeom {
    uint32_t cap2;       // capacity of the shortlist window (power of two >= 2 * nprobe)
    int row_in_lds;      // RM == 1: the approximate row is staged in LDS
    int stage;           // per-probe geometry staged in LDS
    uint32_t stage_rows; // centroid rows the LDS scorer stages at a time (0: no room, pair scorer only)
    uint32_t cand_cap;   // RM == 0: (key, cid) candidates that fit the LDS behind `part` before it is used for anything else
    uint32_t hx_nv;      // exact head evaluation: vectors of the nearest list whose 1-bit estimate is evaluated (0: off); its LDS
                         // scratch (u8 LUT | zero-padded query | three floats per vector) shares the region of the LDS scorer's rows
};
#ifndef RBQ_HX_TRIGGER
#define RBQ_HX_TRIGGER 6
#endif
#ifndef RBQ_HX_REFINE_MULT
#define RBQ_HX_REFINE_MULT 2
#endif
constexpr uint32_t kHxTrigger = RBQ_HX_TRIGGER; // lists alive beyond the head under the Cauchy-Schwarz bound that make the exact evaluation worth its cost
constexpr uint32_t kHxMaxVec = 256;  // one thread per vector

// ---- canonical scores of shortlist entries ----------------------------------------------------------------------------
// Both scorers overwrite keys[e] (approximate key | cid) with the EXACT (score, cid) key of entry e = idx_of(j), j < m, and
// park the canonical squared distance of an inner-product list in the (no longer needed) approximate row.  Same
// accumulators, same order, bit-identical results.
//   pairs : 2 lanes per list straight from global memory, 128 lists per round — every lane walks its row in dependent
//           batches of 16-byte loads, each of which uses a quarter of the 32 cache lines it touches: the round-2 scorer, right
//           when MANY lists are scored (it is bound by the L1 fill rate of those quarter-used lines)
//   staged: `rows` lists at a time are loaded by the whole workgroup with fully coalesced 16-byte loads into LDS (one
//           round trip, whole lines), then 8 lanes per list — one per accumulator of src/math.rs:216-245 — run the 120-step
//           add chains from LDS: ~5 us for a handful of lists against ~45 us for a pair round.  What the lazy selection uses.
template <typename IdxFn>
__device__ __forceinline__ void canon_score_pairs(uint64_t* keys, const float* qrot, const float* __restrict__ cent, uint32_t D, int metric,
                                                  float* grow, uint32_t m, uint32_t tid, IdxFn idx_of) {
    const uint32_t l2 = tid & 1u, grp = tid >> 1;
    for (uint32_t i0 = 0; i0 < m; i0 += kThreads / 2) {
        const uint32_t j = i0 + grp;
        uint32_t cid = 0, e = 0;
        float s = 0.0f;
        if (j < m) {
            e = idx_of(j);
            cid = (uint32_t)keys[e];
            const float* c = cent + (size_t)cid * D;
            if (metric == 0) s = canon_pair2<0>(qrot, c, D, l2);
            else {
                float dl2;
                canon_pair2_ip_l2(qrot, c, D, l2, s, dl2);
                if (l2 == 0) grow[cid] = dl2;
            }
        }
        if (j < m && l2 == 0) keys[e] = make_key(s, cid, metric);
    }
}
template <typename IdxFn>
__device__ __forceinline__ void canon_score_staged(uint64_t* keys, const float* qrot, const float* __restrict__ cent, uint32_t D, int metric,
                                                   float* grow, uint32_t m, uint32_t tid, float* rows, uint32_t RB, IdxFn idx_of) {
    const uint32_t Dp = D + 8u, Dmain = D & ~7u, nv4 = D >> 2; // (+8 floats: the 8 lanes of consecutive rows hit distinct banks)
    for (uint32_t base = 0; base < m; base += RB) {
        const uint32_t nr = m - base < RB ? m - base : RB;
        // (four 16-byte loads per thread in flight: as a plain copy loop every iteration waited for its load in front of its ds_write)
        for (uint32_t x0 = tid; x0 < nr * nv4; x0 += 4 * kThreads) {
            float4 t4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t x = x0 + (uint32_t)u * kThreads;
                t4[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (x < nr * nv4) {
                    const uint32_t r = x / nv4, j4 = x - r * nv4;
                    const uint32_t cid = (uint32_t)keys[idx_of(base + r)];
                    t4[u] = *reinterpret_cast<const float4*>(cent + (size_t)cid * D + 4 * j4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t x = x0 + (uint32_t)u * kThreads;
                if (x < nr * nv4) {
                    const uint32_t r = x / nv4, j4 = x - r * nv4;
                    *reinterpret_cast<float4*>(rows + (size_t)r * Dp + 4 * j4) = t4[u];
                }
            }
        }
        __syncthreads();
        if (tid < nr * 8u) {
            const uint32_t r = tid >> 3, a = tid & 7u;
            const float* c = rows + (size_t)r * Dp;
            float acc = 0.0f, acc2 = 0.0f;
            if (metric == 0) {
#pragma unroll 8
                for (uint32_t i = a; i < Dmain; i += 8) {
                    const float d = qrot[i] - c[i];
                    const float p = d * d;
                    acc = acc + p;
                }
            } else {
#pragma unroll 8
                for (uint32_t i = a; i < Dmain; i += 8) {
                    const float qv = qrot[i], cv = c[i];
                    const float p = qv * cv;
                    const float d = qv - cv;
                    const float r2 = d * d;
                    acc = acc + p;
                    acc2 = acc2 + r2;
                }
            }
            float sum = 0.0f, sum2 = 0.0f;
            if (Dmain) {
                sum = -0.0f; sum2 = -0.0f;
#pragma unroll
                for (int l = 0; l < 8; ++l) {
                    sum = sum + __shfl(acc, l, 8);
                    sum2 = sum2 + __shfl(acc2, l, 8);
                }
            }
            for (uint32_t i = Dmain; i < D; ++i) { // scalar tail
                const float qv = qrot[i], cv = c[i];
                const float d = qv - cv;
                const float r2 = d * d;
                if (metric == 0) sum = sum + r2;
                else {
                    const float p = qv * cv;
                    sum = sum + p;
                    sum2 = sum2 + r2;
                }
            }
            if (a == 0) {
                const uint32_t e = idx_of(base + r);
                const uint32_t cid = (uint32_t)keys[e];
                keys[e] = make_key(sum, cid, metric);
                if (metric == 1) { grow[cid] = sum2; __threadfence_block(); }
            }
        }
        __syncthreads();
    }
}

// wave-wide minimum of one u32 per lane, on DPP (row_shr 1/2/4/8, row_bcast 15/31; the result is read from lane 63)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
    const int id = -1; // 0xffffffff: identity of the unsigned minimum
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x111, 0xf, 0xf, false); x = t < x ? t : x;
    t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x112, 0xf, 0xf, false); x = t < x ? t : x;
    t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x114, 0xf, 0xf, false); x = t < x ? t : x;
    t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x118, 0xf, 0xf, false); x = t < x ? t : x;
    t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x142, 0xa, 0xf, false); x = t < x ? t : x;
    t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x143, 0xc, 0xf, false); x = t < x ? t : x;
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) { // same DPP pattern, sum (missing lanes read 0)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

//   octets: 8 lanes per list straight from global memory, one per accumulator (4-byte loads, the 8 lanes of a list read 32
//           contiguous bytes per step), 32 lists per round, 30 loads in flight per lane: a third of a pair round's time for
//           up to 32 lists — the middle ground between `staged` (<= 8 lists) and `pairs` (> 64).
template <typename IdxFn>
__device__ __forceinline__ void canon_score_octets(uint64_t* keys, const float* qrot, const float* __restrict__ cent, uint32_t D, int metric,
                                                   float* grow, uint32_t m, uint32_t tid, IdxFn idx_of) {
    const uint32_t a = tid & 7u, grp = tid >> 3, Dmain = D & ~7u;
    for (uint32_t i0 = 0; i0 < m; i0 += kThreads / 8) {
        const uint32_t j = i0 + grp;
        if (j < m) { // uniform per 8-lane group
            const uint32_t e = idx_of(j);
            const uint32_t cid = (uint32_t)keys[e];
            const float* c = cent + (size_t)cid * D;
            float acc = 0.0f, acc2 = 0.0f;
            // explicit two-phase steps (all loads of a batch, then its adds in order): left to itself hipcc waits for every
            // load in front of its add
            constexpr int KB = 24;
            for (uint32_t base = a; base < Dmain; base += 8 * KB) {
                float cv[KB];
#pragma unroll
                for (int k = 0; k < KB; ++k) cv[k] = base + 8u * k < Dmain ? c[base + 8u * k] : 0.0f;
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    if (base + 8u * k < Dmain) {
                        const float qv = qrot[base + 8u * k];
                        if (metric == 0) {
                            const float d = qv - cv[k];
                            const float p = d * d;
                            acc = acc + p;
                        } else {
                            const float p = qv * cv[k];
                            const float d = qv - cv[k];
                            const float r2 = d * d;
                            acc = acc + p;
                            acc2 = acc2 + r2;
                        }
                    }
                }
            }
            float sum = 0.0f, sum2 = 0.0f;
            if (Dmain) {
                sum = -0.0f; sum2 = -0.0f;
#pragma unroll
                for (int l = 0; l < 8; ++l) {
                    sum = sum + __shfl(acc, l, 8);
                    sum2 = sum2 + __shfl(acc2, l, 8);
                }
            }
            for (uint32_t i = Dmain; i < D; ++i) { // scalar tail
                const float qv = qrot[i], cv = c[i];
                const float d = qv - cv;
                const float r2 = d * d;
                if (metric == 0) sum = sum + r2;
                else {
                    const float p = qv * cv;
                    sum = sum + p;
                    sum2 = sum2 + r2;
                }
            }
            if (a == 0) {
                keys[e] = make_key(sum, cid, metric);
                if (metric == 1) grow[cid] = sum2;
            }
        }
    }
}

// ascending sort of keys[0..n) (distinct keys; entries n..cap2-1 must be ~0 for the bitonic branch)
__device__ __forceinline__ void sort_keys(uint64_t* keys, uint32_t n, uint32_t cap2, uint32_t tid) {
    if (n <= 2 * kThreads) {
        // rank sort: the keys are distinct (they carry the list id), so the number of smaller keys IS the
        // position; every thread walks the n keys as LDS broadcasts — no barrier per compare-exchange stage
        uint64_t my[2] = {~0ull, ~0ull};
        uint32_t rk[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 2; ++u) if (tid + u * kThreads < n) my[u] = keys[tid + u * kThreads];
        const uint32_t wbase = tid & ~63u;
        // (unrolled: the loads of a group of keys are independent — one LDS round trip per 16 keys, not per key)
        if (n > (uint32_t)kThreads) {
#pragma unroll 8
            for (uint32_t j = 0; j < n; ++j) {
                const uint64_t kj = keys[j];
                rk[0] += kj < my[0] ? 1u : 0u;
                rk[1] += kj < my[1] ? 1u : 0u;
            }
        } else if (wbase < n) {
            const uint4* k2 = reinterpret_cast<const uint4*>(keys); // two keys per 16-byte read; slot n is ~0 (never smaller) when n is odd
            const uint32_t np2 = (n + 1) >> 1;
#pragma unroll 8
            for (uint32_t j = 0; j < np2; ++j) {
                const uint4 kk = k2[j];
                const uint64_t ka = ((uint64_t)kk.y << 32) | kk.x, kb = ((uint64_t)kk.w << 32) | kk.z;
                rk[0] += (ka < my[0] ? 1u : 0u) + (kb < my[0] ? 1u : 0u);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) if (tid + u * kThreads < n) keys[rk[u]] = my[u];
        __syncthreads();
    } else {
        for (uint32_t k = 2; k <= cap2; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < cap2; i += kThreads) {
                    const uint32_t ixj = i ^ j;
                    if (ixj > i) {
                        const uint64_t a = keys[i], b = keys[ixj];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                    }
                }
                __syncthreads();
            }
    }
}


#ifndef RBQ_HX_EX_ALL
#define RBQ_HX_EX_ALL 0 // (1: the refined head vectors with every code unit in flight — 110 registers, four waves per SIMD: -3 % queries/s, measured)
#endif
#ifndef RBQ_HX_GRANULES
#define RBQ_HX_GRANULES 4 // code granules of a head vector requested before its first lookup (head_exact_bound)
#endif
// ---- exact head evaluation (k_select_mfma step 3b) -----------------------------------------------------------------------
// LDS scratch layout (shares the region of the LDS scorer's rows): lut[4 Dc] u8 | sq[ex_qlen] f32 | est[nv] | lb[nv] | ip[nv] | U[nv]
__host__ __device__ inline size_t hx_scratch_bytes(uint32_t D, uint32_t Dc, uint32_t ex_bits, uint32_t nv) {
    return (size_t)Dc * 4 + (ex_bits ? (size_t)ex_qlen(D, ex_bits) * 4 : 0) + (size_t)nv * 16 + 16;
}
__device__ __forceinline__ uint32_t hx_look8(uint32_t x, const uint8_t* p) { // scan.hpp look8 on a plain LDS pointer
    uint32_t s = p[x & 15u];
    s += p[16 + ((x >> 4) & 15u)];
    s += p[32 + ((x >> 8) & 15u)];
    s += p[48 + ((x >> 12) & 15u)];
    s += p[64 + ((x >> 16) & 15u)];
    s += p[80 + ((x >> 20) & 15u)];
    s += p[96 + ((x >> 24) & 15u)];
    s += p[112 + (x >> 28)];
    return s;
}
__device__ __forceinline__ bool hx_tame(float x) { return fabsf(x) < 1e30f; } // far from overflow: the exact value next to it is finite too
// Returns T' (see the kernel), +inf when the list is too short or too few of its vectors have finite bounds.  Every thread of the
// workgroup calls it and gets the same value.  scratch: the LDS region above; part / hist: 256 words of LDS scratch each.
template <class CostOf>
__device__ __forceinline__ float head_exact_bound(const SelectParams& P, const SelectGeom& G, const QueryConsts& qc, const uint64_t* keys,
                                                  const uint32_t* s_head, const uint32_t* s_hgb, const uint32_t* s_hn, const float* s_hcn,
                                                  uint32_t h, float eps, unsigned char* scratch, const float* qrot, uint32_t* part, uint32_t* hist,
                                                  uint32_t q, uint32_t tid, CostOf&& cost_of) {
    const uint32_t D = P.D, Dc = P.Dc, ex_bits = P.ex_bits, top_k = P.top_k;
    // the nearest head list by approximate cost (any head list would do: all of them precede every dead list)
    uint32_t r0 = 0, kbest = 0xffffffffu;
    for (uint32_t r = 0; r < h; ++r) {
        const uint32_t k32 = (uint32_t)(keys[s_head[r]] >> 32);
        if (k32 < kbest) { kbest = k32; r0 = r; }
    }
    const float ci = cost_of(kbest);
    const float g_add = ci + 1.01f * eps; // upper end (estimates and refined distances grow with g_add)
    float ge_lo, ge_hi;
    if (P.metric == 0) { ge_lo = sqrtf(fmaxf(ci - 1.01f * eps, 0.0f)); ge_hi = sqrtf(fmaxf(g_add, 0.0f)); }
    else { const float da = qc.qnorm2 + s_hcn[r0] + 2.0f * ci; ge_lo = sqrtf(fmaxf(da - 4.0f * eps, 0.0f)); ge_hi = sqrtf(fmaxf(da + 4.0f * eps, 0.0f)); }
    const uint32_t gb = s_hgb[r0], nvec = s_hn[r0];
    const uint32_t nv = nvec < G.hx_nv ? nvec : G.hx_nv;
    if (nv < top_k || top_k == 0 || !hx_tame(g_add) || !hx_tame(ge_hi)) return INFINITY; // (uniform)
    // geometry guard: the blocks this pass reads must lie inside the index (a trip is counted and the pass is skipped — it is an
    // optimisation, never needed for correctness)
    if (r0 >= h || (uint64_t)gb + ((nv + 31u) >> 5) > (uint64_t)P.n_blocks) {
        if (tid == 0 && P.fallback_count) atomicAdd(P.fallback_count + 2, 1u);
        return INFINITY;
    }
    if (tid == 0 && P.fallback_count) atomicAdd(P.fallback_count + 3, 1u);
    uint8_t* lutL = scratch;
    float* sq = reinterpret_cast<float*>(scratch + (size_t)Dc * 4);
    const uint32_t qlen = ex_bits ? ex_qlen(D, ex_bits) : 0u;
    float* vEst = sq + qlen;
    float* vLb = vEst + nv;
    float* vIp = vLb + nv;
    float* vU = vIp + nv;
    {
        const uint4* src = reinterpret_cast<const uint4*>(P.lut + (size_t)q * Dc * 4);
        uint4* dst = reinterpret_cast<uint4*>(lutL);
        for (uint32_t i = tid; i < Dc / 4; i += kThreads) dst[i] = src[i];
        for (uint32_t i = tid; i < qlen; i += kThreads) sq[i] = i < D ? qrot[i] : 0.0f;
    }
    __syncthreads();
    const size_t stride = (size_t)Dc * 4 + 384;
    // phase 1: thread v = vector v of the list: accumulate + epilogue of k_scan (lb_of), bounds' ends of g_add / g_err
    if (tid < nv) {
        const uint32_t b = tid >> 5, l32 = tid & 31u;
        const uint8_t* blk = P.blocks + (size_t)(gb + b) * stride;
        const uint32_t G16 = Dc >> 7;
        const uint4* cp = reinterpret_cast<const uint4*>(blk) + l32;
        // the factor rows and RBQ_HX_GRANULES code granules at a time are requested before the first lookup: the loop
        // as hipcc left it waited for every granule in front of its lookups — D/128 dependent global round trips per head evaluation
        const float* fac = reinterpret_cast<const float*>(blk + (size_t)Dc * 4);
        const float f_add = fac[l32], f_rescale = fac[32 + l32], f_error = fac[64 + l32];
        uint2 ytail = make_uint2(0u, 0u);
        if (Dc & 64u) ytail = *(reinterpret_cast<const uint2*>(blk + G16 * 512) + l32);
        uint32_t acc = 0;
        constexpr int KG = RBQ_HX_GRANULES; // code granules in flight (4: 16 registers, the kernel keeps five waves per SIMD)
        for (uint32_t g0 = 0; g0 < G16; g0 += KG) {
            uint4 xx[KG];
#pragma unroll
            for (int u = 0; u < KG; ++u) xx[u] = g0 + (uint32_t)u < G16 ? cp[(g0 + u) * 32] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int u = 0; u < KG; ++u) {
                if (g0 + (uint32_t)u < G16) {
                    const uint8_t* lg = lutL + (g0 + u) * 512;
                    acc += hx_look8(xx[u].x, lg) + hx_look8(xx[u].y, lg + 128) + hx_look8(xx[u].z, lg + 256) + hx_look8(xx[u].w, lg + 384);
                }
            }
        }
        if (Dc & 64u) acc += hx_look8(ytail.x, lutL + G16 * 512) + hx_look8(ytail.y, lutL + G16 * 512 + 128);
        const float ip = fmaf(qc.delta, (float)(acc & 0xffffu), qc.sum_vl);
        const float tt = ip + qc.k1x;
        const float rs = f_rescale * tt;
        float est = f_add + g_add;
        const float e0 = est;
        est = est + rs;
        const float er = fminf(f_error * ge_lo, f_error * ge_hi);
        const float lb = est - er;
        bool ok = hx_tame(f_add) && hx_tame(f_rescale) && hx_tame(f_error) && hx_tame(ip) && hx_tame(rs) && hx_tame(e0) && hx_tame(est) &&
                  hx_tame(f_error * ge_lo) && hx_tame(f_error * ge_hi) && hx_tame(lb) && qc.amax <= 65535.0f;
        if (P.filter) { // search_filtered: only vectors the filter lets through are ever pushed (the scan's own test, src/ivf.rs:2018-2022)
            const uint32_t id32 = (uint32_t)P.ids[(size_t)(gb + b) * 32u + l32];
            ok = ok && ((uint64_t)id32 < P.filter_nbits) && ((P.filter[id32 >> 5] >> (id32 & 31u)) & 1u);
        }
        vEst[tid] = ok ? est : INFINITY;
        vLb[tid] = lb;
        vIp[tid] = ip;
    }
    __syncthreads();
    float* res = reinterpret_cast<float*>(part);
    if (tid == 0) res[0] = INFINITY;
    if (ex_bits == 0) { // distance = estimate: U_v = max(est, lb); T' = the k-th smallest
        if (tid < nv) vU[tid] = vEst[tid] < INFINITY ? fmaxf(vEst[tid], vLb[tid]) : INFINITY;
        __syncthreads();
        if (tid < nv) {
            const float my = vU[tid];
            uint32_t rk = 0;
            for (uint32_t j = 0; j < nv; ++j) { const float o = vU[j]; rk += (o < my || (o == my && j < tid)) ? 1u : 0u; }
            if (rk == top_k - 1u) res[0] = my;
        }
        __syncthreads();
        const float r = res[0];
        __syncthreads(); // (every thread has read the result before anyone reuses the scratch words)
        return r;
    }
    // phase 2: the R smallest estimates are refined (16 lanes per vector, the reference's FMA order: bit-identical to k_scan)
    const uint32_t R0 = (uint32_t)RBQ_HX_REFINE_MULT * top_k > 16u ? (uint32_t)RBQ_HX_REFINE_MULT * top_k : 16u;
    const uint32_t R = R0 < nv ? R0 : nv;
    if (tid < nv) {
        const float my = vEst[tid];
        uint32_t rk = 0;
        for (uint32_t j = 0; j < nv; ++j) { const float o = vEst[j]; rk += (o < my || (o == my && j < tid)) ? 1u : 0u; }
        if (rk < R) hist[rk] = tid;
    }
    __syncthreads();
    {
        const uint32_t grp = tid >> 4, gl = tid & 15u;
        const uint32_t nunits = ex_w4(D, ex_bits);
        const size_t exb = ex_bytes_dev(D, ex_bits);
        for (uint32_t j = grp; j < R; j += kThreads / 16) {
            uint32_t v = hist[j];
            if (v >= nv) { if (gl == 0 && P.fallback_count) atomicAdd(P.fallback_count + 2, 1u); v = 0; } // (guard: cannot happen; never read out of the list)
            const uint32_t slot = (gb + (v >> 5)) * 32u + (v & 31u);
            const uint8_t* ex = P.ex_codes + (size_t)slot * exb;
            const float fa = P.f_add_ex[slot], fr = P.f_rescale_ex[slot]; // (requested with the code units)
            float sacc;
            if (RBQ_HX_EX_ALL && nunits <= (uint32_t)kExRegUnits) { // every unit in flight before the first is decoded (same arithmetic: ex_dot_all)
                uint4 uu[kExRegUnits];
                ex_load_all(uu, ex, gl, nunits);
                sacc = ex_bits == 6 ? ex_dot_all<6>(uu, sq, gl, nunits) : ex_dot_all<2>(uu, sq, gl, nunits);
            } else sacc = ex_bits == 6 ? ex_dot_units<6>(ex, sq, gl, nunits) : ex_dot_units<2>(ex, sq, gl, nunits);
            sacc = group16_reduce(sacc);
            if (gl == 0) {
                float tt2 = qc.scale * vIp[v];
                const float t0 = tt2;
                tt2 = tt2 + sacc;
                const float t1 = tt2;
                tt2 = tt2 + qc.kbx;
                const float a = fa + g_add;
                const float m = fr * tt2;
                const float d = a + m;
                const bool ok = vEst[v] < INFINITY && hx_tame(fa) && hx_tame(fr) && hx_tame(t0) && hx_tame(sacc) && hx_tame(t1) && hx_tame(tt2) &&
                                hx_tame(a) && hx_tame(m) && hx_tame(d);
                vU[j] = ok ? fmaxf(d, vLb[v]) : INFINITY;
            }
        }
    }
    __syncthreads();
    if (tid < R) {
        const float my = vU[tid];
        uint32_t rk = 0;
        for (uint32_t j = 0; j < R; ++j) { const float o = vU[j]; rk += (o < my || (o == my && j < tid)) ? 1u : 0u; }
        if (rk == top_k - 1u) res[0] = my;
    }
    __syncthreads();
    const float r = res[0];
    __syncthreads(); // (every thread has read the result before anyone reuses the scratch words)
    return r;
}

constexpr uint32_t kSelHead = 4;       // nearest lists that are always scored and scanned: their blocks give the bound T_ub
constexpr uint32_t kSelTodoMax = 512;  // lists the lazy path scores (more: the query takes the eager path)
constexpr uint32_t kSelKeptMax = 256;  // scanned lists that are ordered through the side buffer (more: in-place compaction + sort)
constexpr uint32_t kSelZoneMax = 1024; // boundary-zone lists whose membership the lazy path resolves by counting

// dynamic LDS: keys[cap2] u64 | qrot[D] f32 | part[256] u32 | row[nlist] f32 (only when RM == 1) |
//              pgeo[4][nprobe] u32 (only when `stage`: g_add, g_err, first block, vector count of every probe) |
//              rows[stage_rows][D + 8] f32 (LDS scorer)
// RM: where the query's row of approximate scores lives during the selection passes — 2: in registers (nlist <=
// 4096, 16 per thread), 1: staged in LDS, 0: re-read from global memory.
//
// LAZY SELECTION (round 3).  The reference scores every list exactly, sorts, probes the nprobe best in order
// (src/ivf.rs:1782-1857) — and then skips, vector by vector, nearly all of them (97 % of the probed blocks on the bench
// data).  A list whose EVERY vector the reference provably skips contributes nothing to the result but its size to
// skipped_by_lower_bound: it needs neither its exact centroid distance (3840 B of centroid row at D = 960) nor stream
// entries.  With A(c) the approximate score and eps the rigorous bound |A - canonical| <= eps of the header:
//   cost(c) := g_add(c) (L2: the canonical distance; IP: -dot) lies in [+-A - eps, +-A + eps].
//   1. shortlist S = {A within 2 eps of tau} as before, sorted by approximate cost.  Entry i is a CERTAIN member of the probe
//      set iff fewer than nprobe lists can precede it: cost_i + 2 eps < cost of sorted entry nprobe (z0 = their number, a
//      prefix).  The others form the boundary zone (no shortlisted list is a certain non-member, by construction of S).
//   2. head = the first min(kSelHead, z0) entries: scored exactly.  For each of their blocks b, block_ub() bounds
//      max(refined distance, lower bound) of its vectors by U(b) (Cauchy-Schwarz around the list's centroid: about
//      (|r| + |q - c|)^2, i.e. within ~2x of the true distances).  T_ub := the smallest U with at least top_k vectors in
//      head blocks of U(b) <= U.  Claim: once the reference has visited those blocks, its k-th distance is <= T_ub.  (Else
//      each such vector v saw a threshold > T_ub >= lb_v, was evaluated, had a finite distance <= T_ub and was pushed:
//      k pushed distances <= T_ub put the k-th smallest at or below T_ub.)  Thresholds only shrink afterwards.
//   3. a list L outside the head is DEAD iff it certainly comes after every head list (cost_L - eps > max exact head cost)
//      and list_bound_reaches(): every lb of L, for every admissible g_add / g_err, is finite and >= T_ub.  The reference
//      then skips all of L's vectors wherever exactly L sits in its order.  Dead lists are neither scored nor streamed.
//   4. everything not dead is scored exactly.  If a zone list is alive (or diagnostics / the profile want exact
//      membership) ALL zone lists are scored, and a zone list is a member iff fewer than nprobe - z0 zone lists have a
//      smaller exact key.  Scanned lists = members that are not dead, in exact (score, cid) order: the reference's order
//      with the dead lists left out — its thresholds, pushes and results are untouched.
//   diag: dead_skipped[q] = sum of n_c over dead members (added to skipped_by_lower_bound by k_scan).
// Off (eager = every shortlisted list scored, every member streamed, the round-2 behaviour) when P.lazy == 0, with a
// filter (filtered vectors are not pushed: no T_ub), when accu may wrap, and per query when T_ub is infinite or the
// exceptional sizes above are exceeded.
template <int RM>
__global__ RBQ_SEL_BOUNDS void k_select_mfma(const SelectParams P, const SelectGeom G) {
    extern __shared__ __align__(16) unsigned char smraw[];
    const uint32_t nlist = P.nlist, nprobe = P.nprobe, cap2 = G.cap2, D = P.D;
    const int metric = P.metric;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smraw);
    float* qrot = reinterpret_cast<float*>(smraw + (size_t)cap2 * 8);
    uint32_t* part = reinterpret_cast<uint32_t*>(qrot + D);
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_k, s_cnt, s_bad;
    __shared__ unsigned long long s_nvec;
#ifdef RBQ_SEL_STAMPS
    unsigned long long ts[8]; int tn = 0;
#define SSTAMP() ts[tn++] = __builtin_amdgcn_s_memtime()
#else
#define SSTAMP()
#endif
#if defined(RBQ_SEL_STAMPS) && RBQ_SEL_STAMPS == 4
    unsigned long long ls[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // sub-phases of the lazy branch
#define LSTAMP(i) ls[i] = __builtin_amdgcn_s_memtime()
#else
#define LSTAMP(i)
#endif
    SSTAMP();
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    float* grow = P.scores + (size_t)q * nlist;
    float* lrow = reinterpret_cast<float*>(part + kThreads); // RM == 1
    uint32_t* pgeo = reinterpret_cast<uint32_t*>(lrow + (RM == 1 ? nlist : 0u)); // `stage`
    float* rows = reinterpret_cast<float*>(smraw + ((reinterpret_cast<unsigned char*>(pgeo + (G.stage ? 4u * nprobe : 0u)) - smraw + 15) & ~(size_t)15)); // LDS scorer
    constexpr int KPT = 16;
    uint32_t kreg[KPT]; // RM == 2: ordered keys of elements (u>>2)*1024 + 4*tid + (u&3)
    bool bad_load = false;
    // RM == 0 (more lists than fit registers or LDS: cfg5 has 65 536): candidate window, see the prefilter below
    uint64_t* cand = reinterpret_cast<uint64_t*>(pgeo);
    bool use_cand = false;
    uint32_t n_cand = 0;
    // ordered 32-bit key of the approximate score (ascending = better)
    auto okey = [&](float s) -> uint32_t {
        int32_t k = total_key(s);
        if (metric == 1) k = ~k;
        return (uint32_t)k ^ 0x80000000u;
    };
    // approximate COST (L2: the score; IP: -score; ascending = better) back from an ordered key
    auto cost_of = [&](uint32_t k32) -> float {
        int32_t k = (int32_t)(k32 ^ 0x80000000u);
        if (metric == 1) k = ~k;
        const float s = key_to_float(k);
        return metric == 0 ? s : -s;
    };
    if (RM == 2) {
#pragma unroll
        for (int v = 0; v < KPT / 4; ++v) {
            const uint32_t i = v * 1024u + tid * 4u;
            float4 x = make_float4(0, 0, 0, 0);
            if (i + 3 < nlist) x = *reinterpret_cast<const float4*>(grow + i);
            else { if (i < nlist) x.x = grow[i]; if (i + 1 < nlist) x.y = grow[i + 1]; if (i + 2 < nlist) x.z = grow[i + 2]; }
            bad_load |= !finite_f(x.x) || !finite_f(x.y) || !finite_f(x.z) || !finite_f(x.w);
            kreg[4 * v] = okey(x.x); kreg[4 * v + 1] = okey(x.y); kreg[4 * v + 2] = okey(x.z); kreg[4 * v + 3] = okey(x.w);
        }
    } else if (RM == 1) { // the approximate row is read ~6 times: keep it in LDS when it fits
        for (uint32_t i = tid; i < nlist; i += kThreads) lrow[i] = grow[i];
    }
    // body(i, key) for every list of this thread
    auto each_key = [&](auto&& body) {
        if (RM == 2) {
#pragma unroll
            for (int u = 0; u < KPT; ++u) {
                const uint32_t i = (uint32_t)(u >> 2) * 1024u + tid * 4u + (uint32_t)(u & 3);
                if (i < nlist) body(i, kreg[u]);
            }
        } else if (RM == 1) {
            for (uint32_t i = tid; i < nlist; i += kThreads) { const float v = lrow[i]; bad_load |= !finite_f(v); body(i, okey(v)); }
        } else if (use_cand) { // RM == 0 after the prefilter: the few hundred lists that can matter, in LDS
            for (uint32_t j = tid; j < n_cand; j += kThreads) { const uint64_t kc = cand[j]; body((uint32_t)kc, (uint32_t)(kc >> 32)); }
        } else {
            for (uint32_t i = tid; i < nlist; i += kThreads) { const float v = grow[i]; bad_load |= !finite_f(v); body(i, okey(v)); }
        }
    };

    { // (the rotated query: every load of a thread in flight before its first LDS store)
        float qv[4];
        for (uint32_t i0 = tid; i0 < D; i0 += 4 * kThreads) {
#pragma unroll
            for (int u = 0; u < 4; ++u) qv[u] = i0 + (uint32_t)u * kThreads < D ? P.rot[(size_t)q * D + i0 + (uint32_t)u * kThreads] : 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i0 + (uint32_t)u * kThreads < D) qrot[i0 + (uint32_t)u * kThreads] = qv[u];
        }
    }
    if (tid == 0) { s_k = nprobe - 1; s_cnt = 0; s_nvec = 0; s_bad = 0; }
    __syncthreads();

    SSTAMP(); // 1: row + query staged
    // 1. tau = nprobe-th best approximate score (value only).  Radix select over the RANGE the keys actually
    // span: every pass spreads the remaining candidates over 256 bins of [lo, hi] (the top byte of an f32 key
    // is the same for nearly all scores, which would serialise 4096 LDS atomics on one counter), the bin
    // holding rank k is found with a workgroup prefix sum, and [lo, hi] shrinks to that bin.
    bool all = nprobe >= nlist;
    __shared__ uint32_t s_lo, s_hi, s_bin, s_cum, s_h, s_ng, s_w4[4], s_nc;
    // key of rank k (0-based, ascending) among the keys `each` enumerates: the range-adaptive radix select described above
    auto kth_key = [&](auto&& each, uint32_t k) -> uint32_t {
        if (tid == 0) { s_lo = 0xffffffffu; s_hi = 0u; s_ng = 0u; }
        __syncthreads();
        {
            uint32_t kmn = 0xffffffffu, kmx = 0u;
            each([&](uint32_t, uint32_t key) {
                kmn = key < kmn ? key : kmn;
                kmx = key > kmx ? key : kmx;
            });
            const bool bad = bad_load;
            kmn = wave_min_u32(kmn);
            kmx = ~wave_min_u32(~kmx);
            if ((tid & 63u) == 0) { atomicMin(&s_lo, kmn); atomicMax(&s_hi, kmx); }
            if (bad) s_bad = 1;
        }
        __syncthreads();
        uint32_t lo = s_lo, hi = s_hi;
        while (true) {
            const uint32_t span = hi - lo;
            const uint32_t sh = span < 256u ? 0u : (32u - (uint32_t)__builtin_clz(span)) - 8u; // span >> sh in [128, 255]
            hist[tid] = 0;
            __syncthreads();
            each([&](uint32_t, uint32_t key) {
                if (key - lo <= hi - lo) atomicAdd(&hist[(key - lo) >> sh], 1u); // lo <= key <= hi
            });
            __syncthreads();
            const uint32_t h = hist[tid];
            uint32_t total;
            const uint32_t excl = block_scan_excl256(h, s_w4, tid, total);
            if (excl <= k && k < excl + h) { s_bin = tid; s_cum = excl; s_h = h; } // exactly one thread
            __syncthreads();
            k -= s_cum;
            lo += s_bin << sh;
            const uint32_t width = (sh ? (1u << sh) : 1u) - 1u;
            if (hi - lo > width) hi = lo + width;
            if (sh == 0) break; // lo is the key of rank k
            const uint32_t nc = s_h;
            if (nc <= (uint32_t)kThreads) {
                // few candidates left (the usual case after ONE pass): gather them and take the one of rank k
                // by counting — 3 barriers instead of 5 per further radix pass
                uint32_t* cnd = hist; // (every thread has read its bin count; this happens at most once)
                each([&](uint32_t, uint32_t key) {
                    if (key - lo <= hi - lo) cnd[atomicAdd(&s_ng, 1u)] = key;
                });
                __syncthreads();
                if (tid < nc) {
                    const uint32_t my = cnd[tid];
                    uint32_t less = 0, eq = 0;
#pragma unroll 8
                    for (uint32_t j = 0; j < nc; ++j) {
                        const uint32_t kj = cnd[j];
                        less += kj < my ? 1u : 0u;
                        eq += kj == my ? 1u : 0u;
                    }
                    if (less <= k && k < less + eq) s_lo = my; // every such thread holds the same key
                }
                __syncthreads();
                lo = s_lo;
                break;
            }
        }
        __syncthreads(); // (the shared scratch may be reused by the next call)
        return lo;
    };
    const QueryConsts qc = P.consts[q];
    // (+ 16 x 2^-24: up to four split-K parts of the GEMM are added to the row in no fixed order, each addition rounding a sum of
    // magnitude <= 2 (|q|^2 + |c|^2))
    const float eps = ((6.0f * (float)D + 16.0f) * 5.9604645e-8f + 4.0f * 1.52587890625e-5f) * (qc.qnorm2 + P.cnorm2_max) * 1.001f;
    // ordered key of (the score of `key` moved 2 eps towards worse); 0xffffffff if that is not finite
    auto widen = [&](uint32_t key) -> uint32_t {
        int32_t k = (int32_t)(key ^ 0x80000000u);
        if (metric == 1) k = ~k;
        const float sc = key_to_float(k);
        const float lim = metric == 0 ? sc + 2.0f * eps : sc - 2.0f * eps;
        return finite_f(lim) ? okey(lim) : 0xffffffffu;
    };
    // Prefilter for RM == 0 (n_lists beyond registers and LDS): every pass over the row costs 4 * n_lists bytes of global
    // reads and n_lists LDS atomics per query (cfg5, 65 536 lists: the selection was 13 of the step's 24 ms).  Pass A keeps
    // each thread's KL smallest keys in registers; the nprobe-th smallest B of those 256 * KL keys bounds tau from above
    // (at least nprobe keys of the row are <= B).  Pass B gathers the (key, cid) of every list within 2 eps beyond B into
    // LDS: a superset of the shortlist (tau + 2 eps <= B + 2 eps).  Everything after that — the exact tau, the shortlist, the
    // counts — enumerates these few hundred candidates instead of the row.
    constexpr int KL = 8;
    if (RM == 0 && !all && G.cand_cap >= 4u * nprobe && 2u * nprobe <= (uint32_t)(kThreads * KL)) {
        uint32_t lk[KL];
#pragma unroll
        for (int u = 0; u < KL; ++u) lk[u] = 0xffffffffu;
        // (eight loads in flight per thread: left as one load per iteration, every iteration exposed a global round trip in front of
        // its compare — 256 dependent trips per thread and pass at 65 536 lists; round 5)
        constexpr int KU = 8;
        for (uint32_t i0 = tid; i0 < nlist; i0 += kThreads * KU) {
            float vv[KU];
#pragma unroll
            for (int w = 0; w < KU; ++w) { const uint32_t i = i0 + (uint32_t)w * kThreads; vv[w] = i < nlist ? grow[i] : 0.0f; }
#pragma unroll
            for (int w = 0; w < KU; ++w) {
                if (i0 + (uint32_t)w * kThreads < nlist) {
                    const float v = vv[w];
                    bad_load |= !finite_f(v);
                    const uint32_t key = okey(v);
                    if (key < lk[KL - 1]) { // insertion into the ascending register list
                        lk[KL - 1] = key;
#pragma unroll
                        for (int u = KL - 1; u > 0; --u) {
                            const uint32_t a = lk[u - 1], b = lk[u];
                            lk[u - 1] = a < b ? a : b;
                            lk[u] = a < b ? b : a;
                        }
                    }
                }
            }
        }
        const uint32_t B = kth_key([&](auto&& body) {
#pragma unroll
            for (int u = 0; u < KL; ++u) if (lk[u] != 0xffffffffu) body(0u, lk[u]);
        }, nprobe - 1);
        const uint32_t cutB = widen(B);
        if (tid == 0) s_nc = 0;
        __syncthreads();
        if (cutB != 0xffffffffu && s_bad == 0) {
            for (uint32_t i0 = tid; i0 < nlist; i0 += kThreads * KU) {
                float vv[KU];
#pragma unroll
                for (int w = 0; w < KU; ++w) { const uint32_t i = i0 + (uint32_t)w * kThreads; vv[w] = i < nlist ? grow[i] : 0.0f; }
#pragma unroll
                for (int w = 0; w < KU; ++w) {
                    const uint32_t i = i0 + (uint32_t)w * kThreads;
                    const uint32_t key = okey(vv[w]);
                    if (i < nlist && key <= cutB) {
                        const uint32_t pp = atomicAdd(&s_nc, 1u);
                        if (pp < G.cand_cap) cand[pp] = ((uint64_t)key << 32) | i;
                    }
                }
            }
            __syncthreads();
            n_cand = s_nc;
            use_cand = n_cand <= G.cand_cap && n_cand >= nprobe;
        }
        __syncthreads();
    }
    uint32_t tau_key = 0;
    if (!all) tau_key = kth_key(each_key, nprobe - 1);
    else { // (non-finite scores must still be seen)
        each_key([&](uint32_t, uint32_t) {});
        if (bad_load) s_bad = 1;
        __syncthreads();
    }
    SSTAMP(); // 2: radix select done
    // 2. shortlist: approximate score within 2*eps of tau
    const uint32_t cut = all ? 0xffffffffu : widen(tau_key);
    __shared__ uint32_t s_cntle, s_minab; // keys <= tau (more than nprobe: ties at tau) / smallest key above tau
    for (uint32_t i = tid; i < cap2; i += kThreads) keys[i] = ~0ull;
    __shared__ uint32_t s_z0, s_tub, s_flag, s_need, s_dead, s_maxh, s_nh, s_nk, s_head[kSelHead]; // (s_tub, s_maxh: total_cmp keys of floats)
    __shared__ unsigned long long s_memvec;
    if (tid == 0) { s_cntle = 0; s_minab = 0xffffffffu;
                    s_z0 = 0; s_tub = (uint32_t)total_key(INFINITY); s_flag = 0; s_need = 0; s_dead = 0; s_maxh = (uint32_t)total_key(-INFINITY); s_memvec = 0; s_nh = 0; s_nk = 0; }
    if (tid < kSelHead) s_head[tid] = 0xffffffffu;
    __syncthreads();
    bool fallback = s_bad != 0 || P.force_fallback != 0;
    if (!fallback) {
        uint32_t cle = 0, mab = 0xffffffffu;
        each_key([&](uint32_t i, uint32_t key) {
            if (key <= cut) {
                const uint32_t p = atomicAdd(&s_cnt, 1u);
                if (p < cap2) keys[p] = ((uint64_t)key << 32) | i; // approximate key | cid; the exact key after scoring
                if (all || key <= tau_key) ++cle;
                else mab = key < mab ? key : mab;
            }
        });
        cle = wave_sum_u32(cle);
        mab = wave_min_u32(mab);
        if ((tid & 63u) == 0) { if (cle) atomicAdd(&s_cntle, cle); if (mab != 0xffffffffu) atomicMin(&s_minab, mab); }
        __syncthreads();
        fallback = s_cnt > cap2 || s_cnt < nprobe;
    }
    __syncthreads();
    SSTAMP(); // 3: shortlist collected
    const uint32_t l2 = tid & 1u, grp = tid >> 1; // 128 pairs in flight
    uint32_t m_scan = nprobe;      // lists that go to the scan (a prefix of keys[] in exact order at the end)
    uint32_t dbg_tub = 0x7fc00000u, dbg_z = 0; // diagnostics taps: bits of T_ub (NaN: lazy path not entered) | z0, shortlist, scored
    uint32_t dead_vec = 0;         // (thread 0) vectors of dead members
    if (!fallback) {
        const uint32_t n = s_cnt;
        // 3. certain members.  No sort by approximate cost is needed: the (nprobe+1)-th smallest approximate key is tau itself
        // when more than nprobe shortlist keys are <= tau (ties), else the smallest key above tau (both counted by the
        // shortlist pass).
        LSTAMP(0);
        __shared__ uint32_t s_wmin[4][kSelHead], s_todo[kSelTodoMax]; // s_todo: positions of the entries that need an exact score
        const uint32_t nown = (n + kThreads - 1 - tid) / kThreads; // entries tid, tid + 256, ... of this thread (n > tid)
        unsigned long long m_certain = 0ull, m_dead = 0ull, m_apx = 0ull;
        // what the classification (and the head selection) need of this thread's FIRST entry is requested now: one
        // round trip under the passes below instead of one in front of each of them (entries 256.. load on demand)
        BlockSummary ls_own = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0u, 0u};
        uint32_t gb_own = 0, n_own = 0;
        float cn_own = 0.0f;
        if (P.lazy && tid < n) {
            const uint32_t cid = (uint32_t)keys[tid];
            ls_own = P.lsum[cid]; gb_own = P.list_gb0[cid]; n_own = P.list_n[cid];
            if (metric == 1) cn_own = P.cnorm2[cid];
        }
        const uint32_t bound_key = (!all && s_cntle > nprobe) ? tau_key : s_minab;
        const float cost_bound = (all || bound_key == 0xffffffffu) ? INFINITY : cost_of(bound_key);
        {
            uint32_t cnt = 0;
            for (uint32_t u = 0; u < nown && tid < n; ++u) {
                const uint32_t k32 = (uint32_t)(keys[tid + u * kThreads] >> 32);
                if (all || k32 <= tau_key) m_apx |= 1ull << u; // the approximate probe set (exactly nprobe lists unless keys tie at tau)
                // (2.01: the sum below is itself rounded in f32 — up to 2^-24 |cost|, the order of the 0.001 eps of headroom
                // eps carries)
                const bool cert = cost_of(k32) + 2.01f * eps < cost_bound;
                if (cert) { m_certain |= 1ull << u; ++cnt; }
            }
            cnt = wave_sum_u32(cnt);
            if ((tid & 63u) == 0 && cnt) atomicAdd(&s_z0, cnt);
        }
        // the kSelHead smallest distinct approximate keys: per wave on DPP, then the 4 x 4 merged by every thread
        {
            uint32_t prev = 0;
            const uint32_t wave = tid >> 6;
#pragma unroll
            for (uint32_t r = 0; r < kSelHead; ++r) {
                uint32_t best = 0xffffffffu;
                for (uint32_t u = 0; u < nown && tid < n; ++u) {
                    const uint32_t k32 = (uint32_t)(keys[tid + u * kThreads] >> 32);
                    if ((r == 0 || k32 > prev) && k32 < best) best = k32;
                }
                best = wave_min_u32(best);
                if ((tid & 63u) == 0) s_wmin[wave][r] = best;
                prev = best;
            }
        }
        __syncthreads();
        const uint32_t z0 = s_z0; // certain members = the z0 smallest approximate costs
#ifdef RBQ_PREP_STAMPS
        const bool lazy_ok = false;
#else
        const bool lazy_ok = P.lazy != 0 && z0 > 0 && n - z0 <= kSelZoneMax && qc.amax <= 65535.0f && P.top_k > 0;
#endif
        uint32_t h = 0;
        if (lazy_ok) {
            // 4th smallest distinct key level of the workgroup; the head = up to kSelHead CERTAIN entries at or below it
            uint32_t prev = 0, lvl = 0xffffffffu;
#pragma unroll
            for (uint32_t r = 0; r < kSelHead; ++r) {
                uint32_t best = 0xffffffffu;
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (uint32_t j = 0; j < kSelHead; ++j) {
                        const uint32_t v = s_wmin[w][j];
                        if ((r == 0 || v > prev) && v < best) best = v;
                    }
                if (best != 0xffffffffu) lvl = best;
                prev = best;
            }
            for (uint32_t u = 0; u < nown && tid < n; ++u) {
                const uint32_t i = tid + u * kThreads;
                if (((m_certain >> u) & 1ull) && (uint32_t)(keys[i] >> 32) <= lvl) {
                    const uint32_t p = atomicAdd(&s_nh, 1u);
                    if (p < kSelHead) s_head[p] = i;
                }
            }
            __syncthreads();
            h = s_nh < kSelHead ? s_nh : kSelHead;
        }
        const uint32_t hp0 = lazy_ok ? s_head[0] : 0xffffffffu, hp1 = lazy_ok ? s_head[1] : 0xffffffffu,
                       hp2 = lazy_ok ? s_head[2] : 0xffffffffu, hp3 = lazy_ok ? s_head[3] : 0xffffffffu; // positions of the head entries
        auto is_head = [&](uint32_t i) { return i == hp0 || i == hp1 || i == hp2 || i == hp3; };
        bool lazy = false;
        LSTAMP(1);
        if (lazy_ok && h > 0) {
            // head geometry; the head lists are the first entries of the to-score list.  T_ub is computed from their
            // APPROXIMATE costs widened by eps (the upper ends of g_add and g_err: block_ub() grows with both), so that no
            // scoring round sits in front of the classification — everything that needs an exact score is scored in one
            // round afterwards.
            __shared__ uint32_t s_hgb[kSelHead], s_hn[kSelHead];
            __shared__ float s_hcn[kSelHead];
            if (tid < kSelHead) { s_hgb[tid] = 0; s_hn[tid] = 0; s_hcn[tid] = 0.0f; }
            if (tid == 0) s_need = h;
            __syncthreads();
#pragma unroll
            for (uint32_t r = 0; r < kSelHead; ++r) { // the owner of head entry r publishes its geometry
                const uint32_t pos = r == 0 ? hp0 : r == 1 ? hp1 : r == 2 ? hp2 : hp3;
                if (r < h && (pos & (kThreads - 1)) == tid) {
                    uint32_t gb = gb_own, nv = n_own;
                    float cn = cn_own;
                    if (pos >= (uint32_t)kThreads) { const uint32_t cid = (uint32_t)keys[pos]; gb = P.list_gb0[cid]; nv = P.list_n[cid]; if (metric == 1) cn = P.cnorm2[cid]; }
                    s_hgb[r] = gb; s_hn[r] = nv; s_hcn[r] = cn;
                }
            }
            __syncthreads();
            LSTAMP(2);
            // upper ends of g_add / g_err and geometry of head list r
            auto head_info = [&](uint32_t r, float& g_add, float& g_err, uint32_t& gb, uint32_t& nvec) {
                const uint64_t key = keys[s_head[r]];
                const float ci = cost_of((uint32_t)(key >> 32));
                g_add = ci + 1.01f * eps; // (1.01: the rounding of this very sum)
                if (metric == 0) g_err = sqrtf(fmaxf(g_add, 0.0f));
                else g_err = sqrtf(fmaxf(qc.qnorm2 + s_hcn[r] + 2.0f * ci + 4.0f * eps, 0.0f)); // (as in the classification below)
                gb = s_hgb[r]; nvec = s_hn[r];
            };
            // T_ub from up to 256 head blocks: thread t owns candidate block t
            const uint32_t nb0 = (s_hn[0] + 31u) >> 5, nb1 = (s_hn[1] + 31u) >> 5, nb2 = (s_hn[2] + 31u) >> 5, nb3 = (s_hn[3] + 31u) >> 5;
            static_assert(kSelHead == 4, "four head lists");
            const uint32_t tot = nb0 + nb1 + nb2 + nb3;
            const uint32_t ncand = tot < (uint32_t)kThreads ? tot : (uint32_t)kThreads;
            float myU = INFINITY;
            uint32_t myN = 0;
            if (tid < ncand) {
                uint32_t r = 0, b = tid;
                if (b >= nb0) { b -= nb0; r = 1; if (b >= nb1) { b -= nb1; r = 2; if (b >= nb2) { b -= nb2; r = 3; } } }
                float g_add, g_err; uint32_t gb, nvec;
                head_info(r, g_add, g_err, gb, nvec);
                const BlockSummary bs = P.bsum[gb + b];
                const BlockSummaryEx bx = P.bsumx[gb + b];
                myU = block_ub(bs, bx, g_add, g_err, qc, D, P.ex_bits, P.slack) + 1e-4f * eps; // (+: |g_add| in the rounding slack is taken at the upper end)
                myN = (b + 1 == ((nvec + 31u) >> 5)) ? nvec - b * 32u : 32u;
            }
            float* candU = reinterpret_cast<float*>(part);
            candU[tid] = myU; hist[tid] = myN;
            if (tid < h) { // largest possible head cost
                float g_add, g_err; uint32_t gb, nvec;
                head_info(tid, g_add, g_err, gb, nvec);
                atomicMax(reinterpret_cast<int*>(&s_maxh), total_key(g_add)); // (s_maxh starts at the key of -inf)
            }
            __syncthreads();
            if (tid < ncand && myU < INFINITY) {
                uint32_t cum = 0;
#pragma unroll 8
                for (uint32_t j = 0; j < ncand; ++j) cum += candU[j] <= myU ? hist[j] : 0u;
                if (cum >= P.top_k) atomicMin(reinterpret_cast<int*>(&s_tub), total_key(myU));
            }
            __syncthreads();
            // Under a filter the Cauchy-Schwarz bound proves nothing (it counts vectors that may never be pushed): the bound comes from
            // the exact head evaluation alone, which tests every vector's filter bit (round 4); +inf = this query stays eager.
            float T_ub = P.fault_dead_all ? -INFINITY : key_to_float((int32_t)s_tub); // (fault_dead_all: test-only fault injection)
            const float cost_maxh = key_to_float((int32_t)s_maxh);
            bool hx_done = false;
            if (P.filter) { // (uniform)
                T_ub = (G.hx_nv && P.head_exact) ? head_exact_bound(P, G, qc, keys, s_head, s_hgb, s_hn, s_hcn, h, eps, reinterpret_cast<unsigned char*>(rows), qrot, part, hist, q, tid, cost_of) : INFINITY;
                hx_done = true;
                if (P.fault_dead_all) T_ub = -INFINITY; // (the test-only fault injection also under a filter: ADVICE r4)
            }
            lazy = T_ub < INFINITY;
            dbg_tub = __float_as_uint(T_ub);
            if (tid < h) s_todo[tid] = s_head[tid];
            LSTAMP(3);
            if (lazy) {
                // 3. classification of the entries beyond the head against a bound T; what is alive goes straight to the to-score list
                auto classify = [&](float T) {
                    for (uint32_t u = 0; u < nown && tid < n; ++u) {
                        const uint32_t i = tid + u * kThreads;
                        if (is_head(i)) continue;
                        const uint64_t key = keys[i];
                        const uint32_t cid = (uint32_t)key;
                        const float ci = cost_of((uint32_t)(key >> 32));
                        const float clo = ci - 1.01f * eps, chi = ci + 1.01f * eps; // (1.01: the rounding of these sums)
                        float ge_lo, ge_hi;
                        if (metric == 0) { ge_lo = sqrtf(fmaxf(clo, 0.0f)); ge_hi = sqrtf(fmaxf(chi, 0.0f)); }
                        else { // canonical squared distance = |q|^2 + |c|^2 - 2 dot, within 4 eps of this evaluation (rank_mfma.hpp header)
                            const float da = qc.qnorm2 + (u == 0 ? cn_own : P.cnorm2[cid]) + 2.0f * ci;
                            ge_lo = sqrtf(fmaxf(da - 4.0f * eps, 0.0f)); ge_hi = sqrtf(fmaxf(da + 4.0f * eps, 0.0f));
                        }
                        const BlockSummary ls = u == 0 ? ls_own : P.lsum[cid];
                        const bool dead = clo > cost_maxh && list_bound_reaches(ls, clo, chi, ge_lo, ge_hi, qc, T);
                        if (dead) m_dead |= 1ull << u;
                        else {
                            if (!((m_certain >> u) & 1ull)) s_flag = 1u; // a zone list is alive
                            const uint32_t p = atomicAdd(&s_need, 1u);
                            if (p < kSelTodoMax) s_todo[p] = i;
                        }
                    }
                    __syncthreads();
                };
                classify(T_ub);
                // 3b. EXACT HEAD EVALUATION (round 4).  The Cauchy-Schwarz bound ignores the direction of q - c: T_ub is 4-5 x the
                // final k-th distance, and at small dimensions (or 65 536 lists) most probed lists stay alive under it.  When
                // more than kHxTrigger do, the first vectors of the NEAREST head list are evaluated for real: thread v
                // computes vector v's 1-bit estimate and lower bound exactly as k_scan will (same u8 LUT, same operation
                // sequence, g_add / g_err at the ends of their intervals that make both LARGER), the 2k smallest estimates
                // are refined with the ex codes in the reference's FMA order, U_v := max(refined distance, lower bound), and
                // T' := the k-th smallest U_v.  Same claim as for T_ub (any k vectors the reference visits before the dead
                // lists will do): each of them, if it sees a threshold > T' >= lb_v, is evaluated and pushed with a distance
                // <= T'.  Lists are then classified again against min(T_ub, T').
                // (the count is read by every thread and fenced by a barrier BEFORE anyone may change it again: the zone step
                // below adds to s_need, and a thread that read the raised count would take this branch — and its barriers — alone.
                // That race was a rare "memory access fault" in the first builds of this step.)
                const uint32_t need1 = s_need;
                __syncthreads();
                if (G.hx_nv && P.head_exact && !hx_done && need1 - h > kHxTrigger) { // (uniform)
                    const float T2 = head_exact_bound(P, G, qc, keys, s_head, s_hgb, s_hn, s_hcn, h, eps, reinterpret_cast<unsigned char*>(rows), qrot, part, hist, q, tid, cost_of);
                    if (T2 < T_ub) {
                        m_dead = 0ull;
                        if (tid == 0) { s_need = h; s_flag = 0u; }
                        __syncthreads();
                        classify(T2);
                        dbg_tub = __float_as_uint(T2);
                    }
                }
                const bool zone_scored = s_flag != 0u || P.exact_members != 0;
                if (zone_scored) { // (uniform; without diagnostics only when a boundary-zone list is alive) the dead zone lists too
                    for (uint32_t u = 0; u < nown && tid < n; ++u) {
                        const uint32_t i = tid + u * kThreads;
                        if (!is_head(i) && ((m_dead >> u) & 1ull) && !((m_certain >> u) & 1ull)) {
                            const uint32_t p = atomicAdd(&s_need, 1u);
                            if (p < kSelTodoMax) s_todo[p] = i;
                        }
                    }
                    __syncthreads();
                }
                LSTAMP(4);
                lazy = s_need <= kSelTodoMax;
                dbg_z = (z0 & 0x3ffu) | ((n & 0x3ffu) << 10) | (((s_need - h) & 0xfffu) << 20);
                if (lazy) {
                    // 4. score what is needed: the head and everything alive, in one round
                    const uint32_t m = s_need;
                    if (m) { // uniform
                        auto todo_idx = [&](uint32_t j) { return s_todo[j]; };
                        if (G.stage_rows && m <= 2u * G.stage_rows) canon_score_staged(keys, qrot, P.cent, D, metric, grow, m, tid, rows, G.stage_rows, todo_idx);
                        else {
                            if (m <= 64u) canon_score_octets(keys, qrot, P.cent, D, metric, grow, m, tid, todo_idx);
                            else canon_score_pairs(keys, qrot, P.cent, D, metric, grow, m, tid, todo_idx);
                            __threadfence_block();
                            __syncthreads();
                        }
                    }
                    LSTAMP(5);
                    // membership: certain, or a scored zone list with fewer than nprobe - z0 smaller zone keys (the zone
                    // entries are marked in a bitset: only this — rare — step needs to know them from other threads)
                    __shared__ uint32_t s_zone[512];
                    __shared__ unsigned long long s_kept[kSelKeptMax];
                    if (zone_scored) {
                        for (uint32_t w = tid; w < 512; w += kThreads) s_zone[w] = 0u;
                        __syncthreads();
                        for (uint32_t u = 0; u < nown && tid < n; ++u)
                            if (!((m_certain >> u) & 1ull)) { const uint32_t i = tid + u * kThreads; atomicOr(&s_zone[i >> 5], 1u << (i & 31u)); }
                        __syncthreads();
                    }
                    unsigned long long m_keep = 0ull;
                    uint32_t dv = 0;
                    unsigned long long mv = 0;
                    for (uint32_t u = 0; u < nown && tid < n; ++u) {
                        const uint32_t i = tid + u * kThreads;
                        const bool cert = (m_certain >> u) & 1ull, dead = (m_dead >> u) & 1ull;
                        bool member = cert;
                        if (!cert && zone_scored) {
                            const uint64_t my = keys[i];
                            uint32_t less = 0;
                            for (uint32_t j = 0; j < n; ++j) less += ((s_zone[j >> 5] >> (j & 31u)) & 1u) && keys[j] < my ? 1u : 0u;
                            member = less < nprobe - z0;
                        }
                        if (dead && P.audit_dead) { // lazy_audit: the dropped lists, for the oracle to look at (no decision changes)
                            uint32_t* ad = P.audit_dead + (size_t)q * (kAuditCap + 1);
                            const uint32_t pa = atomicAdd(ad, 1u);
                            if (pa < kAuditCap) ad[1 + pa] = (uint32_t)keys[i];
                        }
                        if (member && !dead) {
                            m_keep |= 1ull << u;
                            const uint32_t p = atomicAdd(&s_nk, 1u);
                            if (p < kSelKeptMax) s_kept[p] = keys[i];
                        }
                        // vectors of the probe set: exact under diagnostics; for the profile's algorithmic bytes alone, the
                        // approximate probe set (the boundary zone is resolved only when something in it is alive)
                        if (P.exact_members ? member : (P.prof_total != nullptr && ((m_apx >> u) & 1ull))) {
                            const uint32_t nvec = P.list_n[(uint32_t)keys[i]];
                            mv += nvec;
                            if (dead && P.exact_members) dv += nvec;
                        }
                    }
                    if (dv) atomicAdd(&s_dead, dv);
                    if (mv) atomicAdd(&s_memvec, mv);
                    __syncthreads(); // the kept keys are complete (and every membership count has read the zone keys)
                    if (s_nk <= kSelKeptMax) {
                        // the usual case, a handful of lists: rank sort from the side buffer straight into keys[0 .. m_scan)
                        m_scan = s_nk;
                        if (tid < m_scan) {
                            const unsigned long long my = s_kept[tid];
                            uint32_t rk = 0;
#pragma unroll 4
                            for (uint32_t j = 0; j < m_scan; ++j) rk += s_kept[j] < my ? 1u : 0u;
                            keys[rk] = my;
                        }
                        __syncthreads();
                    } else {
                        // in-place compaction of the kept exact keys, chunk by chunk (a chunk's writes land at or below its
                        // own first entry), then the tail is cleared for the sort
                        uint32_t base = 0;
                        for (uint32_t c0 = 0, u = 0; c0 < n; c0 += kThreads, ++u) {
                            const uint32_t i = c0 + tid;
                            const bool keep = i < n && ((m_keep >> u) & 1ull);
                            const uint64_t v = i < n ? keys[i] : ~0ull;
                            uint32_t total;
                            const uint32_t off = block_scan_excl256(keep ? 1u : 0u, s_w4, tid, total);
                            if (keep) keys[base + off] = v;
                            base += total;
                        }
                        __syncthreads();
                        m_scan = base;
                        for (uint32_t i = m_scan + tid; i < cap2; i += kThreads) keys[i] = ~0ull;
                        __syncthreads();
                        sort_keys(keys, m_scan, cap2, tid);
                    }
                    LSTAMP(6);
                    if (tid == 0) dead_vec = s_dead;
                }
            }
        }
        if (!lazy) {
            // eager: exact canonical scores of the whole shortlist, exact (score, cid) keys, sort; members = the first nprobe
            auto all_idx = [](uint32_t j) { return j; };
            canon_score_pairs(keys, qrot, P.cent, D, metric, grow, n, tid, all_idx);
            __syncthreads();
            SSTAMP(); // 4: canonical scores
            sort_keys(keys, n, cap2, tid);
        }
#ifdef RBQ_SEL_STAMPS
        else SSTAMP();
#endif
    } else {
        // fallback (rare: shortlist overflow or non-finite approximate scores): overwrite this query's row
        // with canonical scores of EVERY list, then the exact 64-bit (score, cid) radix select of k_select
        if (tid == 0 && P.fallback_count) atomicAdd(P.fallback_count, 1u);
        float* row = grow;
        for (uint32_t base = 0; base < nlist; base += kThreads / 2) {
            const uint32_t cid = base + grp;
            if (cid < nlist) {
                const float* c = P.cent + (size_t)cid * D;
                const float s = metric == 0 ? canon_pair2<0>(qrot, c, D, l2) : canon_pair2<1>(qrot, c, D, l2);
                if (l2 == 0) row[cid] = s;
            }
        }
        __threadfence_block();
        __syncthreads();
        __shared__ unsigned long long s_prefix64, s_mask64;
        if (tid == 0) { s_prefix64 = all ? ~0ull : 0ull; s_mask64 = 0; s_k = nprobe - 1; s_cnt = 0; }
        __syncthreads();
        if (!all) {
            for (int pass = 7; pass >= 0; --pass) {
                const int shift = pass * 8;
                hist[tid] = 0;
                __syncthreads();
                const unsigned long long prefix = s_prefix64, mask = s_mask64;
                for (uint32_t i = tid; i < nlist; i += kThreads) {
                    const uint64_t key = make_key(row[i], i, metric);
                    if ((key & mask) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
                }
                __syncthreads();
                if (tid == 0) {
                    uint32_t k = s_k, cum = 0, bb = 0;
                    for (; bb < 256; ++bb) {
                        const uint32_t h = hist[bb];
                        if (k < cum + h) break;
                        cum += h;
                    }
                    s_k = k - cum;
                    s_prefix64 = prefix | ((unsigned long long)bb << shift);
                    s_mask64 = mask | (0xffull << shift);
                }
                __syncthreads();
            }
        }
        const uint64_t kstar = s_prefix64;
        for (uint32_t i = tid; i < cap2; i += kThreads) keys[i] = ~0ull;
        __syncthreads();
        for (uint32_t i = tid; i < nlist; i += kThreads) {
            const uint64_t key = make_key(row[i], i, metric);
            if (key <= kstar) {
                const uint32_t p = atomicAdd(&s_cnt, 1u);
                if (p < cap2) keys[p] = key;
            }
        }
        __syncthreads();
        for (uint32_t k = 2; k <= cap2; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < cap2; i += kThreads) {
                    const uint32_t ixj = i ^ j;
                    if (ixj > i) {
                        const uint64_t a = keys[i], b2 = keys[ixj];
                        const bool up = (i & k) == 0;
                        if ((a > b2) == up) { keys[i] = b2; keys[ixj] = a; }
                    }
                }
                __syncthreads();
            }
    }

    SSTAMP(); // 5: sorted
    // 4. per-probe constants (src/ivf.rs:1850-1857), block counts, work list — as k_select — for the m_scan lists that go
    // to the scan.  IP needs the canonical centroid distance of every such list as well (g_error): parked in the score row
    // by the scorers, fetched into the (free) tail of the key buffer.
    const uint32_t np = m_scan;
    float* dist_ip = reinterpret_cast<float*>(keys + nprobe); // cap2 >= 2*nprobe: room for nprobe floats
    if (metric == 1 && !fallback) {
        __threadfence_block();
        __syncthreads();
        const volatile float* vrow = grow; // written by other threads of this workgroup: read past the vector L1
        for (uint32_t r = tid; r < np; r += kThreads) {
            dist_ip[r] = vrow[(uint32_t)(keys[r] & 0xffffffffu)]; // keys[0..nprobe) and the tail do not overlap
        }
        __syncthreads();
    } else if (metric == 1) {
        for (uint32_t i0 = 0; i0 < np; i0 += kThreads / 2) {
            const uint32_t r = i0 + grp;
            float d = 0.0f;
            if (r < np) {
                const uint32_t cid = (uint32_t)(keys[r] & 0xffffffffu);
                d = canon_pair2<0>(qrot, P.cent + (size_t)cid * D, D, l2);
            }
            __syncthreads(); // all reads of keys[r] of this round done before the tail is written
            if (r < np && l2 == 0) dist_ip[r] = d;
        }
        __syncthreads();
    }
    // per-probe constants from the sorted key (the same operations wherever they are needed)
    auto probe_consts = [&](uint32_t r) -> ProbeInfo {
        const uint64_t key = keys[r];
        int32_t k = (int32_t)((uint32_t)(key >> 32) ^ 0x80000000u);
        if (metric == 1) k = ~k;
        const float s = key_to_float(k);
        float dist, dot;
        // L2: score IS the centroid distance; the dot product only feeds the non-finite lower-bound
        // fallback of the IP metric (src/ivf.rs:2031-2042), so it is not computed here.
        if (metric == 0) { dist = s; dot = 0.0f; }
        else { dot = s; dist = dist_ip[r]; }
        ProbeInfo pi;
        pi.g_add = metric == 0 ? dist : -dot;
        pi.g_err = sqrtf(dist);
        pi.dotqc = dot;
        pi.cid = (uint32_t)(key & 0xffffffffu);
        return pi;
    };
    // first stream position of every probe: workgroup prefix sum over the lists' block counts
    uint32_t* pstart = reinterpret_cast<uint32_t*>(dist_ip + nprobe); // cap2 >= 2*nprobe: room for nprobe words more
    const uint32_t per = (np + kThreads - 1) / kThreads;
    const uint32_t r0 = tid * per < np ? tid * per : np, r1 = (r0 + per < np) ? r0 + per : np;
    uint32_t local = 0;
    unsigned long long local_vec = 0;
    for (uint32_t r = r0; r < r1; ++r) {
        const ProbeInfo pi = probe_consts(r);
        P.probe[(size_t)q * nprobe + r] = pi;
        const uint32_t n = P.list_n[pi.cid];
        if (G.stage) {
            pgeo[r] = __float_as_uint(pi.g_add);
            pgeo[nprobe + r] = __float_as_uint(pi.g_err);
            pgeo[2 * nprobe + r] = P.list_gb0[pi.cid];
            pgeo[3 * nprobe + r] = n;
        }
        local += (n + 31u) >> 5;
        local_vec += n;
    }
    if (local_vec) atomicAdd(&s_nvec, local_vec);
    uint32_t run;
    uint32_t acc0 = block_scan_excl256(local, s_w4, tid, run);
    for (uint32_t r = r0; r < r1; ++r) {
        pstart[r] = acc0;
        acc0 += ((G.stage ? pgeo[3 * nprobe + r] : P.list_n[(uint32_t)(keys[r] & 0xffffffffu)]) + 31u) >> 5;
    }
    if (tid == 0) {
        P.nstream[q] = run;
        // vectors of the probed lists (the scan's algorithmic work): scanned + dead members (exact when exact_members)
        const unsigned long long nv = np == nprobe ? s_nvec : s_memvec;
        P.nvec[q] = nv;
        if (P.prof_total) atomicAdd(P.prof_total + prof_stripe(q), nv);
        if (P.dead_skipped) { // [1]: lists that go to the scan; [2], [3]: diagnostics taps of the lazy selection
            P.dead_skipped[q] = dead_vec; P.dead_skipped[P.nq + q] = np;
            P.dead_skipped[2 * (size_t)P.nq + q] = dbg_tub; P.dead_skipped[3 * (size_t)P.nq + q] = dbg_z;
        }
    }
    __syncthreads();
    // the block stream, four entries per thread and step (their block summaries are all requested before the
    // first bound is evaluated): probe = last r with pstart[r] <= i
    StreamItem* out = P.wl + (size_t)q * P.wl_stride;
    for (uint32_t base = 0; base < run; base += 4 * kThreads) {
        uint32_t rr[4], gbb[4], nvv[4];
        BlockSummary bs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = base + u * kThreads + tid;
            rr[u] = 0; gbb[u] = 0; nvv[u] = 0;
            if (i < run) {
                uint32_t lo = 0, hi = np; // pstart[lo] <= i < pstart[hi] (pstart[np] = run)
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (pstart[mid] <= i) lo = mid; else hi = mid;
                }
                // lists without vectors share their start with the next one: the LAST r with pstart[r] <= i owns i
                uint32_t n, gb;
                if (G.stage) { gb = pgeo[2 * nprobe + lo]; n = pgeo[3 * nprobe + lo]; }
                else {
                    const uint32_t cid = (uint32_t)(keys[lo] & 0xffffffffu);
                    n = P.list_n[cid]; gb = P.list_gb0[cid];
                }
                const uint32_t nb = (n + 31u) >> 5, b = i - pstart[lo];
                rr[u] = lo;
                gbb[u] = gb + b;
                nvv[u] = (b + 1 == nb) ? n - b * 32u : 32u;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) bs[u] = P.bsum[gbb[u]]; // (block 0 for the idle slots)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = base + u * kThreads + tid;
            if (i < run) {
                float g_add, g_err;
                if (G.stage) { g_add = __uint_as_float(pgeo[rr[u]]); g_err = __uint_as_float(pgeo[nprobe + rr[u]]); }
                else { const ProbeInfo pi = probe_consts(rr[u]); g_add = pi.g_add; g_err = pi.g_err; }
                StreamItem wi;
                wi.gblock = gbb[u];
                wi.rank_nvalid = (rr[u] << 6) | nvv[u];
                wi.lbmin = block_lbmin(bs[u], g_add, g_err, qc);
                wi.pad = 0;
                out[i] = wi;
            }
        }
    }
#ifdef RBQ_SEL_STAMPS
    SSTAMP(); // 6: probe info + stream written
    if (tid == 0) {
        unsigned long long pk = 0;
#if RBQ_SEL_STAMPS == 2
        pk = (((ts[0] >> 4) & 0xffffffffull) << 32) | ((ts[6] - ts[0]) & 0xffffffffull); // absolute start (16-cycle units) | duration
        P.nvec[q] = pk;
#elif RBQ_SEL_STAMPS == 3
        P.nvec[q] = ((unsigned long long)m_scan << 32) | s_cnt; // lists scanned | shortlist size
#elif RBQ_SEL_STAMPS == 4
        for (int t = 0; t < 6; ++t) pk |= ((ls[t + 1] > ls[t] ? (ls[t + 1] - ls[t]) >> 7 : 0ull) & 0x3ffull) << (10 * t); // 128-cycle units
        P.nvec[q] = pk;
#else
        for (int t = 0; t < 6; ++t) pk |= (((ts[t + 1] - ts[t]) >> 8) & 0x3ffull) << (10 * t); // 256-cycle units, 10 bits each
        P.nvec[q] = pk | ((unsigned long long)s_cnt << 60);
#endif
    }
#endif
}

// MSTG posting-list scan: the caller supplies each query's posting lists; this emits the
// per-list constants of search_posting_list_fastscan (g_add = l2_distance_sqr(query, centroid) or -dot in
// canonical order, g_error = 0; src/mstg/index.rs:227-231) and the block work list, in the given order.
// dynamic LDS: qrot[D] f32 | part[256] u32
__global__ __launch_bounds__(kThreads) void k_probes_given(const uint32_t* __restrict__ list_ids,
                                                           const uint32_t* __restrict__ list_counts, uint32_t max_lists,
                                                           uint32_t nlist, int metric, const float* __restrict__ rot,
                                                           const float* __restrict__ cent, uint32_t D,
                                                           const uint32_t* __restrict__ list_gb0,
                                                           const uint32_t* __restrict__ list_n,
                                                           ProbeInfo* __restrict__ probe, StreamItem* __restrict__ wl,
                                                           uint64_t wl_stride, uint32_t* __restrict__ nstream,
                                                           const QueryConsts* __restrict__ consts,
                                                           const BlockSummary* __restrict__ bsum) {
    extern __shared__ __align__(16) unsigned char smraw[];
    float* qrot = reinterpret_cast<float*>(smraw);
    uint32_t* part = reinterpret_cast<uint32_t*>(qrot + D);
    const uint32_t q = blockIdx.x, tid = threadIdx.x, l2 = tid & 1u, grp = tid >> 1;
    for (uint32_t i = tid; i < D; i += kThreads) qrot[i] = rot[(size_t)q * D + i];
    __syncthreads();
    const uint32_t cnt = list_counts[q] < max_lists ? list_counts[q] : max_lists;
    const uint32_t* mine = list_ids + (size_t)q * max_lists;
    for (uint32_t i0 = 0; i0 < cnt; i0 += kThreads / 2) {
        const uint32_t r = i0 + grp;
        if (r < cnt) {
            const uint32_t cid = mine[r];
            float s = 0.0f;
            if (cid < nlist) {
                const float* c = cent + (size_t)cid * D;
                s = metric == 0 ? canon_pair2<0>(qrot, c, D, l2) : canon_pair2<1>(qrot, c, D, l2);
            }
            if (l2 == 0) {
                ProbeInfo pi;
                pi.g_add = metric == 0 ? s : -s;
                pi.g_err = 0.0f;
                pi.dotqc = 0.0f;
                pi.cid = cid;
                probe[(size_t)q * max_lists + r] = pi;
            }
        }
    }
    const uint32_t per = (cnt + kThreads - 1) / kThreads;
    const uint32_t r0 = tid * per, r1 = (r0 + per < cnt) ? r0 + per : cnt;
    uint32_t local = 0;
    for (uint32_t r = r0; r < r1; ++r) {
        const uint32_t cid = mine[r];
        if (cid < nlist) local += (list_n[cid] + 31u) >> 5;
    }
    part[tid] = local;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < kThreads; ++i) { const uint32_t v = part[i]; part[i] = run; run += v; }
        nstream[q] = run;
    }
    __syncthreads();
    uint64_t pos = (uint64_t)q * wl_stride + part[tid];
    const QueryConsts qcs = consts[q];
    for (uint32_t r = r0; r < r1; ++r) {
        const uint32_t cid = mine[r];
        if (cid >= nlist) continue;
        const uint32_t n = list_n[cid], gb = list_gb0[cid], nb = (n + 31u) >> 5;
        const ProbeInfo pi = probe[(size_t)q * max_lists + r];
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t nv = (b + 1 == nb) ? n - b * 32u : 32u;
            StreamItem wi;
            wi.gblock = gb + b;
            wi.rank_nvalid = (r << 6) | nv;
            wi.lbmin = block_lbmin(bsum[gb + b], pi.g_add, pi.g_err, qcs);
            wi.pad = 0;
            wl[pos++] = wi;
        }
    }
}

} // namespace rbq
