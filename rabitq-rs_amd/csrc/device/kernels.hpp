// kernels.hpp — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the IVF+RaBitQ query path.
//
// Stages (one launch each per query batch):
//   k_prep         rotate (FHT-Kac / matrix) + query constants + u8 LUT      ref: src/rotation.rs:350-401,
//                                                                                 src/ivf.rs:798-845,862-878
//   k_rank_scores  canonical-order query x centroid scores                   ref: src/ivf.rs:1782-1789, src/math.rs:154-245
//   k_select       nprobe smallest (score,cid) keys, per-list g_add/g_error,
//                  and the per-query block work list                         ref: src/ivf.rs:1791-1857
//   k_scan         THE roofline kernel: streams the probed lists' sign codes,
//                  u8-LUT accumulate from LDS, fused estimator epilogue,
//                  lower-bound pruning, ex-code refine, exact sequential
//                  top-k replay                                              ref: src/ivf.rs:1901-2129, src/simd.rs:972-1184,
//                                                                                 :1835-1915,:2090-2140
// All floating point follows the oracle's operation order; the library is compiled with
// -ffp-contract=off and fuses only where the reference does (explicit fmaf).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "types.hpp"

namespace rbq {

// Block-level bound: accu of any code lies in [amin, amax] and lb is monotone in accu (direction = sign of
// f_rescale), so the epilogue's own operation sequence (compute_batch_distances_u16, AVX2 body: only the first
// op is fused) evaluated on the extremes of every operand is <= lb of every real vector of the block.
__device__ __forceinline__ float block_lbmin(const BlockSummary& bs, float g_add, float g_err, const QueryConsts& qc) {
    const float tA = fmaf(qc.delta, qc.amin, qc.sum_vl) + qc.k1x;
    const float tB = fmaf(qc.delta, qc.amax, qc.sum_vl) + qc.k1x;
    const float r0 = bs.fres_min * tA, r1 = bs.fres_min * tB, r2 = bs.fres_max * tA, r3 = bs.fres_max * tB;
    const float rmin = fminf(fminf(r0, r1), fminf(r2, r3)), rmax = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
    float elo = bs.fadd_min + g_add;
    elo = elo + rmin;
    const float lbmin = elo - bs.ferr_max * g_err;
    float ehi = bs.fadd_max + g_add;
    ehi = ehi + rmax;
    const float lbmax = ehi - bs.ferr_min * g_err;
    const bool fin = isfinite(r0) && isfinite(r1) && isfinite(r2) && isfinite(r3) && isfinite(lbmin) && isfinite(lbmax) &&
                     isfinite(elo) && isfinite(ehi);
    return (bs.usable && fin) ? lbmin : -INFINITY;
}

// ---- bounds used by the lazy probe selection (k_select_mfma) ------------------------------------------------------------
// Upper bound U of max(refined distance, lower bound) over the real vectors of one block of a list whose exact g_add /
// g_err are known, from the block's Cauchy-Schwarz terms (BlockSummaryEx, derivation at k_list_summaries):
//   refined distance <= S + g_add + B |q - c|,   lower bound <= 1-bit estimate <= S1 + g_add + B1 |q - c|
// in exact arithmetic on the kernel's own inputs; |q - c| <= g_err (1 + 1e-4) (g_err = sqrt of the canonical f32 sum of
// squares).  What the scan kernel COMPUTES differs from that by
//   * the u8 quantisation of the LUT (src/ivf.rs:798-845): every codebook entry within delta/2 (+ rounding) of its f32
//     value, D/4 codebooks:  |ip - <q, bit>| <= E_ip = (D/8) delta 1.01 + 1e-4 |q|_1   (the second term also covers the
//     sequential f32 sums behind k1x / kbx and sum_vl),
//   * the f32 summation of the ex-code dot (16-lane FMA chains + tree): <= 1e-3 (2^ex - 1) |q|_1  (actual: < 1.3e-4),
//   * the roundings of the handful of f32 operations of the two formulas: 1e-5 of the sum of the magnitudes involved
//     (each operation contributes at most 6e-8 of its operands), magnitudes bounded through the all-codes ranges.
// +inf when anything is not finite (such a block proves nothing).
__device__ __forceinline__ float block_ub(const BlockSummary& bs, const BlockSummaryEx& bx, float g_add, float g_err,
                                          const QueryConsts& qc, uint32_t D, uint32_t ex_bits, const SlackMul sm = SlackMul()) {
    const float ipA = fmaf(qc.delta, qc.amin, qc.sum_vl), ipB = fmaf(qc.delta, qc.amax, qc.sum_vl);
    const float tmax = fmaxf(fabsf(ipA + qc.k1x), fabsf(ipB + qc.k1x));
    const float ge = g_err * (1.0f + sm.ge * 1e-4f);
    const float E_ip = sm.eip * ((float)D * 0.125f * qc.delta * 1.01f + 1e-4f * qc.q1norm + 1e-6f * (fabsf(qc.sum_vl) + qc.delta * qc.amax));
    const float fres_abs = fmaxf(fabsf(bs.fres_min), fabsf(bs.fres_max)), fadd_abs = fmaxf(fabsf(bs.fadd_min), fabsf(bs.fadd_max));
    float est_hi = bx.S1 + g_add;
    est_hi += bx.B1 * ge;
    est_hi += fres_abs * E_ip;
    est_hi += sm.est * 1e-5f * (fadd_abs + fabsf(g_add) + fres_abs * tmax + fabsf(bx.S1) + bx.B1 * ge);
    const float ferr_abs = fmaxf(fabsf(bs.ferr_min), fabsf(bs.ferr_max));
    const float lb_hi = est_hi + fmaxf(-bs.ferr_min, 0.0f) * ge + sm.lb * 1e-5f * ferr_abs * ge; // lb = est - f_error * g_err
    float d_hi = est_hi; // ex_bits == 0: distance = est
    if (ex_bits) {
        const float cmax = (float)((1u << ex_bits) - 1u);
        const float uA = qc.scale * ipA + qc.exlo + qc.kbx, uB = qc.scale * ipB + qc.exhi + qc.kbx;
        const float umax = fmaxf(fabsf(uA), fabsf(uB));
        const float E_t = qc.scale * E_ip + sm.et * 1e-3f * cmax * qc.q1norm;
        d_hi = bx.S + g_add;
        d_hi += bx.B * ge;
        d_hi += bx.fres_ex_abs * E_t;
        d_hi += sm.dist * 1e-5f * (bx.fadd_ex_abs + fabsf(g_add) + bx.fres_ex_abs * umax + fabsf(bx.S) + bx.B * ge);
    }
    const bool fin = bs.usable && bx.usable && isfinite(est_hi) && isfinite(lb_hi) && isfinite(d_hi) && isfinite(g_add) && isfinite(g_err);
    return fin ? fmaxf(d_hi, lb_hi) : INFINITY;
}
// Whole-list bound: `ls` = the factor ranges over ALL blocks of a list, g_add in [ga_lo, ga_hi], g_err in [ge_lo, ge_hi]
// (0 <= ge_lo).  True iff every vector of the list has a finite lower bound >= T for every admissible (g_add, g_err).
__device__ __forceinline__ bool list_bound_reaches(const BlockSummary& ls, float ga_lo, float ga_hi, float ge_lo, float ge_hi,
                                                   const QueryConsts& qc, float T) {
    const float tA = fmaf(qc.delta, qc.amin, qc.sum_vl) + qc.k1x;
    const float tB = fmaf(qc.delta, qc.amax, qc.sum_vl) + qc.k1x;
    const float r0 = ls.fres_min * tA, r1 = ls.fres_min * tB, r2 = ls.fres_max * tA, r3 = ls.fres_max * tB;
    const float rmin = fminf(fminf(r0, r1), fminf(r2, r3)), rmax = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
    const float e0 = ls.ferr_min * ge_lo, e1 = ls.ferr_min * ge_hi, e2 = ls.ferr_max * ge_lo, e3 = ls.ferr_max * ge_hi;
    const float emin = fminf(fminf(e0, e1), fminf(e2, e3)), emax = fmaxf(fmaxf(e0, e1), fmaxf(e2, e3));
    float elo = ls.fadd_min + ga_lo;
    elo = elo + rmin;
    float ehi = ls.fadd_max + ga_hi;
    ehi = ehi + rmax;
    const float lbmin = elo - emax, lbmax = ehi - emin;
    const bool fin = ls.usable && isfinite(r0) && isfinite(r1) && isfinite(r2) && isfinite(r3) && isfinite(e0) && isfinite(e1) &&
                     isfinite(e2) && isfinite(e3) && isfinite(elo) && isfinite(ehi) && isfinite(lbmin) && isfinite(lbmax);
    return fin && lbmin >= T;
}

// ---- ex-code refine: sum_t code[16t+gl] * q[16t+gl] for one 16-lane group, AVX-512 lane order --------------
// (ip_packed_ex{2,6}_f32, src/simd.rs:1835-1915: one fused multiply-add per 16-dim step per lane, t ascending,
// then the _mm512_reduce_add_ps halving tree.)  Units are walked in a runtime loop (next unit prefetched),
// the CPU codes of a unit are decoded from 4 registers with compile-time shifts.  `sq` is the zero-padded
// rotated query (ex_qlen floats): padded code slots are 0 and 0*q + s == s exactly.
template <int EX>
__device__ __forceinline__ float ex_dot_units(const uint8_t* __restrict__ ex, const float* sq, uint32_t gl, uint32_t nunits) {
    constexpr int CPU = 128 / EX;
    constexpr uint32_t mask = (1u << EX) - 1u;
    const uint4* p = reinterpret_cast<const uint4*>(ex) + gl;
    float sacc = 0.0f;
    uint4 cur = p[0];
#pragma unroll 1
    for (uint32_t j = 0; j < nunits; ++j) {
        const uint4 nxt = p[(j + 1 < nunits ? j + 1 : j) * 16];
        const uint32_t w[5] = {cur.x, cur.y, cur.z, cur.w, 0u};
        const float* qj = sq + (size_t)j * CPU * 16 + gl;
#pragma unroll
        for (int k = 0; k < CPU; ++k) {
            const int bit = k * EX, idx = bit >> 5, sh = bit & 31;
            uint32_t code;
            if (sh + EX <= 32) code = (w[idx] >> sh) & mask;
            else code = ((w[idx] >> sh) | (w[idx + 1] << (32 - sh))) & mask;
            sacc = fmaf((float)code, qj[16 * k], sacc);
        }
        cur = nxt;
    }
    return sacc;
}
// Same arithmetic with all units of a vector resident in registers (nunits <= kExRegUnits), so that the units
// of the NEXT survivor can be in flight while this one is evaluated (heavy tiles: hundreds of survivors).
constexpr int kExRegUnits = 4;
__device__ __forceinline__ void ex_load_all(uint4 (&u)[kExRegUnits], const uint8_t* __restrict__ ex, uint32_t gl, uint32_t nunits) {
    const uint4* p = reinterpret_cast<const uint4*>(ex) + gl;
#pragma unroll
    for (int j = 0; j < kExRegUnits; ++j) u[j] = p[((uint32_t)j < nunits ? j : 0) * 16];
}
template <int EX>
__device__ __forceinline__ float ex_dot_all(const uint4 (&u)[kExRegUnits], const float* sq, uint32_t gl, uint32_t nunits) {
    constexpr int CPU = 128 / EX;
    constexpr uint32_t mask = (1u << EX) - 1u;
    float sacc = 0.0f;
#pragma unroll
    for (int j = 0; j < kExRegUnits; ++j) {
        if ((uint32_t)j < nunits) { // wave-uniform
            const uint32_t w[5] = {u[j].x, u[j].y, u[j].z, u[j].w, 0u};
            const float* qj = sq + j * CPU * 16 + gl;
#pragma unroll
            for (int k = 0; k < CPU; ++k) {
                const int bit = k * EX, idx = bit >> 5, sh = bit & 31;
                uint32_t code;
                if (sh + EX <= 32) code = (w[idx] >> sh) & mask;
                else code = ((w[idx] >> sh) | (w[idx + 1] << (32 - sh))) & mask;
                sacc = fmaf((float)code, qj[16 * k], sacc);
            }
        }
    }
    return sacc;
}
__device__ __forceinline__ float group16_reduce(float sacc) { // _mm512_reduce_add_ps halving tree
    sacc = sacc + __shfl_xor(sacc, 8, 16);
    sacc = sacc + __shfl_xor(sacc, 4, 16);
    sacc = sacc + __shfl_xor(sacc, 2, 16);
    sacc = sacc + __shfl_xor(sacc, 1, 16);
    return sacc;
}

__device__ __forceinline__ int32_t total_key(float x) { // f32::total_cmp ordering key
    int32_t i = __float_as_int(x);
    return i ^ (int32_t)(((uint32_t)(i >> 31)) >> 1);
}
__device__ __forceinline__ float key_to_float(int32_t k) {
    return __int_as_float(k ^ (int32_t)(((uint32_t)(k >> 31)) >> 1));
}
__device__ __forceinline__ bool finite_f(float x) { return (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u; }

// Exclusive prefix sum of one value per thread over a 256-thread workgroup (s_w: 4 words of LDS scratch, free
// again on return); `total` receives the sum of all values.
__device__ __forceinline__ uint32_t block_scan_excl256(uint32_t v, uint32_t* s_w, uint32_t tid, uint32_t& total) {
    static_assert(kThreads == 256, "four waves");
    const uint32_t lane = tid & 63u;
    // wave-inclusive scan on DPP: Hillis-Steele inside each row of 16 lanes (row_shr:1,2,4,8, missing lanes read
    // 0), then lane 15 of row 0/2 into rows 1/3 (row_bcast:15) and lane 31 into rows 2,3 (row_bcast:31)
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    const uint32_t incl = (uint32_t)x;
    if (lane == 63u) s_w[tid >> 6] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t x = s_w[j];
        off += (uint32_t)j < (tid >> 6) ? x : 0u;
        tot += x;
    }
    total = tot;
    __syncthreads();
    return off + incl - v;
}

// ---------------------------------------------------------------------------------------------
// k_prep: one workgroup per query.
// ---------------------------------------------------------------------------------------------
// Synchronisation of the GS threads that share one vector in LDS: a workgroup barrier for GS == 256; for one
// wave (GS == 64) LDS operations of a wave are serviced in issue order, so draining them (which also stops the
// compiler from moving LDS accesses across this point) is enough.
template <int GS>
__device__ __forceinline__ void group_sync() {
    if (GS == 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else __syncthreads();
}

template <int GS>
__device__ __forceinline__ void fht_lds(float* a, uint32_t n, uint32_t tid) {
    for (uint32_t h = 1, lg = 0; h < n; h <<= 1, ++lg) {
        for (uint32_t i = tid; i < n / 2; i += GS) {
            uint32_t j = ((i >> lg) << (lg + 1)) | (i & (h - 1));
            float x = a[j], y = a[j + h];
            a[j] = x + y;
            a[j + h] = x - y;
        }
        group_sync<GS>();
    }
}

// Rotator::rotate_into for one vector, by GS threads (one workgroup or one wave): the rotated vector ends up in x[0..D)
// (LDS); y[0..D) is scratch for the matrix rotator.  Butterflies/adds are the reference's, stage by stage
// (src/rotation.rs:248-401), so the result is bit-identical to the CPU path.
template <int GS>
__device__ __forceinline__ void rotate_into_lds(float* x, float* y, const float* __restrict__ qin, uint32_t dim, uint32_t D,
                                                int rotator, const uint8_t* __restrict__ rot_blob, uint32_t trunc,
                                                float fac, uint32_t tid) {
    if (rotator == 1) { // FhtKacRotator::rotate_into
        for (uint32_t i = tid; i < D; i += GS) x[i] = i < dim ? qin[i] : 0.0f;
        group_sync<GS>();
        const uint32_t fo = D / 8;
        if (trunc == D) {
            for (int r = 0; r < 4; ++r) {
                const uint8_t* f = rot_blob + r * fo;
                for (uint32_t i = tid; i < D; i += GS)
                    if ((f[i >> 3] >> (i & 7)) & 1) x[i] = -x[i];
                group_sync<GS>();
                fht_lds<GS>(x, D, tid);
                for (uint32_t i = tid; i < D; i += GS) x[i] = x[i] * fac;
                group_sync<GS>();
            }
        } else {
            const uint32_t start = D - trunc, half = D / 2;
            for (int r = 0; r < 4; ++r) {
                const uint8_t* f = rot_blob + r * fo;
                for (uint32_t i = tid; i < D; i += GS)
                    if ((f[i >> 3] >> (i & 7)) & 1) x[i] = -x[i];
                group_sync<GS>();
                float* part = (r & 1) ? x + start : x;
                fht_lds<GS>(part, trunc, tid);
                for (uint32_t i = tid; i < trunc; i += GS) part[i] = part[i] * fac;
                group_sync<GS>();
                for (uint32_t i = tid; i < half; i += GS) {
                    float a = x[i], b = x[i + half];
                    x[i] = a + b;
                    x[i + half] = a - b;
                }
                group_sync<GS>();
            }
            for (uint32_t i = tid; i < D; i += GS) x[i] = x[i] * 0.25f;
            group_sync<GS>();
        }
    } else if (rotator == 2) { // RBQ_ROTATOR_NONE: MSTG posting lists are quantised in the raw space
        for (uint32_t i = tid; i < D; i += GS) x[i] = i < dim ? qin[i] : 0.0f;
        group_sync<GS>();
    } else { // MatrixRotator::rotate_into: sequential unfused accumulate per output row
        for (uint32_t i = tid; i < D; i += GS) y[i] = i < dim ? qin[i] : 0.0f;
        group_sync<GS>();
        const float* M = reinterpret_cast<const float*>(rot_blob);
        for (uint32_t r = tid; r < D; r += GS) {
            const float* row = M + (size_t)r * D;
            float acc = 0.0f;
            for (uint32_t c = 0; c < D; ++c) {
                float p = y[c] * row[c];
                acc = acc + p;
            }
            x[r] = acc;
        }
        group_sync<GS>();
    }

}

// canonical single-pair reductions used for the second per-probe quantity
__device__ inline float canon_l2(const float* a, const float* __restrict__ b, uint32_t D) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t Dmain = D & ~7u;
    for (uint32_t i = 0; i < Dmain; i += 8)
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            float d = a[i + l] - b[i + l];
            float p = d * d;
            acc[l] = acc[l] + p;
        }
    float sum = 0.0f;
    if (Dmain) {
        sum = -0.0f;
#pragma unroll
        for (int l = 0; l < 8; ++l) sum = sum + acc[l];
    }
    for (uint32_t i = Dmain; i < D; ++i) {
        float d = a[i] - b[i];
        float p = d * d;
        sum = sum + p;
    }
    return sum;
}
__device__ inline float canon_dot(const float* a, const float* __restrict__ b, uint32_t D) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t Dmain = D & ~7u;
    for (uint32_t i = 0; i < Dmain; i += 8)
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            float p = a[i + l] * b[i + l];
            acc[l] = acc[l] + p;
        }
    float sum = 0.0f;
    if (Dmain) {
        sum = -0.0f;
#pragma unroll
        for (int l = 0; l < 8; ++l) sum = sum + acc[l];
    }
    for (uint32_t i = Dmain; i < D; ++i) {
        float p = a[i] * b[i];
        sum = sum + p;
    }
    return sum;
}

// ---------------------------------------------------------------------------------------------
// k_select: one workgroup per query.  64-bit key = (ordered score << 32) | cid, ascending;
// radix-select the nprobe-th key, collect, bitonic-sort; then per-probe constants and work list.
// dynamic LDS: sel[np2] u64 | qrot[D] f32 | part[kThreads] u32
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t make_key(float score, uint32_t cid, int metric) {
    int32_t k = total_key(score);
    if (metric == 1) k = ~k; // descending score
    return ((uint64_t)((uint32_t)k ^ 0x80000000u) << 32) | cid;
}

} // namespace rbq
