// query_kernels.hpp — the kernels of the query stages in front of the scan that are not templates on the GEMM shape:
// k_prep / k_prep_wave (rotate + constants + LUT), k_rank_scores and k_select (the all-pairs canonical ranking kept as
// the exact_rank option).  Included by k_query.hip only (non-template __global__ functions: one definition).
#pragma once
#include "kernels.hpp"

namespace rbq {

// Range of sum_i code_i * q_i over ALL ex codes (code_i in [0, 2^ex - 1]) from the sums of the positive and of the negative
// query elements, widened by 1e-3 of the largest possible magnitude: that covers the rounding of these two sums and of the
// scan kernel's own 16-lane FMA summation (each below D * 2^-24 <= 1.3e-4 relative for D <= 2048).
__device__ __forceinline__ void ex_dot_range(float sum_pos, float sum_neg, uint32_t ex_bits, float& lo, float& hi) {
    const float cmax = (float)((1u << ex_bits) - 1u);
    const float slack = cmax * (sum_pos - sum_neg) * 1e-3f;
    lo = cmax * sum_neg - slack;
    hi = cmax * sum_pos + slack;
}

__global__ __launch_bounds__(kThreads) void k_prep(const float* __restrict__ queries, uint32_t dim, uint32_t D,
                                                   uint32_t Dc, int rotator, const uint8_t* __restrict__ rot_blob,
                                                   uint32_t trunc, float fac, uint32_t ex_bits,
                                                   float* __restrict__ rot_out, uint8_t* __restrict__ lut_out,
                                                   QueryConsts* __restrict__ consts,
                                                   uint16_t* __restrict__ rot_hi, uint16_t* __restrict__ rot_lo) {
    extern __shared__ __align__(16) float sm[];
    float* x = sm;         // [D]
    float* y = sm + D;     // [D] (matrix rotator input)
    __shared__ int s_kmin, s_kmax;
    __shared__ unsigned int s_amin, s_amax;
    __shared__ float s_sum, s_n2, s_pos, s_neg;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const float* qin = queries + (size_t)q * dim;
#ifdef RBQ_PREP_STAMPS
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif

    rotate_into_lds<kThreads>(x, y, qin, dim, D, rotator, rot_blob, trunc, fac, tid);

    for (uint32_t i = tid; i < D; i += kThreads) {
        rot_out[(size_t)q * D + i] = x[i];
        if (rot_hi) { // split-bf16 image for k_rank_bf16_db
            uint16_t h, l;
            bf16_split(x[i], h, l);
            rot_hi[(size_t)q * D + i] = h;
            rot_lo[(size_t)q * D + i] = l;
        }
    }

#ifdef RBQ_PREP_STAMPS
    const unsigned long long pt1 = __builtin_amdgcn_s_memtime();
#endif
    // QueryPrecomputed::new — strictly sequential sums (Rust iter().sum() folds from -0.0)
    if (tid == 0) {
        float s = -0.0f;
        for (uint32_t i = 0; i < D; ++i) s = s + x[i];
        s_sum = s;
        s_kmin = 0x7fffffff;
        s_kmax = (int)0x80000000;
        s_amin = 0;
        s_amax = 0;
    }
    if (tid == 64) {
        float n2 = -0.0f;
        for (uint32_t i = 0; i < D; ++i) {
            float p = x[i] * x[i];
            n2 = n2 + p;
        }
        s_n2 = n2;
    }
    if (tid == 128) { // sums of the positive / negative elements: range of the ex-code dot product (ex_dot_range)
        float sp = 0.0f, sn = 0.0f;
        for (uint32_t i = 0; i < D; ++i) { sp += fmaxf(x[i], 0.0f); sn += fminf(x[i], 0.0f); }
        s_pos = sp; s_neg = sn;
    }
    __syncthreads();

#ifdef RBQ_PREP_STAMPS
    const unsigned long long pt2 = __builtin_amdgcn_s_memtime();
#endif
    // pack_lut_f32 + QueryLut::new.  D <= 2048 -> at most 2 codebooks per thread.
    const uint32_t ncb = D / 4;
    float l[2][16];
    int kmin = 0x7fffffff, kmax = (int)0x80000000;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        uint32_t c = tid + u * kThreads;
        if (c < ncb) {
            const float q0 = x[4 * c], q1 = x[4 * c + 1], q2 = x[4 * c + 2], q3 = x[4 * c + 3];
            // lut[j] = lut[j - lowbit(j)] + q[KPOS[j]],  KPOS = {3,3,2,3,1,3,2,3,0,3,2,3,1,3,2,3}
            l[u][0] = 0.0f;
            l[u][1] = l[u][0] + q3;
            l[u][2] = l[u][0] + q2;
            l[u][3] = l[u][2] + q3;
            l[u][4] = l[u][0] + q1;
            l[u][5] = l[u][4] + q3;
            l[u][6] = l[u][4] + q2;
            l[u][7] = l[u][6] + q3;
            l[u][8] = l[u][0] + q0;
            l[u][9] = l[u][8] + q3;
            l[u][10] = l[u][8] + q2;
            l[u][11] = l[u][10] + q3;
            l[u][12] = l[u][8] + q1;
            l[u][13] = l[u][12] + q3;
            l[u][14] = l[u][12] + q2;
            l[u][15] = l[u][14] + q3;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                int k = total_key(l[u][j]);
                kmin = k < kmin ? k : kmin;
                kmax = k > kmax ? k : kmax;
            }
        }
    }
    atomicMin(&s_kmin, kmin);
    atomicMax(&s_kmax, kmax);
    __syncthreads();
    const float vl = key_to_float(s_kmin), vr = key_to_float(s_kmax);
    const float delta = (vr - vl) / 255.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        uint32_t c = tid + u * kThreads;
        if (c < ncb) {
            uint32_t w[4] = {0, 0, 0, 0};
            uint32_t emin = 255, emax = 0;
            if (delta > 0.0f) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float v = roundf((l[u][j] - vl) / delta);
                    v = v >= 0.0f ? v : 0.0f; // also maps NaN -> 0 like `as u8`
                    v = v > 255.0f ? 255.0f : v;
                    const uint32_t e = (uint32_t)v;
                    emin = e < emin ? e : emin;
                    emax = e > emax ? e : emax;
                    w[j >> 2] |= e << (8 * (j & 3));
                }
            } else {
                emin = 0;
            }
            atomicAdd(&s_amin, emin);
            atomicAdd(&s_amax, emax);
            // device LUT order: adjacent codebooks swapped (position p holds codebook p^1) so that
            // nibble m of a little-endian code dword indexes table (8*dword + m) directly.
            uint4* dst = reinterpret_cast<uint4*>(lut_out + (size_t)q * Dc * 4 + (size_t)(c ^ 1u) * 16);
            *dst = make_uint4(w[0], w[1], w[2], w[3]);
        } else if (c < Dc / 4) { // code-layout padding (D not a multiple of 64): all-zero tables
            uint4* dst = reinterpret_cast<uint4*>(lut_out + (size_t)q * Dc * 4 + (size_t)(c ^ 1u) * 16);
            *dst = make_uint4(0, 0, 0, 0);
        }
    }
    __syncthreads();
    if (tid == 0) {
        QueryConsts qc;
        qc.amin = (float)s_amin;
        qc.amax = (float)s_amax;
        qc.delta = delta;
        qc.sum_vl = vl * (float)(D / 4);
        qc.qnorm = sqrtf(s_n2);
        qc.qnorm2 = s_n2; qc.q1norm = (s_pos - s_neg) * 1.001f;
        ex_dot_range(s_pos, s_neg, ex_bits, qc.exlo, qc.exhi);
#ifdef RBQ_PREP_STAMPS
        qc.exlo = (float)(pt1 - pt0); qc.exhi = (float)(pt2 - pt1); qc.q1norm = (float)(__builtin_amdgcn_s_memtime() - pt2); // (lazy selection is off in this build)
#endif
        qc.k1x = -0.5f * s_sum;
        const float cb = -((float)(1u << ex_bits) - 0.5f);
        qc.kbx = cb * s_sum;
        qc.scale = (float)(1u << ex_bits);
        consts[q] = qc;
    }
}

// In-register FHT of one wave over n = 64*EPL floats at `part` (LDS): lane holds elements lane*EPL .. +EPL-1, so
// the stages h < EPL are register butterflies and the stages h = EPL*m (m = 1..32) exchange with lane ^ m.  Same
// butterflies in the same stage order as fht_lds (out[j] = x[j] + x[j+h], out[j+h] = x[j] - x[j+h]), followed by
// the reference's rescale x * fac.
// value of lane ^ M: DPP inside a row of 16 lanes (quad_perm for 1 and 2; row_shl:4 / row_shr:4 on alternate quads for 4;
// row_ror:8 for 8), gfx950's v_permlane16_swap / v_permlane32_swap between rows and halves (semantics probed with
// tests/diag/micro/permlane_probe.hip) — no LDS round trip (the first form used ds_bpermute for every stage: 48 of them
// per transform, with one wave per SIMD every one of their latencies was exposed)
template <int M>
__device__ __forceinline__ float lane_xor(float v, uint32_t lane) {
    const int vi = __float_as_int(v);
    if (M == 1) return __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0xB1, 0xf, 0xf, false)); // quad_perm [1,0,3,2]
    if (M == 2) return __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0x4E, 0xf, 0xf, false)); // quad_perm [2,3,0,1]
    if (M == 4) {
        int t = __builtin_amdgcn_update_dpp(vi, vi, 0x104, 0xf, 0x5, false);                       // quads 0,2 <- lane + 4
        t = __builtin_amdgcn_update_dpp(t, vi, 0x114, 0xf, 0xa, false);                            // quads 1,3 <- lane - 4
        return __int_as_float(t);
    }
    if (M == 8) return __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0x128, 0xf, 0xf, false)); // row_ror:8
    if (M == 16) { // r[0] = rows {0,0,2,2} of v, r[1] = rows {1,1,3,3}
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)vi, (unsigned)vi, false, false);
        return __int_as_float((int)((lane & 16u) ? r[0] : r[1]));
    }
    // M == 32: r[0] = the lower half twice, r[1] = the upper half twice
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)vi, (unsigned)vi, false, false);
    return __int_as_float((int)((lane & 32u) ? r[0] : r[1]));
}

// stage h = EPL * M of the transform for this lane's EPL elements: lower partner x + y, upper partner x - y (= p + (-v))
template <int EPL, int M>
__device__ __forceinline__ void fht_exchange(float (&v)[EPL], uint32_t lane) {
    const bool upper = (lane & (uint32_t)M) != 0;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const float p = lane_xor<M>(v[k], lane);
        const float sv = upper ? -v[k] : v[k];
        v[k] = p + sv;
    }
}

// `flip` != null: the flip-sign bits of the elements of `part` (LSB-first, bit i = element i), applied as the elements are
// loaded (flip_sign of the round fused into the transform's own read: no separate pass over the vector)
template <int EPL>
__device__ __forceinline__ void fht_wave(float* part, uint32_t lane, float fac, const uint8_t* flip = nullptr) {
    float v[EPL];
    if (EPL >= 4) {
#pragma unroll
        for (int k = 0; k < EPL; k += 4) {
            const float4 t = *reinterpret_cast<const float4*>(part + lane * EPL + k);
            v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < EPL; ++k) v[k] = part[lane * EPL + k];
    }
    if (flip) {
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const uint32_t i = lane * EPL + k;
            if ((flip[i >> 3] >> (i & 7u)) & 1u) v[k] = -v[k];
        }
    }
#pragma unroll
    for (int h = 1; h < EPL; h <<= 1)
#pragma unroll
        for (int k = 0; k < EPL; ++k)
            if ((k & h) == 0) {
                const float a = v[k], b = v[k + h];
                v[k] = a + b;
                v[k + h] = a - b;
            }
    fht_exchange<EPL, 1>(v, lane);
    fht_exchange<EPL, 2>(v, lane);
    fht_exchange<EPL, 4>(v, lane);
    fht_exchange<EPL, 8>(v, lane);
    fht_exchange<EPL, 16>(v, lane);
    fht_exchange<EPL, 32>(v, lane);
    if (EPL >= 4) {
#pragma unroll
        for (int k = 0; k < EPL; k += 4)
            *reinterpret_cast<float4*>(part + lane * EPL + k) = make_float4(v[k] * fac, v[k + 1] * fac, v[k + 2] * fac, v[k + 3] * fac);
    } else {
#pragma unroll
        for (int k = 0; k < EPL; ++k) part[lane * EPL + k] = v[k] * fac;
    }
}

// FhtKacRotator::rotate_into by one wave; `flips` = the 4*D/8 flip-sign bytes (LDS copy).  Same operations on the same
// values as rotate_into_lds<64> (src/rotation.rs:350-401), with fht_wave in place of the LDS butterflies and the sign flips
// and the final scale FUSED into the passes next to them (a negation commutes with nothing else here: it is applied to
// the very value the separate pass would have negated): round 0's flips by the initial load, round r+1's flips by the
// Kac step of round r as it writes (or, for power-of-two dimensions, by the transform as it reads), the closing * 0.25
// by the last Kac step.  Per round the vector makes 4 passes through LDS as 16-byte accesses instead of 6 as scalars:
// the rotation was 17 of the 27 us of a single query's preparation.
// The initial load (with round 0's flips unless the dimension is a power of two) — a separate step so that the caller can issue
// it BEFORE the barrier behind the copy of the flip bits into LDS: `flips0` may be the global table.
__device__ __forceinline__ void fhtkac_initial_load(float* x, const float* __restrict__ qin, uint32_t dim, uint32_t D, uint32_t trunc,
                                                    const uint8_t* __restrict__ flips0, uint32_t lane) {
    // eight elements per lane are REQUESTED before the first is written to LDS: as a plain loop hipcc waited for every load in front
    // of its ds_write — D / 64 dependent global round trips (15 at D = 960) at the head of every query's preparation (round 5)
    constexpr int KU = 8;
    for (uint32_t i0 = lane; i0 < D; i0 += 64 * KU) {
        float v[KU];
        uint32_t f[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const uint32_t i = i0 + 64u * u;
            v[u] = i < dim ? qin[i] : 0.0f; // (dim <= D)
            f[u] = (trunc != D && i < D) ? (uint32_t)flips0[i >> 3] : 0u;
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const uint32_t i = i0 + 64u * u;
            if (i < D) x[i] = ((f[u] >> (i & 7u)) & 1u) ? -v[u] : v[u];
        }
    }
}
template <int EPL>
__device__ __forceinline__ void rotate_fhtkac_wave(float* x, uint32_t D, const uint8_t* flips, float fac, uint32_t lane) {
    constexpr uint32_t trunc = 64u * EPL;
    const uint32_t fo = D / 8, start = D - trunc, half = D / 2;
    group_sync<64>(); // (fhtkac_initial_load)
    if (trunc == D) { // 4 x (flip, transform of everything, scale)
        for (int r = 0; r < 4; ++r) {
            fht_wave<EPL>(x, lane, fac, flips + r * fo);
            group_sync<64>();
        }
        return;
    }
    for (int r = 0; r < 4; ++r) {
        fht_wave<EPL>((r & 1) ? x + start : x, lane, fac);
        group_sync<64>();
        // Kac step, four pairs (i, i + half) per lane and step; then the next round's flips, or the closing scale
        const uint8_t* fn = flips + (r + 1) * fo;
        for (uint32_t i = 4 * lane; i < half; i += 256) { // half % 32 == 0
            const float4 a = *reinterpret_cast<const float4*>(x + i), b = *reinterpret_cast<const float4*>(x + i + half);
            float s[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}, d[4] = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w};
            if (r < 3) {
                const uint32_t fs = (uint32_t)fn[i >> 3] >> (i & 7u), fd = (uint32_t)fn[(i + half) >> 3] >> ((i + half) & 7u); // (i, half: multiples of 4)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if ((fs >> k) & 1u) s[k] = -s[k];
                    if ((fd >> k) & 1u) d[k] = -d[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) { s[k] = s[k] * 0.25f; d[k] = d[k] * 0.25f; }
            }
            *reinterpret_cast<float4*>(x + i) = make_float4(s[0], s[1], s[2], s[3]);
            *reinterpret_cast<float4*>(x + i + half) = make_float4(d[0], d[1], d[2], d[3]);
        }
        group_sync<64>();
    }
}

// pack_lut_f32 entries of one codebook: lut[j] = lut[j - lowbit(j)] + q[KPOS[j]],  KPOS = {3,3,2,3,1,3,2,3,0,3,2,3,1,3,2,3}
__device__ __forceinline__ void lut_entries(const float* x4, float (&l)[16]) {
    const float q0 = x4[0], q1 = x4[1], q2 = x4[2], q3 = x4[3];
    l[0] = 0.0f;
    l[1] = l[0] + q3;
    l[2] = l[0] + q2;
    l[3] = l[2] + q3;
    l[4] = l[0] + q1;
    l[5] = l[4] + q3;
    l[6] = l[4] + q2;
    l[7] = l[6] + q3;
    l[8] = l[0] + q0;
    l[9] = l[8] + q3;
    l[10] = l[8] + q2;
    l[11] = l[10] + q3;
    l[12] = l[8] + q1;
    l[13] = l[12] + q3;
    l[14] = l[12] + q2;
    l[15] = l[14] + q3;
}

// k_prep for the FHT-Kac and identity rotators with ONE WAVE per query (4 queries per workgroup): no workgroup
// barrier anywhere — the butterflies synchronise through the wave's in-order LDS traffic, the min/max and the
// amin/amax sums are wave reductions, and the two strictly sequential sums (sum q, |q|^2: Rust iter().sum())
// run side by side on lanes 0 and 1 of each wave.  Same arithmetic as k_prep, operation for operation.
// dynamic LDS: 4 x 2 x D floats (vector + squares per wave) | 4*D/8 flip bytes
__global__ __launch_bounds__(kThreads) void k_prep_wave(const float* __restrict__ queries, uint32_t nq, uint32_t dim, uint32_t D,
                                                        uint32_t Dc, int rotator, const uint8_t* __restrict__ rot_blob,
                                                        uint32_t trunc, float fac, uint32_t ex_bits,
                                                        float* __restrict__ rot_out, uint8_t* __restrict__ lut_out,
                                                        QueryConsts* __restrict__ consts,
                                                        uint16_t* __restrict__ rot_hi, uint16_t* __restrict__ rot_lo) {
    extern __shared__ __align__(16) float sm[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t q = blockIdx.x * (kThreads / 64) + wave;
    uint8_t* flips = reinterpret_cast<uint8_t*>(sm + (size_t)2 * (kThreads / 64) * D); // 4*D/8 bytes (FHT-Kac)
#ifdef RBQ_PREP_STAMPS
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
    float* x = sm + (size_t)wave * 2 * D;
    float* x2 = x + D; // squares, for the |q|^2 chain
    const float* qin = queries + (size_t)q * dim;
    const bool wave_fht = rotator == 1 && trunc >= 64 && trunc <= 2048; // (else the generic LDS butterflies)
    if (rotator == 1) {
        for (uint32_t i = threadIdx.x; i < D / 8; i += kThreads) reinterpret_cast<uint32_t*>(flips)[i] = reinterpret_cast<const uint32_t*>(rot_blob)[i]; // (4 D / 8 flip bytes as dwords: one trip)
        // the query is requested together with the flip table (round 0's flips straight from the global table): one
        // round trip in front of the barrier instead of two around it
        if (q < nq && wave_fht) fhtkac_initial_load(x, qin, dim, D, trunc, rot_blob, lane);
        __syncthreads(); // the only workgroup barrier: before any wave can leave
    }
    if (q >= nq) return; // whole wave
    if (rotator == 1) {
        switch (trunc) {
            case 64: rotate_fhtkac_wave<1>(x, D, flips, fac, lane); break;
            case 128: rotate_fhtkac_wave<2>(x, D, flips, fac, lane); break;
            case 256: rotate_fhtkac_wave<4>(x, D, flips, fac, lane); break;
            case 512: rotate_fhtkac_wave<8>(x, D, flips, fac, lane); break;
            case 1024: rotate_fhtkac_wave<16>(x, D, flips, fac, lane); break;
            case 2048: rotate_fhtkac_wave<32>(x, D, flips, fac, lane); break;
            // dim < 64 (transform shorter than a wavefront) or >= 4096: the generic LDS butterflies
            default: rotate_into_lds<64>(x, nullptr, qin, dim, D, rotator, flips, trunc, fac, lane); break;
        }
    } else {
        rotate_into_lds<64>(x, nullptr, qin, dim, D, rotator, rot_blob, trunc, fac, lane);
    }

    float sp = 0.0f, sn = 0.0f; // sums of the positive / negative elements (ex_dot_range)
    for (uint32_t i = lane; i < D; i += 64) {
        const float v = x[i];
        x2[i] = v * v;
        sp += fmaxf(v, 0.0f);
        sn += fminf(v, 0.0f);
        rot_out[(size_t)q * D + i] = v;
        if (rot_hi) { // split-bf16 image for k_rank_bf16_db
            uint16_t h, l;
            bf16_split(v, h, l);
            rot_hi[(size_t)q * D + i] = h;
            rot_lo[(size_t)q * D + i] = l;
        }
    }
    group_sync<64>();
#ifdef RBQ_PREP_STAMPS
    const unsigned long long pt1 = __builtin_amdgcn_s_memtime();
#endif
    // QueryPrecomputed::new — strictly sequential sums (Rust iter().sum() folds from -0.0): lane 0 adds the
    // elements, lane 1 their squares
    float acc = -0.0f;
    if (lane < 2) { // 16 elements per step: four 16-byte LDS reads in flight, then the adds in element order
        const float* src = lane ? x2 : x;
        for (uint32_t i = 0; i < D; i += 16) { // D % 16 == 0
            float4 v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v4[u] = *reinterpret_cast<const float4*>(src + i + 4 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = acc + v4[u].x;
                acc = acc + v4[u].y;
                acc = acc + v4[u].z;
                acc = acc + v4[u].w;
            }
        }
    }
    const float s_sum = __shfl(acc, 0, 64), s_n2 = __shfl(acc, 1, 64);

#ifdef RBQ_PREP_STAMPS
    const unsigned long long pt2 = __builtin_amdgcn_s_memtime();
#endif
    // pack_lut_f32 + QueryLut::new: pass 1 finds the value range, pass 2 quantises
    const uint32_t ncb = D / 4;
    int kmin = 0x7fffffff, kmax = (int)0x80000000;
    for (uint32_t c = lane; c < ncb; c += 64) {
        float l[16];
        lut_entries(x + 4 * c, l);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = total_key(l[j]);
            kmin = k < kmin ? k : kmin;
            kmax = k > kmax ? k : kmax;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int a = __shfl_xor(kmin, d, 64), b = __shfl_xor(kmax, d, 64);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    const float vl = key_to_float(kmin), vr = key_to_float(kmax);
    const float delta = (vr - vl) / 255.0f;
    uint32_t amin = 0, amax = 0;
    for (uint32_t c = lane; c < Dc / 4; c += 64) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (c < ncb) {
            float l[16];
            lut_entries(x + 4 * c, l);
            uint32_t emin = 255, emax = 0;
            if (delta > 0.0f) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float v = roundf((l[j] - vl) / delta);
                    v = v >= 0.0f ? v : 0.0f; // also maps NaN -> 0 like `as u8`
                    v = v > 255.0f ? 255.0f : v;
                    const uint32_t e = (uint32_t)v;
                    emin = e < emin ? e : emin;
                    emax = e > emax ? e : emax;
                    w[j >> 2] |= e << (8 * (j & 3));
                }
            } else {
                emin = 0;
            }
            amin += emin;
            amax += emax;
        }
        // device LUT order: adjacent codebooks swapped (position p holds codebook p^1) so that nibble m of a
        // little-endian code dword indexes table (8*dword + m) directly; padding codebooks are all-zero tables
        *reinterpret_cast<uint4*>(lut_out + (size_t)q * Dc * 4 + (size_t)(c ^ 1u) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        amin += __shfl_xor(amin, d, 64);
        amax += __shfl_xor(amax, d, 64);
        sp += __shfl_xor(sp, d, 64);
        sn += __shfl_xor(sn, d, 64);
    }
    if (lane == 0) {
        QueryConsts qc;
        qc.amin = (float)amin;
        qc.amax = (float)amax;
        qc.delta = delta;
        qc.sum_vl = vl * (float)(D / 4);
        qc.qnorm = sqrtf(s_n2);
        qc.qnorm2 = s_n2; qc.q1norm = (sp - sn) * 1.001f;
        ex_dot_range(sp, sn, ex_bits, qc.exlo, qc.exhi);
#ifdef RBQ_PREP_STAMPS
        qc.exlo = (float)(pt1 - pt0); qc.exhi = (float)(pt2 - pt1); qc.q1norm = (float)(__builtin_amdgcn_s_memtime() - pt2); // (lazy selection is off in this build)
#endif
        qc.k1x = -0.5f * s_sum;
        const float cb = -((float)(1u << ex_bits) - 0.5f);
        qc.kbx = cb * s_sum;
        qc.scale = (float)(1u << ex_bits);
        consts[q] = qc;
    }
}

// ---------------------------------------------------------------------------------------------
// k_rank_scores: scores[q][c] = l2_distance_sqr(rot[q], cent[c]) or dot(...), in the reference's
// AVX2 lane order: 8 strided accumulators, unfused mul/add, lanes summed 0..7, scalar tail.
// Tile 32 queries x 32 centroids per workgroup, 2x2 pairs per thread, 32-dim LDS chunks.
// ---------------------------------------------------------------------------------------------
template <int METRIC>
__global__ __launch_bounds__(kThreads) void k_rank_scores(const float* __restrict__ rot, const float* __restrict__ cent,
                                                          uint32_t nq, uint32_t nlist, uint32_t D,
                                                          float* __restrict__ scores) {
    __shared__ __align__(16) float Qs[32][36];
    __shared__ __align__(16) float Cs[32][36];
    const uint32_t tid = threadIdx.x, tq = tid >> 4, tc = tid & 15;
    const uint32_t q0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const uint32_t Dmain = D & ~7u;
    float acc[2][2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int l = 0; l < 8; ++l) acc[a][b][l] = 0.0f;

    const uint32_t lrow = tid >> 3, lcol = (tid & 7) * 4;
    for (uint32_t k0 = 0; k0 < Dmain; k0 += 32) {
        float4 qv = make_float4(0, 0, 0, 0), cv = make_float4(0, 0, 0, 0);
        const uint32_t k = k0 + lcol;
        if (q0 + lrow < nq) {
            const float* p = rot + (size_t)(q0 + lrow) * D + k;
            if (k + 4 <= Dmain && (D & 3) == 0) qv = *reinterpret_cast<const float4*>(p);
            else {
                if (k < Dmain) qv.x = p[0];
                if (k + 1 < Dmain) qv.y = p[1];
                if (k + 2 < Dmain) qv.z = p[2];
                if (k + 3 < Dmain) qv.w = p[3];
            }
        }
        if (c0 + lrow < nlist) {
            const float* p = cent + (size_t)(c0 + lrow) * D + k;
            if (k + 4 <= Dmain && (D & 3) == 0) cv = *reinterpret_cast<const float4*>(p);
            else {
                if (k < Dmain) cv.x = p[0];
                if (k + 1 < Dmain) cv.y = p[1];
                if (k + 2 < Dmain) cv.z = p[2];
                if (k + 3 < Dmain) cv.w = p[3];
            }
        }
        *reinterpret_cast<float4*>(&Qs[lrow][lcol]) = qv;
        *reinterpret_cast<float4*>(&Cs[lrow][lcol]) = cv;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float qa[2][8], cb[2][8];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float4 v0 = *reinterpret_cast<const float4*>(&Qs[2 * tq + a][8 * t]);
                float4 v1 = *reinterpret_cast<const float4*>(&Qs[2 * tq + a][8 * t + 4]);
                qa[a][0] = v0.x; qa[a][1] = v0.y; qa[a][2] = v0.z; qa[a][3] = v0.w;
                qa[a][4] = v1.x; qa[a][5] = v1.y; qa[a][6] = v1.z; qa[a][7] = v1.w;
                float4 w0 = *reinterpret_cast<const float4*>(&Cs[2 * tc + a][8 * t]);
                float4 w1 = *reinterpret_cast<const float4*>(&Cs[2 * tc + a][8 * t + 4]);
                cb[a][0] = w0.x; cb[a][1] = w0.y; cb[a][2] = w0.z; cb[a][3] = w0.w;
                cb[a][4] = w1.x; cb[a][5] = w1.y; cb[a][6] = w1.z; cb[a][7] = w1.w;
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int l = 0; l < 8; ++l) {
                        float p;
                        if (METRIC == 0) {
                            float d = qa[a][l] - cb[b][l];
                            p = d * d;
                        } else {
                            p = qa[a][l] * cb[b][l];
                        }
                        acc[a][b][l] = acc[a][b][l] + p;
                    }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t qi = q0 + 2 * tq + a, ci = c0 + 2 * tc + b;
            if (qi < nq && ci < nlist) {
                float sum = 0.0f;
                if (Dmain > 0) {
                    sum = -0.0f;
#pragma unroll
                    for (int l = 0; l < 8; ++l) sum = sum + acc[a][b][l];
                }
                for (uint32_t i = Dmain; i < D; ++i) { // scalar tail
                    float x = rot[(size_t)qi * D + i], y = cent[(size_t)ci * D + i], p;
                    if (METRIC == 0) {
                        float d = x - y;
                        p = d * d;
                    } else {
                        p = x * y;
                    }
                    sum = sum + p;
                }
                scores[(size_t)qi * nlist + ci] = sum;
            }
        }
}


__global__ __launch_bounds__(kThreads) void k_select(const float* __restrict__ scores, uint32_t nlist, uint32_t nprobe,
                                                     uint32_t np2, int metric, const float* __restrict__ rot,
                                                     const float* __restrict__ cent, uint32_t D,
                                                     const uint32_t* __restrict__ list_gb0,
                                                     const uint32_t* __restrict__ list_n,
                                                     ProbeInfo* __restrict__ probe, StreamItem* __restrict__ wl,
                                                     uint64_t wl_stride, uint32_t* __restrict__ nstream,
                                                     unsigned long long* __restrict__ nvec_probed,
                                                     unsigned long long* __restrict__ prof_total,
                                                     const QueryConsts* __restrict__ consts,
                                                     const BlockSummary* __restrict__ bsum, uint64_t* gsel) {
    // gsel != null: the key window [nq][np2] lives in global memory (nprobe beyond what the LDS of one compute unit holds:
    // the reference only clamps nprobe to the number of lists, src/ivf.rs:1791 — slow here, but served)
    extern __shared__ __align__(16) unsigned char smraw[];
    uint64_t* sel = gsel ? gsel + (size_t)blockIdx.x * np2 : reinterpret_cast<uint64_t*>(smraw);
    float* qrot = reinterpret_cast<float*>(smraw + (gsel ? (size_t)0 : (size_t)np2 * 8));
    uint32_t* part = reinterpret_cast<uint32_t*>(qrot + D);
    __shared__ uint32_t hist[256];
    __shared__ uint64_t s_prefix, s_mask;
    __shared__ uint32_t s_k, s_cnt;
    __shared__ unsigned long long s_nvec;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const float* sc = scores + (size_t)q * nlist;

    for (uint32_t i = tid; i < D; i += kThreads) qrot[i] = rot[(size_t)q * D + i];
    if (tid == 0) { s_prefix = 0; s_mask = 0; s_k = nprobe - 1; s_cnt = 0; s_nvec = 0; }
    __syncthreads();

    if (nprobe < nlist) {
        for (int pass = 7; pass >= 0; --pass) {
            const int shift = pass * 8;
            hist[tid] = 0;
            __syncthreads();
            const uint64_t prefix = s_prefix, mask = s_mask;
            for (uint32_t i = tid; i < nlist; i += kThreads) {
                uint64_t key = make_key(sc[i], i, metric);
                if ((key & mask) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t k = s_k, cum = 0, b = 0;
                for (; b < 256; ++b) {
                    uint32_t h = hist[b];
                    if (k < cum + h) break;
                    cum += h;
                }
                s_k = k - cum;
                s_prefix = prefix | ((uint64_t)b << shift);
                s_mask = mask | (0xffull << shift);
            }
            __syncthreads();
        }
    } else if (tid == 0) {
        s_prefix = ~0ull;
    }
    __syncthreads();
    const uint64_t kstar = s_prefix;
    for (uint32_t i = tid; i < np2; i += kThreads) sel[i] = ~0ull;
    if (gsel) __threadfence_block();
    __syncthreads();
    for (uint32_t i = tid; i < nlist; i += kThreads) {
        uint64_t key = make_key(sc[i], i, metric);
        if (key <= kstar) {
            uint32_t pos = atomicAdd(&s_cnt, 1u);
            if (pos < np2) sel[pos] = key;
        }
    }
    if (gsel) __threadfence_block(); // (global window: the workgroup's own writes are read back by other lanes)
    __syncthreads();
    // bitonic sort ascending
    for (uint32_t k = 2; k <= np2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < np2; i += kThreads) {
                uint32_t ixj = i ^ j;
                if (ixj > i) {
                    uint64_t a = sel[i], b = sel[ixj];
                    bool up = (i & k) == 0;
                    if ((a > b) == up) { sel[i] = b; sel[ixj] = a; }
                }
            }
            if (gsel) __threadfence_block();
            __syncthreads();
        }

    // per-probe constants (src/ivf.rs:1850-1857) + block counts
    const uint32_t per = (nprobe + kThreads - 1) / kThreads;
    const uint32_t r0 = tid * per, r1 = (r0 + per < nprobe) ? r0 + per : nprobe;
    uint32_t local = 0;
    unsigned long long local_vec = 0;
    for (uint32_t r = r0; r < r1; ++r) {
        uint32_t cid = (uint32_t)(sel[r] & 0xffffffffu);
        float s = sc[cid], dist, dot;
        const float* c = cent + (size_t)cid * D;
        // L2: score IS the centroid distance; the dot product only feeds the non-finite lower-bound
        // fallback of the IP metric (src/ivf.rs:2031-2042), so it is not computed here.
        if (metric == 0) { dist = s; dot = 0.0f; }
        else { dot = s; dist = canon_l2(qrot, c, D); }
        ProbeInfo pi;
        pi.g_add = metric == 0 ? dist : -dot;
        pi.g_err = sqrtf(dist);
        pi.dotqc = dot;
        pi.cid = cid;
        probe[(size_t)q * nprobe + r] = pi;
        local += (list_n[cid] + 31u) >> 5;
        local_vec += list_n[cid];
    }
    part[tid] = local;
    if (local_vec) atomicAdd(&s_nvec, local_vec);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < kThreads; ++i) { uint32_t v = part[i]; part[i] = run; run += v; }
        nstream[q] = run;
        nvec_probed[q] = s_nvec; // sum of n_c over the probed lists: the scan's algorithmic work
        if (prof_total) atomicAdd(prof_total + prof_stripe(q), s_nvec);
    }
    __syncthreads();
    uint64_t pos = (uint64_t)q * wl_stride + part[tid];
    const QueryConsts qc = consts[q];
    for (uint32_t r = r0; r < r1; ++r) {
        uint32_t cid = (uint32_t)(sel[r] & 0xffffffffu);
        uint32_t n = list_n[cid], gb = list_gb0[cid], nb = (n + 31u) >> 5;
        const ProbeInfo pi = probe[(size_t)q * nprobe + r]; // written by this thread above
        for (uint32_t b = 0; b < nb; ++b) {
            uint32_t nv = (b + 1 == nb) ? n - b * 32u : 32u;
            StreamItem wi;
            wi.gblock = gb + b;
            wi.rank_nvalid = (r << 6) | nv;
            wi.lbmin = block_lbmin(bsum[gb + b], pi.g_add, pi.g_err, qc);
            wi.pad = 0;
            wl[pos++] = wi;
        }
    }
}


} // namespace rbq
