// latency.hpp — the latency-first front of a SMALL call (a handful of queries): k_lat_front does, in ONE launch, what k_prep_wave
// and the ranking GEMM do in two — DynamicRotator::rotate (src/rotation.rs:350-401), QueryPrecomputed::new (src/ivf.rs:862-878),
// pack_lut_f32 + QueryLut::new (src/simd.rs:818-840, src/ivf.rs:798-845) and the centroid scores of src/ivf.rs:1782-1835 in the
// reference's own summation order (src/math.rs:154-245: eight strided accumulators, unfused, lanes summed 0..7, scalar tail).
//
// Why.  One query through the batch kernels is four dependent launches of single workgroups whose waves execute a few thousand
// instructions each, one after the other: the preparation alone is 13 us of ONE wave (rotation, two 960-step serial sums, 2 x 240
// IEEE divisions), the MFMA GEMM + radix shortlist for 4096 x 960 MACs is pure overhead (tools/lat_trace.py).  Here the
// query's work is spread over the chip instead:
//   (ceil(n_lists / 32) + 1) x nq workgroups of 256 threads (lat_front_grid).  EVERY workgroup rotates its query itself (wave 0, ~2 us: cheaper
//   than a launch boundary), then
//   * workgroups 0 .. G-1 score 32 lists each, eight lanes per list straight from global memory (64 loads in flight per lane), and
//     write the EXACT canonical score into the row the probe selection reads — the selection's rigorous |A - canonical| <= eps
//     contract holds with room to spare, nothing downstream changes;
//   * workgroup G writes the rotated query, the query constants and the u8 LUT: the two strictly sequential sums on two lanes of
//     wave 0 WHILE waves 1-3 find the LUT's value range; the quantisation by all four waves.
// Same arithmetic as k_prep / k_prep_wave, operation for operation (the stage-level parity test compares bits).
#pragma once
#include "query_kernels.hpp"

namespace rbq {

constexpr uint32_t kLatLists = 32; // lists scored per workgroup (eight lanes each)

struct LatFrontParams {
    const float* queries; // [nq][dim]
    uint32_t nq, dim, D, Dc;
    int rotator;          // 1 = FhtKac, 2 = none (the matrix rotator is O(D^2) per query: k_prep serves it)
    const uint8_t* rot_blob;
    uint32_t trunc;
    float fac;
    uint32_t ex_bits;
    float* rot;           // [nq][D]
    uint8_t* lut;         // [nq][4Dc]
    QueryConsts* consts;  // [nq]
    const float* cent;    // [nlist][D] rotated centroids
    uint32_t nlist;
    int metric;
    float* scores;        // [nq][nlist] exact canonical scores (L2: squared distance; IP: dot)
    uint16_t *rot_hi, *rot_lo; // null, or the split-bf16 image of the rotated query (the ranking GEMM's operand: `scorers` = 0)
    float* zero_scores;   // null, or the score rows [nq][nlist] to clear (a split-K ranking GEMM adds its parts to them)
    uint32_t scorers;     // G = ceil(nlist / 32) scoring workgroups per query, or 0: preparation only, the ranking GEMM follows (medium
                          // batches: a workgroup per query finishes a query's preparation in ~2/3 of the time one wave of k_prep_wave needs)
};
// Workgroup -> (query, role) mapping.  Consecutive workgroup ids go to consecutive XCDs (8 of them, each with its own L2), so with
// several queries the id is cut as  id = (slot * 8 + xcd),  slot = (role block * nq + query):  the nq workgroups that score the SAME 32
// lists for the nq queries land on one XCD, back to back — the centroid rows come from HBM once and from that L2 nq - 1 times.
__host__ __device__ inline uint32_t lat_front_grid(uint32_t scorers, uint32_t nq) {
    return scorers ? ((scorers + 1u + 7u) / 8u) * 8u * nq : nq;
}

// dynamic LDS: x[D] f32 (the rotated query) | x2[D] f32 (its squares: the |q|^2 chain) | 4*D/8 flip bytes
__global__ __launch_bounds__(kThreads) void k_lat_front(const LatFrontParams P) {
    extern __shared__ __align__(16) float sm[];
    __shared__ int s_kmin, s_kmax;
    __shared__ unsigned int s_amin, s_amax;
    __shared__ float s_sum, s_n2, s_sp, s_sn;
    const uint32_t D = P.D, Dc = P.Dc, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t G = P.scorers;
    uint32_t q = blockIdx.x, g = 0;
    if (G) {
        const uint32_t slot = blockIdx.x >> 3;
        q = slot % P.nq;
        g = (slot / P.nq) * 8u + (blockIdx.x & 7u);
        if (g > G) return; // (padding of the role count to a multiple of 8)
    }
    float* x = sm;
    float* x2 = sm + D;
    uint8_t* flips = reinterpret_cast<uint8_t*>(sm + (size_t)2 * D);
    const float* qin = P.queries + (size_t)q * P.dim;
    const uint32_t trunc = P.trunc;
#ifdef RBQ_PREP_STAMPS
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
    const bool wave_fht = P.rotator == 1 && trunc >= 64 && trunc <= 2048;
    if (P.rotator == 1) {
        for (uint32_t i = tid; i < D / 8; i += kThreads) reinterpret_cast<uint32_t*>(flips)[i] = reinterpret_cast<const uint32_t*>(P.rot_blob)[i]; // (4 D / 8 flip bytes as dwords: one trip)
        if (wave == 0 && wave_fht) fhtkac_initial_load(x, qin, P.dim, D, trunc, P.rot_blob, lane);
        __syncthreads();
    }
    if (wave == 0) { // k_prep_wave's rotation, by one wave (no barrier inside)
        if (P.rotator == 1) {
            switch (trunc) {
                case 64: rotate_fhtkac_wave<1>(x, D, flips, P.fac, lane); break;
                case 128: rotate_fhtkac_wave<2>(x, D, flips, P.fac, lane); break;
                case 256: rotate_fhtkac_wave<4>(x, D, flips, P.fac, lane); break;
                case 512: rotate_fhtkac_wave<8>(x, D, flips, P.fac, lane); break;
                case 1024: rotate_fhtkac_wave<16>(x, D, flips, P.fac, lane); break;
                case 2048: rotate_fhtkac_wave<32>(x, D, flips, P.fac, lane); break;
                default: rotate_into_lds<64>(x, nullptr, qin, P.dim, D, P.rotator, flips, trunc, P.fac, lane); break;
            }
        } else {
            rotate_into_lds<64>(x, nullptr, qin, P.dim, D, P.rotator, P.rot_blob, trunc, P.fac, lane);
        }
    }
    __syncthreads(); // x[0..D) = the rotated query

    if (g < G) {
        // ---- scorer: list g*32 + (tid >> 3), accumulator lane tid & 7 (canon_score_octets' loop, the score goes to the row)
        const uint32_t a = tid & 7u, cid = g * kLatLists + (tid >> 3), Dmain = D & ~7u;
        if (cid >= P.nlist) return; // uniform per eight-lane group
        const float* c = P.cent + (size_t)cid * D;
        float acc = 0.0f;
        constexpr int KB = 64; // loads in flight per lane: D <= 512 in one round trip, 960 in two
        for (uint32_t base = a; base < Dmain; base += 8 * KB) {
            float cv[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) cv[k] = base + 8u * k < Dmain ? c[base + 8u * k] : 0.0f;
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                if (base + 8u * k < Dmain) {
                    const float qv = x[base + 8u * k];
                    if (P.metric == 0) {
                        const float d = qv - cv[k];
                        const float p = d * d;
                        acc = acc + p;
                    } else {
                        const float p = qv * cv[k];
                        acc = acc + p;
                    }
                }
            }
        }
        float sum = 0.0f;
        if (Dmain) {
            sum = -0.0f;
#pragma unroll
            for (int l = 0; l < 8; ++l) sum = sum + __shfl(acc, l, 8);
        }
        for (uint32_t i = Dmain; i < D; ++i) { // scalar tail
            const float qv = x[i], cv = c[i];
            if (P.metric == 0) {
                const float d = qv - cv;
                const float p = d * d;
                sum = sum + p;
            } else {
                const float p = qv * cv;
                sum = sum + p;
            }
        }
        if (a == 0) P.scores[(size_t)q * P.nlist + cid] = sum;
        return;
    }

    // ---- workgroup G: rotated query, constants, LUT of query q
#ifdef RBQ_PREP_STAMPS
    const unsigned long long pt1 = __builtin_amdgcn_s_memtime();
#endif
    for (uint32_t i = tid; i < D; i += kThreads) {
        const float v = x[i];
        x2[i] = v * v;
        P.rot[(size_t)q * D + i] = v;
        if (P.rot_hi) { // split-bf16 image for k_rank_bf16_db
            uint16_t h, l;
            bf16_split(v, h, l);
            P.rot_hi[(size_t)q * D + i] = h;
            P.rot_lo[(size_t)q * D + i] = l;
        }
    }
    if (P.zero_scores) { // (the row a split-K ranking GEMM adds its parts to)
        float* zr = P.zero_scores + (size_t)q * P.nlist;
        if ((P.nlist & 3u) == 0u) for (uint32_t i = tid; i < P.nlist / 4; i += kThreads) reinterpret_cast<float4*>(zr)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        else for (uint32_t i = tid; i < P.nlist; i += kThreads) zr[i] = 0.0f;
    }
    if (tid == 0) { s_kmin = 0x7fffffff; s_kmax = (int)0x80000000; s_amin = 0; s_amax = 0; }
    __syncthreads();
    const uint32_t ncb = D / 4;
    if (wave == 0) {
        // sums of the positive / negative elements (ex_dot_range), k_prep_wave's order: strided per lane, xor tree
        float sp = 0.0f, sn = 0.0f;
        for (uint32_t i = lane; i < D; i += 64) {
            const float v = x[i];
            sp += fmaxf(v, 0.0f);
            sn += fminf(v, 0.0f);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            sp += __shfl_xor(sp, d, 64);
            sn += __shfl_xor(sn, d, 64);
        }
        // QueryPrecomputed::new — strictly sequential sums (Rust iter().sum() folds from -0.0): lane 0 adds the elements,
        // lane 1 their squares; 16 elements per step, the adds in element order
        float acc = -0.0f;
        if (lane < 2) {
            // the next 16 elements are requested before the current 16 are added: the chain of 960 dependent adds never waits for LDS
            const float* src = lane ? x2 : x;
            float4 cur[4], nxt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) cur[u] = *reinterpret_cast<const float4*>(src + 4 * u);
            for (uint32_t i = 0; i < D; i += 16) { // D % 16 == 0
                const uint32_t in = i + 16 < D ? i + 16 : i;
#pragma unroll
                for (int u = 0; u < 4; ++u) nxt[u] = *reinterpret_cast<const float4*>(src + in + 4 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc = acc + cur[u].x;
                    acc = acc + cur[u].y;
                    acc = acc + cur[u].z;
                    acc = acc + cur[u].w;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
            }
        }
        if (lane == 0) { s_sum = acc; s_sp = sp; s_sn = sn; }
        if (lane == 1) s_n2 = acc;
    } else {
        // pack_lut_f32 + QueryLut::new, pass 1 (value range) by waves 1-3 while wave 0 runs the serial sums
        int kmin = 0x7fffffff, kmax = (int)0x80000000;
        for (uint32_t c = tid - 64u; c < ncb; c += kThreads - 64u) {
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
        if (lane == 0) { atomicMin(&s_kmin, kmin); atomicMax(&s_kmax, kmax); }
    }
    __syncthreads();
#ifdef RBQ_PREP_STAMPS
    const unsigned long long pt2 = __builtin_amdgcn_s_memtime();
#endif
    const float vl = key_to_float(s_kmin), vr = key_to_float(s_kmax);
    const float delta = (vr - vl) / 255.0f;
    uint32_t amin = 0, amax = 0;
    for (uint32_t c = tid; c < Dc / 4; c += kThreads) { // pass 2: quantise (all four waves)
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
        // device LUT order: adjacent codebooks swapped (position p holds codebook p^1); padding codebooks are all-zero tables
        *reinterpret_cast<uint4*>(P.lut + (size_t)q * Dc * 4 + (size_t)(c ^ 1u) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        amin += __shfl_xor(amin, d, 64);
        amax += __shfl_xor(amax, d, 64);
    }
    if (lane == 0) { atomicAdd(&s_amin, amin); atomicAdd(&s_amax, amax); }
    __syncthreads();
    if (tid == 0) {
        QueryConsts qc;
        qc.amin = (float)s_amin;
        qc.amax = (float)s_amax;
        qc.delta = delta;
        qc.sum_vl = vl * (float)(D / 4);
        qc.qnorm = sqrtf(s_n2);
        qc.qnorm2 = s_n2; qc.q1norm = (s_sp - s_sn) * 1.001f;
        ex_dot_range(s_sp, s_sn, P.ex_bits, qc.exlo, qc.exhi);
#ifdef RBQ_PREP_STAMPS
        qc.exlo = (float)(pt1 - pt0); qc.exhi = (float)(pt2 - pt1); qc.q1norm = (float)(__builtin_amdgcn_s_memtime() - pt2); // (lazy selection is off in this build)
#endif
        qc.k1x = -0.5f * s_sum;
        const float cb = -((float)(1u << P.ex_bits) - 0.5f);
        qc.kbx = cb * s_sum;
        qc.scale = (float)(1u << P.ex_bits);
        P.consts[q] = qc;
    }
}

} // namespace rbq
