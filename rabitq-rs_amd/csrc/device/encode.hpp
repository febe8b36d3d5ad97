// encode.hpp — GPU-side encoder: the quantisation loop of IvfRabitqIndex::train_with_clusters
// (src/ivf.rs:1025-1215) for the "faster" configuration (constant rescale factor t_const), writing the device
// layout of rbq_api.hip directly.
//
//   k_rotate_rows   Rotator::rotate_into for a set of rows (k_prep's rotation code)
//   k_encode        quantize_with_centroid (src/quantizer.rs:140-262, :264-308, :429-535): one THREAD per vector —
//                   every reduction of the reference is a sequential chain (iter().sum()) or the 8-accumulator
//                   AVX2 dot (src/math.rs:154-245), so a lane walks its vector in order while 64-dim tiles of 64
//                   vectors are staged through LDS with coalesced loads.  Two passes over the rotated rows: the
//                   residual norm has to be known before the ex codes can be formed.
//   k_pack_ex       raw ex codes -> lane-major 128-bit units (the layout relayout_ex produces on the host)
//   k_block_summary factor ranges of every block (BlockSummary)
// The arithmetic is the CPU builder's (rbq_build.cpp), expression for expression; -ffp-contract=off.
#pragma once
#include "kernels.hpp"
#include "scan.hpp"

namespace rbq {

// rows[r] = rotate(src[map ? map[r] : r]) for r < nrows; rows of padding slots (map[r] == kNoSrc) are skipped
__global__ __launch_bounds__(kThreads) void k_rotate_rows(const float* __restrict__ src, const uint32_t* __restrict__ map,
                                                          uint32_t dim, uint32_t D, int rotator,
                                                          const uint8_t* __restrict__ rot_blob, uint32_t trunc, float fac,
                                                          float* __restrict__ rows) {
    extern __shared__ __align__(16) float sm_rot[];
    float* x = sm_rot;
    float* y = sm_rot + D;
    const uint32_t r = blockIdx.x, tid = threadIdx.x;
    const uint32_t s = map ? map[r] : r;
    if (s == kNoSrc) return; // uniform
    rotate_into_lds<kThreads>(x, y, src + (size_t)s * dim, dim, D, rotator, rot_blob, trunc, fac, tid);
    for (uint32_t i = tid; i < D; i += kThreads) rows[(size_t)r * D + i] = x[i];
}

constexpr int kEncTile = 64;             // dims per LDS tile
constexpr int kEncLd = kEncTile + 1;     // row stride (floats): lane i reads word i*65 + k -> conflict-free

// dot8 state: acc[l] += a*b for element index i with l = i % 8 (src/math.rs AVX2 lane order); D % 8 == 0
struct Dot8 {
    float a[8];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int l = 0; l < 8; ++l) a[l] = 0.0f;
    }
    __device__ __forceinline__ float finish(bool any) const {
        float sum = 0.0f;
        if (any) {
            sum = -0.0f;
#pragma unroll
            for (int l = 0; l < 8; ++l) sum = sum + a[l];
        }
        return sum;
    }
};

// SCATTER = false: block-ordered chunk (rbq_index_build_device) — row r IS chunk-local slot r, the 32 lanes of a
//   half-wave share one block and therefore one centroid row.
// SCATTER = true: streamed build (rbq_build_stream_push) — rows are the chunk's vectors sorted by slot, row r goes
//   to GLOBAL slot row_slot[r]; every row stages its own centroid tile.  Same arithmetic, expression for expression.
template <bool SCATTER>
__global__ __launch_bounds__(kEncThreads) void k_encode(EncodeParams P) {
    __shared__ float s_x[kEncThreads * kEncLd];
    __shared__ float s_c[SCATTER ? kEncThreads * kEncLd : 2 * kEncTile];
    const uint32_t tid = threadIdx.x, half = tid >> 5;
    const uint32_t slot = blockIdx.x * kEncThreads + tid;         // row (block-ordered mode: chunk-local slot)
    const uint32_t nblk = (P.nslots + 31u) / 32u;
    const uint32_t src = (slot < P.nslots) ? P.slot_src[slot] : kNoSrc;
    const bool valid = src != kNoSrc;
    const uint32_t oslot = SCATTER ? (valid ? P.row_slot[slot] : 0u) : slot; // output slot in the views of P
    const uint32_t blk = SCATTER ? (oslot >> 5) : blockIdx.x * 2 + half;
    const uint32_t v = SCATTER ? (oslot & 31u) : (tid & 31u);
    const bool has_blk = SCATTER ? valid : blk < nblk;
    const uint32_t D = P.D, Dc = P.Dc, ex_bits = P.ex_bits;
    const size_t stride = (size_t)Dc * 4 + 384;
    uint8_t* rec = P.blocks + (size_t)blk * stride;
    const float F32_EPS = 1.1920929e-07f, K_CONST_EPSILON = 1.9f;

    // coalesced staging of one tile: 64 rows x 256 B, 16 lanes per row; the centroid rows of the two blocks
    auto stage = [&](uint32_t t0) {
        __syncthreads();
        const uint32_t w = D - t0 < (uint32_t)kEncTile ? D - t0 : (uint32_t)kEncTile; // tile width (multiple of 16)
#pragma unroll 4
        for (uint32_t it = 0; it < 16; ++it) {
            const uint32_t r = it * 4 + (tid >> 4), k4 = (tid & 15u) * 4;
            const uint32_t gs = blockIdx.x * kEncThreads + r;
            float4 x = make_float4(0, 0, 0, 0);
            if (gs < P.nslots && k4 < w && P.slot_src[gs] != kNoSrc) x = *reinterpret_cast<const float4*>(P.rows + (size_t)gs * D + t0 + k4);
            float* d = s_x + r * kEncLd + k4;
            d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
        }
        if (SCATTER) {
#pragma unroll 4
            for (uint32_t it = 0; it < 16; ++it) {
                const uint32_t r = it * 4 + (tid >> 4), k4 = (tid & 15u) * 4;
                const uint32_t gs = blockIdx.x * kEncThreads + r;
                float4 c = make_float4(0, 0, 0, 0);
                if (gs < P.nslots && k4 < w && P.slot_src[gs] != kNoSrc)
                    c = *reinterpret_cast<const float4*>(P.centroids + (size_t)P.block_list[P.row_slot[gs] >> 5] * D + t0 + k4);
                float* d = s_c + r * kEncLd + k4;
                d[0] = c.x; d[1] = c.y; d[2] = c.z; d[3] = c.w;
            }
        } else
        for (uint32_t i = tid; i < 2u * kEncTile; i += kEncThreads) {
            const uint32_t h = i / kEncTile, k = i % kEncTile, b = blockIdx.x * 2 + h;
            s_c[i] = (b < nblk && k < w) ? P.centroids[(size_t)P.block_list[b] * D + t0 + k] : 0.0f;
        }
        __syncthreads();
        return w;
    };

    // ---- pass A: residual, sign bits, |r|^2 chain, the five dots of compute_one_bit_factors
    float n2 = -0.0f;
    Dot8 d_l2, d_xu, d_rx, d_cx, d_rc;
    d_l2.init(); d_xu.init(); d_rx.init(); d_cx.init(); d_rc.init();
    uint32_t gran[4] = {0, 0, 0, 0}; // 16 code bytes (128 dims) being assembled
    for (uint32_t t0 = 0; t0 < D; t0 += kEncTile) {
        const uint32_t w = stage(t0);
        const float* xr = s_x + tid * kEncLd;
        const float* cr = SCATTER ? s_c + tid * kEncLd : s_c + half * kEncTile;
        for (uint32_t k0 = 0; k0 < w; k0 += 8) {
            uint32_t byte = 0;
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const float c = cr[k0 + l];
                const float r = xr[k0 + l] - c;
                const bool bit = r >= 0.0f;
                byte |= (bit ? 1u : 0u) << (7 - l);
                const float na = fabsf(r);
                { const float p = na * na; n2 = n2 + p; }
                const float xb = (bit ? 1.0f : 0.0f) - 0.5f;
                { const float p = r * r; d_l2.a[l] = d_l2.a[l] + p; }
                { const float p = xb * xb; d_xu.a[l] = d_xu.a[l] + p; }
                { const float p = r * xb; d_rx.a[l] = d_rx.a[l] + p; }
                { const float p = c * xb; d_cx.a[l] = d_cx.a[l] + p; }
                { const float p = r * c; d_rc.a[l] = d_rc.a[l] + p; }
            }
            const uint32_t col = (t0 + k0) >> 3; // byte column of the vector's packed sign code
            gran[(col & 15u) >> 2] |= byte << (8 * (col & 3u));
            if ((col & 15u) == 15u || t0 + k0 + 8 == D) { // granule complete (or the code ends)
                const uint32_t g = col >> 4, G16 = Dc >> 7;
                if (valid) {
                    if (g < G16) *reinterpret_cast<uint4*>(rec + (size_t)g * 512 + v * 16) = make_uint4(gran[0], gran[1], gran[2], gran[3]);
                    else *reinterpret_cast<uint2*>(rec + (size_t)G16 * 512 + v * 8) = make_uint2(gran[0], gran[1]);
                }
                gran[0] = gran[1] = gran[2] = gran[3] = 0;
            }
        }
    }
    const bool any8 = D >= 8;
    const float l2_sqr = d_l2.finish(any8), xu_norm_sqr = d_xu.finish(any8), ip_resi_xucb = d_rx.finish(any8);
    const float ip_cent_xucb = d_cx.finish(any8), dot_res_cent = d_rc.finish(any8);
    const float l2_norm = sqrtf(l2_sqr);
    const float norm = sqrtf(n2);

    // ---- pass B: ex codes (t_const), ipnorm chain in f64, the two dots of compute_extended_factors
    float ipnorm_inv = 1.0f;
    float f_add_ex = 0.0f, f_rescale_ex = 0.0f;
    if (ex_bits > 0) { // uniform
        const bool coded = norm > F32_EPS;
        const int32_t max_val = (1 << ex_bits) - 1;
        const double t = (double)P.t_const;
        const float cb = -((float)(1u << ex_bits) - 0.5f);
        double ipnorm = 0.0;
        Dot8 d_ipr, d_ipc;
        d_ipr.init(); d_ipc.init();
        for (uint32_t t0 = 0; t0 < D; t0 += kEncTile) {
            const uint32_t w = stage(t0);
            const float* xr = s_x + tid * kEncLd;
            const float* cr = SCATTER ? s_c + tid * kEncLd : s_c + half * kEncTile;
            for (uint32_t k0 = 0; k0 < w; k0 += 16) {
                uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
                for (int l = 0; l < 16; ++l) {
                    const float c = cr[k0 + l];
                    const float r = xr[k0 + l] - c;
                    const bool bit = r >= 0.0f;
                    uint32_t code = 0;
                    if (coded) {
                        const float na = fabsf(r) / norm;
                        int32_t cur = (int32_t)(t * (double)na + 1e-5);
                        if (cur > max_val) cur = max_val;
                        ipnorm += ((double)cur + 0.5) * (double)na;
                        code = (uint32_t)cur;
                        if (r < 0.0f) code = (~code) & (uint32_t)max_val;
                    }
                    const float xu = (float)(uint16_t)(code + ((bit ? 1u : 0u) << ex_bits)) + cb;
                    { const float p = r * xu; d_ipr.a[l & 7] = d_ipr.a[l & 7] + p; }
                    { const float p = c * xu; d_ipc.a[l & 7] = d_ipc.a[l & 7] + p; }
                    pk[l >> 2] |= code << (8 * (l & 3));
                }
                if (valid) *reinterpret_cast<uint4*>(P.raw_ex + (size_t)slot * D + t0 + k0) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        }
        if (coded) {
            ipnorm_inv = (isfinite(ipnorm) && ipnorm > 0.0) ? (float)(1.0 / ipnorm) : 1.0f;
            if (!isfinite(ipnorm_inv)) ipnorm_inv = 1.0f;
        }
        const float ip_r = d_ipr.finish(any8), ip_c = d_ipc.finish(any8);
        const float safe = fabsf(ip_r) <= F32_EPS ? INFINITY : ip_r;
        if (P.metric == 0) {
            f_add_ex = l2_sqr + 2.0f * l2_sqr * ip_c / safe;
            f_rescale_ex = -2.0f * l2_norm * ipnorm_inv;
        } else {
            f_add_ex = 1.0f - dot_res_cent + l2_sqr * ip_c / safe;
            f_rescale_ex = -l2_norm * ipnorm_inv;
        }
    }

    // ---- compute_one_bit_factors
    float f_add, f_rescale, f_error;
    {
        float denom = ip_resi_xucb;
        if (fabsf(denom) <= F32_EPS) denom = INFINITY;
        float tmp_error = 0.0f;
        if (D > 1) {
            const float ratio = ((l2_sqr * xu_norm_sqr) / (denom * denom)) - 1.0f;
            if (isfinite(ratio) && ratio > 0.0f)
                tmp_error = l2_norm * K_CONST_EPSILON * sqrtf(fmaxf(ratio / (float)(D - 1), 0.0f));
        }
        if (P.metric == 0) {
            f_add = l2_sqr + 2.0f * l2_sqr * ip_cent_xucb / denom;
            f_rescale = -2.0f * l2_sqr / denom;
            f_error = 2.0f * tmp_error;
        } else {
            f_add = 1.0f - dot_res_cent + l2_sqr * ip_cent_xucb / denom;
            f_rescale = -l2_sqr / denom;
            f_error = tmp_error;
        }
    }
    if (has_blk) {
        float* fac = reinterpret_cast<float*>(rec + (size_t)Dc * 4);
        fac[v] = valid ? f_add : 0.0f;
        fac[32 + v] = valid ? f_rescale : 0.0f;
        fac[64 + v] = valid ? f_error : 0.0f;
        const uint32_t s = SCATTER ? oslot : blk * 32 + v;
        P.ids[s] = valid ? P.src_base + src : ~0ull;
        if (ex_bits) {
            P.f_add_ex[s] = valid ? f_add_ex : 0.0f;
            P.f_rescale_ex[s] = valid ? f_rescale_ex : 0.0f;
        }
    }
}

// raw ex codes [slot][D] u8 -> [slot][unit][lane][16 B]; 16 lanes per vector, 16 vectors per workgroup
// (row_slot != null: row `slot` of the raw codes goes to the global slot row_slot[slot] — streamed build)
__global__ __launch_bounds__(256) void k_pack_ex(const uint8_t* __restrict__ raw, const uint32_t* __restrict__ slot_src,
                                                 const uint32_t* __restrict__ row_slot,
                                                 uint32_t nslots, uint32_t D, uint32_t ex_bits, uint8_t* __restrict__ ex) {
    const uint32_t slot = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15u;
    if (slot >= nslots) return;
    const uint32_t w4 = ex_w4(D, ex_bits), cpu = ex_cpu(ex_bits);
    const size_t exd = (size_t)w4 * 256;
    const bool valid = slot_src[slot] != kNoSrc;
    if (row_slot && !valid) return;
    uint4* dst = reinterpret_cast<uint4*>(ex + (size_t)(row_slot ? row_slot[slot] : slot) * exd) + l;
    const uint8_t* src = raw + (size_t)slot * D + l;
    uint32_t t = 0;
    for (uint32_t unit = 0; unit < w4; ++unit) {
        uint32_t u[5] = {0, 0, 0, 0, 0};
        for (uint32_t k = 0; k < cpu && t < D / 16; ++k, ++t) {
            const uint32_t code = valid ? src[16 * t] : 0u;
            const uint32_t bit = k * ex_bits, idx = bit >> 5, sh = bit & 31u;
            u[idx] |= code << sh;
            if (sh + ex_bits > 32) u[idx + 1] |= code >> (32 - sh);
        }
        dst[unit * 16] = make_uint4(u[0], u[1], u[2], u[3]);
    }
}

// BlockSummary of every block: one 32-lane half-wave per block
__global__ __launch_bounds__(256) void k_block_summary(const uint8_t* __restrict__ blocks, const uint32_t* __restrict__ block_nvalid,
                                                       uint32_t nblocks, uint32_t Dc, BlockSummary* __restrict__ bsum) {
    const uint32_t b = blockIdx.x * 8 + (threadIdx.x >> 5), v = threadIdx.x & 31u;
    if (b >= nblocks) return;
    const float* fac = reinterpret_cast<const float*>(blocks + (size_t)b * ((size_t)Dc * 4 + 384) + (size_t)Dc * 4);
    const bool real = v < block_nvalid[b];
    const float a = fac[v], r = fac[32 + v], e = fac[64 + v];
    float amin = real ? a : INFINITY, amax = real ? a : -INFINITY;
    float rmin = real ? r : INFINITY, rmax = real ? r : -INFINITY;
    float emin = real ? e : INFINITY, emax = real ? e : -INFINITY;
    uint32_t bad = real && (!finite_f(a) || !finite_f(r) || !finite_f(e)) ? 1u : 0u;
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) {
        amin = fminf(amin, __shfl_xor(amin, d, 32)); amax = fmaxf(amax, __shfl_xor(amax, d, 32));
        rmin = fminf(rmin, __shfl_xor(rmin, d, 32)); rmax = fmaxf(rmax, __shfl_xor(rmax, d, 32));
        emin = fminf(emin, __shfl_xor(emin, d, 32)); emax = fmaxf(emax, __shfl_xor(emax, d, 32));
        bad |= __shfl_xor(bad, d, 32);
    }
    if (v == 0) {
        BlockSummary bs;
        bs.usable = bad ? 0u : 1u; bs.pad = 0;
        if (bad) { amin = amax = rmin = rmax = emin = emax = 0.0f; }
        bs.fadd_min = amin; bs.fadd_max = amax; bs.fres_min = rmin; bs.fres_max = rmax; bs.ferr_min = emin; bs.ferr_max = emax;
        bsum[b] = bs;
    }
}

// One workgroup per list (lazy probe selection, rank_mfma.hpp): BlockSummaryEx of every block and the factor ranges of
// the WHOLE list (a BlockSummary over its blocks' summaries; an empty list gets an unusable one).
//
// BlockSummaryEx regroups the estimator around the list's centroid c.  With u' the centred total code of a vector
// (u'_i = (bit_i << ex) + ex_i - (2^ex - 0.5), src/quantizer.rs) and ubar_i = bit_i - 0.5, for any rotated query q
//   refined distance = f_add_ex + g_add + f_rescale_ex * <q, u'>    (src/ivf.rs:2086-2099; scale*ip + ex_dot + kbx = <q, u'>)
//                    = [f_add_ex + f_rescale_ex * <c, u'>] + g_add + f_rescale_ex * <q - c, u'>
//                   <=  S  + g_add + B * |q - c|,      S = f_add_ex + f_rescale_ex <c, u'>,  B = |f_rescale_ex| |u'|
//   1-bit estimate  <=  S1 + g_add + B1 * |q - c|,     S1 = f_add + f_rescale <c, ubar>,     B1 = |f_rescale| sqrt(D) / 2
// by Cauchy-Schwarz — against the ALL-codes range |q|_1 * 63.5 of block_lbmin this is 3-10x tighter, which is what makes
// the select-time bound of the k-th distance (T_ub) useful.  <c, u'>, <c, ubar> and |u'|^2 are accumulated in f64 from
// the device-layout codes; S, B, S1, B1 are rounded up.  |q - c| is the probed list's g_error (src/ivf.rs:1856).
// 16 lanes per vector, lane l owns dims 16t + l (the layout of the ex codes, types.hpp).
__global__ __launch_bounds__(256) void k_list_summaries(const uint8_t* __restrict__ blocks, const uint8_t* __restrict__ ex,
                                                        const float* __restrict__ fadd_ex, const float* __restrict__ fres_ex,
                                                        const float* __restrict__ cent, const BlockSummary* __restrict__ bsum,
                                                        const uint32_t* __restrict__ list_gb0, const uint32_t* __restrict__ list_n,
                                                        uint32_t D, uint32_t Dc, uint32_t ex_bits,
                                                        BlockSummaryEx* __restrict__ bsumx, BlockSummary* __restrict__ lsum) {
    extern __shared__ __align__(16) float s_cent[]; // [D]
    __shared__ float sS[32], sB[32], sS1[32], sB1[32], sFa[32], sFr[32];
    __shared__ uint32_t sBad[32];
    const uint32_t c = blockIdx.x, tid = threadIdx.x, grp = tid >> 4, l = tid & 15u;
    const uint32_t n = list_n[c], gb0 = list_gb0[c], nb = (n + 31u) >> 5;
    for (uint32_t i = tid; i < D; i += 256) s_cent[i] = cent[(size_t)c * D + i];
    __syncthreads();
    const size_t stride = (size_t)Dc * 4 + 384, exd = ex_bytes_dev(D, ex_bits);
    const uint32_t G16 = Dc >> 7, cpu = ex_cpu(ex_bits), nt = D / 16;
    const double centre = (double)(1u << ex_bits) - 0.5;
    for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t nv = (b + 1 == nb) ? n - b * 32u : 32u;
        const uint8_t* blk = blocks + (size_t)(gb0 + b) * stride;
        const float* fac = reinterpret_cast<const float*>(blk + (size_t)Dc * 4);
#pragma unroll 1
        for (uint32_t half = 0; half < 2; ++half) {
            const uint32_t v = grp + 16u * half;
            if (v < nv) { // uniform per 16-lane group
                const size_t slot = (size_t)(gb0 + b) * 32 + v;
                const uint4* exu = reinterpret_cast<const uint4*>(ex + slot * exd) + l;
                double dcu = 0.0, dcb = 0.0; // <c, u'>, <c, ubar>
                uint32_t u2x4 = 0;          // sum (2 u'_i)^2: |u'|^2 * 4, exact in 32 bits (D <= 2048, |2u'| <= 127)
                uint4 unit = make_uint4(0, 0, 0, 0);
                for (uint32_t t = 0; t < nt; ++t) {
                    const uint32_t i = 16u * t + l, col = i >> 3, g = col >> 4, cc = col & 15u;
                    uint32_t word;
                    if (g < G16) word = reinterpret_cast<const uint32_t*>(blk + (size_t)g * 512 + v * 16)[cc >> 2];
                    else word = reinterpret_cast<const uint32_t*>(blk + (size_t)G16 * 512 + v * 8)[cc >> 2];
                    const uint32_t bit = (word >> (8u * (cc & 3u) + 7u - (i & 7u))) & 1u;
                    uint32_t code = 0;
                    if (ex_bits) {
                        const uint32_t k = t % cpu;
                        if (k == 0) unit = exu[(t / cpu) * 16];
                        const uint32_t pos = k * ex_bits, idx = pos >> 5, sh = pos & 31u;
                        const uint32_t w0 = idx == 0 ? unit.x : idx == 1 ? unit.y : idx == 2 ? unit.z : unit.w;
                        const uint32_t w1 = idx == 0 ? unit.y : idx == 1 ? unit.z : idx == 2 ? unit.w : 0u;
                        uint32_t raw = w0 >> sh;
                        if (sh + ex_bits > 32) raw |= w1 << (32 - sh);
                        code = raw & ((1u << ex_bits) - 1u);
                    }
                    const double ci = (double)s_cent[i];
                    const double up = (double)((bit << ex_bits) + code) - centre;
                    dcu += ci * up;
                    dcb += ci * ((double)bit - 0.5);
                    const int tw = 2 * (int)((bit << ex_bits) + code) - (int)((2u << ex_bits) - 1u);
                    u2x4 += (uint32_t)(tw * tw);
                }
#pragma unroll
                for (int d = 8; d > 0; d >>= 1) {
                    dcu += __shfl_xor(dcu, d, 16);
                    dcb += __shfl_xor(dcb, d, 16);
                    u2x4 += __shfl_xor(u2x4, d, 16);
                }
                if (l == 0) {
                    const float fa = fac[v], fr = fac[32 + v];
                    const float fax = ex_bits ? fadd_ex[slot] : 0.0f, frx = ex_bits ? fres_ex[slot] : 0.0f;
                    const double S = (double)fax + (double)frx * dcu, B = fabs((double)frx) * sqrt((double)u2x4 * 0.25);
                    const double S1 = (double)fa + (double)fr * dcb, B1 = fabs((double)fr) * sqrt((double)D) * 0.5;
                    // rounded up (the f64 values themselves carry ~1e-13 relative error: one more ulp of f32 covers it)
                    auto up32 = [](double x) { float f = (float)x; if ((double)f < x) f = nextafterf(f, INFINITY); return nextafterf(f, INFINITY); };
                    sS[v] = up32(S); sB[v] = up32(B); sS1[v] = up32(S1); sB1[v] = up32(B1);
                    sFa[v] = fabsf(fax); sFr[v] = fabsf(frx);
                    sBad[v] = (finite_f(fa) && finite_f(fr) && finite_f(fax) && finite_f(frx) && isfinite(S) && isfinite(B) && isfinite(S1)) ? 0u : 1u;
                }
            } else if (l == 0) {
                sS[v] = -INFINITY; sB[v] = 0.0f; sS1[v] = -INFINITY; sB1[v] = 0.0f; sFa[v] = 0.0f; sFr[v] = 0.0f; sBad[v] = 0u;
            }
        }
        __syncthreads();
        if (tid < 32) {
            float S = sS[tid], B = sB[tid], S1 = sS1[tid], B1 = sB1[tid], Fa = sFa[tid], Fr = sFr[tid];
            uint32_t bad = sBad[tid];
#pragma unroll
            for (int d = 16; d > 0; d >>= 1) {
                S = fmaxf(S, __shfl_xor(S, d, 32)); B = fmaxf(B, __shfl_xor(B, d, 32));
                S1 = fmaxf(S1, __shfl_xor(S1, d, 32)); B1 = fmaxf(B1, __shfl_xor(B1, d, 32));
                Fa = fmaxf(Fa, __shfl_xor(Fa, d, 32)); Fr = fmaxf(Fr, __shfl_xor(Fr, d, 32));
                bad |= __shfl_xor(bad, d, 32);
            }
            if (tid == 0) {
                BlockSummaryEx bx;
                bx.S = S; bx.B = B; bx.S1 = S1; bx.B1 = B1; bx.fadd_ex_abs = Fa; bx.fres_ex_abs = Fr; bx.usable = bad ? 0u : 1u; bx.pad = 0;
                bsumx[gb0 + b] = bx;
            }
        }
        __syncthreads();
    }
    if (tid < 64) {
        const uint32_t lane = tid;
        float amin = INFINITY, amax = -INFINITY, rmin = INFINITY, rmax = -INFINITY, emin = INFINITY, emax = -INFINITY;
        uint32_t bad = nb == 0 ? 1u : 0u;
        for (uint32_t b = lane; b < nb; b += 64) {
            const BlockSummary bs = bsum[gb0 + b];
            bad |= bs.usable ? 0u : 1u;
            amin = fminf(amin, bs.fadd_min); amax = fmaxf(amax, bs.fadd_max);
            rmin = fminf(rmin, bs.fres_min); rmax = fmaxf(rmax, bs.fres_max);
            emin = fminf(emin, bs.ferr_min); emax = fmaxf(emax, bs.ferr_max);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            amin = fminf(amin, __shfl_xor(amin, d, 64)); amax = fmaxf(amax, __shfl_xor(amax, d, 64));
            rmin = fminf(rmin, __shfl_xor(rmin, d, 64)); rmax = fmaxf(rmax, __shfl_xor(rmax, d, 64));
            emin = fminf(emin, __shfl_xor(emin, d, 64)); emax = fmaxf(emax, __shfl_xor(emax, d, 64));
            bad |= __shfl_xor(bad, d, 64);
        }
        if (lane == 0) {
            BlockSummary ls;
            ls.usable = bad ? 0u : 1u; ls.pad = 0;
            if (bad) { amin = amax = rmin = rmax = emin = emax = 0.0f; }
            ls.fadd_min = amin; ls.fadd_max = amax; ls.fres_min = rmin; ls.fres_max = rmax; ls.ferr_min = emin; ls.ferr_max = emax;
            lsum[c] = ls;
        }
    }
}

// histogram of the assignment; sets *err when an assignment is out of range
__global__ void k_count_assign(const uint32_t* __restrict__ assign, uint64_t n, uint32_t nlist, uint32_t* __restrict__ counts,
                               uint32_t* __restrict__ err) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = assign[i];
        if (c >= nlist) { *err = 1u; continue; }
        atomicAdd(&counts[c], 1u);
    }
}
__global__ void k_iota(uint32_t* __restrict__ x, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) x[i] = (uint32_t)i;
}
// sorted position -> slot: slot_src[gb0[c]*32 + (pos - vstart[c])] = source index
__global__ void k_scatter_slots(const uint32_t* __restrict__ sorted_list, const uint32_t* __restrict__ sorted_src, uint64_t n,
                                const uint32_t* __restrict__ list_gb0, const uint64_t* __restrict__ vstart,
                                uint32_t* __restrict__ slot_src) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = sorted_list[i];
        slot_src[(uint64_t)list_gb0[c] * 32 + (i - vstart[c])] = sorted_src[i];
    }
}

} // namespace rbq
