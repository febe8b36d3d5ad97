// relayout.hpp — reference layout -> device layout on the GPU (rbq_index_create / rbq_index_load_rbq1), plus the
// small index-build helpers that used to run on one host thread.
//
// The reference keeps, per list (ClusterData, src/ivf.rs:205-242): FastScan-interleaved batch records
// (pack_codes, src/simd.rs:864-904), one packed ex code per vector (src/simd.rs:2478-2541, :2601-2695), ids and the
// ex factors in vector order.  The host uploads those bytes unchanged, a chunk of lists at a time; these kernels
// undo the pshufb interleave into lane-major code granules, re-pack the ex codes lane-major, and spread the
// per-vector arrays over the 32-padded slots.  Byte-for-byte the arrays the host loops (round 1) produced.
#pragma once
#include "kernels.hpp"

namespace rbq {

// inverse of pack_codes: KPERM0[j] = (j>>1) + 8*(j&1)  =>  j = 2*(u&7) + (u>>3)
__device__ __forceinline__ uint32_t fastscan_byte_dev(const uint8_t* __restrict__ packed, uint32_t col, uint32_t v) {
    const uint32_t u = v & 15u, j = 2u * (u & 7u) + (u >> 3);
    const uint32_t a = packed[col * 32 + j], b = packed[col * 32 + 16 + j];
    const uint32_t hi = v < 16 ? (a & 15u) : (a >> 4);
    const uint32_t lo = v < 16 ? (b & 15u) : (b >> 4);
    return (hi << 4) | lo;
}

// one half-wave per block, lane = vector: reference record [D*4 codes | 96 f32] -> device block [Dc*4 | 96 f32]
__global__ __launch_bounds__(256) void k_relayout_blocks(const uint8_t* __restrict__ recs, uint32_t nb, uint32_t D, uint32_t Dc,
                                                         uint8_t* __restrict__ blocks) {
    const uint32_t b = blockIdx.x * 8 + (threadIdx.x >> 5), v = threadIdx.x & 31u;
    if (b >= nb) return;
    const uint8_t* rec = recs + (size_t)b * ((size_t)D * 4 + 384);
    uint8_t* dst = blocks + (size_t)b * ((size_t)Dc * 4 + 384);
    const uint32_t ncol = D / 8, G16 = Dc >> 7;
    for (uint32_t g = 0; g < G16; ++g) {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t c = 0; c < 16; ++c) {
            const uint32_t col = g * 16 + c;
            if (col < ncol) w[c >> 2] |= fastscan_byte_dev(rec, col, v) << (8 * (c & 3));
        }
        *reinterpret_cast<uint4*>(dst + (size_t)g * 512 + v * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (Dc & 64u) {
        uint32_t w[2] = {0, 0};
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) {
            const uint32_t col = G16 * 16 + c;
            if (col < ncol) w[c >> 2] |= fastscan_byte_dev(rec, col, v) << (8 * (c & 3));
        }
        *reinterpret_cast<uint2*>(dst + (size_t)G16 * 512 + v * 8) = make_uint2(w[0], w[1]);
    }
    // factor rows: 384 bytes, byte-addressed on the source side (an RBQ1 stream has no alignment)
    const uint8_t* fs = rec + (size_t)D * 4;
    uint32_t* fd = reinterpret_cast<uint32_t*>(dst + (size_t)Dc * 4);
#pragma unroll
    for (uint32_t r = 0; r < 3; ++r) {
        const uint8_t* p = fs + (r * 32 + v) * 4;
        fd[r * 32 + v] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    }
}

// 16 lanes per vector: reference packed ex code (dense, `exb` bytes per vector, vector order) -> lane-major units
// of the slot.  Lane l owns dims 16t+l.  2-bit: code = (w >> (8*(l&3) + 2*(l>>2))) & 3 with w the LE word of step t;
// 6-bit: low4 from byte (l&7) of the u64 (high nibble for l >= 8), top2 from byte 8+(l&3), bits 2*(l>>2).
__global__ __launch_bounds__(256) void k_relayout_ex(const uint8_t* __restrict__ exsrc, const uint64_t* __restrict__ block_dense0,
                                                     const uint32_t* __restrict__ block_nv, uint32_t nb, uint32_t D,
                                                     uint32_t ex_bits, uint8_t* __restrict__ ex) {
    const uint32_t slot = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15u;
    const uint32_t b = slot >> 5, v = slot & 31u;
    if (b >= nb) return;
    const uint32_t w4 = ex_w4(D, ex_bits), cpu = ex_cpu(ex_bits);
    const size_t exd = (size_t)w4 * 256, exb = (size_t)D * ex_bits / 8;
    uint4* dst = reinterpret_cast<uint4*>(ex + (size_t)slot * exd) + l;
    const bool valid = v < block_nv[b];
    const uint8_t* src = exsrc + (block_dense0[b] + v) * exb;
    uint32_t t = 0;
    for (uint32_t unit = 0; unit < w4; ++unit) {
        uint32_t u[5] = {0, 0, 0, 0, 0};
        for (uint32_t k = 0; k < cpu && t < D / 16; ++k, ++t) {
            uint32_t code = 0;
            if (valid) {
                if (ex_bits == 2) {
                    code = ((uint32_t)src[t * 4 + (l & 3u)] >> (2 * (l >> 2))) & 3u;
                } else {
                    const uint32_t lo = src[t * 12 + (l & 7u)], hi = src[t * 12 + 8 + (l & 3u)];
                    code = ((lo >> (l < 8 ? 0 : 4)) & 15u) | (((hi >> (2 * (l >> 2))) & 3u) << 4);
                }
            }
            const uint32_t bit = k * ex_bits, idx = bit >> 5, sh = bit & 31u;
            u[idx] |= code << sh;
            if (sh + ex_bits > 32) u[idx + 1] |= code >> (32 - sh);
        }
        dst[unit * 16] = make_uint4(u[0], u[1], u[2], u[3]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_spread(const T* __restrict__ src, const uint64_t* __restrict__ block_dense0,
                                                const uint32_t* __restrict__ block_nv, uint32_t nb, T fill, T* __restrict__ dst) {
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x, b = slot >> 5, v = slot & 31u;
    if (b >= nb) return;
    dst[slot] = v < block_nv[b] ? src[block_dense0[b] + v] : fill;
}

// squared norms (f64 accumulate) and split-bf16 image of the rotated centroids
__global__ __launch_bounds__(256) void k_centroid_arrays(const float* __restrict__ cent, uint32_t nlist, uint32_t D,
                                                         float* __restrict__ cnorm2, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (c >= nlist) return;
    const float* row = cent + (size_t)c * D;
    double a = 0.0;
    for (uint32_t i = lane; i < D; i += 64) {
        const float x = row[i];
        a += (double)x * (double)x;
        uint16_t h, l;
        bf16_split(x, h, l);
        hi[(size_t)c * D + i] = h;
        lo[(size_t)c * D + i] = l;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) a += __shfl_xor(a, d, 64);
    if (lane == 0) cnorm2[c] = (float)a;
}

// ---- streamed build: slots of one pushed chunk -------------------------------------------------------------------
// rows of the chunk sorted by (list, source index): first row of every list present in the chunk
__global__ void k_chunk_first(const uint32_t* __restrict__ sorted_list, uint32_t n, uint32_t* __restrict__ chunk_first) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (r == 0 || sorted_list[r - 1] != sorted_list[r]) chunk_first[sorted_list[r]] = r;
}
// slot of row r = first slot of its list + vectors of that list pushed by earlier chunks + rank inside this chunk
__global__ void k_chunk_slots(const uint32_t* __restrict__ sorted_list, const uint32_t* __restrict__ sorted_src, uint32_t n,
                              const uint32_t* __restrict__ list_gb0, const uint32_t* __restrict__ list_cursor,
                              const uint32_t* __restrict__ chunk_first, uint32_t* __restrict__ row_src, uint32_t* __restrict__ row_slot) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t c = sorted_list[r];
    row_src[r] = sorted_src[r];
    row_slot[r] = list_gb0[c] * 32u + list_cursor[c] + (r - chunk_first[c]);
}
__global__ void k_chunk_advance(const uint32_t* __restrict__ sorted_list, uint32_t n, const uint32_t* __restrict__ chunk_first,
                                uint32_t* __restrict__ list_cursor) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (r + 1 == n || sorted_list[r + 1] != sorted_list[r]) list_cursor[sorted_list[r]] += r - chunk_first[sorted_list[r]] + 1u;
}

// ---- optional full-precision rerank (north_star item 3; NOT part of the reference, default off) -------------------
// One workgroup per query: exact squared distance / dot of the query (input space) with the raw vector of every
// returned id in the canonical 8-accumulator order of math::l2_distance_sqr / dot (src/math.rs:154-245), then the
// top_k (<= 1024 here) results re-sorted by (exact score, position) — ascending distance for L2, descending for IP.
__global__ __launch_bounds__(kThreads) void k_rerank(const float* __restrict__ queries, uint32_t dim, const float* __restrict__ raw,
                                                     uint64_t n_raw, int metric, uint32_t top_k, uint64_t* __restrict__ ids,
                                                     float* __restrict__ scores, const uint32_t* __restrict__ counts) {
    extern __shared__ __align__(16) unsigned char smraw[];
    float* sq = reinterpret_cast<float*>(smraw);                         // [dim]
    unsigned long long* key = reinterpret_cast<unsigned long long*>(sq + ((dim + 1u) & ~1u)); // [top_k]
    unsigned long long* idv = key + top_k;                                // [top_k]
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t cnt = counts[q] < top_k ? counts[q] : top_k;
    for (uint32_t i = tid; i < dim; i += kThreads) sq[i] = queries[(size_t)q * dim + i];
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += kThreads) {
        const uint64_t id = ids[(size_t)q * top_k + i];
        float s = __int_as_float(0x7fc00000);
        if (id < n_raw) s = metric == 0 ? canon_l2(sq, raw + id * dim, dim) : canon_dot(sq, raw + id * dim, dim);
        int32_t k = total_key(s);
        if (metric == 1) k = ~k;
        key[i] = ((unsigned long long)((uint32_t)k ^ 0x80000000u) << 32) | i;
        idv[i] = id;
    }
    __syncthreads();
    // rank by counting (keys are distinct: they carry the position)
    for (uint32_t i = tid; i < cnt; i += kThreads) {
        const unsigned long long my = key[i];
        uint32_t r = 0;
        for (uint32_t j = 0; j < cnt; ++j) r += key[j] < my ? 1u : 0u;
        int32_t k = (int32_t)((uint32_t)(my >> 32) ^ 0x80000000u);
        if (metric == 1) k = ~k;
        ids[(size_t)q * top_k + r] = idv[i];
        scores[(size_t)q * top_k + r] = key_to_float(k);
    }
}

} // namespace rbq
