// k_build.hip — translation unit of the index-build kernels: GPU-side encoder (encode.hpp), reference-layout ->
// device-layout converters and build helpers (relayout.hpp), the rocPRIM sort they use.  gfx950 only.
#include <hip/hip_runtime.h>
#include <cstring>

#include "launch.hpp"
#include "kernels.hpp"
#include "encode.hpp"
#include "relayout.hpp"
#include <rocprim/rocprim.hpp>

namespace rbq {

hipError_t launch_rotate_rows(const float* src, const uint32_t* map, uint32_t nrows, uint32_t dim, uint32_t D, int rotator,
                              const uint8_t* rot_blob, uint32_t trunc, float fac, float* rows, hipStream_t s) {
    if (!nrows) return hipSuccess;
    hipLaunchKernelGGL(k_rotate_rows, dim3(nrows), dim3(kThreads), (size_t)D * 4 * 2, s, src, map, dim, D, rotator, rot_blob, trunc,
                       fac, rows);
    return hipGetLastError();
}
hipError_t launch_encode(const EncodeParams& P, hipStream_t s) {
    if (!P.nslots) return hipSuccess;
    const dim3 grid((P.nslots + kEncThreads - 1) / kEncThreads);
    if (P.row_slot) hipLaunchKernelGGL(k_encode<true>, grid, dim3(kEncThreads), 0, s, P);
    else hipLaunchKernelGGL(k_encode<false>, grid, dim3(kEncThreads), 0, s, P);
    return hipGetLastError();
}
hipError_t launch_pack_ex(const uint8_t* raw, const uint32_t* slot_src, const uint32_t* row_slot, uint32_t nrows, uint32_t D,
                          uint32_t ex_bits, uint8_t* ex, hipStream_t s) {
    if (!nrows) return hipSuccess;
    hipLaunchKernelGGL(k_pack_ex, dim3((nrows + 15) / 16), dim3(256), 0, s, raw, slot_src, row_slot, nrows, D, ex_bits, ex);
    return hipGetLastError();
}
hipError_t launch_block_summary(const uint8_t* blocks, const uint32_t* block_nv, uint32_t nblocks, uint32_t Dc, BlockSummary* bsum,
                                hipStream_t s) {
    if (!nblocks) return hipSuccess;
    hipLaunchKernelGGL(k_block_summary, dim3((nblocks + 7) / 8), dim3(256), 0, s, blocks, block_nv, nblocks, Dc, bsum);
    return hipGetLastError();
}
hipError_t launch_list_summaries(const uint8_t* blocks, const uint8_t* ex, const float* fadd_ex, const float* fres_ex, const float* cent,
                                 const BlockSummary* bsum, const uint32_t* list_gb0, const uint32_t* list_n, uint32_t nlist, uint32_t D,
                                 uint32_t Dc, uint32_t ex_bits, BlockSummaryEx* bsumx, BlockSummary* lsum, hipStream_t s) {
    if (!nlist) return hipSuccess;
    hipLaunchKernelGGL(k_list_summaries, dim3(nlist), dim3(256), (size_t)D * 4, s, blocks, ex, fadd_ex, fres_ex, cent, bsum, list_gb0,
                       list_n, D, Dc, ex_bits, bsumx, lsum);
    return hipGetLastError();
}
hipError_t launch_count_assign(const uint32_t* assign, uint64_t n, uint32_t nlist, uint32_t* counts, uint32_t* err, hipStream_t s) {
    hipLaunchKernelGGL(k_count_assign, dim3(1024), dim3(256), 0, s, assign, n, nlist, counts, err);
    return hipGetLastError();
}
hipError_t launch_iota(uint32_t* x, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_iota, dim3(1024), dim3(256), 0, s, x, n);
    return hipGetLastError();
}
hipError_t launch_scatter_slots(const uint32_t* sorted_list, const uint32_t* sorted_src, uint64_t n, const uint32_t* list_gb0,
                                const uint64_t* vstart, uint32_t* slot_src, hipStream_t s) {
    hipLaunchKernelGGL(k_scatter_slots, dim3(1024), dim3(256), 0, s, sorted_list, sorted_src, n, list_gb0, vstart, slot_src);
    return hipGetLastError();
}
hipError_t sort_pairs_u32(void* tmp, size_t* tmp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                          uint32_t* vals_out, size_t n, unsigned bits, hipStream_t s) {
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, bits, s);
}
hipError_t launch_chunk_first(const uint32_t* sorted_list, uint32_t n, uint32_t* chunk_first, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_chunk_first, dim3((n + 255) / 256), dim3(256), 0, s, sorted_list, n, chunk_first);
    return hipGetLastError();
}
hipError_t launch_chunk_slots(const uint32_t* sorted_list, const uint32_t* sorted_src, uint32_t n, const uint32_t* list_gb0,
                              const uint32_t* list_cursor, const uint32_t* chunk_first, uint32_t* row_src, uint32_t* row_slot,
                              hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_chunk_slots, dim3((n + 255) / 256), dim3(256), 0, s, sorted_list, sorted_src, n, list_gb0, list_cursor,
                       chunk_first, row_src, row_slot);
    return hipGetLastError();
}
hipError_t launch_chunk_advance(const uint32_t* sorted_list, uint32_t n, const uint32_t* chunk_first, uint32_t* list_cursor,
                                hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_chunk_advance, dim3((n + 255) / 256), dim3(256), 0, s, sorted_list, n, chunk_first, list_cursor);
    return hipGetLastError();
}
hipError_t launch_relayout_blocks(const uint8_t* recs, uint32_t nb, uint32_t D, uint32_t Dc, uint8_t* blocks, hipStream_t s) {
    if (!nb) return hipSuccess;
    hipLaunchKernelGGL(k_relayout_blocks, dim3((nb + 7) / 8), dim3(256), 0, s, recs, nb, D, Dc, blocks);
    return hipGetLastError();
}
hipError_t launch_relayout_ex(const uint8_t* exsrc, const uint64_t* block_dense0, const uint32_t* block_nv, uint32_t nb, uint32_t D,
                              uint32_t ex_bits, uint8_t* ex, hipStream_t s) {
    if (!nb || !ex_bits) return hipSuccess;
    hipLaunchKernelGGL(k_relayout_ex, dim3(nb * 2), dim3(256), 0, s, exsrc, block_dense0, block_nv, nb, D, ex_bits, ex);
    return hipGetLastError();
}
hipError_t launch_spread_u64(const uint64_t* src, const uint64_t* block_dense0, const uint32_t* block_nv, uint32_t nb, uint64_t fill,
                             uint64_t* dst, hipStream_t s) {
    if (!nb) return hipSuccess;
    hipLaunchKernelGGL(k_spread<uint64_t>, dim3((nb + 7) / 8), dim3(256), 0, s, src, block_dense0, block_nv, nb, fill, dst);
    return hipGetLastError();
}
hipError_t launch_spread_f32(const float* src, const uint64_t* block_dense0, const uint32_t* block_nv, uint32_t nb, float fill,
                             float* dst, hipStream_t s) {
    if (!nb) return hipSuccess;
    hipLaunchKernelGGL(k_spread<float>, dim3((nb + 7) / 8), dim3(256), 0, s, src, block_dense0, block_nv, nb, fill, dst);
    return hipGetLastError();
}
hipError_t launch_centroid_arrays(const float* cent, uint32_t nlist, uint32_t D, float* cnorm2, uint16_t* hi, uint16_t* lo,
                                  hipStream_t s) {
    hipLaunchKernelGGL(k_centroid_arrays, dim3((nlist + 3) / 4), dim3(256), 0, s, cent, nlist, D, cnorm2, hi, lo);
    return hipGetLastError();
}
hipError_t launch_rerank(const float* queries, uint32_t nq, uint32_t dim, const float* raw, uint64_t n_raw, int metric,
                         uint32_t top_k, uint64_t* ids, float* scores, const uint32_t* counts, hipStream_t s) {
    if (!nq) return hipSuccess;
    const size_t lds = (size_t)((dim + 1u) & ~1u) * 4 + (size_t)top_k * 16;
    hipLaunchKernelGGL(k_rerank, dim3(nq), dim3(kThreads), lds, s, queries, dim, raw, n_raw, metric, top_k, ids, scores, counts);
    return hipGetLastError();
}

} // namespace rbq
