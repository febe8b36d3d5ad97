/*
 * rbq_ref.h — CPU oracle API (TEST INFRASTRUCTURE ONLY; see rbq_ref.c header).
 * Shares only the plain data-contract structs of include/rbq.h with the product.
 */
#ifndef RBQ_REF_H
#define RBQ_REF_H
#include <stddef.h>
#include <stdint.h>
#include "../include/rbq.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    float sum_q, query_norm, k1x_sum_q, kbx_sum_q, binary_scale;
} ref_query_consts;

float    ref_dot(const float* a, const float* b, size_t len);
float    ref_l2_distance_sqr(const float* a, const float* b, size_t len);
uint32_t ref_floor_log2(uint64_t x);
uint32_t ref_padded_dim(uint32_t dim, int rotator);
void     ref_fht(float* data, size_t n);
void     ref_fht_kac_rotate(uint32_t dim, uint32_t D, const uint8_t* flip, const float* in, float* out);
void     ref_matrix_rotate(uint32_t dim, uint32_t D, const float* matrix, const float* in, float* out);
void     ref_rotate(const rbq_header* h, const float* in, float* out);
void     ref_pack_lut_f32(const float* q, size_t D, float* lut);
void     ref_query_lut(const float* q, size_t D, uint8_t* lut8, float* delta_out, float* sum_vl_out);
void     ref_query_precompute(const float* q, size_t D, uint32_t ex_bits, ref_query_consts* c);
void     ref_accumulate_batch_scalar(const uint8_t* codes, const uint8_t* lut, size_t D, uint16_t* res);
void     ref_accumulate_batch_shuffle_emul(const uint8_t* codes, const uint8_t* lut, size_t D, uint16_t* res);
void     ref_accumulate_batch(const uint8_t* codes, const uint8_t* lut, size_t D, uint16_t* res);
int      ref_simd_level(void);
void     ref_force_simd_level(int level);
void     ref_unpack_single_vector_bytes(const uint8_t* packed, int vec_idx, size_t dim_bytes, uint8_t* out_bytes);
void     ref_compute_batch_distances(const uint16_t* accu, float delta, float sum_vl,
                                     const float* f_add, const float* f_rescale, const float* f_error,
                                     float g_add, float g_error, float k1x,
                                     float* ip, float* est, float* lb);
float    ref_ip_packed_ex2(const float* q, const uint8_t* code, size_t D);
float    ref_ip_packed_ex6(const float* q, const uint8_t* code, size_t D);
float    ref_ex_dot(const float* q, const uint8_t* code, size_t D, uint32_t ex_bits);
int      ref_heap_trace(const float* dist, const uint64_t* ids, size_t n, uint32_t top_k, uint64_t* out_ids, float* out_dist,
                        uint32_t* out_len);
/* numeric variants of the reference (rbq_ref.c, "Numeric variants"): a bit mask, 0 = the default (target-cpu=native on AVX-512) */
enum { REF_VAR_EX_AVX2 = 1, REF_VAR_EX_SCALAR = 2, REF_VAR_EPI_SCALAR = 4, REF_VAR_CONTRACT = 8 };
void     ref_set_variant(int mask);
int      ref_get_variant(void);
float    ref_ip_packed_ex_avx2_order(const float* q, const uint8_t* code, size_t D, uint32_t ex_bits);
float    ref_ip_packed_ex_avx2_real(const float* q, const uint8_t* code, size_t D, uint32_t ex_bits); /* real AVX2 instructions */
float    ref_ip_packed_ex_scalar_order(const float* q, const uint8_t* code, size_t D, uint32_t ex_bits);
int      ref_have_avx512(void);
float    ref_reduce_add_16(const float* s);                 /* the halving tree the oracle uses for _mm512_reduce_add_ps */
float    ref_reduce_add_16_avx512(const float* s);          /* the compiler's own _mm512_reduce_add_ps (0 without AVX-512) */
float    ref_ip_packed_ex_avx512(const float* q, const uint8_t* code, size_t D, uint32_t ex_bits); /* real 16-lane FMA + reduce */
size_t   ref_select_probes(const rbq_header* h, const rbq_list_view* lists, const float* rq,
                           uint32_t nprobe_in, uint32_t* out_cids);
int      ref_search(const rbq_header* h, const rbq_list_view* lists, const float* query, uint32_t query_dim,
                    uint32_t top_k, uint32_t nprobe, const uint32_t* filter_words, uint64_t filter_nbits,
                    uint64_t* out_ids, float* out_scores, uint32_t* out_count, rbq_diag* diag);
int      ref_search_lists(const rbq_header* h, const rbq_list_view* lists, const float* query, uint32_t query_dim,
                          uint32_t top_k, uint32_t nprobe, const uint32_t* filter_words, uint64_t filter_nbits,
                          uint64_t* out_ids, float* out_scores, uint32_t* out_count,
                          uint32_t* probe_cids, uint32_t* probe_evaluated, uint32_t* n_probed); /* per probed list: vectors not skipped */
int      ref_search_batch(const rbq_header* h, const rbq_list_view* lists, const float* queries, uint64_t nq,
                          uint32_t query_dim, uint32_t top_k, uint32_t nprobe,
                          const uint32_t* filter_words, uint64_t filter_nbits,
                          uint64_t* out_ids, float* out_scores, uint32_t* out_counts, rbq_diag* diag,
                          int nthreads);
int      ref_num_threads(void);
int      ref_search_naive(const rbq_header* h, const rbq_list_view* lists, const float* query, uint32_t query_dim,
                          uint32_t top_k, uint32_t nprobe, uint64_t* out_ids, float* out_scores, uint32_t* out_count);

int      ref_posting_scan(const rbq_header* h, const rbq_list_view* lists, const float* query, uint32_t query_dim,
                          uint32_t top_k, const uint32_t* list_ids, uint32_t n_sel,
                          uint64_t* out_ids, float* out_scores, uint32_t* out_count);
int      ref_posting_scan_batch(const rbq_header* h, const rbq_list_view* lists, const float* queries, uint64_t nq,
                                uint32_t query_dim, uint32_t top_k, const uint32_t* list_ids, const uint32_t* list_counts,
                                uint32_t max_lists, uint64_t* out_ids, float* out_scores, uint32_t* out_counts, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
