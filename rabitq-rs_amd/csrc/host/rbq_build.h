/* rbq_build.h — C ABI of the CPU index builder (train-time harness; see rbq_build.cpp). */
#ifndef RBQ_BUILD_H
#define RBQ_BUILD_H
#include <stdint.h>
#include "../../../include/rbq.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct rbq_built rbq_built;

/* IvfRabitqIndex::train_with_clusters (reference src/ivf.rs:1025-1103). centroids are in the
 * ORIGINAL space ([nlist][dim]); data [n][dim]; assignments [n] < nlist. */
int rbq_build_train_with_clusters(const float* data, uint64_t n, uint32_t dim,
                                  const float* centroids, uint64_t nlist, const uint32_t* assignments,
                                  uint32_t total_bits, uint8_t metric, uint8_t rotator_type,
                                  uint64_t seed, int use_faster_config, rbq_built** out);
const rbq_header*    rbq_built_header(const rbq_built* b);
const rbq_list_view* rbq_built_lists(const rbq_built* b);
float                rbq_built_t_const(const rbq_built* b);
void                 rbq_built_free(rbq_built* b);
/* IvfRabitqIndex::save_to_writer (src/ivf.rs:1317-1474) into a malloc'd buffer. */
int  rbq_built_save_rbq1(const rbq_built* b, uint8_t** bytes, uint64_t* len);
void rbq_build_free_bytes(uint8_t* p);

void rbq_build_pack_binary_code(const uint8_t* bits, uint8_t* packed, uint64_t dim);
void rbq_build_pack_ex_code_1bit(const uint16_t* c, uint8_t* p, uint64_t dim);
void rbq_build_pack_ex_code_2bit(const uint16_t* c, uint8_t* p, uint64_t dim);
void rbq_build_pack_ex_code_6bit(const uint16_t* c, uint8_t* p, uint64_t dim);
void rbq_build_pack_codes(const uint8_t* codes, uint64_t num_vectors, uint64_t dim_bytes, uint8_t* packed);
uint32_t rbq_build_crc32(const uint8_t* p, uint64_t n);
void rbq_build_rotate(const rbq_header* h, const float* in, float* out);
int  rbq_build_kmeans(const float* data, uint64_t n, uint32_t dim, uint64_t k, int iters, uint64_t seed,
                      float* centroids, uint32_t* assignments);
#ifdef __cplusplus
}
#endif
#endif
