// rbq_hostcheck.cpp — C-ABI test shim over rbq_host_logic.hpp (the GPU-free host logic of librbq.so), built by
// tests/test_sanitizers.py with -fsanitize=address,undefined.  TEST INFRASTRUCTURE: not linked into the product.
#include "rbq_host_logic.hpp"

extern "C" {

// rbq1_parse + a read of EVERY byte each list view claims (an out-of-range view is an ASan report, not a silent pass)
int rbq_hostcheck_parse(const void* bytes, size_t len, char* detail, size_t detail_cap, uint64_t* n_lists, uint64_t* n_vectors,
                        uint64_t* checksum) {
    rbq_header h;
    std::vector<rbq_host::ListSrc> lists;
    std::string msg;
    const int rc = rbq_host::rbq1_parse(bytes, len, &h, &lists, &msg);
    if (detail && detail_cap) {
        const size_t c = std::min(detail_cap - 1, msg.size());
        std::memcpy(detail, msg.data(), c);
        detail[c] = 0;
    }
    if (rc != RBQ_OK) return rc;
    uint64_t sum = 0, nv = 0;
    const size_t D = h.padded_dim, stride = D * 4 + 384, exb = h.ex_bits ? D * h.ex_bits / 8 : 0;
    auto eat = [&](const uint8_t* p, size_t n) { for (size_t i = 0; i < n; ++i) sum += p[i]; };
    eat(h.rotator_blob, h.rotator_len);
    for (const rbq_host::ListSrc& L : lists) {
        eat(L.centroid, D * 4);
        eat(L.ids, L.n * 8);
        eat(L.batch_data, ((L.n + 31) / 32) * stride);
        for (uint64_t v = 0; v < L.n; ++v) eat(L.ex + v * L.ex_stride, exb);
        eat(L.fadd, L.n * 4);
        eat(L.fres, L.n * 4);
        nv += L.n;
    }
    if (n_lists) *n_lists = lists.size();
    if (n_vectors) *n_vectors = nv;
    if (checksum) *checksum = sum;
    return RBQ_OK;
}

uint32_t rbq_hostcheck_crc32(const void* p, size_t n) { return rbq_host::crc32_ieee((const uint8_t*)p, n); }

void rbq_hostcheck_outpack(uint64_t n, uint32_t top_k, int diag, uint64_t out[5]) {
    const rbq_host::OutPack op(n, top_k, diag != 0);
    out[0] = op.o_ids; out[1] = op.o_scores; out[2] = op.o_counts; out[3] = op.o_diag; out[4] = op.total;
}
void rbq_hostcheck_shard(uint64_t r, uint64_t R, uint64_t nq, uint64_t out[2]) { rbq_host::shard_range(r, R, nq, &out[0], &out[1]); }
// sub-batch plan of a call: writes at most cap (first, count) pairs, returns the number of sub-batches
uint64_t rbq_hostcheck_plan(uint64_t nq, uint64_t forced, uint64_t* out, uint64_t cap) {
    const auto plan = rbq_host::subbatch_plan(nq, forced);
    for (size_t i = 0; i < plan.size() && i < cap; ++i) { out[2 * i] = plan[i].first; out[2 * i + 1] = plan[i].second; }
    return plan.size();
}

} // extern "C"
