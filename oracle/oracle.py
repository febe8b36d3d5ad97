"""ctypes binding of the CPU oracle (oracle/librbq_ref.so). TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the
product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("RBQ_REF_LIB") or os.path.join(_HERE, "librbq_ref.so")  # (override: the sanitizer build)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        f32p, u8p, vp = C.c_void_p, C.c_void_p, C.c_void_p
        L.ref_dot.restype = C.c_float
        L.ref_dot.argtypes = [f32p, f32p, C.c_size_t]
        L.ref_l2_distance_sqr.restype = C.c_float
        L.ref_l2_distance_sqr.argtypes = [f32p, f32p, C.c_size_t]
        L.ref_floor_log2.restype = C.c_uint32
        L.ref_floor_log2.argtypes = [C.c_uint64]
        L.ref_padded_dim.restype = C.c_uint32
        L.ref_padded_dim.argtypes = [C.c_uint32, C.c_int]
        L.ref_fht.argtypes = [f32p, C.c_size_t]
        L.ref_fht_kac_rotate.argtypes = [C.c_uint32, C.c_uint32, u8p, f32p, f32p]
        L.ref_matrix_rotate.argtypes = [C.c_uint32, C.c_uint32, f32p, f32p, f32p]
        L.ref_rotate.argtypes = [vp, f32p, f32p]
        L.ref_pack_lut_f32.argtypes = [f32p, C.c_size_t, f32p]
        L.ref_query_lut.argtypes = [f32p, C.c_size_t, u8p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ref_query_precompute.argtypes = [f32p, C.c_size_t, C.c_uint32, vp]
        for n in ("ref_accumulate_batch_scalar", "ref_accumulate_batch_shuffle_emul", "ref_accumulate_batch"):
            getattr(L, n).argtypes = [u8p, u8p, C.c_size_t, vp]
        L.ref_simd_level.restype = C.c_int
        L.ref_force_simd_level.argtypes = [C.c_int]
        L.ref_unpack_single_vector_bytes.argtypes = [u8p, C.c_int, C.c_size_t, u8p]
        L.ref_compute_batch_distances.argtypes = [vp, C.c_float, C.c_float, f32p, f32p, f32p,
                                                  C.c_float, C.c_float, C.c_float, f32p, f32p, f32p]
        for n in ("ref_ip_packed_ex2", "ref_ip_packed_ex6"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [f32p, u8p, C.c_size_t]
        L.ref_heap_trace.restype = C.c_int
        L.ref_heap_trace.argtypes = [f32p, vp, C.c_size_t, C.c_uint32, vp, f32p, vp]
        L.ref_have_avx512.restype = C.c_int
        for n in ("ref_reduce_add_16", "ref_reduce_add_16_avx512"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [f32p]
        L.ref_ip_packed_ex_avx512.restype = C.c_float
        L.ref_ip_packed_ex_avx512.argtypes = [f32p, u8p, C.c_size_t, C.c_uint32]
        for n in ("ref_ip_packed_ex_avx2_order", "ref_ip_packed_ex_avx2_real", "ref_ip_packed_ex_scalar_order"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [f32p, u8p, C.c_size_t, C.c_uint32]
        L.ref_set_variant.argtypes = [C.c_int]
        L.ref_get_variant.restype = C.c_int
        L.ref_ex_dot.restype = C.c_float
        L.ref_ex_dot.argtypes = [f32p, u8p, C.c_size_t, C.c_uint32]
        L.ref_select_probes.restype = C.c_size_t
        L.ref_select_probes.argtypes = [vp, vp, f32p, C.c_uint32, vp]
        L.ref_search.restype = C.c_int
        L.ref_search.argtypes = [vp, vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint64,
                                 vp, vp, vp, vp]
        L.ref_search_lists.restype = C.c_int
        L.ref_search_lists.argtypes = [vp, vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, vp, vp, vp, vp, vp]
        L.ref_search_batch.restype = C.c_int
        L.ref_search_batch.argtypes = [vp, vp, f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp,
                                       C.c_uint64, vp, vp, vp, vp, C.c_int]
        L.ref_num_threads.restype = C.c_int
        L.ref_search_naive.restype = C.c_int
        L.ref_search_naive.argtypes = [vp, vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
        L.ref_posting_scan_batch.restype = C.c_int
        L.ref_posting_scan_batch.argtypes = [vp, vp, f32p, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32,
                                             vp, vp, vp, C.c_int]
        _LIB = L
    return _LIB


class QueryConsts(C.Structure):
    _fields_ = [("sum_q", C.c_float), ("query_norm", C.c_float), ("k1x_sum_q", C.c_float),
                ("kbx_sum_q", C.c_float), ("binary_scale", C.c_float)]


def _p(a):
    return a.ctypes.data if a is not None else None


def _addr(ptr):
    return C.cast(ptr, C.c_void_p)


def search_batch(built, queries, top_k, nprobe, filter_words=None, filter_nbits=0, want_diag=False,
                 nthreads=0):
    """Oracle `batch_search` over a builder.BuiltIndex (or anything with hdr_ptr/lists_ptr).
    Returns (rc, ids[nq,k] u64, scores[nq,k] f32, counts[nq] u32, diag[nq,3] u64 | None)."""
    q = np.ascontiguousarray(queries, dtype=np.float32)
    if q.ndim == 1:
        q = q[None, :]
    nq, qd = q.shape
    ids = np.full((nq, max(top_k, 1)), np.iinfo(np.uint64).max, np.uint64)[:, :top_k].copy()
    scores = np.full((nq, top_k), np.nan, np.float32)
    counts = np.zeros(nq, np.uint32)
    diag = np.zeros((nq, 3), np.uint64) if want_diag else None
    fw = np.ascontiguousarray(filter_words, dtype=np.uint32) if filter_words is not None else None
    rc = lib().ref_search_batch(_addr(built.hdr_ptr), _addr(built.lists_ptr), _p(q), nq, qd, top_k, nprobe,
                                _p(fw), filter_nbits, _p(ids), _p(scores), _p(counts), _p(diag), nthreads)
    return rc, ids, scores, counts, diag


def search_lists(built, query, top_k, nprobe, filter_words=None, filter_nbits=0):
    """One query: (probe order cids, per probed list the number of its vectors the reference did NOT skip by the lower bound)."""
    q = np.ascontiguousarray(query, dtype=np.float32)
    ids = np.zeros(max(top_k, 1), np.uint64)
    scores = np.zeros(max(top_k, 1), np.float32)
    cnt = C.c_uint32()
    cids = np.zeros(built.n_lists, np.uint32)
    ev = np.zeros(built.n_lists, np.uint32)
    n = C.c_uint32()
    fw = np.ascontiguousarray(filter_words, dtype=np.uint32) if filter_words is not None else None
    rc = lib().ref_search_lists(_addr(built.hdr_ptr), _addr(built.lists_ptr), _p(q), q.shape[0], top_k, nprobe, _p(fw), filter_nbits,
                                _p(ids), _p(scores), C.byref(cnt), _p(cids), _p(ev), C.byref(n))
    assert rc == 0, rc
    return cids[:n.value].copy(), ev[:n.value].copy()


def search_naive(built, query, top_k, nprobe):
    q = np.ascontiguousarray(query, dtype=np.float32)
    ids = np.zeros(top_k, np.uint64)
    scores = np.zeros(top_k, np.float32)
    cnt = C.c_uint32()
    rc = lib().ref_search_naive(_addr(built.hdr_ptr), _addr(built.lists_ptr), _p(q), q.shape[0], top_k, nprobe,
                                _p(ids), _p(scores), C.byref(cnt))
    return rc, ids[:cnt.value], scores[:cnt.value]


def rotate(built, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(built.padded_dim, np.float32)
    lib().ref_rotate(_addr(built.hdr_ptr), _p(x), _p(out))
    return out


def query_lut(rq):
    rq = np.ascontiguousarray(rq, dtype=np.float32)
    D = rq.shape[0]
    lut = np.empty(4 * D, np.uint8)
    d, s = C.c_float(), C.c_float()
    lib().ref_query_lut(_p(rq), D, _p(lut), C.byref(d), C.byref(s))
    return lut, d.value, s.value


def query_precompute(rq, ex_bits):
    rq = np.ascontiguousarray(rq, dtype=np.float32)
    qc = QueryConsts()
    lib().ref_query_precompute(_p(rq), rq.shape[0], ex_bits, C.byref(qc))
    return qc


def select_probes(built, rq, nprobe):
    rq = np.ascontiguousarray(rq, dtype=np.float32)
    out = np.empty(built.n_lists, np.uint32)
    n = lib().ref_select_probes(_addr(built.hdr_ptr), _addr(built.lists_ptr), _p(rq), nprobe, _p(out))
    return out[:n].copy()


def posting_scan_batch(built, queries, top_k, list_ids, list_counts, nthreads=0):
    """Oracle MSTG posting-list scan. list_ids: [nq, max_lists] u32, list_counts: [nq] u32."""
    q = np.ascontiguousarray(queries, dtype=np.float32)
    li = np.ascontiguousarray(list_ids, dtype=np.uint32)
    lc = np.ascontiguousarray(list_counts, dtype=np.uint32)
    nq, qd = q.shape
    ids = np.full((nq, top_k), np.iinfo(np.uint64).max, np.uint64)
    scores = np.full((nq, top_k), np.nan, np.float32)
    counts = np.zeros(nq, np.uint32)
    rc = lib().ref_posting_scan_batch(_addr(built.hdr_ptr), _addr(built.lists_ptr), _p(q), nq, qd, top_k, _p(li), _p(lc),
                                      li.shape[1], _p(ids), _p(scores), _p(counts), nthreads)
    return rc, ids, scores, counts


VARIANTS = {"default": 0, "ex_avx2": 1, "ex_scalar": 2, "epilogue_scalar": 4, "ex_scalar+epilogue_scalar": 6, "contract": 8,
            "ex_avx2+contract": 9}


class variant:
    """with oracle.variant("ex_avx2"): ...   — one of the reference's other compile-time numeric variants (rbq_ref.c)"""

    def __init__(self, name_or_mask):
        self.mask = VARIANTS[name_or_mask] if isinstance(name_or_mask, str) else int(name_or_mask)

    def __enter__(self):
        self.old = lib().ref_get_variant()
        lib().ref_set_variant(self.mask)
        return self

    def __exit__(self, *a):
        lib().ref_set_variant(self.old)


def variant_diff_table(built, queries, top_k, nprobe, names=None):
    """For every numeric variant: fraction of queries whose returned ids differ from the default variant's, fraction whose score
    bits differ, and the largest relative score difference among queries with the same ids."""
    rc, ids0, sc0, cnt0, _ = search_batch(built, queries, top_k, nprobe)
    assert rc == 0
    out = {}
    for name in (names or [n for n in VARIANTS if n != "default"]):
        with variant(name):
            rc, ids, sc, cnt, _ = search_batch(built, queries, top_k, nprobe)
        assert rc == 0
        same_ids = (ids == ids0).all(axis=1) & (cnt == cnt0)
        bits_same = np.array([np.array_equal(sc[q, :cnt0[q]].view(np.uint32), sc0[q, :cnt0[q]].view(np.uint32)) for q in range(len(ids))])
        rel = 0.0
        rel_scale = 0.0  # against max(|score|, the batch's median |score|): near-exact hits have scores near 0, where a relative
        valid = np.arange(sc0.shape[1])[None, :] < cnt0[:, None]  # difference measures cancellation, not the variant
        med = float(np.median(np.abs(sc0[valid]))) if valid.any() else 1.0
        for q in np.nonzero(same_ids)[0]:
            c = int(cnt0[q])
            if c:
                d = np.abs(sc[q, :c] - sc0[q, :c])
                rel = max(rel, float(np.max(d / np.maximum(np.abs(sc0[q, :c]), 1e-30))))
                rel_scale = max(rel_scale, float(np.max(d / np.maximum(np.abs(sc0[q, :c]), med))))
        out[name] = {"queries": int(len(ids)), "ids_differ_frac": float(1.0 - same_ids.mean()),
                     "score_bits_differ_frac": float(1.0 - (bits_same & same_ids).mean()), "max_rel_score_diff_same_ids": rel,
                     "max_score_diff_over_scale_same_ids": rel_scale}
    return out
