"""`IvfRabitqIndex` query façade over the C ABI (include/rbq.h)."""
import ctypes as C
import os

import numpy as np

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
LIB_PATH = os.environ.get("RBQ_LIB_PATH") or os.path.join(_HERE, "csrc", "librbq.so")  # override: kernel A/B builds


def _hip_runtime_of_torch_first():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so and asks for it by the
    unversioned name; librbq.so asks for libamdhip64.so.7.  If torch is loaded first, librbq.so binds to torch's copy
    (same SONAME) and the process has one runtime.  The other way round the loader maps a SECOND runtime for torch,
    which then sees no GPU ("No HIP GPUs are available": found by tests/diag/soak.py, whose first seed used this
    library before torch).  So when torch is installed it is imported before librbq.so is mapped; without torch
    (a C / Rust host, or a torch-less Python) there is only the system runtime and nothing to order."""
    import sys
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass


def lib():
    """Load csrc/librbq.so (HIP, gfx950). Fails loudly — there is no CPU fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} missing: the HIP extension must be built first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        _hip_runtime_of_torch_first()
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.rbq_index_create.restype = C.c_int
        L.rbq_index_create.argtypes = [vp, vp, C.c_int, vp, C.POINTER(vp)]
        L.rbq_index_load_rbq1.restype = C.c_int
        L.rbq_index_load_rbq1.argtypes = [vp, C.c_size_t, C.c_int, vp, C.POINTER(vp)]
        L.rbq_index_destroy.argtypes = [vp]
        for n in ("rbq_index_len", "rbq_index_cluster_count"):
            getattr(L, n).restype = C.c_uint64
            getattr(L, n).argtypes = [vp]
        for n in ("rbq_index_dim", "rbq_index_padded_dim"):
            getattr(L, n).restype = C.c_uint32
            getattr(L, n).argtypes = [vp]
        L.rbq_search_batch.restype = C.c_int
        L.rbq_search_batch.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint64,
                                       vp, vp, vp, vp]
        L.rbq_search_batch_device.restype = C.c_int
        L.rbq_search_batch_device.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp,
                                              C.c_uint64, vp, vp, vp, vp, vp]
        L.rbq_posting_scan_batch.restype = C.c_int
        L.rbq_posting_scan_batch.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32, vp, vp, vp]
        L.rbq_profile_begin.argtypes = [vp]
        L.rbq_profile_end.argtypes = [vp]
        L.rbq_profile_stage_ms.restype = C.c_double
        L.rbq_profile_stage_ms.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint64)]
        L.rbq_profile_scan_bytes.restype = C.c_uint64
        L.rbq_profile_scan_bytes.argtypes = [vp]
        L.rbq_debug_rank_fallbacks.restype = C.c_uint64
        L.rbq_debug_rank_fallbacks.argtypes = [vp]
        L.rbq_debug_stage_resources.restype = C.c_int
        L.rbq_debug_stage_resources.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, vp]
        for n_ in ("rbq_debug_head_exact_evaluations", "rbq_debug_head_exact_guard_trips"):
            getattr(L, n_).restype = C.c_uint64
            getattr(L, n_).argtypes = [vp]
        L.rbq_debug_tie_log_stats.restype = None
        L.rbq_debug_tie_log_stats.argtypes = [vp, vp]
        L.rbq_debug_bounce_copies.restype = C.c_uint64
        L.rbq_debug_bounce_copies.argtypes = []
        L.rbq_index_build_device.restype = C.c_int
        L.rbq_index_build_device.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_float, C.c_int, vp]
        L.rbq_debug_copy_index.restype = C.c_int
        L.rbq_debug_copy_index.argtypes = [vp, C.c_char_p, vp, C.c_uint64]
        L.rbq_profile_set_sampling.restype = None
        L.rbq_profile_set_sampling.argtypes = [vp, C.c_uint32]
        L.rbq_profile_select_stages.restype = None
        L.rbq_profile_select_stages.argtypes = [vp, C.c_uint32]
        L.rbq_debug_copy_workspace.restype = C.c_int
        L.rbq_debug_copy_workspace.argtypes = [vp, vp, C.c_char_p, vp, C.c_uint64]
        L.rbq_debug_heap_restarts.restype = C.c_uint64
        L.rbq_debug_heap_restarts.argtypes = [vp]
        L.rbq_debug_set_option.restype = C.c_int
        L.rbq_debug_set_option.argtypes = [vp, C.c_char_p, C.c_int]
        L.rbq_strerror.restype = C.c_char_p
        L.rbq_strerror.argtypes = [C.c_int]
        L.rbq_last_error_detail.restype = C.c_int
        L.rbq_last_error_detail.argtypes = [C.c_char_p, C.c_size_t]
        L.rbq_abi_version.restype = C.c_uint32
        L.rbq_index_device_count.restype = C.c_uint32
        L.rbq_index_device_count.argtypes = [vp]
        L.rbq_build_stream_begin.restype = C.c_int
        L.rbq_build_stream_begin.argtypes = [vp, vp, vp, C.c_float, C.c_int, C.POINTER(vp)]
        L.rbq_build_stream_push.restype = C.c_int
        L.rbq_build_stream_push.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64]
        L.rbq_build_stream_finish.restype = C.c_int
        L.rbq_build_stream_finish.argtypes = [vp, C.c_int, vp, C.POINTER(vp)]
        L.rbq_build_stream_abort.restype = None
        L.rbq_build_stream_abort.argtypes = [vp]
        L.rbq_release_stream.restype = C.c_int
        L.rbq_release_stream.argtypes = [vp, vp]
        L.rbq_host_alloc.restype = vp
        L.rbq_host_alloc.argtypes = [C.c_size_t]
        L.rbq_host_free.restype = None
        L.rbq_host_free.argtypes = [vp]
        L.rbq_index_set_rerank_vectors.restype = C.c_int
        L.rbq_index_set_rerank_vectors.argtypes = [vp, vp, C.c_uint64]
        L.rbq_profile_stage_samples.restype = C.c_uint64
        L.rbq_profile_stage_samples.argtypes = [vp, C.c_char_p, vp, C.c_uint64]
        L.rbq_profile_counters.restype = C.c_int
        L.rbq_profile_counters.argtypes = [vp, vp, C.c_uint32]
        _LIB = L
    return _LIB


def _detail():
    buf = C.create_string_buffer(512)
    lib().rbq_last_error_detail(buf, 512)
    return buf.value.decode("utf-8", "replace")


def _check(rc):
    if rc != _abi.RBQ_OK:
        from . import RabitqError
        raise RabitqError(rc, _detail())


def _addr(ptr):
    return C.cast(ptr, C.c_void_p)


class IvfRabitqIndex:
    """Device-resident IVF+RaBitQ index; query methods mirror reference src/ivf.rs:1705-1752."""

    def __init__(self, handle):
        self._h = handle

    # -- construction -------------------------------------------------------------
    @staticmethod
    def _devices(device, devices):
        """(n_devices, int array | None): `devices` = list of HIP ordinals (one replica each), else `device`."""
        if devices is not None:
            devices = list(devices)
            return len(devices), (C.c_int * len(devices))(*devices)
        return 1, ((C.c_int * 1)(device) if device is not None else None)

    @classmethod
    def from_built(cls, built, device=None, devices=None):
        """Upload a builder.BuiltIndex (ClusterData-shaped host arrays) via rbq_index_create."""
        h = C.c_void_p()
        n, dev = cls._devices(device, devices)
        _check(lib().rbq_index_create(_addr(built.hdr_ptr), _addr(built.lists_ptr), n, dev, C.byref(h)))
        return cls(h)

    @classmethod
    def build_on_device(cls, hdr_ptr, centroids, d_data, d_assign, n, t_const, device=0):
        """GPU-side encoder (rbq_index_build_device): `hdr_ptr` is a ctypes pointer to an rbq_header (dim,
        padded_dim, metric, rotator + blob, ex_bits, n_lists), `centroids` a host [n_lists][dim] f32 array,
        `d_data` / `d_assign` device pointers to [n][dim] f32 vectors and [n] u32 cluster ids."""
        cent = np.ascontiguousarray(centroids, dtype=np.float32)
        h = C.c_void_p()
        _check(lib().rbq_index_build_device(_addr(hdr_ptr), cent.ctypes.data, C.c_void_p(d_data), C.c_void_p(d_assign),
                                            int(n), float(t_const), int(device), C.byref(h)))
        return cls(h)

    def debug_copy_index(self, name, out):
        """Diagnostic: copy one of the index's device arrays into the numpy array `out` (exact size)."""
        _check(lib().rbq_debug_copy_index(self._h, name.encode(), out.ctypes.data, out.nbytes))
        return out

    @classmethod
    def load_from_bytes(cls, data, device=None, devices=None):
        """`load_from_reader` (src/ivf.rs:1484-1702) straight into HBM."""
        h = C.c_void_p()
        n, dev = cls._devices(device, devices)
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        _check(lib().rbq_index_load_rbq1(buf, len(data), n, dev, C.byref(h)))
        return cls(h)

    @classmethod
    def load_from_path(cls, path, device=None):
        try:
            with open(path, "rb") as f:
                data = f.read()
        except OSError as e:
            from . import RabitqError
            raise RabitqError(_abi.RBQ_IO, str(e))
        return cls.load_from_bytes(data, device)

    # -- accessors ----------------------------------------------------------------
    def __len__(self):
        return lib().rbq_index_len(self._h)

    def is_empty(self):
        return len(self) == 0

    def cluster_count(self):
        return lib().rbq_index_cluster_count(self._h)

    @property
    def dim(self):
        return lib().rbq_index_dim(self._h)

    @property
    def padded_dim(self):
        return lib().rbq_index_padded_dim(self._h)

    # -- queries ------------------------------------------------------------------
    def batch_search_raw(self, queries, params, filter_words=None, filter_nbits=0, want_diag=False):
        """Returns (ids[nq,k] u64, scores[nq,k] f32, counts[nq] u32, diag[nq,3] u64|None)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq, qd = q.shape
        k = params.top_k
        ids = np.empty((nq, k), np.uint64)
        scores = np.empty((nq, k), np.float32)
        counts = np.zeros(nq, np.uint32)
        diag = np.zeros((nq, 3), np.uint64) if want_diag else None
        fw = np.ascontiguousarray(filter_words, dtype=np.uint32) if filter_words is not None else None
        rc = lib().rbq_search_batch(self._h, q.ctypes.data, nq, qd, k, params.nprobe,
                                    fw.ctypes.data if fw is not None else None, filter_nbits,
                                    ids.ctypes.data, scores.ctypes.data, counts.ctypes.data,
                                    diag.ctypes.data if diag is not None else None)
        _check(rc)
        return ids, scores, counts, diag

    def _results(self, ids, scores, counts, q):
        from . import SearchResult
        return [SearchResult(int(ids[q, i]), float(scores[q, i])) for i in range(int(counts[q]))]

    def search(self, query, params):
        """`IvfRabitqIndex::search` (src/ivf.rs:1705-1711)."""
        ids, scores, counts, _ = self.batch_search_raw(np.asarray(query, np.float32)[None, :], params)
        return self._results(ids, scores, counts, 0)

    def search_filtered(self, query, params, allowed_ids):
        """`search_filtered` (src/ivf.rs:1723-1730); `allowed_ids` plays the RoaringBitmap."""
        allowed = np.asarray(sorted(set(int(i) for i in allowed_ids)), dtype=np.uint64)
        nbits = int(allowed.max()) + 1 if allowed.size else 0
        words = np.zeros((nbits + 31) // 32 or 1, np.uint32)
        if allowed.size:
            np.bitwise_or.at(words, (allowed >> np.uint64(5)).astype(np.int64),
                             (np.uint32(1) << (allowed & np.uint64(31)).astype(np.uint32)))
        ids, scores, counts, _ = self.batch_search_raw(np.asarray(query, np.float32)[None, :], params,
                                                       words, nbits)
        return self._results(ids, scores, counts, 0)

    def batch_search(self, queries, params):
        """`batch_search` (src/ivf.rs:1743-1752): per-query results in input order."""
        ids, scores, counts, _ = self.batch_search_raw(queries, params)
        return [self._results(ids, scores, counts, q) for q in range(ids.shape[0])]

    def batch_query(self, queries, top_k, nprobe):
        """Python-binding shape of the reference (`batch_query`, src/python_bindings.rs:593-665):
        list of (k,2) f32 arrays [id, score] (ids cast to f32 exactly as the reference does)."""
        from . import SearchParams
        ids, scores, counts, _ = self.batch_search_raw(queries, SearchParams(top_k, nprobe))
        out = []
        for q in range(ids.shape[0]):
            c = int(counts[q])
            out.append(np.stack([ids[q, :c].astype(np.float32), scores[q, :c]], axis=1))
        return out

    def posting_scan(self, queries, top_k, list_ids, list_counts):
        """MSTG posting-list scan (reference src/mstg/index.rs:149-330): the caller has already chosen each
        query's posting lists. Returns (ids[nq,k] u64, distances[nq,k] f32 ascending, counts[nq] u32)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        li = np.ascontiguousarray(list_ids, dtype=np.uint32)
        lc = np.ascontiguousarray(list_counts, dtype=np.uint32)
        nq, qd = q.shape
        ids = np.empty((nq, top_k), np.uint64)
        scores = np.empty((nq, top_k), np.float32)
        counts = np.zeros(nq, np.uint32)
        _check(lib().rbq_posting_scan_batch(self._h, q.ctypes.data, nq, qd, top_k, li.ctypes.data, lc.ctypes.data,
                                            li.shape[1], ids.ctypes.data, scores.ctypes.data, counts.ctypes.data))
        return ids, scores, counts

    # -- device-pointer entry (bench / torch interop) --------------------------------
    def search_batch_device(self, d_queries, nq, query_dim, top_k, nprobe, d_ids, d_scores, d_counts,
                            stream=None, d_filter=None, filter_nbits=0, d_diag=None):
        _check(lib().rbq_search_batch_device(self._h, d_queries, nq, query_dim, top_k, nprobe, d_filter,
                                             filter_nbits, d_ids, d_scores, d_counts, d_diag, stream))

    def profile_begin(self, stages=("prep", "rank", "select", "scan"), every=1):
        mask = sum(1 << ("prep", "rank", "select", "scan").index(s) for s in stages)
        lib().rbq_profile_select_stages(self._h, mask)
        lib().rbq_profile_set_sampling(self._h, every)
        lib().rbq_profile_begin(self._h)

    def profile_end(self):
        lib().rbq_profile_end(self._h)

    def profile_stage(self, name):
        n = C.c_uint64()
        ms = lib().rbq_profile_stage_ms(self._h, name.encode(), C.byref(n))
        return ms, n.value

    def profile_stage_samples(self, name):
        """durations (ms) of the individual timed launches of a stage between profile_begin/end"""
        n = lib().rbq_profile_stage_samples(self._h, name.encode(), None, 0)
        out = np.zeros(max(int(n), 1), np.float32)
        lib().rbq_profile_stage_samples(self._h, name.encode(), out.ctypes.data, int(n))
        return out[:int(n)]

    def set_option(self, name, value):
        _check(lib().rbq_debug_set_option(self._h, name.encode(), int(value)))

    def stage_resources(self, nq, top_k, nprobe):
        """{stage: {workgroups, threads, vgprs, lds_bytes, scratch_bytes}} of the kernels a call of this shape launches (nothing runs)"""
        out = np.zeros((4, 6), np.uint32)
        _check(lib().rbq_debug_stage_resources(self._h, nq, top_k, nprobe, out.ctypes.data))
        return {s: {"workgroups": int(o[0]), "threads": int(o[1]), "vgprs": int(o[2]), "lds_bytes": int(o[3]), "scratch_bytes": int(o[4])}
                for s, o in zip(("prep", "rank", "select", "scan"), out)}

    def head_exact_stats(self):
        """(queries whose probe selection ran the exact head evaluation, guard trips — must be 0)"""
        return int(lib().rbq_debug_head_exact_evaluations(self._h)), int(lib().rbq_debug_head_exact_guard_trips(self._h))

    def tie_log_stats(self):
        """k_scan's tie log: {replays, entries, heap_ops, overflows} since the index was created"""
        out = np.zeros(4, np.uint64)
        lib().rbq_debug_tie_log_stats(self._h, out.ctypes.data)
        return dict(zip(("replays", "entries", "heap_ops", "overflows"), (int(v) for v in out)))

    def rank_fallbacks(self):
        return lib().rbq_debug_rank_fallbacks(self._h)

    def debug_copy_workspace(self, stream, name, out):
        """Diagnostic: copy an intermediate device buffer of `stream`'s workspace into the numpy array `out`."""
        _check(lib().rbq_debug_copy_workspace(self._h, C.c_void_p(stream), name.encode(), out.ctypes.data, out.nbytes))
        return out

    def heap_restarts(self):
        return lib().rbq_debug_heap_restarts(self._h)

    def profile_scan_bytes(self):
        return lib().rbq_profile_scan_bytes(self._h)

    def device_count(self):
        return lib().rbq_index_device_count(self._h)

    def profile_counters(self):
        """dict of the scan's traffic counters between profile_begin/end (rbq_profile_counters)."""
        out = (C.c_uint64 * 8)()
        _check(lib().rbq_profile_counters(self._h, out, 8))
        names = ("vectors_probed", "code_blocks", "meta_blocks", "stream_entries", "ex_evals", "queries")
        return {k: int(out[i]) for i, k in enumerate(names)}

    def release_stream(self, stream):
        _check(lib().rbq_release_stream(self._h, C.c_void_p(stream)))

    def set_rerank_vectors(self, vectors=None, device_ptr=None, n=0):
        """OPTIONAL extension (default off, not reference behaviour): attach raw vectors for the exact rerank.
        `vectors` = host array [n][dim], or `device_ptr` + n; None detaches."""
        if device_ptr is not None:
            _check(lib().rbq_index_set_rerank_vectors(self._h, C.c_void_p(device_ptr), int(n)))
        elif vectors is None:
            _check(lib().rbq_index_set_rerank_vectors(self._h, None, 0))
        else:
            v = np.ascontiguousarray(vectors, dtype=np.float32)
            _check(lib().rbq_index_set_rerank_vectors(self._h, v.ctypes.data, v.shape[0]))

    def close(self):
        if self._h:
            lib().rbq_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StreamBuilder:
    """Streamed GPU-side encoder (rbq_build_stream_*): `train_with_clusters` (src/ivf.rs:1025-1215) with the
    vectors delivered chunk by chunk, for data sets that do not fit in HBM at once."""

    def __init__(self, hdr_ptr, centroids, list_sizes, t_const, device=0):
        cent = np.ascontiguousarray(centroids, dtype=np.float32)
        ls = np.ascontiguousarray(list_sizes, dtype=np.uint32)
        self._b = C.c_void_p()
        _check(lib().rbq_build_stream_begin(_addr(hdr_ptr), cent.ctypes.data, ls.ctypes.data, float(t_const), int(device),
                                            C.byref(self._b)))

    def push(self, vectors, assign, first_id, count=None):
        """vectors / assign: numpy arrays (host) or integer device pointers (then `count` is required)."""
        if isinstance(vectors, np.ndarray):
            v = np.ascontiguousarray(vectors, dtype=np.float32)
            a = np.ascontiguousarray(assign, dtype=np.uint32)
            _check(lib().rbq_build_stream_push(self._b, v.ctypes.data, a.ctypes.data, int(first_id), v.shape[0]))
        else:
            _check(lib().rbq_build_stream_push(self._b, C.c_void_p(vectors), C.c_void_p(assign), int(first_id), int(count)))

    def finish(self, devices=None):
        h = C.c_void_p()
        if devices is None:
            _check(lib().rbq_build_stream_finish(self._b, 1, None, C.byref(h)))
        else:
            devices = list(devices)
            _check(lib().rbq_build_stream_finish(self._b, len(devices), (C.c_int * len(devices))(*devices), C.byref(h)))
        self._b = None
        return IvfRabitqIndex(h)

    def abort(self):
        if self._b:
            lib().rbq_build_stream_abort(self._b)
            self._b = None

    def __del__(self):
        try:
            self.abort()
        except Exception:
            pass
