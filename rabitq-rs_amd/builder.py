"""CPU index builder (train-time harness) — ctypes binding of csrc/host/rbq_build.cpp.

Mirrors `IvfRabitqIndex::train_with_clusters` (reference src/ivf.rs:1025-1103). Training
stays on the CPU; the result is the reference's `ClusterData` byte layout, ready for
rbq_index_create (GPU) or for serialisation as RBQ1 v3."""
import ctypes as C
import os

import numpy as np

from ._abi import Header, ListView, RBQ_OK

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("RBQ_BUILD_LIB") or os.path.join(_HERE, "csrc", "librbq_build.so")  # (override: the sanitizer build)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        L.rbq_build_train_with_clusters.restype = C.c_int
        L.rbq_build_train_with_clusters.argtypes = [
            C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
            C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]
        L.rbq_built_header.restype = C.POINTER(Header)
        L.rbq_built_header.argtypes = [C.c_void_p]
        L.rbq_built_lists.restype = C.POINTER(ListView)
        L.rbq_built_lists.argtypes = [C.c_void_p]
        L.rbq_built_t_const.restype = C.c_float
        L.rbq_built_t_const.argtypes = [C.c_void_p]
        L.rbq_built_list_recon.restype = C.c_int
        L.rbq_built_list_recon.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float))]
        L.rbq_built_free.argtypes = [C.c_void_p]
        L.rbq_built_save_rbq1.restype = C.c_int
        L.rbq_built_save_rbq1.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)]
        L.rbq_build_free_bytes.argtypes = [C.POINTER(C.c_uint8)]
        L.rbq_build_crc32.restype = C.c_uint32
        L.rbq_build_crc32.argtypes = [C.c_void_p, C.c_uint64]
        L.rbq_build_kmeans.restype = C.c_int
        L.rbq_build_kmeans.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.c_uint64,
                                       C.c_void_p, C.c_void_p]
        for name in ("rbq_build_pack_binary_code", "rbq_build_pack_ex_code_1bit",
                     "rbq_build_pack_ex_code_2bit", "rbq_build_pack_ex_code_6bit"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.rbq_build_pack_codes.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.rbq_build_rotate.argtypes = [C.POINTER(Header), C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class BuiltIndex:
    """Host-resident index in the reference's ClusterData layout."""

    def __init__(self, handle):
        self._h = handle
        self.hdr_ptr = lib().rbq_built_header(handle)
        self.lists_ptr = lib().rbq_built_lists(handle)

    @property
    def header(self):
        return self.hdr_ptr.contents

    @property
    def hdr(self):
        return self.hdr_ptr.contents

    @property
    def t_const(self):
        """Constant rescale factor of the faster config (0.0 when it was not used)."""
        return float(lib().rbq_built_t_const(self._h))

    @property
    def dim(self):
        return self.header.dim

    @property
    def padded_dim(self):
        return self.header.padded_dim

    @property
    def n_lists(self):
        return self.header.n_lists

    def __len__(self):
        return self.header.n_vectors

    def list_sizes(self):
        return np.array([self.lists_ptr[i].n for i in range(self.n_lists)], dtype=np.int64)

    def list_ids(self, c):
        lv = self.lists_ptr[c]
        return np.ctypeslib.as_array(lv.ids, shape=(lv.n,)).copy() if lv.n else np.zeros(0, np.uint64)

    def centroid(self, c):
        return np.ctypeslib.as_array(self.lists_ptr[c].centroid, shape=(self.padded_dim,)).copy()

    def list_arrays(self, c):
        """Every array of ClusterData c (src/ivf.rs:205-242) as numpy copies: centroid, ids, batch_data, ex_codes [n][D*ex/8],
        f_add_ex, f_rescale_ex, delta, vl."""
        lv = self.lists_ptr[c]
        n, D, ex = int(lv.n), int(self.padded_dim), int(self.header.ex_bits)
        exb = D * ex // 8
        arr = lambda p, shape, dt: (np.ctypeslib.as_array(p, shape=shape).copy() if shape[0] else np.zeros(shape, dt))  # noqa: E731
        d, v = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        assert lib().rbq_built_list_recon(self._h, c, C.byref(d), C.byref(v)) == RBQ_OK
        return {"centroid": self.centroid(c), "ids": self.list_ids(c),
                "batch_data": arr(lv.batch_data, (int(lv.batch_len),), np.uint8),
                "ex_codes": arr(lv.ex_codes, (n, exb), np.uint8) if exb and n else np.zeros((n, exb), np.uint8),
                "f_add_ex": arr(lv.f_add_ex, (n,), np.float32), "f_rescale_ex": arr(lv.f_rescale_ex, (n,), np.float32),
                "delta": arr(d, (n,), np.float32), "vl": arr(v, (n,), np.float32)}

    def rotator_blob(self):
        h = self.header
        return bytes(np.ctypeslib.as_array(h.rotator_blob, shape=(int(h.rotator_len),))) if h.rotator_len else b""

    def rotate(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(self.padded_dim, np.float32)
        lib().rbq_build_rotate(self.hdr_ptr, x.ctypes.data, out.ctypes.data)
        return out

    def save_rbq1(self):
        p = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        rc = lib().rbq_built_save_rbq1(self._h, C.byref(p), C.byref(n))
        assert rc == RBQ_OK
        data = bytes(C.cast(p, C.POINTER(C.c_uint8 * n.value)).contents)
        lib().rbq_build_free_bytes(p)
        return data

    def close(self):
        if self._h:
            lib().rbq_built_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def train_with_clusters(data, centroids, assignments, total_bits, metric, rotator_type, seed,
                        use_faster_config):
    data = np.ascontiguousarray(data, dtype=np.float32)
    centroids = np.ascontiguousarray(centroids, dtype=np.float32)
    assignments = np.ascontiguousarray(assignments, dtype=np.uint32)
    n, dim = data.shape
    h = C.c_void_p()
    rc = lib().rbq_build_train_with_clusters(
        data.ctypes.data, n, dim, centroids.ctypes.data, centroids.shape[0], assignments.ctypes.data,
        total_bits, metric, rotator_type, seed, int(use_faster_config), C.byref(h))
    if rc != RBQ_OK:
        from . import RabitqError
        raise RabitqError(rc, "train_with_clusters rejected its configuration")
    return BuiltIndex(h)


def kmeans(data, k, iters=10, seed=0):
    data = np.ascontiguousarray(data, dtype=np.float32)
    n, dim = data.shape
    cent = np.empty((k, dim), np.float32)
    assign = np.empty(n, np.uint32)
    rc = lib().rbq_build_kmeans(data.ctypes.data, n, dim, k, iters, seed, cent.ctypes.data, assign.ctypes.data)
    assert rc == RBQ_OK
    return cent, assign


def train(data, nlist, total_bits, metric, rotator_type, seed, use_faster_config, kmeans_iters=10):
    """`IvfRabitqIndex::train` (src/ivf.rs:950-1021) with a plain Lloyd k-means harness."""
    cent, assign = kmeans(data, nlist, kmeans_iters, seed ^ 0x5A5A5A5A5A5A5A5A)
    return train_with_clusters(data, cent, assign, total_bits, metric, rotator_type, seed, use_faster_config)
