"""ctypes mirror of include/rbq.h (plain data-contract structs + error codes)."""
import ctypes as C

RBQ_OK, RBQ_DIMENSION_MISMATCH, RBQ_INVALID_CONFIG, RBQ_EMPTY_INDEX, RBQ_IO, \
    RBQ_INVALID_PERSISTENCE, RBQ_DEVICE = range(7)
METRIC_L2, METRIC_IP = 0, 1
ROTATOR_MATRIX, ROTATOR_FHT_KAC = 0, 1
BATCH = 32


class Header(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("padded_dim", C.c_uint32),
                ("metric", C.c_uint8), ("rotator", C.c_uint8), ("ex_bits", C.c_uint8), ("reserved", C.c_uint8),
                ("n_vectors", C.c_uint64), ("n_lists", C.c_uint64),
                ("rotator_blob", C.POINTER(C.c_uint8)), ("rotator_len", C.c_uint64)]


class ListView(C.Structure):
    _fields_ = [("centroid", C.POINTER(C.c_float)), ("n", C.c_uint64), ("ids", C.POINTER(C.c_uint64)),
                ("batch_data", C.POINTER(C.c_uint8)), ("batch_len", C.c_uint64),
                ("ex_codes", C.POINTER(C.c_uint8)), ("f_add_ex", C.POINTER(C.c_float)),
                ("f_rescale_ex", C.POINTER(C.c_float))]


class Diag(C.Structure):
    _fields_ = [("estimated", C.c_uint64), ("skipped_by_lower_bound", C.c_uint64),
                ("extended_evaluations", C.c_uint64)]
