"""rabitq_rs_amd — MI355X-native IVF+RaBitQ candidate-scan engine (host-side Python mirror).

Mirrors the reference's public query API for this path:
  SearchParams{top_k,nprobe}   src/ivf.rs:23-26,137-141
  SearchResult{id,score}       src/ivf.rs:143-148
  RabitqError                  src/lib.rs:39-57
  Metric                       src/lib.rs:18-37
  IvfRabitqIndex.search / search_filtered / batch_search / len / cluster_count
                               src/ivf.rs:1705-1752,1218-1230
All compute goes through the C ABI of include/rbq.h (csrc/librbq.so, hand-written HIP for
gfx950). There is no CPU fallback: if the HIP library is missing or no GPU is present the
calls raise.
"""
from dataclasses import dataclass

import numpy as np

from . import _abi
from ._abi import (METRIC_IP, METRIC_L2, ROTATOR_FHT_KAC, ROTATOR_MATRIX, RBQ_OK)

ROTATOR_NONE = 2


class Metric:
    L2 = METRIC_L2
    InnerProduct = METRIC_IP


class RotatorType:
    MatrixRotator = ROTATOR_MATRIX
    FhtKacRotator = ROTATOR_FHT_KAC
    NoRotation = ROTATOR_NONE  # MSTG posting lists (quantised in the raw space)


_ERR_NAMES = {
    _abi.RBQ_DIMENSION_MISMATCH: "DimensionMismatch",
    _abi.RBQ_INVALID_CONFIG: "InvalidConfig",
    _abi.RBQ_EMPTY_INDEX: "EmptyIndex",
    _abi.RBQ_IO: "Io",
    _abi.RBQ_INVALID_PERSISTENCE: "InvalidPersistence",
    _abi.RBQ_DEVICE: "Device",
}


class RabitqError(Exception):
    """`RabitqError` (src/lib.rs:39-57) + Device for HIP failures."""

    def __init__(self, code, detail=""):
        self.code = code
        self.kind = _ERR_NAMES.get(code, f"Unknown({code})")
        self.detail = detail
        super().__init__(f"{self.kind}: {detail}" if detail else self.kind)


@dataclass(frozen=True)
class SearchParams:
    top_k: int
    nprobe: int


@dataclass(frozen=True)
class SearchResult:
    id: int
    score: float


from .index import IvfRabitqIndex, StreamBuilder  # noqa: E402
from . import builder  # noqa: E402,F401

__all__ = ["Metric", "RotatorType", "RabitqError", "SearchParams", "SearchResult", "IvfRabitqIndex",
           "StreamBuilder", "builder"]
