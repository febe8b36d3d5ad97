"""Generates tests/golden/oracle_vectors.npz: seeded inputs and what THIS repository's CPU oracle / CPU builder
return for them.  They pin the oracle (and through it the GPU path) against regressions of this code base; they
are NOT outputs of the reference implementation, which cannot be built here.
Run from the repository root:  python tests/golden/make_oracle_vectors.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import build_index, make_dataset  # noqa: E402

CASES = [  # name, n, dim, nlist, bits, metric, rotator, nq, top_k, nprobe
    ("l2_7bit_d64", 1500, 64, 12, 7, 0, 1, 16, 10, 6),
    ("ip_3bit_d100", 1200, 100, 10, 3, 1, 1, 12, 5, 4),
    ("l2_1bit_matrix_d48", 900, 48, 8, 1, 0, 0, 8, 10, 8),
]


def run_case(n, dim, nlist, bits, metric, rot, nq, top_k, nprobe, seed=2026):
    data, built = build_index(n=n, dim=dim, nlist=nlist, total_bits=bits, metric=metric, rotator=rot, seed=seed,
                              normalize=(metric == 1))
    q = make_dataset(nq, dim, max(nlist // 4, 1), seed + 1, normalize=(metric == 1))
    rc, ids, sc, cnt, diag = oracle.search_batch(built, q, top_k, nprobe, want_diag=True)
    assert rc == 0
    rq0 = oracle.rotate(built, q[0])
    lut, delta, sum_vl = oracle.query_lut(rq0)
    return built, q, dict(ids=ids, scores=sc, counts=cnt, diag=diag, rot0=rq0, lut0=lut,
                          lutc0=np.array([delta, sum_vl], np.float32))


def main():
    out = {}
    for name, *cfg in CASES:
        _, _, r = run_case(*cfg)
        for k, v in r.items():
            out[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
