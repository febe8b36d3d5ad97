"""The reference picks its arithmetic at COMPILE time (cfg(target_feature) in src/simd.rs, .cargo/config.toml:11-14): the oracle and
the kernels restate the `target-cpu=native` build on an AVX-512 host.  The oracle also restates the other bodies behind
`ref_set_variant` (oracle/rbq_ref.c, "Numeric variants"); these tests pin those restatements and check that the variants differ
from the default only where they should (CPU only; the table for DESIGN.md comes from tools/variant_table.py)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from conftest import build_index, make_dataset


def _codes(rng, D, ex_bits):
    return rng.integers(0, 256, D * ex_bits // 8, dtype=np.uint8)


@pytest.mark.parametrize("ex_bits", [2, 6])
@pytest.mark.parametrize("D", [64, 128, 960, 1536])
def test_avx2_order_restatement_equals_real_avx2_instructions(ex_bits, D):
    """ip_packed_ex{2,6}_f32_avx2 (src/simd.rs:1722-1825): the scalar restatement of its lane order against the same instruction
    sequence on real AVX2 — magnitudes spanning 12 orders, which do distinguish summation orders."""
    L = oracle.lib()
    rng = np.random.default_rng(100 + D + ex_bits)
    for trial in range(50):
        q = (rng.standard_normal(D) * 10.0 ** rng.integers(-6, 6, D)).astype(np.float32)
        code = _codes(rng, D, ex_bits)
        a = L.ref_ip_packed_ex_avx2_order(q.ctypes.data, code.ctypes.data, D, ex_bits)
        b = L.ref_ip_packed_ex_avx2_real(q.ctypes.data, code.ctypes.data, D, ex_bits)
        assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32)


@pytest.mark.parametrize("ex_bits", [2, 6])
def test_ex_dot_variants_agree_where_arithmetic_is_exact(ex_bits):
    """Integer-valued queries: every product and partial sum is exact, so all three summation orders give the same value (the
    reference's own KATs: 1920 / 9600, src/simd.rs:3222-3258) — and on general inputs they differ in the last bits only."""
    L = oracle.lib()
    rng = np.random.default_rng(5)
    D = 960
    fns = [L.ref_ip_packed_ex_avx2_order, L.ref_ip_packed_ex_scalar_order, L.ref_ip_packed_ex_avx512]
    q = rng.integers(-8, 9, D).astype(np.float32)
    code = _codes(rng, D, ex_bits)
    vals = {float(f(q.ctypes.data, code.ctypes.data, D, ex_bits)) for f in fns}
    vals.add(float(L.ref_ex_dot(q.ctypes.data, code.ctypes.data, D, ex_bits)))
    assert len(vals) == 1
    differ = 0
    for trial in range(200):
        q = rng.standard_normal(D).astype(np.float32)
        code = _codes(rng, D, ex_bits)
        v = [np.float32(f(q.ctypes.data, code.ctypes.data, D, ex_bits)) for f in fns[:2]] + [np.float32(L.ref_ex_dot(q.ctypes.data, code.ctypes.data, D, ex_bits))]
        differ += len({x.view(np.uint32).item() for x in v}) > 1
        scale = float(np.abs(q).sum()) * ((1 << ex_bits) - 1)  # (sum of |terms| bounds the rounding of ANY order; sums cancel)
        assert abs(float(v[0]) - float(v[2])) <= 1e-5 * scale and abs(float(v[1]) - float(v[2])) <= 1e-5 * scale
    assert differ > 50  # they ARE different orders


def test_variant_switch_is_scoped_and_default_is_unchanged():
    data, built = build_index(n=3000, dim=128, nlist=24, total_bits=7, seed=77)
    q = make_dataset(64, 128, 6, 78)
    rc, ids0, sc0, cnt0, d0 = oracle.search_batch(built, q, 10, 8, want_diag=True)
    with oracle.variant("ex_avx2"):
        assert oracle.lib().ref_get_variant() == 1
    assert oracle.lib().ref_get_variant() == 0
    rc, ids1, sc1, cnt1, d1 = oracle.search_batch(built, q, 10, 8, want_diag=True)
    assert np.array_equal(ids0, ids1) and np.array_equal(sc0.view(np.uint32), sc1.view(np.uint32)) and np.array_equal(d0, d1)


@pytest.mark.parametrize("bits,metric", [(7, 0), (3, 1), (1, 0)])
def test_variants_differ_only_where_expected(bits, metric):
    """Seeded generator, every variant against the default:
    * 1-bit indexes never evaluate ex codes: the ex-dot variants change NOTHING;
    * every variant moves scores by a few ulps at most (same formula, other rounding) and ids only at near-ties."""
    data, built = build_index(n=6000, dim=192, nlist=32, total_bits=bits, metric=metric, normalize=(metric == 1), seed=300 + bits)
    q = np.concatenate([make_dataset(96, 192, 8, 301, normalize=(metric == 1)), data[:32] + np.float32(1e-3)])
    tab = oracle.variant_diff_table(built, q, 10, 12)
    for name, row in tab.items():
        # a few ulps of the distance scale (near-exact hits have scores near 0: their RELATIVE difference between two variants of the
        # reference itself exceeds north_star's 1e-4 — cancellation, reported by tools/variant_table.py)
        assert row["max_score_diff_over_scale_same_ids"] < 1e-5, (name, row)
        assert row["ids_differ_frac"] <= 0.1, (name, row)
    if bits == 1:
        for name in ("ex_avx2", "ex_scalar"):
            assert tab[name]["ids_differ_frac"] == 0.0 and tab[name]["score_bits_differ_frac"] == 0.0, (name, tab[name])
    else:
        assert tab["ex_avx2"]["score_bits_differ_frac"] > 0.0  # a different summation order does show in the last bits
    assert tab["contract"]["score_bits_differ_frac"] > 0.0
