"""GPU parity tests proper: the HIP path (through the C ABI, include/rbq.h) against the CPU oracle on the
same seeded inputs.  Bar: ids identical, counts identical, scores within 1e-4 relative (BASELINE.json
north_star); the diag counters (SearchDiagnostics, reference src/ivf.rs:150-155) are compared exactly too.
Run with `pytest -m gpu` on an MI355X."""
import numpy as np
import pytest

import oracle
import rabitq_rs_amd as rq
from conftest import build_index, make_dataset

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north_star tolerance for f32 scores


def _compare(built, idx, queries, top_k, nprobe, filter_words=None, filter_nbits=0):
    rc, oids, osc, ocnt, odiag = oracle.search_batch(built, queries, top_k, nprobe, filter_words, filter_nbits,
                                                     want_diag=True)
    assert rc == 0
    ids, sc, cnt, diag = idx.batch_search_raw(queries, rq.SearchParams(top_k, nprobe), filter_words, filter_nbits,
                                              want_diag=True)
    assert np.array_equal(cnt, ocnt), f"counts differ: {np.nonzero(cnt != ocnt)[0][:10]}"
    bad = np.nonzero((ids != oids).any(axis=1))[0]
    assert bad.size == 0, f"ids differ for queries {bad[:10]}: gpu={ids[bad[0]]} oracle={oids[bad[0]]}"
    for q in range(ids.shape[0]):
        c = int(cnt[q])
        np.testing.assert_allclose(sc[q, :c], osc[q, :c], rtol=RTOL, atol=0)
        assert np.isnan(sc[q, c:]).all() and (ids[q, c:] == np.iinfo(np.uint64).max).all()
    assert np.array_equal(diag, odiag), "SearchDiagnostics counters differ"
    # the same call without diagnostics: k_scan then merges whole refine batches data-parallel for 64 <= top_k <= 256
    # (RankRun, scan.hpp) and keeps the block-level bound under a filter — results must not change by a bit
    ids2, sc2, cnt2, _ = idx.batch_search_raw(queries, rq.SearchParams(top_k, nprobe), filter_words, filter_nbits,
                                              want_diag=False)
    assert np.array_equal(cnt2, cnt) and np.array_equal(ids2, ids), "results differ without diagnostics"
    assert np.array_equal(sc2.view(np.uint32), sc.view(np.uint32)), "scores differ without diagnostics"
    return ids, sc, cnt


CASES = [
    # n, dim, nlist, bits, metric, rotator, nq, top_k, nprobe, uniform
    pytest.param(10000, 128, 256, 7, 0, 1, 64, 10, 32, True, id="cfg1_10k_d128_7bit_L2"),
    pytest.param(6000, 960, 48, 7, 0, 1, 48, 10, 12, False, id="gist_shape_d960_7bit_L2"),
    pytest.param(6000, 960, 48, 3, 1, 1, 48, 10, 16, False, id="gist_shape_d960_3bit_IP"),
    pytest.param(5000, 768, 40, 7, 0, 1, 32, 10, 10, False, id="d768_7bit_L2"),
    pytest.param(4000, 128, 32, 1, 0, 1, 32, 10, 8, False, id="d128_1bit_L2"),
    pytest.param(4000, 128, 32, 1, 1, 1, 32, 10, 8, False, id="d128_1bit_IP"),
    pytest.param(4000, 100, 32, 7, 0, 1, 32, 10, 8, False, id="d100_pad128_kac_7bit_L2"),
    pytest.param(3000, 200, 24, 3, 0, 1, 32, 5, 24, False, id="d200_pad256_3bit_all_lists"),
    pytest.param(3000, 64, 24, 7, 1, 0, 32, 10, 6, False, id="matrix_rotator_d64_7bit_IP"),
    pytest.param(3000, 48, 24, 3, 0, 0, 32, 10, 6, False, id="matrix_rotator_d48_codepad_3bit_L2"),
    pytest.param(5000, 320, 40, 7, 0, 1, 16, 100, 20, False, id="d320_top100"),
    pytest.param(2000, 64, 16, 7, 0, 1, 16, 1, 4, False, id="top1"),
    # FHT-Kac transforms shorter than a wavefront (dim < 64: trunc = 16 / 32, padded to 64) take the generic LDS
    # butterflies in k_prep_wave; tests/diag/soak.py found them going through the 2048-point branch (seeds 1004, 1009)
    pytest.param(600, 16, 13, 1, 0, 1, 30, 5, 6, False, id="kac_d16_trunc16_1bit_L2"),
    pytest.param(1500, 24, 12, 7, 1, 1, 20, 10, 5, False, id="kac_d24_trunc16_7bit_IP"),
    pytest.param(1500, 40, 12, 3, 0, 1, 20, 10, 12, False, id="kac_d40_trunc32_3bit_L2"),
]


@pytest.mark.parametrize("n,dim,nlist,bits,metric,rot,nq,top_k,nprobe,uniform", CASES)
def test_search_matches_oracle(n, dim, nlist, bits, metric, rot, nq, top_k, nprobe, uniform):
    data, built = build_index(n=n, dim=dim, nlist=nlist, total_bits=bits, metric=metric, rotator=rot,
                              uniform=uniform, normalize=(metric == 1), seed=1000 + dim + bits)
    idx = rq.IvfRabitqIndex.from_built(built)
    assert len(idx) == n and idx.cluster_count() == nlist and idx.dim == dim
    queries = make_dataset(nq, dim, max(nlist // 4, 1), 4242, normalize=(metric == 1), uniform=uniform)
    _compare(built, idx, queries, top_k, nprobe)
    # queries taken from the data itself (near-zero residuals, exact hits)
    _compare(built, idx, data[:8], top_k, nprobe)
    idx.close()


def test_filter_without_diag_uses_block_bound():
    """Filtered search without diagnostics keeps the block-level lower bound enabled (it is switched off
    only when per-candidate filter tests are needed for exact diag counters)."""
    data, built = build_index(n=6000, dim=128, nlist=48, total_bits=7)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(32, 128, 12, 5)
    allowed = np.arange(0, 6000, 2)
    nbits = 6000
    words = np.zeros((nbits + 31) // 32, np.uint32)
    np.bitwise_or.at(words, allowed >> 5, (np.uint32(1) << (allowed & 31).astype(np.uint32)))
    rc, oids, osc, ocnt, _ = oracle.search_batch(built, q, 10, 24, words, nbits)
    ids, sc, cnt, _ = idx.batch_search_raw(q, rq.SearchParams(10, 24), words, nbits, want_diag=False)
    assert np.array_equal(ids, oids) and np.array_equal(cnt, ocnt)
    np.testing.assert_allclose(sc, osc, rtol=RTOL)
    idx.close()


def test_duplicate_vectors_exact_ties():
    """Identical vectors produce identical codes, factors and distances: the heap's tie handling (Rust std
    BinaryHeap sift order, restated in oracle and kernel) must agree element for element."""
    base = make_dataset(700, 64, 4, 21)
    data = np.concatenate([base, base, base[:300]], axis=0)
    _, built = build_index(nlist=12, total_bits=7, data=data, dim=64)
    idx = rq.IvfRabitqIndex.from_built(built)
    _compare(built, idx, base[:48], 10, 6)
    _compare(built, idx, base[:16], 3, 12)
    # the sorted-run fast path must have noticed the bit-identical distances and re-run those queries with the
    # exact BinaryHeap emulation; forcing that emulation from the start gives the same answer
    assert idx.heap_restarts() > 0
    idx.set_option("exact_heap", 1)
    before = idx.heap_restarts()
    _compare(built, idx, base[:48], 10, 6)
    assert idx.heap_restarts() == before
    idx.close()


def test_rank_kernels_agree():
    """Split-bf16 GEMM shortlist, f32 GEMM shortlist and the exact all-pairs ranking select the same probes
    (ids and diagnostics equal the oracle's), and the shortlist never needs its fallback on ordinary data."""
    data, built = build_index(n=20000, dim=256, nlist=256, total_bits=7, seed=17)
    q = make_dataset(200, 256, 32, 18)
    idx = rq.IvfRabitqIndex.from_built(built)
    _compare(built, idx, q, 10, 32)
    assert idx.rank_fallbacks() == 0
    idx.set_option("f32_rank", 1)
    _compare(built, idx, q, 10, 32)
    assert idx.rank_fallbacks() == 0
    idx.set_option("exact_rank", 1)
    _compare(built, idx, q, 10, 32)
    idx.close()


def test_u16_accumulator_wrap_dim1536():
    """D > 1028: the u8-LUT sum can exceed 65535 and wraps exactly like the reference's u16 lanes
    (src/simd.rs:1016-1110); the block bound must switch itself off (amax > 65535)."""
    data, built = build_index(n=1500, dim=1536, nlist=8, total_bits=7, seed=5)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(16, 1536, 2, 6)
    _compare(built, idx, q, 10, 4)
    idx.close()


def test_rank_fallback_path_matches(monkeypatch):
    """The all-lists canonical fallback of the MFMA shortlist selector returns the same probes."""
    data, built = build_index(n=5000, dim=128, nlist=64, total_bits=7)
    q = make_dataset(64, 128, 16, 9)
    monkeypatch.setenv("RBQ_FORCE_RANK_FALLBACK", "1")
    idx = rq.IvfRabitqIndex.from_built(built)
    _compare(built, idx, q, 10, 16)
    assert idx.rank_fallbacks() >= 64
    idx.close()
    monkeypatch.delenv("RBQ_FORCE_RANK_FALLBACK")
    monkeypatch.setenv("RBQ_EXACT_RANK", "1")
    idx = rq.IvfRabitqIndex.from_built(built)
    _compare(built, idx, q, 10, 16)
    idx.close()


def test_near_duplicate_centroids_overflow_shortlist():
    """Many (near-)equal centroid scores overflow the shortlist window and must take the exact fallback."""
    rng = np.random.default_rng(3)
    base = rng.standard_normal((40, 64)).astype(np.float32)
    cent = np.repeat(base, 8, axis=0) + 1e-6 * rng.standard_normal((320, 64)).astype(np.float32)
    data = cent[rng.integers(0, 320, 6000)] + 0.05 * rng.standard_normal((6000, 64)).astype(np.float32)
    assign = np.argmin(((data[:, None, :] - cent[None, :, :]) ** 2).sum(-1), axis=1).astype(np.uint32)
    built = rq.builder.train_with_clusters(data, cent, assign, 7, 0, 1, 11, True)
    idx = rq.IvfRabitqIndex.from_built(built)
    _compare(built, idx, data[:64], 10, 20)
    idx.close()


def test_host_chunking_large_nq():
    """rbq_search_batch processes 16384 queries per device pass: cross a chunk boundary."""
    data, built = build_index(n=1200, dim=64, nlist=8, total_bits=3)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(16384 + 300, 64, 2, 17)
    ids, sc, cnt, _ = idx.batch_search_raw(q, rq.SearchParams(5, 3))
    rc, oids, osc, ocnt, _ = oracle.search_batch(built, q, 5, 3)
    assert np.array_equal(ids, oids) and np.array_equal(cnt, ocnt)
    idx.close()


def test_nprobe_clamp_and_small_counts():
    data, built = build_index(n=300, dim=64, nlist=8, total_bits=7)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(8, 64, 2, 7)
    _compare(built, idx, q, 10, 0)       # nprobe clamped up to 1
    _compare(built, idx, q, 10, 1000)    # clamped down to nlist
    _compare(built, idx, q, 400, 8)      # top_k > n: counts < top_k, padded with UINT64_MAX / NaN
    idx.close()


def test_filtered_search_matches_oracle():
    data, built = build_index(n=5000, dim=128, nlist=32, total_bits=7)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(16, 128, 8, 11)
    rng = np.random.default_rng(5)
    allowed = rng.choice(5000, 700, replace=False)
    nbits = int(allowed.max()) + 1
    words = np.zeros((nbits + 31) // 32, np.uint32)
    np.bitwise_or.at(words, allowed >> 5, (np.uint32(1) << (allowed & 31).astype(np.uint32)))
    ids, sc, cnt = _compare(built, idx, q, 10, 16, words, nbits)
    assert set(ids[cnt[:, None] > np.arange(10)[None, :]].tolist()) <= set(allowed.tolist())
    # empty filter -> empty result (reference src/tests.rs:753-909)
    _compare(built, idx, q, 10, 16, np.zeros(1, np.uint32), 0)
    idx.close()


def test_rbq1_round_trip_identical_results():
    """save -> load_rbq1 -> search equals search on the created index (reference src/tests.rs:394-431)."""
    data, built = build_index(n=3000, dim=128, nlist=24, total_bits=7)
    blob = built.save_rbq1()
    a = rq.IvfRabitqIndex.from_built(built)
    b = rq.IvfRabitqIndex.load_from_bytes(blob)
    q = make_dataset(32, 128, 6, 3)
    ra = a.batch_search_raw(q, rq.SearchParams(10, 8))
    rb = b.batch_search_raw(q, rq.SearchParams(10, 8))
    assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[2], rb[2])
    assert np.array_equal(ra[1].view(np.uint32), rb[1].view(np.uint32))  # bit-equal scores
    _compare(built, b, q, 10, 8)
    a.close(); b.close()


def test_api_semantics_on_device():
    data, built = build_index(n=1000, dim=64, nlist=8, total_bits=7)
    idx = rq.IvfRabitqIndex.from_built(built)
    with pytest.raises(rq.RabitqError) as e:
        idx.search(np.zeros(63, np.float32), rq.SearchParams(10, 4))
    assert e.value.kind == "DimensionMismatch" and "expected 64, got 63" in e.value.detail
    assert idx.search(data[0], rq.SearchParams(0, 4)) == []  # top_k == 0 -> Ok(vec![])
    res = idx.search(data[0], rq.SearchParams(5, 8))
    assert len(res) == 5 and all(isinstance(r.id, int) for r in res)
    assert [r.score for r in res] == sorted(r.score for r in res)  # L2: ascending distance
    bq = idx.batch_query(data[:3], 5, 8)
    assert len(bq) == 3 and bq[0].shape == (5, 2) and bq[0].dtype == np.float32
    idx.close()


def test_large_batch_property_checks():
    """BASELINE-sized batch (nq=1024) through size-independent properties: sortedness, ids drawn from the
    probed lists, idempotence, and batch == one-at-a-time."""
    data, built = build_index(n=20000, dim=128, nlist=64, total_bits=7)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(1024, 128, 16, 77)
    ids, sc, cnt, _ = idx.batch_search_raw(q, rq.SearchParams(10, 16))
    ids2, sc2, cnt2, _ = idx.batch_search_raw(q, rq.SearchParams(10, 16))
    assert np.array_equal(ids, ids2) and np.array_equal(sc.view(np.uint32), sc2.view(np.uint32))
    assert (cnt == 10).all()
    assert (np.diff(sc, axis=1) >= 0).all()
    for i in (0, 511, 1023):
        one = idx.batch_search_raw(q[i], rq.SearchParams(10, 16))
        assert np.array_equal(one[0][0], ids[i])
    rc, oids, osc, ocnt, _ = oracle.search_batch(built, q, 10, 16)
    assert np.array_equal(ids, oids)
    np.testing.assert_allclose(sc, osc, rtol=RTOL)
    idx.close()


# ---- MSTG posting-list scan ------------------------------------------------------------------
def _mstg_case(metric, bits, dim=128, n=6000, nlist=48, nq=40, seed=31):
    rng = np.random.default_rng(seed)
    data = make_dataset(n, dim, 12, seed, normalize=(metric == 1))
    cent, assign = rq.builder.kmeans(data, nlist, 5, seed)
    built = rq.builder.train_with_clusters(data, cent, assign, bits, metric, rq.RotatorType.NoRotation, seed, True)
    q = make_dataset(nq, dim, 12, seed + 1, normalize=(metric == 1))
    # stand-in for MstgIndex's HNSW centroid search + dynamic pruning: nearest centroids, ragged counts
    d = ((q[:, None, :] - cent[None, :, :]) ** 2).sum(-1)
    order = np.argsort(d, axis=1).astype(np.uint32)
    max_lists = 12
    counts = rng.integers(1, max_lists + 1, nq).astype(np.uint32)
    counts[0] = 0  # a query with no selected list
    lists = order[:, :max_lists].copy()
    return built, q, lists, counts


@pytest.mark.parametrize("metric,bits", [(0, 7), (1, 7), (0, 1), (0, 3)])
def test_mstg_posting_scan_matches_oracle(metric, bits):
    built, q, lists, counts = _mstg_case(metric, bits)
    _mstg_compare(built, q, lists, counts, metric)


def _mstg_compare(built, q, lists, counts, metric, top_ks=(10, 100)):
    idx = rq.IvfRabitqIndex.from_built(built)
    for top_k in top_ks:
        rc, oids, osc, ocnt = oracle.posting_scan_batch(built, q, top_k, lists, counts)
        assert rc == 0
        ids, sc, cnt = idx.posting_scan(q, top_k, lists, counts)
        assert np.array_equal(cnt, ocnt) and cnt[0] == 0
        # the oracle's list a little past the cut: a candidate TIED with the last returned one may be inside for one
        # side and outside for the other — the reference partitions with select_nth_unstable_by on the distance alone
        # (src/mstg/index.rs:185-203), so which of them is returned is not defined there either
        ext = 64
        rc2, xids, xsc, xcnt = oracle.posting_scan_batch(built, q, top_k + ext, lists, counts)
        assert rc2 == 0
        for i in range(len(q)):
            c = int(cnt[i])
            if c == 0:
                continue
            assert np.array_equal(xsc[i, :c].view(np.uint32), osc[i, :c].view(np.uint32))  # the longer list extends the shorter
            assert np.array_equal(sc[i, :c].view(np.uint32) & 0x7fffffff if metric == 0 else sc[i, :c].view(np.uint32),
                                  osc[i, :c].view(np.uint32) & 0x7fffffff if metric == 0 else osc[i, :c].view(np.uint32))
            assert (np.diff(sc[i, :c]) >= 0).all()
            if metric == 0:
                assert (sc[i, :c] >= 0).all()  # L2 estimates clamped at 0
            # ids agree wherever the distance is unique (the reference leaves ties unordered) — unique also with
            # respect to the first candidate past the cut
            uniq = np.ones(c, bool)
            uniq[1:] &= sc[i, 1:c] != sc[i, :c - 1]
            uniq[:-1] &= sc[i, :c - 1] != sc[i, 1:c]
            xc = int(xcnt[i])
            cut_tie = xc > c and xsc[i, c] == osc[i, c - 1]
            if cut_tie:
                assert xc < top_k + ext or xsc[i, xc - 1] != osc[i, c - 1], "tie group longer than the look-ahead"
                uniq &= sc[i, :c] != osc[i, c - 1]
            assert np.array_equal(ids[i, :c][uniq], oids[i, :c][uniq])
            if not cut_tie:
                assert sorted(ids[i, :c].tolist()) == sorted(oids[i, :c].tolist())
            else:
                # below the tied distance the id sets are equal; at it, the device's ids are candidates that the
                # oracle's longer list holds at exactly that distance, and as many of them as the oracle returned
                d = osc[i, c - 1]
                lo_g, lo_o = sc[i, :c] != d, osc[i, :c] != d
                assert sorted(ids[i, :c][lo_g].tolist()) == sorted(oids[i, :c][lo_o].tolist())
                pool = set(xids[i, :xc][xsc[i, :xc] == d].tolist())
                tied = ids[i, :c][~lo_g].tolist()
                assert len(tied) == int((~lo_o).sum()) and len(set(tied)) == len(tied) and set(tied) <= pool
    idx.close()


def test_mstg_posting_scan_rejects_rotated_index():
    data, built = build_index(n=500, dim=64, nlist=4, total_bits=7)
    idx = rq.IvfRabitqIndex.from_built(built)
    with pytest.raises(rq.RabitqError) as e:
        idx.posting_scan(data[:2], 5, np.zeros((2, 2), np.uint32), np.ones(2, np.uint32))
    assert e.value.kind == "InvalidConfig"
    idx.close()


@pytest.mark.parametrize("wg_prep", [0, 1])
def test_concurrent_streams_match_oracle(wg_prep):
    """rbq_search_batch_device on several caller streams at once (bench.py's pipelining): every stream's result
    equals the oracle's.  Kernels of different streams share CUs and SIMDs here, which once exposed a
    cross-kernel corruption (see rank_mfma.hpp) that no single-stream test can see; wg_prep=1 forces the
    workgroup-per-query k_prep, the kernel it had been seen in, beside the x16 bf16 MFMA GEMM."""
    import torch
    dev = torch.device("cuda", 0)
    data, built = build_index(n=60000, dim=960, nlist=256, total_bits=7, seed=23)
    idx = rq.IvfRabitqIndex.from_built(built)
    idx.set_option("wg_prep", wg_prep)
    nq, top_k, nprobe, ns = 512, 10, 32, 3
    q = make_dataset(nq, 960, 64, 24)
    rc, oids, osc, ocnt, _ = oracle.search_batch(built, q, top_k, nprobe)
    assert rc == 0
    qd = torch.from_numpy(q).to(dev)
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(ns - 1)]
    d_ids = [torch.zeros(nq, top_k, dtype=torch.int64, device=dev) for _ in range(ns)]
    d_sc = [torch.zeros(nq, top_k, dtype=torch.float32, device=dev) for _ in range(ns)]
    d_cnt = [torch.zeros(nq, dtype=torch.int32, device=dev) for _ in range(ns)]
    for rep in range(12):
        for i in range(ns):
            idx.search_batch_device(qd.data_ptr(), nq, 960, top_k, nprobe, d_ids[i].data_ptr(), d_sc[i].data_ptr(),
                                    d_cnt[i].data_ptr(), stream=streams[i].cuda_stream)
        torch.cuda.synchronize(dev)
        for i in range(ns):
            got = d_ids[i].cpu().numpy().view(np.uint64)
            bad = np.nonzero((got != oids).any(axis=1))[0]
            assert bad.size == 0, f"rep {rep} stream {i}: ids differ for queries {bad[:10]}"
            assert np.array_equal(d_cnt[i].cpu().numpy().view(np.uint32), ocnt)
            d_ids[i].zero_()
    idx.close()


@pytest.mark.parametrize("nlist", [5000, 17000])
def test_many_lists_select_row_modes(nlist):
    """nlist > 4096 keeps the approximate score row in LDS, nlist > 16384 re-reads it from global memory
    (k_select_mfma<1>/<0>); both must select the reference's probes."""
    import torch
    dim, n = 64, 3 * nlist
    data = make_dataset(n, dim, 64, 31)
    rng = np.random.default_rng(32)
    cent = data[rng.choice(n, nlist, replace=False)].copy()
    x, c = torch.from_numpy(data).cuda(), torch.from_numpy(cent).cuda()
    assign = torch.cdist(x, c).argmin(dim=1).cpu().numpy().astype(np.uint32)
    built = rq.builder.train_with_clusters(data, cent, assign, 7, 0, 1, 33, True)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(48, dim, 64, 34)
    _compare(built, idx, q, 10, 40)
    assert idx.rank_fallbacks() == 0
    idx.close()


ENC_CASES = [
    # n, dim, nlist, bits, metric, rotator
    pytest.param(6000, 960, 40, 7, 0, 1, id="enc_d960_7bit_L2"),
    pytest.param(5000, 960, 40, 3, 1, 1, id="enc_d960_3bit_IP"),
    pytest.param(4000, 100, 24, 7, 0, 1, id="enc_d100_pad128_7bit_L2"),
    pytest.param(4000, 128, 32, 1, 0, 1, id="enc_d128_1bit_L2"),
    pytest.param(3000, 48, 24, 3, 0, 0, id="enc_matrix_d48_3bit_L2"),
    pytest.param(3000, 64, 20, 7, 1, 0, id="enc_matrix_d64_7bit_IP"),
]


@pytest.mark.parametrize("n,dim,nlist,bits,metric,rot", ENC_CASES)
def test_device_encoder_matches_cpu_builder(n, dim, nlist, bits, metric, rot):
    """rbq_index_build_device (GPU quantize_with_centroid, faster config) produces, array for array, the index
    that rbq_index_create builds from the CPU builder's ClusterData — codes, ex codes, factors, ids, block
    summaries — and therefore the same search results."""
    import torch
    data = make_dataset(n, dim, max(nlist // 4, 1), 41, normalize=(metric == 1))
    cent, assign = rq.builder.kmeans(data, nlist, 5, 42)
    built = rq.builder.train_with_clusters(data, cent, assign, bits, metric, rot, 43, True)
    ref = rq.IvfRabitqIndex.from_built(built)
    xd = torch.from_numpy(data).cuda()
    ad = torch.from_numpy(assign.astype(np.int32)).cuda()
    enc = rq.IvfRabitqIndex.build_on_device(built.hdr_ptr, cent, xd.data_ptr(), ad.data_ptr(), n, built.t_const)
    hdr = built.hdr
    D, ex = hdr.padded_dim, hdr.ex_bits
    Dc = (D + 63) // 64 * 64
    gb0 = ref.debug_copy_index("list_gb0", np.empty(nlist, np.uint32))
    ln = ref.debug_copy_index("list_n", np.empty(nlist, np.uint32))
    nblocks = int(((ln + 31) // 32).sum())
    cpu_u = 128 // ex if ex else 1
    w4 = ((D // 16 + cpu_u - 1) // cpu_u) if ex else 0
    sizes = {"list_gb0": nlist * 4, "list_n": nlist * 4, "centroids": nlist * D * 4, "blocks": nblocks * (Dc * 4 + 384),
             "ids": nblocks * 32 * 8, "bsum": nblocks * 32}
    if ex:
        sizes.update({"ex": nblocks * 32 * w4 * 256, "fadd_ex": nblocks * 32 * 4, "fres_ex": nblocks * 32 * 4})
    for name, nbytes in sizes.items():
        a = ref.debug_copy_index(name, np.empty(nbytes, np.uint8))
        b = enc.debug_copy_index(name, np.empty(nbytes, np.uint8))
        bad = np.nonzero(a != b)[0]
        assert bad.size == 0, f"{name}: {bad.size} bytes differ, first at {bad[:5]}"
    q = make_dataset(32, dim, max(nlist // 4, 1), 44, normalize=(metric == 1))
    _compare(built, enc, q, 10, min(8, nlist))
    ref.close(); enc.close()


def test_device_encoder_rejects_bad_input():
    """rbq_index_build_device: an assignment outside [0, n_lists) and a missing constant rescale factor are
    configuration errors (RabitqError::InvalidConfig), reported without touching the GPU index."""
    import torch
    data, built = build_index(n=2000, dim=64, nlist=8, total_bits=7, seed=51)
    cent = np.zeros((8, 64), np.float32)
    xd = torch.from_numpy(data).cuda()
    bad = torch.full((2000,), 8, dtype=torch.int32).cuda()
    with pytest.raises(rq.RabitqError) as e:
        rq.IvfRabitqIndex.build_on_device(built.hdr_ptr, cent, xd.data_ptr(), bad.data_ptr(), 2000, built.t_const)
    assert e.value.kind == "InvalidConfig"
    ok = torch.zeros(2000, dtype=torch.int32).cuda()
    with pytest.raises(rq.RabitqError) as e:
        rq.IvfRabitqIndex.build_on_device(built.hdr_ptr, cent, xd.data_ptr(), ok.data_ptr(), 2000, 0.0)
    assert e.value.kind == "InvalidConfig"


def _random_case(seed):
    rng = np.random.default_rng(seed)
    rot = int(rng.integers(0, 2))
    dim = int(rng.choice([16, 24, 40, 64, 96, 100, 128, 200, 256, 384, 512, 768])) if rot == 1 else int(rng.choice([16, 32, 48, 64, 96, 128]))
    bits = int(rng.choice([1, 3, 7]))
    metric = int(rng.integers(0, 2))
    nlist = int(rng.integers(2, 60))
    n = int(rng.integers(max(nlist, 40), 4000))
    nq = int(rng.integers(1, 40))
    top_k = int(rng.choice([1, 2, 5, 10, 17, 64, 100]))  # (63/64/128/129/255/256 boundaries: test_register_sorted_run_*)
    nprobe = int(rng.integers(1, nlist + 3))
    return n, dim, nlist, bits, metric, rot, nq, top_k, nprobe


@pytest.mark.parametrize("seed", list(range(100, 220)) + [1004, 1009])
def test_random_configurations_match_oracle(seed):
    """Seeded random shapes (dimension padding, tiny lists, nprobe > nlist, top_k beyond the candidate count,
    top_k >= 64 = LDS heap, both metrics, both rotators): ids, counts, scores and diagnostics equal the oracle's."""
    n, dim, nlist, bits, metric, rot, nq, top_k, nprobe = _random_case(seed)
    data, built = build_index(n=n, dim=dim, nlist=nlist, total_bits=bits, metric=metric, rotator=rot, seed=seed,
                              normalize=(metric == 1))
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(nq, dim, max(nlist // 4, 1), seed + 1000, normalize=(metric == 1))
    _compare(built, idx, q, top_k, nprobe)
    idx.close()


def test_gpu_matches_committed_vectors():
    """REGRESSION guard, not parity evidence: tests/golden/oracle_vectors.npz holds outputs of THIS repository's oracle for seeded
    cases (the reference's tests hold no end-to-end search outputs).  The GPU path must keep reproducing them whatever oracle
    library is on the box."""
    import importlib.util
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_oracle_vectors", os.path.join(g, "make_oracle_vectors.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    want = np.load(os.path.join(g, "oracle_vectors.npz"))
    for name, n, dim, nlist, bits, metric, rot, nq, top_k, nprobe in gen.CASES:
        data, built = build_index(n=n, dim=dim, nlist=nlist, total_bits=bits, metric=metric, rotator=rot, seed=2026,
                                  normalize=(metric == 1))
        q = make_dataset(nq, dim, max(nlist // 4, 1), 2027, normalize=(metric == 1))
        idx = rq.IvfRabitqIndex.from_built(built)
        ids, sc, cnt, diag = idx.batch_search_raw(q, rq.SearchParams(top_k, nprobe), want_diag=True)
        assert np.array_equal(ids, want[f"{name}/ids"]), name
        assert np.array_equal(cnt, want[f"{name}/counts"]), name
        assert np.array_equal(diag, want[f"{name}/diag"]), name
        w = want[f"{name}/scores"]
        ok = np.isfinite(w)
        np.testing.assert_allclose(sc[ok], w[ok], rtol=RTOL, atol=0)
        idx.close()


@pytest.mark.parametrize("nprobe", [4096, 5000, 8192])
def test_nprobe_up_to_the_limit(nprobe):
    """Thousands of probes per query (the reference clamps nprobe to n_lists and nothing else): the shortlist window of
    k_select_mfma grows to 16 384 keys; above 8192 the exact all-pairs ranking with its key window in global memory takes
    over (nprobe = 8193 and nprobe = n_lists = 9000)."""
    import torch
    nlist, dim = 9000, 64
    n = 3 * nlist
    data = make_dataset(n, dim, 64, 65)
    rng = np.random.default_rng(66)
    cent = data[rng.choice(n, nlist, replace=False)].copy()
    x, c = torch.from_numpy(data).cuda(), torch.from_numpy(cent).cuda()
    assign = torch.cdist(x, c).argmin(dim=1).cpu().numpy().astype(np.uint32)
    built = rq.builder.train_with_clusters(data, cent, assign, 7, 0, 1, 67, True)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(6, dim, 64, 68)
    _compare(built, idx, q, 10, nprobe)
    if nprobe == 8192:
        _compare(built, idx, q[:3], 10, 8193)
        _compare(built, idx, q[:3], 10, 20000)  # clamped to n_lists = 9000 (src/ivf.rs:1791)
    idx.close()


@pytest.mark.parametrize("nprobe", [300, 600, 800])
def test_large_nprobe_select_paths(nprobe):
    """nprobe > 256: more than one probe per selector thread, 1024/2048-entry shortlist windows (bitonic sort
    instead of the rank sort), staged probe geometry of several KB; nprobe == nlist takes the all-lists branch."""
    import torch
    nlist, dim = 800, 64
    n = 12 * nlist
    data = make_dataset(n, dim, 64, 61)
    rng = np.random.default_rng(62)
    cent = data[rng.choice(n, nlist, replace=False)].copy()
    x, c = torch.from_numpy(data).cuda(), torch.from_numpy(cent).cuda()
    assign = torch.cdist(x, c).argmin(dim=1).cpu().numpy().astype(np.uint32)
    built = rq.builder.train_with_clusters(data, cent, assign, 7, 0, 1, 63, True)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(24, dim, 64, 64)
    _compare(built, idx, q, 10, nprobe)
    assert idx.rank_fallbacks() == 0
    idx.close()


@pytest.mark.parametrize("top_k", [63, 64, 500, 4096, 16384])
def test_large_top_k(top_k):
    """top_k <= 256 keeps the top-k in the replay wave's registers (sorted run / RankRun; RegHeap after a tie), above that
    the exact BinaryHeap emulation runs on an LDS array from the first candidate (one lane); 16384 exceeds the number of
    probed candidates here (counts < top_k, NaN / u64::MAX padding)."""
    data, built = build_index(n=6000, dim=64, nlist=24, total_bits=7, seed=71)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(12, 64, 6, 72)
    _compare(built, idx, q, top_k, 8)
    idx.close()


def test_top_k_6000_with_evictions_and_limit():
    """A heap of 6000 entries in LDS that fills up and evicts (every list probed: 20 000 candidates per query); beyond the LDS
    (top_k 16385, 19999: heap in global memory, evictions; 20000 and 30000: every candidate returned) the call is served as the
    reference serves it; only top_k > 2^20 is RBQ_INVALID_CONFIG."""
    data, built = build_index(n=20000, dim=64, nlist=16, total_bits=3, seed=73)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(6, 64, 6, 74)
    ids, sc, cnt = _compare(built, idx, q, 6000, 16)
    assert (cnt == 6000).all()
    for top_k in (16385, 19999, 20000, 30000):
        ids, sc, cnt = _compare(built, idx, q[:3], top_k, 16)
        assert (cnt == min(top_k, 20000)).all()
    with pytest.raises(rq.RabitqError) as e:
        idx.batch_search_raw(q, rq.SearchParams((1 << 20) + 1, 4))
    assert e.value.kind == "InvalidConfig" and "2^20" in e.value.detail
    idx.close()


def test_empty_lists_and_single_vector_lists():
    """Lists without vectors (their probes contribute no blocks) and one-vector lists, through both builders."""
    import torch
    rng = np.random.default_rng(81)
    n, dim, nlist = 3000, 64, 40
    data = make_dataset(n, dim, 8, 82)
    cent = data[rng.choice(n, nlist, replace=False)].copy()
    assign = rng.integers(0, nlist, n).astype(np.uint32)
    assign[assign % 5 == 0] = 1          # lists 0, 5, 10, ... stay empty
    assign[assign == 7] = 1
    assign[0] = 7                         # list 7 holds exactly one vector
    built = rq.builder.train_with_clusters(data, cent, assign, 7, 0, 1, 83, True)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(20, dim, 8, 84)
    _compare(built, idx, q, 10, 40)      # every list probed, the empty ones included
    _compare(built, idx, q, 10, 9)
    xd = torch.from_numpy(data).cuda(); ad = torch.from_numpy(assign.astype(np.int32)).cuda()
    enc = rq.IvfRabitqIndex.build_on_device(built.hdr_ptr, cent, xd.data_ptr(), ad.data_ptr(), n, built.t_const)
    _compare(built, enc, q, 10, 40)
    idx.close(); enc.close()


def test_bench_two_rank_rehearsal():
    """bench.py's N > 1 control flow (device build on every rank, barriers, per-stream gather buffers, rank-0-only
    sections) with two ranks sharing this GPU over gloo — RCCL itself needs one GPU per rank and is left to the
    8-GPU run; what must not happen there is a rank-0-only collective or a crash in the N > 1 branches."""
    import json
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RBQ_BENCH_REHEARSAL="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-extras", "--nbatches", "3", "--min-seconds", "0", "--no-latency"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["pruned"]["launches"] == 4 and d["config"]["distinct_query_batches"] == 3
    assert d["roofline"]["frac"] <= 1.0 and d["roofline"]["ids_identical_to_product_configuration"]
    assert 0.0 <= d["pruned"]["block_skip_frac"] < 1.0 and d["recall_at_10"] > 0.9


def test_bench_rccl_single_rank_group():
    """The RCCL calls of bench.py's N > 1 path — init_process_group("nccl", device_id), one all_gather per bucket of batches on
    the exchange stream behind the six search streams, barrier, all_reduce, teardown — executed for real on this one-GPU box
    with a one-rank communicator (RCCL refuses two ranks on one device: that is what the gloo rehearsal above is for)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RBQ_BENCH_FORCE_DIST="1")
    env.pop("RBQ_BENCH_REHEARSAL", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "1", "--steps", "12", "--warmup", "2", "--no-extras", "--no-cpu", "--nbatches", "4", "--min-seconds", "0", "--no-latency"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["pruned"]["launches"] == 12 and d["recall_at_10"] > 0.9
    assert d["roofline"]["ids_identical_to_product_configuration"] and d["rccl_world_size"] == 1 and len(d["per_rank_queries_per_s"]) == 1


def test_library_first_then_torch_in_a_fresh_process():
    """A fresh Python process that searches through librbq.so BEFORE it ever imports torch must still find the GPU
    from torch afterwards (one HIP runtime per process: rabitq_rs_amd.index._hip_runtime_of_torch_first).  This
    pytest process cannot show it — collection imports torch first — so it runs in a child."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "assert 'torch' not in sys.modules\n"
        "import conftest, numpy as np, rabitq_rs_amd as rq\n"
        "data, built = conftest.build_index(n=600, dim=64, nlist=6, total_bits=7)\n"
        "idx = rq.IvfRabitqIndex.from_built(built)\n"
        "ids, sc, cnt = idx.batch_search_raw(data[:4], rq.SearchParams(5, 3))[:3]\n"
        "assert (cnt == 5).all()\n"
        "import torch\n"
        "x = torch.from_numpy(data).cuda()\n"
        "print('ok', torch.cuda.device_count(), float(x.sum().item()) == float(x.cpu().sum().item()))\n"
    ) % (root, os.path.join(root, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1].startswith("ok 1")
