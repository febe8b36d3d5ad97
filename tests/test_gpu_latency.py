"""GPU parity tests of the latency-first path of SMALL calls (round 5; csrc/device/latency.hpp): one launch rotates the query,
builds its constants and LUT and scores every list in the reference's summation order (src/rotation.rs:350-401, src/ivf.rs:798-878,
:1782-1835, src/math.rs:154-245).  Bar: the stage outputs equal the oracle's bit for bit, the results equal the oracle's AND the
batch path's (option latency_path = 0) bit for bit, at nq in {1, 3, 4, 8} incl. degenerate queries."""
import numpy as np
import pytest

import oracle
import rabitq_rs_amd as rq
from conftest import build_index, make_dataset
from test_gpu_parity import _compare

pytestmark = pytest.mark.gpu

CASES = [
    # n, dim, nlist, bits, metric, rotator, top_k, nprobe, uniform
    pytest.param(10000, 128, 256, 7, 0, 1, 10, 32, True, id="cfg1_10k_d128_7bit_L2"),
    pytest.param(6000, 960, 48, 7, 0, 1, 10, 12, False, id="gist_shape_d960_7bit_L2"),
    pytest.param(6000, 960, 48, 3, 1, 1, 10, 16, False, id="gist_shape_d960_3bit_IP"),
    pytest.param(5000, 768, 40, 7, 0, 1, 10, 10, False, id="d768_7bit_L2"),
    pytest.param(4000, 128, 33, 1, 1, 1, 10, 8, False, id="d128_1bit_IP_33_lists"),
    pytest.param(4000, 100, 32, 7, 0, 1, 10, 8, False, id="d100_pad128_kac_7bit_L2"),
    pytest.param(3000, 200, 24, 3, 0, 1, 5, 24, False, id="d200_pad256_3bit_all_lists"),
    pytest.param(3000, 64, 24, 7, 1, 0, 10, 6, False, id="matrix_rotator_d64_7bit_IP_batch_path_serves_it"),
    pytest.param(5000, 320, 40, 7, 0, 1, 100, 20, False, id="d320_top100"),
    pytest.param(1500, 24, 12, 7, 1, 1, 10, 5, False, id="kac_d24_trunc16_7bit_IP"),
    pytest.param(9000, 128, 1000, 7, 0, 1, 10, 64, False, id="d128_1000_lists"),
]


def _front_taken(idx, nq, top_k, nprobe):
    r = idx.stage_resources(nq, top_k, nprobe)
    return r["rank"]["workgroups"] == 0 and r["prep"]["workgroups"] > 0


@pytest.mark.parametrize("n,dim,nlist,bits,metric,rot,top_k,nprobe,uniform", CASES)
def test_latency_path_matches_oracle_and_batch_path(n, dim, nlist, bits, metric, rot, top_k, nprobe, uniform):
    data, built = build_index(n=n, dim=dim, nlist=nlist, total_bits=bits, metric=metric, rotator=rot, uniform=uniform,
                              normalize=(metric == 1), seed=5100 + dim + bits)
    idx = rq.IvfRabitqIndex.from_built(built)
    assert _front_taken(idx, 1, top_k, nprobe) == (rot == 1)  # the matrix rotator is O(D^2) per query: k_prep serves it
    assert not _front_taken(idx, 64, top_k, nprobe)           # batches keep the GEMM
    for nq in (1, 3, 4, 8):  # (8: beyond the front's window — the batch path with the workgroup-per-query preparation)
        q = make_dataset(nq, dim, max(nlist // 4, 1), 5200 + nq, normalize=(metric == 1), uniform=uniform)
        ids, sc, cnt = _compare(built, idx, q, top_k, nprobe)
        idx.set_option("latency_path", 0)
        ids0, sc0, cnt0, _ = idx.batch_search_raw(q, rq.SearchParams(top_k, nprobe))
        idx.set_option("latency_path", 1)
        assert np.array_equal(ids, ids0) and np.array_equal(cnt, cnt0) and np.array_equal(sc.view(np.uint32), sc0.view(np.uint32))
    idx.close()


STAGE_CASES = [
    pytest.param(8000, 128, 64, 7, 0, 16, id="d128_7bit_L2"),
    pytest.param(6000, 960, 48, 7, 0, 12, id="d960_7bit_L2"),
    pytest.param(6000, 960, 48, 3, 1, 16, id="d960_3bit_IP"),
    pytest.param(3000, 100, 24, 3, 1, 6, id="kac_d100_pad128_3bit_IP"),
    pytest.param(1500, 40, 12, 3, 0, 12, id="kac_d40_trunc32_3bit_L2"),
]


@pytest.mark.parametrize("n,dim,nlist,bits,metric,nprobe", STAGE_CASES)
def test_latency_front_stage_outputs(n, dim, nlist, bits, metric, nprobe):
    """Rotated query, LUT bytes, query constants and the score of EVERY list (not only the probed ones) against the oracle's stage
    functions, bit for bit; the constants the lazy selection derives its slack from (q1norm, exlo, exhi, amin, amax) against the
    batch kernel's."""
    import torch
    data, built = build_index(n=n, dim=dim, nlist=nlist, total_bits=bits, metric=metric, rotator=1, normalize=(metric == 1), seed=5300 + dim + bits)
    idx = rq.IvfRabitqIndex.from_built(built)
    nq, top_k = 4, 10
    q = make_dataset(nq, dim, max(nlist // 4, 1), 5353, normalize=(metric == 1))
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(q).to(dev)
    out = (torch.zeros(nq, top_k, dtype=torch.int64, device=dev), torch.zeros(nq, top_k, dtype=torch.float32, device=dev),
           torch.zeros(nq, dtype=torch.int32, device=dev))
    st = torch.cuda.Stream(dev)
    D = built.padded_dim
    Dc = (D + 63) // 64 * 64
    taps = {}
    for lat in (0, 1):
        idx.set_option("latency_path", lat)
        torch.cuda.synchronize(dev)
        idx.search_batch_device(qd.data_ptr(), nq, dim, top_k, nprobe, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), stream=st.cuda_stream)
        torch.cuda.synchronize(dev)
        taps[lat] = (idx.debug_copy_workspace(st.cuda_stream, "rot", np.empty((nq, D), np.float32)),
                     idx.debug_copy_workspace(st.cuda_stream, "lut", np.empty((nq, Dc * 4), np.uint8)),
                     idx.debug_copy_workspace(st.cuda_stream, "consts", np.empty((nq, 12), np.float32)),
                     idx.debug_copy_workspace(st.cuda_stream, "scores", np.empty((nq, nlist), np.float32)),
                     out[0].cpu().numpy().copy(), out[1].cpu().numpy().copy(), out[2].cpu().numpy().copy())
    rot_d, lut_d, consts_d, scores_d = taps[1][:4]
    assert np.array_equal(rot_d.view(np.uint32), taps[0][0].view(np.uint32)) and np.array_equal(lut_d, taps[0][1])
    assert np.array_equal(consts_d.view(np.uint32), taps[0][2].view(np.uint32)), "query constants differ from k_prep_wave's"
    for a, b in zip(taps[0][4:], taps[1][4:]):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    for i in range(nq):
        r = oracle.rotate(built, q[i])
        assert np.array_equal(r.view(np.uint32), rot_d[i].view(np.uint32))
        lut, delta, sum_vl = oracle.query_lut(r)
        assert np.array_equal(lut_d[i].reshape(Dc // 4, 16)[np.arange(D // 4) ^ 1], lut.reshape(D // 4, 16))
        qc = oracle.query_precompute(r, built.hdr.ex_bits)
        want = np.array([delta, sum_vl, qc.k1x_sum_q, qc.kbx_sum_q, qc.binary_scale, qc.query_norm], np.float32)
        assert np.array_equal(consts_d[i, :6].view(np.uint32), want.view(np.uint32))
        for cid in range(nlist):  # (IP: the selection overwrites the scores of the lists it scans with their distances — skip those)
            c = built.centroid(cid)
            dist = np.float32(oracle.lib().ref_l2_distance_sqr(r.ctypes.data, c.ctypes.data, D))
            dot = np.float32(oracle.lib().ref_dot(r.ctypes.data, c.ctypes.data, D))
            got = scores_d[i, cid]
            assert got == (dist if metric == 0 else dot) or (metric == 1 and got == dist), (i, cid, got, dist, dot)
    idx.close()


def test_latency_path_degenerate_queries():
    """Zero, huge, NaN and Inf queries through the latency path: whatever the batch path returns (itself compared with the oracle
    by test_degenerate_queries), bit for bit."""
    data, built = build_index(n=4000, dim=128, nlist=32, total_bits=7, seed=5400)
    idx = rq.IvfRabitqIndex.from_built(built)
    base = make_dataset(8, 128, 8, 5401)
    qs = base.copy()
    qs[0] = 0.0
    qs[1] *= 1e30
    qs[2, 5] = np.nan
    qs[3, 7] = np.inf
    qs[4, 9] = -np.inf
    qs[5] = 1e-30
    for nq in (1, 4):
        for first in range(0, 8, nq):
            q = qs[first:first + nq]
            a = idx.batch_search_raw(q, rq.SearchParams(10, 8), want_diag=True)
            idx.set_option("latency_path", 0)
            b = idx.batch_search_raw(q, rq.SearchParams(10, 8), want_diag=True)
            idx.set_option("latency_path", 1)
            for x, y in zip(a, b):
                assert np.array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8)), (nq, first)
    rc, oids, osc, ocnt, _ = oracle.search_batch(built, qs[:2], 10, 8)
    ids, sc, cnt, _ = idx.batch_search_raw(qs[:2], rq.SearchParams(10, 8))
    assert np.array_equal(ids, oids) and np.array_equal(cnt, ocnt)
    idx.close()


# ---- tie log of k_scan (round 5): a tied query replays its logged candidates through the BinaryHeap emulation ---------------------------
@pytest.mark.parametrize("top_k,dim,bits", [(10, 64, 7), (3, 64, 7), (100, 64, 7), (128, 64, 7), (200, 64, 3), (10, 960, 7), (100, 960, 7), (16, 128, 1)])
def test_tie_log_replay_matches_oracle(top_k, dim, bits):
    """Duplicated vectors: equal distances inside the top-k, i.e. results that depend on the layout of the reference's BinaryHeap
    (src/ivf.rs:904-931, :2116-2126).  k_scan (scan_wave = 0) without diagnostics settles them by replaying the tie log through the
    lane-parallel heap (top_k = 128: the LDS heap; 200: four registers per lane; 960 dimensions: the instantiations at their register
    limit; 1 bit: no refinement, the estimates are logged): ids, scores and counts equal the oracle's; the same call with the log
    switched off (the query is scanned again) and with a log too small for the query (overflow: scanned again) give the same bits."""
    base = make_dataset(900, dim, 4, 5500 + top_k)
    data = np.concatenate([base, base, base[:400]], axis=0)
    _, built = build_index(nlist=12, total_bits=bits, data=data, dim=dim)
    idx = rq.IvfRabitqIndex.from_built(built)
    idx.set_option("scan_wave", 0)
    q = np.ascontiguousarray(base[:40])
    s0 = idx.tie_log_stats()
    ids, sc, cnt = _compare(built, idx, q, top_k, 8)
    s1 = idx.tie_log_stats()
    assert s1["replays"] > s0["replays"] and s1["overflows"] == s0["overflows"], (s0, s1)
    assert s1["entries"] > s1["heap_ops"] >= s1["replays"]  # (some logged candidates are skipped or pushed-and-popped: no heap operation)
    idx.set_option("tie_log", 0)
    a = idx.batch_search_raw(q, rq.SearchParams(top_k, 8))
    s2 = idx.tie_log_stats()
    assert s2["replays"] == s1["replays"]
    idx.set_option("tie_log", 1)
    idx.set_option("tie_log_cap", 8)
    b = idx.batch_search_raw(q, rq.SearchParams(top_k, 8))
    s3 = idx.tie_log_stats()
    assert s3["overflows"] > s2["overflows"], (s2, s3)
    idx.set_option("tie_log_cap", 0)
    for x in (a, b):
        assert np.array_equal(x[0], ids) and np.array_equal(x[2], cnt) and np.array_equal(x[1].view(np.uint32), sc.view(np.uint32))
    idx.close()


def test_tie_log_seeded_sweep():
    """Random small indexes with heavy duplication, random top_k / nprobe: the tie log's answer equals the oracle's on every one."""
    rng = np.random.default_rng(5600)
    for it in range(12):
        dim = int(rng.choice([32, 64, 128, 200]))
        nb = int(rng.integers(200, 600))
        base = make_dataset(nb, dim, 3, 5601 + it)
        reps = int(rng.integers(2, 5))
        data = np.concatenate([base] * reps, axis=0)
        bits = int(rng.choice([1, 3, 7]))
        metric = int(rng.integers(0, 2))
        if metric == 1:
            data = data / np.linalg.norm(data, axis=1, keepdims=True)
        nlist = int(rng.integers(4, 14))
        _, built = build_index(nlist=nlist, total_bits=bits, data=np.ascontiguousarray(data.astype(np.float32)), dim=dim, metric=metric)
        idx = rq.IvfRabitqIndex.from_built(built)
        idx.set_option("scan_wave", 0)
        top_k = int(rng.choice([1, 2, 7, 10, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256]))
        nprobe = int(rng.integers(1, nlist + 1))
        q = np.ascontiguousarray(data[rng.integers(0, len(data), 24)].astype(np.float32))
        _compare(built, idx, q, top_k, nprobe)
        idx.close()


# ---- split-K ranking GEMM of small calls (round 5) -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,dim,nlist,bits,metric,top_k,nprobe", [
    pytest.param(8000, 128, 70, 7, 0, 10, 16, id="d128_70_lists_L2"),
    pytest.param(6000, 960, 48, 7, 0, 10, 12, id="d960_L2"),
    pytest.param(6000, 960, 48, 3, 1, 10, 16, id="d960_3bit_IP"),
    pytest.param(9000, 256, 1000, 7, 0, 10, 64, id="d256_1000_lists"),
    pytest.param(5000, 768, 40, 7, 1, 100, 10, id="d768_IP_top100"),
])
def test_split_k_rank_gemm_small_calls(n, dim, nlist, bits, metric, top_k, nprobe):
    """Calls of up to 256 queries split the K loop of the split-bf16 ranking GEMM over grid.z; the parts are added atomically (in no
    fixed order) to score rows the preparation kernel cleared.  The approximate scores only feed the shortlist: ids, scores, counts and
    diagnostics equal the oracle's with 2 and 4 forced parts, with the default choice and without the split, bit for bit between them."""
    data, built = build_index(n=n, dim=dim, nlist=nlist, total_bits=bits, metric=metric, normalize=(metric == 1), seed=5700 + dim + nlist)
    idx = rq.IvfRabitqIndex.from_built(built)
    for nq in (5, 33, 64, 200):
        q = make_dataset(nq, dim, max(nlist // 4, 1), 5750 + nq, normalize=(metric == 1))
        outs = []
        for ks in (1, 0, 2, 4):  # 1 = by batch size (the default), 0 = never, 2 / 4 = forced
            idx.set_option("rank_ksplit", ks)
            outs.append(_compare(built, idx, q, top_k, nprobe))
        idx.set_option("rank_ksplit", 1)
        for o in outs[1:]:
            assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[2], outs[0][2]) and np.array_equal(o[1].view(np.uint32), outs[0][1].view(np.uint32))
    assert idx.rank_fallbacks() == 0
    idx.close()
