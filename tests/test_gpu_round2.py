"""GPU parity tests added in round 2: stage-level taps (rotation, LUT bytes, query constants, probe order against the
oracle's ref_rotate / ref_query_lut / ref_query_precompute / ref_select_probes), the multi-replica handle, the streamed
encoder, the pipelined host entry point, the cfg5-shaped index (nlist 65 536, nprobe 512, d 768), the Matrix-rotator
multi-stream path, the traffic counters and the optional rerank.  Same bar as test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

import oracle
import rabitq_rs_amd as rq
from conftest import build_index, make_dataset
from test_gpu_parity import RTOL, _compare

pytestmark = pytest.mark.gpu


# ---- stage-level parity -------------------------------------------------------------------------------------------
STAGE_CASES = [
    # n, dim, nlist, bits, metric, rotator, nq, nprobe, uniform, faster
    pytest.param(10000, 128, 256, 7, 0, 1, 32, 32, True, False, id="cfg1_readme_quickstart_not_faster"),
    pytest.param(8000, 128, 64, 7, 0, 1, 48, 16, False, True, id="cfg2_shape_d128_7bit_L2"),
    pytest.param(6000, 960, 48, 7, 0, 1, 40, 12, False, True, id="cfg3_shape_d960_7bit_L2"),
    pytest.param(6000, 960, 48, 3, 1, 1, 40, 16, False, True, id="cfg4_shape_d960_3bit_IP"),
    pytest.param(3000, 96, 24, 7, 0, 0, 24, 6, False, True, id="matrix_rotator_d96_7bit_L2"),
    pytest.param(3000, 100, 24, 3, 1, 1, 24, 6, False, True, id="kac_d100_pad128_3bit_IP"),
]


@pytest.mark.parametrize("n,dim,nlist,bits,metric,rot,nq,nprobe,uniform,faster", STAGE_CASES)
def test_stage_level_parity(n, dim, nlist, bits, metric, rot, nq, nprobe, uniform, faster):
    """Every intermediate of the query path against the oracle's own stage functions, not only the final ids:
    rotated query (bitwise; src/rotation.rs:350-401), u8 LUT bytes + delta + sum_vl (src/ivf.rs:798-845),
    QueryPrecomputed (src/ivf.rs:862-878) and the ordered probe list (src/ivf.rs:1782-1835)."""
    import torch
    data, built = build_index(n=n, dim=dim, nlist=nlist, total_bits=bits, metric=metric, rotator=rot, uniform=uniform,
                              normalize=(metric == 1), seed=3000 + dim + bits, faster=faster)
    idx = rq.IvfRabitqIndex.from_built(built)
    idx.set_option("lazy_select", 0)  # the full ordered probe list is wanted here; tests/test_gpu_round3.py covers the lazy selection
    q = make_dataset(nq, dim, max(nlist // 4, 1), 3131, normalize=(metric == 1), uniform=uniform)
    top_k = 10
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(q).to(dev)
    d_ids = torch.zeros(nq, top_k, dtype=torch.int64, device=dev)
    d_sc = torch.zeros(nq, top_k, dtype=torch.float32, device=dev)
    d_cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    idx.search_batch_device(qd.data_ptr(), nq, dim, top_k, nprobe, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr(),
                            stream=st.cuda_stream)
    torch.cuda.synchronize(dev)
    D = built.padded_dim
    Dc = (D + 63) // 64 * 64
    ex = built.hdr.ex_bits
    rot_d = idx.debug_copy_workspace(st.cuda_stream, "rot", np.empty((nq, D), np.float32))
    lut_d = idx.debug_copy_workspace(st.cuda_stream, "lut", np.empty((nq, Dc * 4), np.uint8))
    consts_d = idx.debug_copy_workspace(st.cuda_stream, "consts", np.empty((nq, 12), np.float32))
    probe_d = idx.debug_copy_workspace(st.cuda_stream, "probe", np.empty((nq, nprobe, 4), np.uint32))
    for i in range(nq):
        r = oracle.rotate(built, q[i])
        assert np.array_equal(r.view(np.uint32), rot_d[i].view(np.uint32)), f"query {i}: rotated vector differs bitwise"
        lut, delta, sum_vl = oracle.query_lut(r)
        # device LUT order: position p holds codebook p ^ 1; padding codebooks (D..Dc) are zero tables
        dev_lut = lut_d[i].reshape(Dc // 4, 16)
        ref_lut = lut.reshape(D // 4, 16)
        assert np.array_equal(dev_lut[np.arange(D // 4) ^ 1], ref_lut), f"query {i}: LUT bytes differ"
        assert not dev_lut[D // 4:].any()  # (D/4 is even: the pair swap never mixes real and padding tables)
        qc = oracle.query_precompute(r, ex)
        got = consts_d[i]
        want = np.array([delta, sum_vl, qc.k1x_sum_q, qc.kbx_sum_q, qc.binary_scale, qc.query_norm], np.float32)
        assert np.array_equal(got[:6].view(np.uint32), want.view(np.uint32)), f"query {i}: consts {got[:6]} vs {want}"
        # accu bounds: every u8 sum lies in [amin, amax]
        assert got[10] == ref_lut.min(axis=1).astype(np.float64).sum() and got[11] == ref_lut.max(axis=1).astype(np.float64).sum()
        probes = oracle.select_probes(built, r, nprobe)
        assert np.array_equal(probe_d[i, :len(probes), 3], probes), f"query {i}: probe order differs"
        # g_add / g_err of every probed list (src/ivf.rs:1850-1857), canonical order of src/math.rs
        pf = probe_d[i].view(np.float32)
        for rnk, cid in enumerate(probes[:8]):
            c = built.centroid(int(cid))
            dist = oracle.lib().ref_l2_distance_sqr(r.ctypes.data, c.ctypes.data, D)
            dot = oracle.lib().ref_dot(r.ctypes.data, c.ctypes.data, D)
            g_add = np.float32(dist) if metric == 0 else np.float32(-dot)
            assert pf[rnk, 0] == g_add and pf[rnk, 1] == np.sqrt(np.float32(dist))
    rc, oids, osc, ocnt, _ = oracle.search_batch(built, q, top_k, nprobe)
    assert np.array_equal(d_ids.cpu().numpy().view(np.uint64), oids) and np.array_equal(d_cnt.cpu().numpy().view(np.uint32), ocnt)
    idx.release_stream(st.cuda_stream)
    with pytest.raises(rq.RabitqError):
        idx.debug_copy_workspace(st.cuda_stream, "rot", np.empty((nq, D), np.float32))  # released
    idx.close()


# ---- multi-replica handle (n_devices > 1) on ONE GPU ---------------------------------------------------------------
def test_multi_replica_handle_same_gpu():
    """rbq_index_create(..., n_devices=2, devices=[0,0]): the second replica is a device-to-device copy, a batch is
    sharded [r*nq/2, (r+1)*nq/2) over the replicas by rbq_search_batch; results equal the oracle's and the
    single-replica handle's, array for array."""
    data, built = build_index(n=9000, dim=128, nlist=64, total_bits=7, seed=201)
    one = rq.IvfRabitqIndex.from_built(built)
    two = rq.IvfRabitqIndex.from_built(built, devices=[0, 0])
    assert one.device_count() == 1 and two.device_count() == 2 and len(two) == len(one)
    q = make_dataset(333, 128, 16, 202)  # odd: ragged shards and sub-batches
    for top_k, nprobe in ((10, 16), (100, 8)):
        ids, sc, cnt = _compare(built, two, q, top_k, nprobe)
        ids1, sc1, cnt1, _ = one.batch_search_raw(q, rq.SearchParams(top_k, nprobe))
        assert np.array_equal(ids, ids1) and np.array_equal(cnt, cnt1)
        assert np.array_equal(sc.view(np.uint32), sc1.view(np.uint32))
    ln = one.debug_copy_index("list_n", np.empty(64, np.uint32))
    nblocks = int(((ln + 31) // 32).sum())
    two.set_option("debug_replica", 1)
    for name, nbytes in (("blocks", nblocks * (128 * 4 + 384)), ("ids", nblocks * 32 * 8), ("ex", nblocks * 32 * 1 * 256),
                         ("bsum", nblocks * 32), ("centroids", 64 * 128 * 4)):
        a = one.debug_copy_index(name, np.empty(nbytes, np.uint8))
        b = two.debug_copy_index(name, np.empty(nbytes, np.uint8))
        assert np.array_equal(a, b), name
    with pytest.raises(rq.RabitqError):
        two.set_option("debug_replica", 2)
    # a filtered, diagnosed batch through both replicas
    allowed = np.arange(0, 9000, 3)
    words = np.zeros((9000 + 31) // 32, np.uint32)
    np.bitwise_or.at(words, allowed >> 5, (np.uint32(1) << (allowed & 31).astype(np.uint32)))
    _compare(built, two, q[:101], 10, 12, words, 9000)
    # RBQ1 bytes into two replicas
    three = rq.IvfRabitqIndex.load_from_bytes(built.save_rbq1(), devices=[0, 0, 0])
    assert three.device_count() == 3
    _compare(built, three, q[:50], 10, 16)
    one.close(); two.close(); three.close()


# ---- streamed encoder ------------------------------------------------------------------------------------------------
STREAM_CASES = [
    pytest.param(7000, 960, 40, 7, 0, 1, id="stream_d960_7bit_L2"),
    pytest.param(5000, 100, 24, 3, 1, 1, id="stream_d100_pad128_3bit_IP"),
    pytest.param(4000, 128, 32, 1, 0, 1, id="stream_d128_1bit_L2"),
    pytest.param(3000, 64, 20, 7, 1, 0, id="stream_matrix_d64_7bit_IP"),
]


@pytest.mark.parametrize("n,dim,nlist,bits,metric,rot", STREAM_CASES)
def test_stream_builder_matches_one_shot(n, dim, nlist, bits, metric, rot):
    """rbq_build_stream_begin/push/finish with ragged chunks (host arrays, then device pointers) produces, array for
    array, the index of rbq_index_build_device and of rbq_index_create over the CPU builder's ClusterData."""
    import torch
    data = make_dataset(n, dim, max(nlist // 4, 1), 241, normalize=(metric == 1))
    cent, assign = rq.builder.kmeans(data, nlist, 5, 242)
    built = rq.builder.train_with_clusters(data, cent, assign, bits, metric, rot, 243, True)
    ref = rq.IvfRabitqIndex.from_built(built)
    sizes = np.bincount(assign, minlength=nlist).astype(np.uint32)
    sb = rq.StreamBuilder(built.hdr_ptr, cent, sizes, built.t_const)
    cuts = [0, 1, 700, 701, n // 2, n - 13, n]
    xd = torch.from_numpy(data).cuda()
    ad = torch.from_numpy(assign.astype(np.int32)).cuda()
    for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if k % 2 == 0:
            sb.push(data[a:b], assign[a:b], a)
        else:  # device pointers
            sb.push(xd[a:b].contiguous().data_ptr(), ad[a:b].contiguous().data_ptr(), a, b - a)
    enc = sb.finish()
    hdr = built.hdr
    D, ex = hdr.padded_dim, hdr.ex_bits
    Dc = (D + 63) // 64 * 64
    ln = ref.debug_copy_index("list_n", np.empty(nlist, np.uint32))
    nblocks = int(((ln + 31) // 32).sum())
    cpu_u = 128 // ex if ex else 1
    w4 = ((D // 16 + cpu_u - 1) // cpu_u) if ex else 0
    sizes_b = {"list_gb0": nlist * 4, "list_n": nlist * 4, "centroids": nlist * D * 4, "blocks": nblocks * (Dc * 4 + 384),
               "ids": nblocks * 32 * 8, "bsum": nblocks * 32}
    if ex:
        sizes_b.update({"ex": nblocks * 32 * w4 * 256, "fadd_ex": nblocks * 32 * 4, "fres_ex": nblocks * 32 * 4})
    for name, nbytes in sizes_b.items():
        a = ref.debug_copy_index(name, np.empty(nbytes, np.uint8))
        b = enc.debug_copy_index(name, np.empty(nbytes, np.uint8))
        bad = np.nonzero(a != b)[0]
        assert bad.size == 0, f"{name}: {bad.size} bytes differ, first at {bad[:5]}"
    q = make_dataset(32, dim, max(nlist // 4, 1), 244, normalize=(metric == 1))
    _compare(built, enc, q, 10, min(8, nlist))
    ref.close(); enc.close()


def test_stream_builder_rejects_protocol_errors():
    import torch  # noqa: F401
    data, built = build_index(n=2000, dim=64, nlist=8, total_bits=7, seed=251)
    cent, assign = rq.builder.kmeans(data, 8, 3, 252)
    sizes = np.bincount(assign, minlength=8).astype(np.uint32)
    sb = rq.StreamBuilder(built.hdr_ptr, cent, sizes, built.t_const)
    sb.push(data[:1000], assign[:1000], 0)
    with pytest.raises(rq.RabitqError) as e:       # not every announced vector pushed
        sb.finish()
    assert e.value.kind == "InvalidConfig"
    with pytest.raises(rq.RabitqError) as e:       # ids must ascend
        sb.push(data[:10], assign[:10], 5)
    assert e.value.kind == "InvalidConfig"
    with pytest.raises(rq.RabitqError) as e:       # list id out of range
        sb.push(data[1000:1010], np.full(10, 8, np.uint32), 1000)
    assert e.value.kind == "InvalidConfig"
    sb.abort()
    sb = rq.StreamBuilder(built.hdr_ptr, cent, sizes, built.t_const)
    with pytest.raises(rq.RabitqError) as e:       # a list outgrows its announced size
        sb.push(data[:2000], np.zeros(2000, np.uint32), 0)
    assert e.value.kind == "InvalidConfig"
    sb.abort()
    with pytest.raises(rq.RabitqError):            # 7-bit codes need the constant rescale factor
        rq.StreamBuilder(built.hdr_ptr, cent, sizes, 0.0)


# ---- host entry point ---------------------------------------------------------------------------------------------------
def test_host_pipeline_pageable_and_pinned_buffers():
    """rbq_search_batch cuts the batch into sub-batches pipelined over several streams; pageable caller buffers are
    staged through pinned memory, page-locked ones (rbq_host_alloc) are DMA-ed directly.  Ragged sizes, diagnostics
    and filters go through both; results equal the oracle's."""
    data, built = build_index(n=12000, dim=128, nlist=64, total_bits=7, seed=261)
    idx = rq.IvfRabitqIndex.from_built(built)
    lib = rq.index.lib()
    for nq in (1, 257, 1500, 4500):
        q = make_dataset(nq, 128, 16, 262 + nq)
        ids, sc, cnt = _compare(built, idx, q, 10, 12)
        # the same call on page-locked buffers
        top_k = 10
        nb = [nq * 128 * 4, nq * top_k * 8, nq * top_k * 4, nq * 4, nq * 24]
        ptrs = [lib.rbq_host_alloc(b) for b in nb]
        assert all(ptrs)
        C.memmove(ptrs[0], q.ctypes.data, nb[0])
        rc = lib.rbq_search_batch(idx._h, ptrs[0], nq, 128, top_k, 12, None, 0, ptrs[1], ptrs[2], ptrs[3], ptrs[4])
        assert rc == 0
        pid = np.ctypeslib.as_array(C.cast(ptrs[1], C.POINTER(C.c_uint64)), shape=(nq, top_k)).copy()
        psc = np.ctypeslib.as_array(C.cast(ptrs[2], C.POINTER(C.c_float)), shape=(nq, top_k)).copy()
        pcn = np.ctypeslib.as_array(C.cast(ptrs[3], C.POINTER(C.c_uint32)), shape=(nq,)).copy()
        assert np.array_equal(pid, ids) and np.array_equal(pcn, cnt) and np.array_equal(psc.view(np.uint32), sc.view(np.uint32))
        for p in ptrs:
            lib.rbq_host_free(p)
    idx.close()


def test_host_entry_many_caller_threads():
    """Several host threads inside rbq_search_batch on one handle (Rayon workers calling search, src/ivf.rs:1748)."""
    import threading
    data, built = build_index(n=8000, dim=96, nlist=40, total_bits=3, seed=271)
    idx = rq.IvfRabitqIndex.from_built(built)
    qs = [make_dataset(nq, 96, 10, 272 + nq) for nq in (700, 1300, 64, 2048)]
    want = [oracle.search_batch(built, q, 10, 9) for q in qs]
    errs = []

    def work(i):
        try:
            for _ in range(4):
                ids, sc, cnt, _ = idx.batch_search_raw(qs[i], rq.SearchParams(10, 9))
                assert np.array_equal(ids, want[i][1]) and np.array_equal(cnt, want[i][3])
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    idx.close()


# ---- cfg5-shaped index ------------------------------------------------------------------------------------------------------
def test_cfg5_shaped_many_lists_nprobe512():
    """BASELINE config 5 in shape (nlist = 65 536, nprobe = 512, d = 768, 7-bit, L2) at a size the oracle covers:
    k_select_mfma<0> (the approximate score row re-read from global memory) with a 512-probe block stream."""
    import torch
    nlist, dim, n = 65536, 768, 300000
    rng = np.random.default_rng(281)
    data = make_dataset(n, dim, 2048, 282)
    cent = data[rng.choice(n, nlist, replace=False)].copy()
    cent += 0.01 * rng.standard_normal(cent.shape).astype(np.float32)
    x, c = torch.from_numpy(data).cuda(), torch.from_numpy(cent).cuda()
    assign = torch.empty(n, dtype=torch.int64, device="cuda")
    cn = (c * c).sum(1)
    for s in range(0, n, 8192):
        assign[s:s + 8192] = (cn[None, :] - 2.0 * (x[s:s + 8192] @ c.T)).argmin(1)
    assign = assign.cpu().numpy().astype(np.uint32)
    del x, c
    torch.cuda.empty_cache()
    built = rq.builder.train_with_clusters(data, cent, assign, 7, 0, 1, 283, True)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(24, dim, 2048, 284)
    _compare(built, idx, q, 10, 512)
    _compare(built, idx, data[:8], 100, 512)
    assert idx.rank_fallbacks() == 0
    # the same index from the streamed encoder, searched with the same parameters
    sizes = np.bincount(assign, minlength=nlist).astype(np.uint32)
    sb = rq.StreamBuilder(built.hdr_ptr, cent, sizes, built.t_const)
    for a in range(0, n, 70000):
        sb.push(data[a:a + 70000], assign[a:a + 70000], a)
    enc = sb.finish()
    _compare(built, enc, q, 10, 512)
    idx.close(); enc.close()


# ---- Matrix rotator on concurrent streams (k_prep: the workgroup-per-query preparation) ------------------------------
def test_matrix_rotator_concurrent_streams():
    """The Matrix rotator is prepared by k_prep (one workgroup per query) — the kernel that once produced wrong LUT
    bytes beside the x16 bf16 MFMA GEMM of a neighbouring stream (rank_mfma.hpp).  Several caller streams at once,
    every result equal to the oracle's."""
    import torch
    dev = torch.device("cuda", 0)
    data, built = build_index(n=40000, dim=256, nlist=512, total_bits=7, rotator=0, seed=291)
    idx = rq.IvfRabitqIndex.from_built(built)
    nq, top_k, nprobe, ns = 1024, 10, 32, 4
    q = make_dataset(nq, 256, 128, 292)
    rc, oids, osc, ocnt, _ = oracle.search_batch(built, q, top_k, nprobe)
    assert rc == 0
    qd = torch.from_numpy(q).to(dev)
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    d_ids = [torch.zeros(nq, top_k, dtype=torch.int64, device=dev) for _ in range(ns)]
    d_sc = [torch.zeros(nq, top_k, dtype=torch.float32, device=dev) for _ in range(ns)]
    d_cnt = [torch.zeros(nq, dtype=torch.int32, device=dev) for _ in range(ns)]
    torch.cuda.synchronize(dev)
    for rep in range(10):
        for i in range(ns):
            idx.search_batch_device(qd.data_ptr(), nq, 256, top_k, nprobe, d_ids[i].data_ptr(), d_sc[i].data_ptr(),
                                    d_cnt[i].data_ptr(), stream=streams[i].cuda_stream)
        torch.cuda.synchronize(dev)
        for i in range(ns):
            got = d_ids[i].cpu().numpy().view(np.uint64)
            bad = np.nonzero((got != oids).any(axis=1))[0]
            assert bad.size == 0, f"rep {rep} stream {i}: ids differ for queries {bad[:10]}"
            assert np.array_equal(d_cnt[i].cpu().numpy().view(np.uint32), ocnt)
            d_ids[i].zero_()
    idx.close()


# ---- traffic counters ------------------------------------------------------------------------------------------------------
def test_profile_traffic_counters():
    """rbq_profile_counters: with the block bound off every probed block's codes are requested (code blocks == stream
    entries == sum of ceil(n_c/32) over probed lists); with it on the kernel fetches a subset, results unchanged."""
    data, built = build_index(n=20000, dim=128, nlist=64, total_bits=7, seed=301)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(200, 128, 16, 302)
    sizes = built.list_sizes()
    D = built.padded_dim
    counters = {}
    for bound in (0, 1):
        idx.set_option("block_bound", bound)
        idx.profile_begin()
        ids, sc, cnt, _ = idx.batch_search_raw(q, rq.SearchParams(10, 16))
        idx.profile_end()
        counters[bound] = (idx.profile_counters(), idx.profile_scan_bytes(), ids)
    c0, bytes0, ids0 = counters[0]
    c1, bytes1, ids1 = counters[1]
    assert np.array_equal(ids0, ids1)
    blocks = 0
    vectors = 0
    for i in range(q.shape[0]):
        probes = oracle.select_probes(built, oracle.rotate(built, q[i]), 16)
        blocks += int(((sizes[probes] + 31) // 32).sum())
        vectors += int(sizes[probes].sum())
    assert c0["queries"] == 200 and c1["queries"] == 200
    assert c0["vectors_probed"] == vectors and bytes0 == vectors * (D // 8 + 12)
    # bound on = lazy probe selection on: lists proved skipped as a whole never enter the stream, and without diagnostics
    # the profile counts the APPROXIMATE probe set (boundary lists may differ from the exact one by a list or two)
    assert abs(c1["vectors_probed"] - vectors) <= 0.02 * vectors and bytes1 == c1["vectors_probed"] * (D // 8 + 12)
    assert c0["stream_entries"] == blocks and c0["code_blocks"] == blocks and c0["meta_blocks"] == blocks
    assert c1["stream_entries"] <= blocks and 0 < c1["code_blocks"] <= c1["stream_entries"]  # (how much is pruned depends on the data)
    assert c0["ex_evals"] >= c1["ex_evals"] > 0
    idx.close()


# ---- optional rerank (extension, default off) ---------------------------------------------------------------------------
@pytest.mark.parametrize("metric", [0, 1])
def test_optional_rerank_default_off(metric):
    """Without attached vectors (the default) results are the reference's.  With them, the SAME ids come back
    re-scored with the exact distance (canonical order of src/math.rs) and re-sorted."""
    data, built = build_index(n=6000, dim=96, nlist=32, total_bits=7, metric=metric, normalize=(metric == 1), seed=311)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(40, 96, 8, 312, normalize=(metric == 1))
    ids, sc, cnt = _compare(built, idx, q, 20, 8)
    idx.set_rerank_vectors(data)
    rid, rsc, rcnt, _ = idx.batch_search_raw(q, rq.SearchParams(20, 8))
    assert np.array_equal(rcnt, cnt)
    L = oracle.lib()
    for i in range(q.shape[0]):
        c = int(cnt[i])
        assert sorted(rid[i, :c].tolist()) == sorted(ids[i, :c].tolist())
        exact = np.array([(L.ref_l2_distance_sqr if metric == 0 else L.ref_dot)(q[i].ctypes.data, data[int(j)].ctypes.data, 96)
                          for j in rid[i, :c]], np.float32)
        assert np.array_equal(exact.view(np.uint32), rsc[i, :c].view(np.uint32))
        assert (np.diff(rsc[i, :c]) >= 0).all() if metric == 0 else (np.diff(rsc[i, :c]) <= 0).all()
    idx.set_option("rerank", 0)
    _compare(built, idx, q, 20, 8)
    idx.set_rerank_vectors(None)
    with pytest.raises(rq.RabitqError):
        idx.set_option("rerank", 1)
    idx.close()


# ---- top_k up to 256 in registers ---------------------------------------------------------------------------------
@pytest.mark.parametrize("top_k", [64, 65, 100, 128, 129, 192, 200, 256, 257])
def test_register_sorted_run_up_to_256(top_k):
    """64 < top_k <= 256 keeps the top-k as a sorted run over four registers per lane of the replay wave (k_scan<..., TR=4>;
    the reference's own benchmark setting is top_k = 100, examples/recall_qps_sweep.rs:120); 257 falls back to the LDS heap.
    Run boundaries (64/65, 128/129, 192, 256) and partially filled runs (fewer candidates than top_k) included."""
    data, built = build_index(n=9000, dim=128, nlist=48, total_bits=7, seed=401)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(40, 128, 12, 402)
    _compare(built, idx, q, top_k, 12)
    _compare(built, idx, q[:8], top_k, 1)       # one list: fewer candidates than top_k for most queries
    _compare(built, idx, data[:8], top_k, 48)   # every list
    # (f32 distances among hundreds of results coincide now and then; such queries re-run with the exact heap — still equal)
    idx.close()


@pytest.mark.parametrize("top_k,bits,metric", [(64, 7, 0), (100, 7, 0), (100, 3, 1), (200, 1, 0), (255, 7, 0), (256, 7, 1), (300, 3, 0)])
def test_register_sorted_run_ties_restart(top_k, bits, metric):
    """Duplicate vectors give bit-identical distances: the sorted run reports the tie, the query is re-run with the exact
    BinaryHeap emulation — in the replay wave's registers up to top_k = 255 (four registers per lane), in LDS from 256 —
    and matches the oracle element for element."""
    base = make_dataset(900, 64, 6, 411, normalize=(metric == 1))
    data = np.concatenate([base, base[:500], base[:200]], axis=0)
    _, built = build_index(nlist=10, total_bits=bits, data=data, dim=64, metric=metric)
    idx = rq.IvfRabitqIndex.from_built(built)
    _compare(built, idx, base[:40], top_k, 6)
    assert idx.heap_restarts() > 0 or top_k > 256  # (above 256 the exact heap runs from the first candidate: nothing to restart)
    idx.close()


# ---- BASELINE sizes through size-independent properties -----------------------------------------------------------------
@pytest.mark.parametrize("name,n,dim,nlist,bits,metric,nprobe,batch", [
    ("cfg2_sift1m", 1_000_000, 128, 1024, 7, 0, 64, 1024),
    ("cfg3_gist1m", 1_000_000, 960, 4096, 7, 0, 128, 4096),
    ("cfg4_gist1m_ip_3bit", 1_000_000, 960, 4096, 3, 1, 256, 2048),
])
def test_full_size_baseline_configs_properties(name, n, dim, nlist, bits, metric, nprobe, batch):
    """BASELINE.json's configurations at FULL size (1 M vectors; the index built by the device encoder, whose arrays
    are byte-identical to the CPU build's — test_device_encoder_matches_cpu_builder), where the oracle would take minutes:
    properties that hold at any size —
      * a query's result does not depend on its batch (permuted batch, sub-batch, one at a time),
      * diagnostic switches change the work, never the result (block bound off, exact all-pairs ranking, f32 GEMM,
        exact heap from the first candidate, workgroup-per-query preparation),
      * the host entry point (pageable buffers, sub-batch pipeline, diagnostics) returns the device path's bits,
      * results are well-formed: counts == top_k, distances ascending (L2) / scores descending (IP), ids distinct and
        inside the index; recall@10 against exact f32 brute force is at the level bench.py reports.
    And, since round 3, the oracle itself on the first 128 queries: the CPU build of the same index (arrays byte-identical to
    the device encoder's) searched by oracle.search_batch — ids, counts and SearchDiagnostics exactly, scores at 1e-4.
    (bench.py also compares two 1024-query batches of the full-size cfg3 index with the oracle on every run.)"""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    mix = bench.Mixture(torch, dev, dim, nlist, "mixture_id32", metric == 1)
    x = mix.draw(n, 20260105)
    cent, assign = bench.kmeans_gpu(torch, x, nlist, 2, 20260103)
    xs = mix.draw(max(2 * nlist, 4096), 99).cpu().numpy()
    small = rq.builder.train_with_clusters(xs, cent.cpu().numpy(), (np.arange(xs.shape[0]) % nlist).astype(np.uint32), bits, metric, 1,
                                           20260104, True)
    a32 = assign.to(torch.int32).contiguous()
    idx = rq.IvfRabitqIndex.build_on_device(small.hdr_ptr, cent.cpu().numpy(), x.data_ptr(), a32.data_ptr(), n, small.t_const)
    top_k = 10
    q = mix.draw(batch, 20260102).contiguous()
    qh = q.cpu().numpy()

    def dev_search(qd, nq, index=idx):
        ids = torch.empty(nq, top_k, dtype=torch.int64, device=dev)
        sc = torch.empty(nq, top_k, dtype=torch.float32, device=dev)
        cnt = torch.empty(nq, dtype=torch.int32, device=dev)
        index.search_batch_device(qd.data_ptr(), nq, dim, top_k, nprobe, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr())
        torch.cuda.synchronize(dev)
        return ids.cpu().numpy().view(np.uint64), sc.cpu().numpy(), cnt.cpu().numpy().view(np.uint32)

    ids, sc, cnt = dev_search(q, batch)
    # well-formed
    assert (cnt == top_k).all()
    assert ((np.diff(sc, axis=1) >= 0) if metric == 0 else (np.diff(sc, axis=1) <= 0)).all()
    assert (ids < n).all() and all(len(set(r.tolist())) == top_k for r in ids[::97])
    # recall against exact brute force
    gt = bench.exact_topk(torch, x, q, top_k, metric).cpu().numpy()
    rec = bench.recall_of(ids, gt, top_k)
    assert rec > (0.93 if bits == 3 else 0.95), rec
    # the oracle at full size: CPU build (seconds on the box's cores) + 128 queries
    built = rq.builder.train_with_clusters(x.cpu().numpy(), cent.cpu().numpy(), assign.cpu().numpy().astype(np.uint32), bits, metric, 1,
                                           20260104, True)
    del x
    torch.cuda.empty_cache()
    nor = 128
    rc, oids, osc, ocnt, odiag = oracle.search_batch(built, qh[:nor], top_k, nprobe, want_diag=True)
    assert rc == 0
    assert np.array_equal(ids[:nor], oids) and np.array_equal(cnt[:nor], ocnt), f"{name}: ids differ from the oracle at full size"
    np.testing.assert_allclose(sc[:nor], osc, rtol=RTOL, atol=0)
    gid, gsc, gcnt, gdiag = idx.batch_search_raw(qh[:nor], rq.SearchParams(top_k, nprobe), want_diag=True)
    assert np.array_equal(gid, oids) and np.array_equal(gdiag, odiag), f"{name}: SearchDiagnostics differ from the oracle at full size"
    del built
    # batch independence: permuted batch, a ragged sub-batch, single queries
    perm = np.random.default_rng(5).permutation(batch)
    pid, psc, _ = dev_search(q[torch.from_numpy(perm).to(dev)].contiguous(), batch)
    assert np.array_equal(pid, ids[perm]) and np.array_equal(psc.view(np.uint32), sc[perm].view(np.uint32))
    sid, ssc, _ = dev_search(q[100:357].contiguous(), 257)
    assert np.array_equal(sid, ids[100:357]) and np.array_equal(ssc.view(np.uint32), sc[100:357].view(np.uint32))
    for i in (0, batch // 2, batch - 1):
        oid, osc, _ = dev_search(q[i:i + 1].contiguous(), 1)
        assert np.array_equal(oid[0], ids[i]) and np.array_equal(osc[0].view(np.uint32), sc[i].view(np.uint32))
    # diagnostic switches: same bits, different work
    for opt, val, back in (("block_bound", 0, 1), ("exact_rank", 1, 0), ("f32_rank", 1, 0), ("exact_heap", 1, 0), ("wg_prep", 1, 0),
                           ("lazy_select", 0, 1)):
        idx.set_option(opt, val)
        nq_o = 512 if opt == "exact_rank" else batch  # (the all-pairs canonical ranking is the slow one)
        o_ids, o_sc, o_cnt = dev_search(q[:nq_o].contiguous(), nq_o)
        idx.set_option(opt, back)
        assert np.array_equal(o_ids, ids[:nq_o]) and np.array_equal(o_sc.view(np.uint32), sc[:nq_o].view(np.uint32)), opt
    assert idx.rank_fallbacks() == 0
    # host entry point (pageable buffers; sub-batches of 1024 over several lanes), with diagnostics
    hid, hsc, hcnt, hdiag = idx.batch_search_raw(qh, rq.SearchParams(top_k, nprobe), want_diag=True)
    assert np.array_equal(hid, ids) and np.array_equal(hsc.view(np.uint32), sc.view(np.uint32)) and np.array_equal(hcnt, cnt)
    assert (hdiag[:, 0] >= top_k).all() and (hdiag[:, 1] > 0).all()            # estimated >= results, something was pruned
    assert (hdiag[:, 2] >= hdiag[:, 0]).all() if bits > 1 else (hdiag[:, 2] == 0).all()  # every estimate was an ex evaluation
    idx.close()


def test_create_destroy_cycles_release_device_memory():
    """Handles, replicas, stream workspaces, pooled lanes (pinned staging included) and stream builders give their
    device memory back: 25 create / search / destroy cycles end where they started."""
    import torch
    data, built = build_index(n=20000, dim=256, nlist=64, total_bits=7, seed=501)
    q = make_dataset(300, 256, 16, 502)
    cent, assign = rq.builder.kmeans(data, 64, 2, 503)
    sizes = np.bincount(assign, minlength=64).astype(np.uint32)
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(q).to(dev)
    out = (torch.empty(300, 10, dtype=torch.int64, device=dev), torch.empty(300, 10, dtype=torch.float32, device=dev),
           torch.empty(300, dtype=torch.int32, device=dev))

    def cycle():
        idx = rq.IvfRabitqIndex.from_built(built, devices=[0, 0])
        idx.batch_search_raw(q, rq.SearchParams(10, 8), want_diag=True)
        streams = [torch.cuda.Stream(dev) for _ in range(3)]
        for s in streams:
            idx.search_batch_device(qd.data_ptr(), 300, 256, 10, 8, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), stream=s.cuda_stream)
        torch.cuda.synchronize(dev)
        idx.release_stream(streams[0].cuda_stream)   # the others are freed with the handle
        idx.close()
        sb = rq.StreamBuilder(built.hdr_ptr, cent, sizes, built.t_const)
        sb.push(data[:5000], assign[:5000], 0)
        sb.abort()                                   # an unfinished builder

    for _ in range(3):
        cycle()
    torch.cuda.synchronize(dev)
    free0 = torch.cuda.mem_get_info(dev)[0]
    for _ in range(25):
        cycle()
    torch.cuda.synchronize(dev)
    free1 = torch.cuda.mem_get_info(dev)[0]
    assert free0 - free1 < 64 << 20, f"device memory not returned: {(free0 - free1) >> 20} MiB after 25 cycles"
