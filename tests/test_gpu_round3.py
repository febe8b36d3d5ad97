"""GPU parity tests added in round 3: the lazy probe selection (lists that are provably skipped as a whole are neither
scored nor streamed), bench.py's self-launch and in-library modes, the small-batch path.  Same bar as test_gpu_parity.py:
ids / counts / diagnostics identical to the oracle, scores within 1e-4 (bit-equal between the GPU's own modes)."""
import numpy as np
import pytest

import oracle
import rabitq_rs_amd as rq
from conftest import build_index, make_dataset
from test_gpu_parity import _compare

pytestmark = pytest.mark.gpu


def _probe_taps(idx, built, q, top_k, nprobe, want_diag):
    """one device-entry call; returns (scanned list ids per query, dead_skipped, ids, diag)"""
    import torch
    dev = torch.device("cuda", 0)
    nq, dim = q.shape
    qd = torch.from_numpy(q).to(dev)
    d_ids = torch.zeros(nq, top_k, dtype=torch.int64, device=dev)
    d_sc = torch.zeros(nq, top_k, dtype=torch.float32, device=dev)
    d_cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    d_diag = torch.zeros(nq, 3, dtype=torch.int64, device=dev) if want_diag else None
    st = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    idx.search_batch_device(qd.data_ptr(), nq, dim, top_k, nprobe, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr(),
                            stream=st.cuda_stream, d_diag=d_diag.data_ptr() if want_diag else None)
    torch.cuda.synchronize(dev)
    probe = idx.debug_copy_workspace(st.cuda_stream, "probe", np.empty((nq, nprobe, 4), np.uint32))
    ds = idx.debug_copy_workspace(st.cuda_stream, "dead_skipped", np.empty((2, nq), np.uint32))
    idx.release_stream(st.cuda_stream)
    scanned = [probe[i, :ds[1, i], 3].copy() for i in range(nq)]
    gadd = [probe[i, :ds[1, i], 0].view(np.float32).copy() for i in range(nq)]
    return scanned, gadd, ds[0], d_ids.cpu().numpy().view(np.uint64), (d_diag.cpu().numpy().view(np.uint64) if want_diag else None)


LAZY_CASES = [
    # n, dim, nlist, bits, metric, nq, top_k, nprobe
    pytest.param(40000, 128, 256, 7, 0, 96, 10, 64, id="d128_7bit_L2"),
    pytest.param(30000, 960, 128, 7, 0, 48, 10, 48, id="d960_7bit_L2"),
    pytest.param(30000, 960, 128, 3, 1, 48, 10, 64, id="d960_3bit_IP"),
    pytest.param(40000, 128, 256, 1, 0, 64, 10, 64, id="d128_1bit_L2"),
    pytest.param(40000, 128, 256, 7, 0, 48, 100, 64, id="d128_7bit_L2_top100"),
    pytest.param(30000, 256, 300, 7, 1, 48, 10, 300, id="d256_7bit_IP_all_lists"),
    pytest.param(20000, 64, 700, 3, 0, 40, 5, 600, id="d64_3bit_L2_nprobe600"),
]


@pytest.mark.parametrize("n,dim,nlist,bits,metric,nq,top_k,nprobe", LAZY_CASES)
def test_lazy_selection_matches_oracle_and_eager(n, dim, nlist, bits, metric, nq, top_k, nprobe):
    """Lazy probe selection (rank_mfma.hpp): results and diagnostics equal the oracle's; the lists that go to the scan are
    a subsequence of the reference's probe order (src/ivf.rs:1803-1835) with bit-exact g_add; dead_skipped is the size of
    the probed lists left out; and lazy_select = 0 (every probed list scored and streamed) gives the same bits."""
    data, built = build_index(n=n, dim=dim, nlist=nlist, total_bits=bits, metric=metric, normalize=(metric == 1), seed=900 + dim + bits)
    idx = rq.IvfRabitqIndex.from_built(built)
    # queries from the data's own mixture (conftest.make_dataset draws new component means per seed: such queries are far
    # from every list and nothing can be pruned): perturbed data points
    rng = np.random.default_rng(901)
    q = data[rng.choice(n, nq, replace=False)] + 0.05 * rng.standard_normal((nq, dim)).astype(np.float32)
    if metric == 1:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = np.ascontiguousarray(q, dtype=np.float32)
    ids, sc, cnt = _compare(built, idx, q, top_k, nprobe)          # lazy (default), with and without diagnostics
    D = built.padded_dim
    ln = built.list_sizes()
    dropped = 0
    for want_diag in (True, False):
        scanned, gadd, dead, ids_d, diag = _probe_taps(idx, built, q, top_k, nprobe, want_diag)
        assert np.array_equal(ids_d, ids)
        for i in range(nq):
            r = oracle.rotate(built, q[i])
            probes = oracle.select_probes(built, r, nprobe)
            pos = {int(c): k for k, c in enumerate(probes)}
            got = [int(c) for c in scanned[i]]
            assert all(c in pos for c in got), f"query {i}: scanned a list outside the reference's probe set"
            order = [pos[c] for c in got]
            assert order == sorted(order) and len(set(order)) == len(order), f"query {i}: scan order is not the reference's"
            for k, c in enumerate(got[:6]):
                cv = built.centroid(c)
                dist = oracle.lib().ref_l2_distance_sqr(r.ctypes.data, cv.ctypes.data, D)
                dot = oracle.lib().ref_dot(r.ctypes.data, cv.ctypes.data, D)
                assert gadd[i][k] == (np.float32(dist) if metric == 0 else np.float32(-dot))
            if want_diag:
                left_out = sorted(set(int(c) for c in probes) - set(got))
                assert int(dead[i]) == int(ln[left_out].sum()), f"query {i}: dead_skipped {dead[i]} vs {ln[left_out].sum()}"
                dropped += len(left_out)
    assert dropped > 0, "the lazy selection never dropped a list on this data: the test exercises nothing"
    idx.set_option("lazy_select", 0)
    ids2, sc2, cnt2 = _compare(built, idx, q, top_k, nprobe)
    assert np.array_equal(ids2, ids) and np.array_equal(cnt2, cnt) and np.array_equal(sc2.view(np.uint32), sc.view(np.uint32))
    scanned, _, dead, _, _ = _probe_taps(idx, built, q, top_k, nprobe, True)
    assert all(len(s_) == min(nprobe, nlist) for s_ in scanned) and not dead.any()
    idx.close()


def test_lazy_selection_duplicate_and_tied_centroids():
    """Near-duplicate centroids put many lists inside the 2-eps window of each other and of tau: the boundary zone is
    large, members must be resolved exactly, ties in the exact keys fall back on the list id."""
    rng = np.random.default_rng(5)
    base = make_dataset(12000, 64, 6, 41)
    data = np.concatenate([base, base + 1e-6 * rng.standard_normal(base.shape).astype(np.float32)])
    _, built = build_index(nlist=96, total_bits=7, data=data, dim=64, seed=43)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(64, 64, 6, 44)
    for nprobe in (8, 40, 96):
        _compare(built, idx, q, 10, nprobe)
    idx.close()


def test_lazy_selection_small_lists_and_large_top_k():
    """top_k larger than the head lists hold: no finite select-time bound exists, the query takes the eager path; top_k
    just inside it: the bound comes from several blocks."""
    data, built = build_index(n=9000, dim=128, nlist=128, total_bits=7, seed=77)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(40, 128, 32, 78)
    for top_k in (1, 30, 64, 200, 1000):
        _compare(built, idx, q, top_k, 48)
    idx.close()


def test_lazy_selection_off_with_filter():
    """A filter disables the lazy selection (filtered vectors are never pushed, so the select-time bound of the k-th
    distance does not hold): results and diagnostics still equal the oracle's."""
    data, built = build_index(n=20000, dim=128, nlist=128, total_bits=7, seed=79)
    idx = rq.IvfRabitqIndex.from_built(built)
    q = make_dataset(48, 128, 32, 80)
    allowed = np.arange(0, 20000, 7)
    words = np.zeros((20000 + 31) // 32, np.uint32)
    np.bitwise_or.at(words, allowed >> 5, (np.uint32(1) << (allowed & 31).astype(np.uint32)))
    _compare(built, idx, q, 10, 64, words, 20000)
    idx.close()


# ---- bench.py: self-launch and in-library modes -----------------------------------------------------------------------
def _run_bench(args, env_extra, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_gpus_2_starts_itself():
    """`python bench.py --gpus 2` with NO torchrun wrapper: bench.py starts its two ranks as a child process (before it
    touches the GPU), relays their JSON line and exit code.  Two ranks share this one GPU over gloo here
    (RBQ_BENCH_REHEARSAL); on an 8-GPU node the same path runs RCCL."""
    d = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--no-extras", "--nbatches", "3", "--min-seconds", "0", "--no-latency",
                    "--n", "200000", "--nlist", "1024", "--nprobe", "32"], {"RBQ_BENCH_REHEARSAL": "1"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["per_rank_queries_per_s"]) == 2 and all(v > 0 for v in d["per_rank_queries_per_s"])
    assert d["rccl_world_size"] is None  # gloo rehearsal: no RCCL group
    assert d["timed_regions"] == 1 and d["pruned"]["launches"] == 4


def test_bench_in_library_two_replicas():
    """--in-library: ONE process, ONE handle with two replicas (devices [0, 0] on this one-GPU box), host buffers through
    rbq_search_batch; ids identical to one replica serving the same queries through the device entry."""
    d = _run_bench(["--in-library", "--gpus", "2", "--steps", "6", "--warmup", "2", "--min-seconds", "0", "--n", "200000", "--nlist", "1024",
                    "--nprobe", "32", "--nbatches", "4"], {"RBQ_BENCH_REHEARSAL": "1"})
    assert d["n_gpus"] == 2 and d["replicas"] == 2 and d["value"] > 0
    assert d["ids_identical_to_one_replica_device_entry"] and d["recall_at_10"] > 0.9
    assert set(d["latency"]) >= {"1", "8", "64", "256"} and d["latency"]["1"]["p50_us"] > 0


# ---- compile-time scan instantiations for the common padded dimensions ---------------------------------------------------
@pytest.mark.parametrize("dim,bits,metric", [(256, 7, 0), (384, 3, 1), (512, 7, 0), (1024, 7, 0), (1024, 1, 1), (1536, 3, 0), (500, 7, 0)])
def test_scan_dimension_instantiations(dim, bits, metric):
    """k_scan<D, ex, TR> for D in {256, 384, 512, 1024, 1536} (k_scan2.hip: rolling code window, immediate LUT offsets), and
    a dimension that has no instantiation (500 -> padded 512 with dim < padded_dim; 320, 200 elsewhere take the
    runtime-dimension kernel): the reference serves every multiple of 64 with one body (src/simd.rs:972-1014)."""
    data, built = build_index(n=5000, dim=dim, nlist=32, total_bits=bits, metric=metric, normalize=(metric == 1), seed=600 + dim + bits)
    idx = rq.IvfRabitqIndex.from_built(built)
    rng = np.random.default_rng(601)
    q = data[rng.choice(5000, 24, replace=False)] + 0.05 * rng.standard_normal((24, dim)).astype(np.float32)
    if metric == 1:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = np.ascontiguousarray(q, dtype=np.float32)
    for top_k, nprobe in ((10, 8), (100, 12), (200, 32)):
        _compare(built, idx, q, top_k, nprobe)
    idx.close()


# ---- replicas on DISTINCT devices: runs only where the box has two GPUs -------------------------------
def test_replicas_on_two_devices():
    """devices = [0, 1]: the second replica is a cross-device copy (peer copy, or the pinned bounce when the devices cannot
    reach each other), every replica launches on its own device with its own LDS attribute cache, and caller buffers are
    reached from both.  Skipped on a one-GPU box — until it has run somewhere, N distinct devices are covered by code review
    and by the [0, 0] tests only (include/rbq.h says so)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    data, built = build_index(n=9000, dim=128, nlist=64, total_bits=7, seed=201)
    one = rq.IvfRabitqIndex.from_built(built)
    two = rq.IvfRabitqIndex.from_built(built, devices=[0, 1])
    assert two.device_count() == 2
    rng = np.random.default_rng(77)
    q = np.ascontiguousarray(data[rng.choice(9000, 333, replace=False)] + 0.05 * rng.standard_normal((333, 128)).astype(np.float32))
    for top_k, nprobe in ((10, 16), (100, 8), (5000, 64)):  # (5000: the large-LDS heap — the LDS limit is raised per device)
        ids, sc, cnt = _compare(built, two, q, top_k, nprobe)
        ids1, sc1, cnt1, _ = one.batch_search_raw(q, rq.SearchParams(top_k, nprobe))
        assert np.array_equal(ids, ids1) and np.array_equal(cnt, cnt1) and np.array_equal(sc.view(np.uint32), sc1.view(np.uint32))
    ln = one.debug_copy_index("list_n", np.empty(64, np.uint32))
    nblocks = int(((ln + 31) // 32).sum())
    two.set_option("debug_replica", 1)
    for name, nbytes in (("blocks", nblocks * (128 * 4 + 384)), ("ids", nblocks * 32 * 8), ("bsum", nblocks * 32), ("centroids", 64 * 128 * 4)):
        a = one.debug_copy_index(name, np.empty(nbytes, np.uint8))
        b = two.debug_copy_index(name, np.empty(nbytes, np.uint8))
        assert np.array_equal(a, b), name
    one.close(); two.close()


# ---- traffic counters: striped, and switchable -----------------------------------------------------------------------------
def test_profile_counters_striped_and_switchable():
    """The counters of an open profile live in 64 stripes (query index mod 64) summed by the host: the totals are what one set
    of counters gave (compared with the oracle's probe sets, block bound off), whatever the batch size does to the stripes;
    with the option profile_counters = 0 an open profile keeps its stage timings and counts nothing; results never change."""
    data, built = build_index(n=12000, dim=128, nlist=48, total_bits=7, seed=341)
    idx = rq.IvfRabitqIndex.from_built(built)
    sizes = built.list_sizes()
    rng = np.random.default_rng(342)
    q = np.ascontiguousarray(data[rng.choice(12000, 150, replace=False)] + 0.05 * rng.standard_normal((150, 128)).astype(np.float32))
    idx.set_option("block_bound", 0)
    want_blocks = want_vectors = 0
    for i in range(q.shape[0]):
        probes = oracle.select_probes(built, oracle.rotate(built, q[i]), 12)
        want_blocks += int(((sizes[probes] + 31) // 32).sum())
        want_vectors += int(sizes[probes].sum())
    ref_ids = None
    for nq_call in (150, 64, 7):  # (whole batch; one stripe round; ragged: stripes filled unevenly)
        idx.profile_begin()
        ids = []
        for a0 in range(0, 150, nq_call):
            ids.append(idx.batch_search_raw(q[a0:a0 + nq_call], rq.SearchParams(10, 12))[0])
        idx.profile_end()
        c = idx.profile_counters()
        ids = np.concatenate(ids)
        assert c["queries"] == 150 and c["vectors_probed"] == want_vectors and c["stream_entries"] == want_blocks == c["code_blocks"]
        ref_ids = ids if ref_ids is None else ref_ids
        assert np.array_equal(ids, ref_ids)
    idx.set_option("profile_counters", 0)
    idx.profile_begin()
    ids = idx.batch_search_raw(q, rq.SearchParams(10, 12))[0]
    idx.profile_end()
    c = idx.profile_counters()
    assert np.array_equal(ids, ref_ids) and c["queries"] == 0 and c["vectors_probed"] == 0 and c["code_blocks"] == 0
    ms, launches = idx.profile_stage("scan")
    assert launches >= 1 and ms > 0
    idx.set_option("profile_counters", 1)
    idx.close()
