"""Diagnostic soak: the seeded random configurations of tests/test_gpu_parity.py over a long seed range, in one
process (python tests/diag/soak.py FIRST LAST).  Prints the seeds whose result differs from the oracle.
Import paths are set up exactly as tests/conftest.py does (oracle/oracle.py must win over the oracle/ directory).
A harness error (the same non-assertion exception three times in a row) aborts the run: it says nothing about parity."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (first, as under pytest: one HIP runtime per process, see rabitq_rs_amd/index.py)
import conftest  # noqa: F401  (inserts ROOT and ROOT/oracle into sys.path in the order the tests use)
import __graft_entry__ as g
g.build_cpu_libs()
import oracle
assert hasattr(oracle, "search_batch"), "wrong `oracle` module on sys.path: %r" % (oracle,)
import test_gpu_parity as t

first, last = int(sys.argv[1]), int(sys.argv[2])
wide = len(sys.argv) > 3 and sys.argv[3] in ("wide", "streams", "mstg", "ties", "lists", "threads", "lazy")
lazy_mode = len(sys.argv) > 3 and sys.argv[3] == "lazy"
threads_mode = len(sys.argv) > 3 and sys.argv[3] == "threads"
lists_mode = len(sys.argv) > 3 and sys.argv[3] == "lists"
ties_mode = len(sys.argv) > 3 and sys.argv[3] == "ties"
mstg_mode = len(sys.argv) > 3 and sys.argv[3] == "mstg"
streams_mode = len(sys.argv) > 3 and sys.argv[3] == "streams"
import numpy as np


def wide_case(seed):
    """A wider draw than the test's generator: padded dimensions up to 2048, odd dimensions, up to 300 lists and
    30 000 vectors, top_k around the 63/64 heap switch, random id filters (also shorter than the id range), and the
    device encoder in place of the CPU builder's arrays."""
    rng = np.random.default_rng(seed)
    rot = int(rng.integers(0, 4) != 0)
    dim = int(rng.choice([16, 24, 33, 40, 63, 64, 65, 100, 128, 130, 200, 256, 300, 384, 512, 700, 768, 960, 1024, 1100,
                          1536, 2000, 2048])) if rot == 1 else int(rng.choice([16, 32, 48, 64, 96, 128, 192, 256]))
    bits = int(rng.choice([1, 3, 7]))
    metric = int(rng.integers(0, 2))
    nlist = int(rng.integers(2, 300))
    nmax = max(nlist + 1, min(30000, 6_000_000 // dim))
    n = int(rng.integers(nlist, nmax))
    nq = int(rng.integers(1, 64))
    top_k = int(rng.choice([1, 2, 5, 10, 17, 63, 64, 65, 100, 128, 129, 200, 255, 256, 257, 300]))  # register top-k: 1 / 2 / 4 per lane, LDS above 256
    nprobe = int(rng.integers(1, nlist + 3))
    filt = float(rng.choice([0.0, 0.0, 0.0, 0.01, 0.1, 0.5, 0.9]))
    short = bool(rng.integers(0, 2))
    enc = bool(rng.integers(0, 3) == 0)
    rbq1 = (not enc) and bool(rng.integers(0, 3) == 0)  # through save_rbq1 -> rbq_index_load_rbq1
    raw_ip = bool(rng.integers(0, 3) == 0)    # inner product over vectors that are NOT normalised
    uniform = bool(rng.integers(0, 6) == 0)   # all-positive uniform data instead of the mixture
    return dict(raw_ip=raw_ip, uniform=uniform, rbq1=rbq1, n=n, dim=dim, nlist=nlist, bits=bits, metric=metric, rot=rot, nq=nq, top_k=top_k, nprobe=nprobe,
                filt=filt, short=short, enc=enc)


stats = dict(queries=0, results=0, encoder=0, rbq1=0, filtered=0)


def streams_case(seed):
    """Larger indexes searched on four caller streams at once (bench.py's pipelining), several rounds: kernels of
    different batches share CUs, which is where cross-kernel interference would show."""
    rng = np.random.default_rng(seed)
    dim = int(rng.choice([128, 384, 768, 960, 960, 1536]))
    bits = int(rng.choice([1, 3, 7, 7]))
    metric = int(rng.integers(0, 2))
    nlist = int(rng.choice([64, 128, 256, 512]))
    n = int(rng.integers(20000, max(20001, min(150000, 60_000_000 // dim))))
    nq = int(rng.choice([128, 256, 512, 1024]))
    top_k = int(rng.choice([1, 10, 10, 100]))
    nprobe = int(rng.choice([8, 16, 32, 64]))
    return dict(n=n, dim=dim, nlist=nlist, bits=bits, metric=metric, nq=nq, top_k=top_k, nprobe=nprobe)


def ties_case(seed):
    """Indexes made of a few distinct vectors repeated many times: every distance occurs in runs of equal values, so the
    top-k goes through the exact-heap re-run (the reference's BinaryHeap order decides which of the tied ids stay)."""
    rng = np.random.default_rng(seed)
    dim = int(rng.choice([32, 64, 128, 200, 384, 960]))
    nlist = int(rng.integers(2, 40))
    n = int(rng.integers(max(200, nlist * 4), 6000))
    return dict(dim=dim, nlist=nlist, n=n, distinct=int(rng.integers(3, max(4, n // 8))), bits=int(rng.choice([1, 3, 7])),
                metric=int(rng.integers(0, 2)), nq=int(rng.integers(1, 40)), top_k=int(rng.choice([1, 2, 5, 10, 17, 63, 64, 100, 128, 129, 255, 256, 300])),
                nprobe=int(rng.integers(1, nlist + 1)), from_data=bool(rng.integers(0, 2)))


def run_ties(seed):
    import rabitq_rs_amd as rq
    c = ties_case(seed)
    rng = np.random.default_rng(seed + 3)
    base = conftest.make_dataset(c["distinct"], c["dim"], max(c["nlist"] // 2, 1), seed, normalize=(c["metric"] == 1))
    data = base[rng.integers(0, c["distinct"], c["n"])].copy()
    data, built = conftest.build_index(n=c["n"], dim=c["dim"], nlist=c["nlist"], total_bits=c["bits"], metric=c["metric"],
                                       rotator=1, seed=seed, normalize=False, data=data)
    idx = rq.IvfRabitqIndex.from_built(built)
    # queries: members of the index (distance ties at zero as well) or fresh draws
    q = data[rng.integers(0, c["n"], c["nq"])].copy() if c["from_data"] else \
        conftest.make_dataset(c["nq"], c["dim"], max(c["nlist"] // 2, 1), seed + 1000, normalize=(c["metric"] == 1))
    ids, sc, cnt = t._compare(built, idx, q, c["top_k"], c["nprobe"])
    stats["queries"] += len(q)
    stats["results"] += int(cnt.sum())
    stats["restarts"] = stats.get("restarts", 0) + int(idx.heap_restarts())
    idx.close()


def lists_case(seed):
    """Many lists and large nprobe: the three row modes of k_select_mfma (scores in registers up to 4096 lists, LDS row up
    to 16384, global row beyond), shortlists past 512 entries (bitonic sort instead of the rank sort), lists of 0-3
    vectors."""
    rng = np.random.default_rng(seed)
    nlist = int(rng.choice([300, 1000, 3000, 4096, 4097, 6000, 12000, 16384, 16385, 20000]))
    dim = int(rng.choice([16, 32, 64, 128]))
    n = int(nlist * rng.choice([1, 2, 4]) + rng.integers(0, 500))
    nprobe = int(min(nlist, rng.choice([1, 7, 64, 128, 255, 256, 257, 300, 600, 1000, 2048])))
    return dict(nlist=nlist, dim=dim, n=n, nprobe=nprobe, bits=int(rng.choice([1, 3, 7])), metric=int(rng.integers(0, 2)),
                rot=int(rng.integers(0, 4) != 0), nq=int(rng.integers(1, 24)), top_k=int(rng.choice([1, 10, 64, 100])))


def run_lists(seed):
    import rabitq_rs_amd as rq
    c = lists_case(seed)
    data, built = conftest.build_index(n=c["n"], dim=c["dim"], nlist=c["nlist"], total_bits=c["bits"], metric=c["metric"],
                                       rotator=c["rot"], seed=seed, normalize=(c["metric"] == 1))
    idx = rq.IvfRabitqIndex.from_built(built)
    q = conftest.make_dataset(c["nq"], c["dim"], max(c["nlist"] // 4, 1), seed + 1000, normalize=(c["metric"] == 1))
    ids, sc, cnt = t._compare(built, idx, q, c["top_k"], c["nprobe"])
    stats["queries"] += len(q)
    stats["results"] += int(cnt.sum())
    stats["fallbacks"] = stats.get("fallbacks", 0) + int(idx.rank_fallbacks())
    idx.close()


def threads_case(seed):
    """One handle, three caller threads inside rbq_search_batch at the same time, each with its own batch size, top_k
    and nprobe (the handle is re-entrant: every call takes a workspace and a stream from the pool)."""
    rng = np.random.default_rng(seed)
    dim = int(rng.choice([64, 128, 384, 960]))
    nlist = int(rng.integers(8, 200))
    n = int(rng.integers(nlist * 8, max(nlist * 8 + 1, min(40000, 8_000_000 // dim))))
    calls = [dict(nq=int(rng.choice([1, 3, 17, 64, 200, 700])), top_k=int(rng.choice([1, 10, 64, 100])),
                  nprobe=int(rng.integers(1, nlist + 1))) for _ in range(3)]
    return dict(dim=dim, nlist=nlist, n=n, bits=int(rng.choice([1, 3, 7])), metric=int(rng.integers(0, 2)), calls=calls)


def run_threads(seed):
    import threading
    import rabitq_rs_amd as rq
    c = threads_case(seed)
    data, built = conftest.build_index(n=c["n"], dim=c["dim"], nlist=c["nlist"], total_bits=c["bits"], metric=c["metric"],
                                       rotator=1, seed=seed, normalize=(c["metric"] == 1))
    idx = rq.IvfRabitqIndex.from_built(built)
    qs = [conftest.make_dataset(k["nq"], c["dim"], max(c["nlist"] // 4, 1), seed + 1000 + i, normalize=(c["metric"] == 1))
          for i, k in enumerate(c["calls"])]
    want = [oracle.search_batch(built, qs[i], k["top_k"], k["nprobe"]) for i, k in enumerate(c["calls"])]
    got = [None] * 3
    errs = []

    def work(i):
        try:
            k = c["calls"][i]
            for _ in range(4):
                got[i] = idx.batch_search_raw(qs[i], rq.SearchParams(k["top_k"], k["nprobe"]))
                ids, sc, cnt = got[i][:3]
                rc, oids, osc, ocnt = want[i][:4]
                assert rc == 0 and np.array_equal(cnt, ocnt), "counts differ (thread %d)" % i
                assert np.array_equal(ids, oids), "ids differ (thread %d)" % i
                for qi in range(len(qs[i])):
                    kk = int(cnt[qi])
                    np.testing.assert_allclose(sc[qi, :kk], osc[qi, :kk], rtol=t.RTOL, atol=0)
        except BaseException as e:  # re-raised in the main thread
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    if errs:
        raise errs[0]
    stats["queries"] += 4 * sum(k["nq"] for k in c["calls"])
    idx.close()


def mstg_case(seed):
    rng = np.random.default_rng(seed)
    return dict(metric=int(rng.integers(0, 2)), bits=int(rng.choice([1, 3, 7])),
                dim=int(rng.choice([64, 128, 192, 256, 384, 768])), nlist=int(rng.integers(3, 120)),
                n=int(rng.integers(300, 20000)), nq=int(rng.integers(2, 60)), top_k=int(rng.choice([1, 5, 10, 64, 100, 300])))


def run_mstg(seed):
    c = mstg_case(seed)
    built, q, lists, counts = t._mstg_case(c["metric"], c["bits"], dim=c["dim"], n=max(c["n"], c["nlist"] * 2), nlist=c["nlist"],
                                           nq=c["nq"], seed=seed)
    t._mstg_compare(built, q, lists, counts, c["metric"], top_ks=(c["top_k"],))
    stats["queries"] += len(q)


def run_streams(seed):
    import rabitq_rs_amd as rq
    c = streams_case(seed)
    dev = torch.device("cuda", 0)
    data, built = conftest.build_index(n=c["n"], dim=c["dim"], nlist=c["nlist"], total_bits=c["bits"], metric=c["metric"],
                                       rotator=1, seed=seed, normalize=(c["metric"] == 1))
    idx = rq.IvfRabitqIndex.from_built(built)
    nq, top_k, nprobe, ns = c["nq"], c["top_k"], c["nprobe"], 4
    q = conftest.make_dataset(nq, c["dim"], max(c["nlist"] // 4, 1), seed + 1000, normalize=(c["metric"] == 1))
    rc, oids, osc, ocnt, _ = oracle.search_batch(built, q, top_k, nprobe)
    assert rc == 0
    qd = torch.from_numpy(q).to(dev)
    ss = [torch.cuda.Stream(dev) for _ in range(ns)]
    d_ids = [torch.zeros(nq, top_k, dtype=torch.int64, device=dev) for _ in range(ns)]
    d_sc = [torch.zeros(nq, top_k, dtype=torch.float32, device=dev) for _ in range(ns)]
    d_cnt = [torch.zeros(nq, dtype=torch.int32, device=dev) for _ in range(ns)]
    torch.cuda.synchronize(dev)
    for rep in range(6):
        for i in range(ns):
            idx.search_batch_device(qd.data_ptr(), nq, c["dim"], top_k, nprobe, d_ids[i].data_ptr(), d_sc[i].data_ptr(),
                                    d_cnt[i].data_ptr(), stream=ss[i].cuda_stream)
        torch.cuda.synchronize(dev)
        for i in range(ns):
            got = d_ids[i].cpu().numpy().view(np.uint64)
            cnt = d_cnt[i].cpu().numpy().view(np.uint32)
            assert np.array_equal(cnt, ocnt), f"rep {rep} stream {i}: counts differ"
            bad = np.nonzero((got != oids).any(axis=1))[0]
            assert bad.size == 0, f"rep {rep} stream {i}: ids differ for queries {bad[:10]}"
            sc = d_sc[i].cpu().numpy()
            for qi in range(nq):
                k = int(cnt[qi])
                np.testing.assert_allclose(sc[qi, :k], osc[qi, :k], rtol=t.RTOL, atol=0)
            stats["queries"] += nq
            stats["results"] += int(cnt.sum())
    idx.close()


def lazy_case(seed):
    """In-distribution queries (perturbed data points: the lists near a query are really near, the far ones are dropped by the
    lazy probe selection) over both metrics, raw and normalised inner product, all bit widths, clustered and uniform data,
    large nprobe, top_k from 1 to past what the head lists hold; the queries of every other mode come from a different
    mixture and drop nothing."""
    rng = np.random.default_rng(seed)
    rot = int(rng.integers(0, 5) != 0)
    dim = int(rng.choice([16, 40, 64, 100, 128, 200, 256, 384, 512, 768, 960, 1024, 1536])) if rot == 1 else int(rng.choice([16, 32, 64, 96, 128]))
    nlist = int(rng.choice([8, 20, 50, 100, 200, 300, 600, 1000, 4097, 17000]))
    nmax = max(nlist * 3, min(40000, 8_000_000 // dim))
    n = int(rng.integers(nlist * 2, nmax))
    if nlist >= 4097:
        dim = min(dim, 128)
        n = int(nlist * rng.choice([2, 3]))
    return dict(rot=rot, dim=dim, nlist=nlist, n=n, bits=int(rng.choice([1, 3, 7])), metric=int(rng.integers(0, 2)),
                raw_ip=bool(rng.integers(0, 3) == 0), uniform=bool(rng.integers(0, 8) == 0), nq=int(rng.integers(1, 48)),
                top_k=int(rng.choice([1, 2, 5, 10, 10, 17, 64, 100, 200, 300, 1000])),
                nprobe=int(min(nlist, rng.choice([1, 2, 4, 8, 16, 32, 64, 128, 300, 600, 2000]))),
                noise=float(rng.choice([0.0, 0.01, 0.05, 0.2, 1.0])), gen=int(rng.choice([2, 4, 8, 50])))


def run_lazy(seed):
    import rabitq_rs_amd as rq
    c = lazy_case(seed)
    norm = c["metric"] == 1 and not c["raw_ip"]
    data = conftest.make_dataset(c["n"], c["dim"], max(c["nlist"] // c["gen"], 1), seed, normalize=norm, uniform=c["uniform"])
    data, built = conftest.build_index(n=c["n"], dim=c["dim"], nlist=c["nlist"], total_bits=c["bits"], metric=c["metric"],
                                       rotator=c["rot"], seed=seed, data=data)
    idx = rq.IvfRabitqIndex.from_built(built)
    rng = np.random.default_rng(seed + 11)
    q = data[rng.integers(0, c["n"], c["nq"])] + c["noise"] * rng.standard_normal((c["nq"], c["dim"])).astype(np.float32)
    if norm:
        q = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-20)
    q = np.ascontiguousarray(q, dtype=np.float32)
    ids, sc, cnt = t._compare(built, idx, q, c["top_k"], c["nprobe"])
    idx.set_option("lazy_select", 0)
    ids2, sc2, cnt2, _ = idx.batch_search_raw(q, rq.SearchParams(c["top_k"], c["nprobe"]))
    assert np.array_equal(ids2, ids) and np.array_equal(cnt2, cnt) and np.array_equal(sc2.view(np.uint32), sc.view(np.uint32)), "eager selection differs"
    stats["queries"] += len(q)
    stats["results"] += int(cnt.sum())
    idx.close()


def run_wide(seed):
    import rabitq_rs_amd as rq
    c = wide_case(seed)
    norm = c["metric"] == 1 and not c["raw_ip"]
    data, built = conftest.build_index(n=c["n"], dim=c["dim"], nlist=c["nlist"], total_bits=c["bits"], metric=c["metric"],
                                       rotator=c["rot"], seed=seed, normalize=norm, uniform=c["uniform"])
    if c["enc"]:
        import torch
        # build_index clusters internally; redo the clustering here so that the assignment is known
        cent, assign = rq.builder.kmeans(data, c["nlist"], 5, seed + 7)
        built = rq.builder.train_with_clusters(data, cent, assign, c["bits"], c["metric"], c["rot"], seed + 8, True)
        xd = torch.from_numpy(data).cuda()
        ad = torch.from_numpy(assign.astype(np.int32)).cuda()
        idx = rq.IvfRabitqIndex.build_on_device(built.hdr_ptr, cent, xd.data_ptr(), ad.data_ptr(), c["n"], built.t_const)
    elif c["rbq1"]:
        idx = rq.IvfRabitqIndex.load_from_bytes(built.save_rbq1())
    else:
        idx = rq.IvfRabitqIndex.from_built(built)
    q = conftest.make_dataset(c["nq"], c["dim"], max(c["nlist"] // 4, 1), seed + 1000, normalize=norm, uniform=c["uniform"])
    words, nbits = None, 0
    if c["filt"] > 0.0:
        rng = np.random.default_rng(seed + 5)
        allowed = rng.choice(c["n"], max(1, int(c["filt"] * c["n"])), replace=False)
        nbits = int(allowed.max()) + 1
        if c["short"]:
            nbits = max(1, nbits // 2)
            allowed = allowed[allowed < nbits]
        words = np.zeros((nbits + 31) // 32, np.uint32)
        if allowed.size:
            np.bitwise_or.at(words, allowed >> 5, (np.uint32(1) << (allowed & 31).astype(np.uint32)))
    ids, sc, cnt = t._compare(built, idx, q, c["top_k"], c["nprobe"], words, nbits)
    if words is not None:
        # without the diagnostic counters a filtered search keeps the block-level bound (k_scan: bound_ok): same results
        ids2, sc2, cnt2 = idx.batch_search_raw(q, rq.SearchParams(c["top_k"], c["nprobe"]), words, nbits)[:3]
        assert np.array_equal(cnt2, cnt), "counts differ (filtered, no diagnostics)"
        assert np.array_equal(ids2, ids), "ids differ (filtered, no diagnostics)"
        assert np.array_equal(sc2.view(np.uint32), sc.view(np.uint32)), "scores differ (filtered, no diagnostics)"
    idx.close()
    stats["queries"] += len(q)
    stats["results"] += int(cnt.sum())
    stats["encoder"] += int(c["enc"])
    stats["rbq1"] += int(c["rbq1"])
    stats["filtered"] += int(c["filt"] > 0.0)

bad, harness = [], []
t0 = time.time()
for seed in range(first, last):
    try:
        if lazy_mode:
            run_lazy(seed)
        elif threads_mode:
            run_threads(seed)
        elif lists_mode:
            run_lists(seed)
        elif ties_mode:
            run_ties(seed)
        elif mstg_mode:
            run_mstg(seed)
        elif streams_mode:
            run_streams(seed)
        elif wide:
            run_wide(seed)
        else:
            t.test_random_configurations_match_oracle(seed)
        harness = []
    except AssertionError:
        bad.append(seed)
        harness = []
        print("MISMATCH seed", seed, (lazy_case(seed) if lazy_mode else threads_case(seed) if threads_mode else lists_case(seed) if lists_mode else ties_case(seed) if ties_mode else mstg_case(seed) if mstg_mode else streams_case(seed) if streams_mode else wide_case(seed)) if wide else t._random_case(seed), traceback.format_exc().splitlines()[-1][:300], flush=True)
    except Exception:
        msg = traceback.format_exc().splitlines()[-1][:300]
        print("ERROR seed", seed, (lazy_case(seed) if lazy_mode else threads_case(seed) if threads_mode else lists_case(seed) if lists_mode else ties_case(seed) if ties_mode else mstg_case(seed) if mstg_mode else streams_case(seed) if streams_mode else wide_case(seed)) if wide else t._random_case(seed), msg, flush=True)
        if os.environ.get("SOAK_TB"):
            traceback.print_exc()
        harness.append(msg)
        if len(harness) >= 3 and len(set(harness[-3:])) == 1:
            print("aborting: harness error, no parity information in this run")
            sys.exit(2)
    if (seed - first) % 50 == 49:
        print("... %d seeds, %d mismatches, %.0f s" % (seed - first + 1, len(bad), time.time() - t0), flush=True)
print("done: %d seeds, mismatches: %s" % (last - first, bad))
if wide:
    print("compared:", stats)
sys.exit(1 if bad else 0)
