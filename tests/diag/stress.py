"""Diagnostic: repeat one full-size batch many times and compare every run with the oracle (ids exact)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import rabitq_rs_amd as rq
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle
a = bench.parse()
reps = int(os.environ.get("STRESS_REPS", "30"))
dev = torch.device("cuda", 0)
x = bench.mixture(torch, dev, a.n, a.dim, a.nlist, 20260105, False)
cent, assign = bench.kmeans_gpu(torch, x, a.nlist, 6, 20260103)
built = rq.builder.train_with_clusters(x.cpu().numpy(), cent.cpu().numpy(), assign.cpu().numpy().astype(np.uint32), a.bits, 0, 1, 20260104, True)
idx = rq.IvfRabitqIndex.from_built(built)
for kv in filter(None, os.environ.get('STRESS_OPTS', '').split(',')):
    k, v = kv.split('=')
    idx.set_option(k, int(v))
q = bench.mixture(torch, dev, a.batch, a.dim, a.nlist, 20260102, False).cpu().numpy()
rc, oids, osc, ocnt, odiag = oracle.search_batch(built, q, a.top_k, a.nprobe, want_diag=True)
bad_total = 0
for r in range(0 if os.environ.get('STRESS_SKIP_SYNC') else reps):
    ids, sc, cnt, diag = idx.batch_search_raw(q, rq.SearchParams(a.top_k, a.nprobe), want_diag=(r % 2 == 0))
    bad = np.nonzero((ids != oids).any(axis=1))[0]
    if r % 2 == 0:
        badd = np.nonzero((diag.astype(np.uint64) != odiag.astype(np.uint64)).any(axis=1))[0]
    else:
        badd = []
    if len(bad) or len(badd):
        bad_total += 1
        print("rep", r, "id mismatches at queries", bad[:8], "diag mismatches", list(badd[:8]))
        for b in bad[:2]:
            print("  gpu", ids[b], sc[b]); print("  ref", oids[b], osc[b])
        for b in list(badd[:2]):
            print("  diag gpu", diag[b], "ref", odiag[b])
print("reps", reps, "bad reps", bad_total, "heap_restarts", idx.heap_restarts())

# asynchronous multi-stream path (bench.py's step): every stream's output must equal the oracle's too
ns = int(os.environ.get("STRESS_STREAMS", "3"))
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(ns - 1)]
qd = torch.from_numpy(q).to(dev)
d_ids = [torch.zeros(a.batch, a.top_k, dtype=torch.int64, device=dev) for _ in range(ns)]
d_sc = [torch.zeros(a.batch, a.top_k, dtype=torch.float32, device=dev) for _ in range(ns)]
d_cnt = [torch.zeros(a.batch, dtype=torch.int32, device=dev) for _ in range(ns)]
bad_async = 0
for r in range(reps):
    for i in range(ns):
        idx.search_batch_device(qd.data_ptr(), a.batch, a.dim, a.top_k, a.nprobe, d_ids[i].data_ptr(), d_sc[i].data_ptr(),
                                d_cnt[i].data_ptr(), stream=streams[i].cuda_stream)
    torch.cuda.synchronize(dev)
    for i in range(ns):
        got = d_ids[i].cpu().numpy().view(np.uint64)
        bad = np.nonzero((got != oids).any(axis=1))[0]
        if len(bad):
            bad_async += 1
            print("async rep", r, "stream", i, "mismatching queries", len(bad), bad[:8])
        d_ids[i].zero_()
print("async reps", reps, "streams", ns, "bad", bad_async, "rank_fallbacks", idx.rank_fallbacks())
