"""Diagnostic: repeat one full-size batch many times and compare every run with the oracle (ids exact)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import rabitq_rs_amd as rq
from oracle import oracle
a = bench.parse()
reps = int(os.environ.get("STRESS_REPS", "30"))
dev = torch.device("cuda", 0)
x = bench.mixture(torch, dev, a.n, a.dim, a.nlist, 20260105, False)
cent, assign = bench.kmeans_gpu(torch, x, a.nlist, 6, 20260103)
built = rq.builder.train_with_clusters(x.cpu().numpy(), cent.cpu().numpy(), assign.cpu().numpy().astype(np.uint32), a.bits, 0, 1, 20260104, True)
idx = rq.IvfRabitqIndex.from_built(built)
q = bench.mixture(torch, dev, a.batch, a.dim, a.nlist, 20260102, False).cpu().numpy()
rc, oids, osc, ocnt, odiag = oracle.search_batch(built, q, a.top_k, a.nprobe, want_diag=True)
bad_total = 0
for r in range(reps):
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
