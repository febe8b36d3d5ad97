"""Diagnostic: one seed of the MSTG posting-scan soak, with the differing rows printed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import conftest  # noqa: F401
import numpy as np
import oracle
import rabitq_rs_amd as rq
import test_gpu_parity as t
sys.argv = [sys.argv[0], "0", "0", "mstg"] + sys.argv[1:]
seed = int(sys.argv[4])
rng = np.random.default_rng(seed)
c = dict(metric=int(rng.integers(0, 2)), bits=int(rng.choice([1, 3, 7])), dim=int(rng.choice([64, 128, 192, 256, 384, 768])),
         nlist=int(rng.integers(3, 120)), n=int(rng.integers(300, 20000)), nq=int(rng.integers(2, 60)),
         top_k=int(rng.choice([1, 5, 10, 64, 100, 300])))
print(c)
built, q, lists, counts = t._mstg_case(c["metric"], c["bits"], dim=c["dim"], n=max(c["n"], c["nlist"] * 2), nlist=c["nlist"], nq=c["nq"], seed=seed)
idx = rq.IvfRabitqIndex.from_built(built)
top_k, metric = c["top_k"], c["metric"]
rc, oids, osc, ocnt = oracle.posting_scan_batch(built, q, top_k, lists, counts)
ids, sc, cnt = idx.posting_scan(q, top_k, lists, counts)
print("counts equal:", np.array_equal(cnt, ocnt), "cnt[0]", cnt[0])
for i in range(len(q)):
    k = int(cnt[i])
    if k == 0 or cnt[i] != ocnt[i]:
        if cnt[i] != ocnt[i]:
            print("query", i, "count", cnt[i], "oracle", ocnt[i])
        continue
    a = sc[i, :k].view(np.uint32) & (0x7fffffff if metric == 0 else 0xffffffff)
    b = osc[i, :k].view(np.uint32) & (0x7fffffff if metric == 0 else 0xffffffff)
    if not np.array_equal(a, b):
        j = np.nonzero(a != b)[0]
        print("query", i, "score bits differ at", j[:8], "gpu", sc[i, j[:4]], "oracle", osc[i, j[:4]], "ids", ids[i, j[:4]], oids[i, j[:4]])
    if not (np.diff(sc[i, :k]) >= 0).all():
        print("query", i, "not sorted at", np.nonzero(np.diff(sc[i, :k]) < 0)[0][:5])
    if metric == 0 and not (sc[i, :k] >= 0).all():
        print("query", i, "negative L2 at", np.nonzero(sc[i, :k] < 0)[0][:5], sc[i, :k][sc[i, :k] < 0][:5])
    uniq = np.ones(k, bool)
    uniq[1:] &= sc[i, 1:k] != sc[i, :k - 1]
    uniq[:-1] &= sc[i, :k - 1] != sc[i, 1:k]
    if not np.array_equal(ids[i, :k][uniq], oids[i, :k][uniq]):
        j = np.nonzero(ids[i, :k] != oids[i, :k])[0]
        print("query", i, "ids differ at unique distances:", j[:8], "gpu", ids[i, j[:4]], sc[i, j[:4]], "oracle", oids[i, j[:4]], osc[i, j[:4]], "count", k, "lists", counts[i])
    if sorted(ids[i, :k].tolist()) != sorted(oids[i, :k].tolist()):
        sa, sb = set(ids[i, :k].tolist()), set(oids[i, :k].tolist())
        print("query", i, "id SETS differ: only gpu", sorted(sa - sb)[:6], "only oracle", sorted(sb - sa)[:6], "k", k,
              "last gpu", sc[i, k - 3:k], "last oracle", osc[i, k - 3:k])
