"""Diagnostic (GPU box): how many probed LISTS have at least one block that can pass the threshold?  Reads the block stream
(every entry carries the block-level lower bound k_select wrote) of one cold batch and compares each list's smallest bound
with the query's final k-th distance T (and with 2T / 4T: what a looser, select-time upper bound of T would still prune).
Usage: python tools/live_lists.py [bench args]   (same workload arguments as bench.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import rabitq_rs_amd as rq

a = bench.parse()
dev = torch.device("cuda", 0)
mix = bench.Mixture(torch, dev, a.dim, a.nlist, a.dataset, a.metric == 1)
x = mix.draw(a.n, 20260105)
cent, assign = bench.kmeans_gpu(torch, x, a.nlist, a.kmeans_iters, 20260103)
xs = mix.draw(max(2 * a.nlist, 4096), 99).cpu().numpy()
small = rq.builder.train_with_clusters(xs, cent.cpu().numpy(), (np.arange(len(xs)) % a.nlist).astype(np.uint32), a.bits, a.metric, 1, 20260104, True)
idx = rq.IvfRabitqIndex.build_on_device(small.hdr_ptr, cent.cpu().numpy(), x.data_ptr(), assign.to(torch.int32).contiguous().data_ptr(), a.n, small.t_const)
del x
nq = min(a.batch, 256)
q = mix.draw(nq, 20260102).contiguous()
s = torch.cuda.Stream(dev)
o = (torch.zeros(nq, a.top_k, dtype=torch.int64, device=dev), torch.zeros(nq, a.top_k, dtype=torch.float32, device=dev), torch.zeros(nq, dtype=torch.int32, device=dev))
idx.search_batch_device(q.data_ptr(), nq, a.dim, a.top_k, a.nprobe, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), stream=s.cuda_stream)
torch.cuda.synchronize(dev)
ln = idx.debug_copy_index("list_n", np.empty(a.nlist, np.uint32))
nb = np.sort((ln.astype(np.int64) + 31) // 32)[::-1]
stride = max(int(nb[:min(a.nprobe, a.nlist)].sum()), 1)
ns = idx.debug_copy_workspace(s.cuda_stream, "nstream", np.empty(nq, np.uint32))
wl = idx.debug_copy_workspace(s.cuda_stream, "wl", np.empty((nq, stride, 4), np.uint32))
sc = o[1].cpu().numpy()
cnt = o[2].cpu().numpy()
live = {1: [], 2: [], 4: []}
liveb = {1: [], 2: [], 4: []}
nl = []
for i in range(nq):
    if cnt[i] < a.top_k:
        continue
    T = abs(float(sc[i, a.top_k - 1])) if a.metric == 0 else -float(sc[i, a.top_k - 1])
    e = wl[i, :ns[i]]
    rank = e[:, 1] >> 6
    lb = e[:, 2].view(np.float32)
    nl.append(len(np.unique(rank)))
    for f in live:
        thr = T * f if T > 0 else T / f
        m = lb < thr
        live[f].append(len(np.unique(rank[m])))
        liveb[f].append(int(m.sum()))
print("queries %d, probed lists/query %.1f, blocks/query %.1f" % (len(nl), np.mean(nl), ns.mean()))
for f in live:
    print("threshold %d x final T: live lists/query mean %.1f p90 %.0f max %d (%.1f %% of probed); live blocks/query mean %.1f (%.2f %%)" % (
        f, np.mean(live[f]), np.percentile(live[f], 90), np.max(live[f]), 100 * np.mean(live[f]) / np.mean(nl), np.mean(liveb[f]), 100 * np.mean(liveb[f]) / ns.mean()))
