"""Diagnostic (GPU box): distribution of the lazy probe selection's per-query work on a bench-shaped index: lists scored beyond
the head, lists scanned, T_ub / final T.  Usage: python tools/lazy_taps.py [bench args]"""
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
nq = a.batch
q = mix.draw(nq, 20260102).contiguous()
s = torch.cuda.Stream(dev)
o = (torch.zeros(nq, a.top_k, dtype=torch.int64, device=dev), torch.zeros(nq, a.top_k, dtype=torch.float32, device=dev), torch.zeros(nq, dtype=torch.int32, device=dev))
idx.search_batch_device(q.data_ptr(), nq, a.dim, a.top_k, a.nprobe, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), stream=s.cuda_stream)
torch.cuda.synchronize(dev)
ds = idx.debug_copy_workspace(s.cuda_stream, "dead_skipped", np.empty((4, nq), np.uint32))
ns = idx.debug_copy_workspace(s.cuda_stream, "nstream", np.empty(nq, np.uint32))
tub = ds[2].view(np.float32)
sc = np.abs(o[1].cpu().numpy()[:, a.top_k - 1])
scored = (ds[3] >> 20).astype(np.int64)
z0 = (ds[3] & 1023).astype(np.int64); nsh = ((ds[3] >> 10) & 1023).astype(np.int64)
pc = lambda v: "mean %.1f p50 %d p90 %d p99 %d max %d" % (v.mean(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max())
print("lazy path entered (T_ub finite): %.1f %% of %d queries" % (100 * np.isfinite(tub).mean(), nq))
print("lists scored beyond the head:", pc(scored))
print("lists scanned:", pc(ds[1].astype(np.int64)), " of nprobe", a.nprobe)
print("stream entries:", pc(ns.astype(np.int64)))
print("shortlist:", pc(nsh), " certain:", pc(z0))
ok = np.isfinite(tub) & (sc > 0)
print("T_ub / final k-th distance: mean %.2f p90 %.2f" % ((tub[ok] / sc[ok]).mean(), np.percentile(tub[ok] / sc[ok], 90)))
