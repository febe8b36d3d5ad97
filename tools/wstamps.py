"""Diagnostic: run batches through a -DRBQ_WSTAMPS build (k_scanw, scanw.hpp) and print where a query's wave spends its cycles.
  RBQ_LIB_PATH=.../librbq_wst.so python tools/wstamps.py [bench args]   (WSTAMPS_LOAD=N: N other streams keep the chip busy meanwhile)"""
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
cent, assign = bench.kmeans_gpu(torch, x, a.nlist, 6, 20260103)
xs = mix.draw(max(2 * a.nlist, 8192), 99).cpu().numpy()
small = rq.builder.train_with_clusters(xs, cent.cpu().numpy(), (np.arange(len(xs)) % a.nlist).astype(np.uint32), a.bits, a.metric, 1, 20260104, True)
idx = rq.IvfRabitqIndex.build_on_device(small.hdr_ptr, cent.cpu().numpy(), x.data_ptr(), assign.to(torch.int32).contiguous().data_ptr(), a.n, small.t_const)
del x
idx.set_option("scan_wave", 1)
for kv in a.option:
    k, v = kv.split("=")
    idx.set_option(k, int(v))
qs = [mix.draw(a.batch, 20260102 + i).cpu().numpy() for i in range(3)]
idx.set_option('host_subbatch', 1 << 20)  # one sub-batch: the diag slots carry the stamps of one launch
nload = int(os.environ.get("WSTAMPS_LOAD", "0"))
stop = [False]
if nload:  # background load: device-resident batches on nload streams, as in the bench
    import threading
    qd = mix.draw(8 * a.batch, 777).contiguous().view(8, a.batch, a.dim)
    streams = [torch.cuda.Stream(dev) for _ in range(nload)]
    outs = [(torch.empty(a.batch, a.top_k, dtype=torch.int64, device=dev), torch.empty(a.batch, a.top_k, dtype=torch.float32, device=dev),
             torch.empty(a.batch, dtype=torch.int32, device=dev)) for _ in range(nload)]
    def load():
        i = 0
        while not stop[0]:
            s = i % nload
            idx.search_batch_device(qd[i % 8].data_ptr(), a.batch, a.dim, a.top_k, a.nprobe, outs[s][0].data_ptr(), outs[s][1].data_ptr(), outs[s][2].data_ptr(),
                                    stream=streams[s].cuda_stream)
            i += 1
            if i % nload == 0:
                streams[0].synchronize()
    th = threading.Thread(target=load); th.start()
    import time; time.sleep(0.5)
for q in qs:
    ids, sc, cnt, diag = idx.batch_search_raw(q, rq.SearchParams(a.top_k, a.nprobe), want_diag=True)
stop[0] = True
if nload:
    th.join()
d = diag.astype(np.uint64)
lo = lambda c: (d[:, c] & 0xffffffff).astype(np.float64); hi = lambda c: (d[:, c] >> 32).astype(np.float64)
if os.environ.get("WSTAMPS_MODE") == "2":
    nround = ((d[:, 2] >> 16) & 0xffff).astype(np.float64)
    print("k_scanw refine rounds, cycles/query: collect + permutes %.0f  loads + dot + reduce %.0f  replay %.0f  (all rounds %.0f; %.1f rounds) -> per round %.0f / %.0f / %.0f" % (
        lo(0).mean(), hi(0).mean(), lo(1).mean(), hi(1).mean(), nround.mean(), lo(0).mean() / nround.mean(), hi(0).mean() / nround.mean(), lo(1).mean() / nround.mean()))
    sys.exit(0)
total, fill, tile, rounds = lo(0), hi(0), lo(1), hi(1)
ntile = (d[:, 2] & 0xffff).astype(np.float64); nround = ((d[:, 2] >> 16) & 0xffff).astype(np.float64)
nlive = ((d[:, 2] >> 32) & 0xffff).astype(np.float64); ncand = (d[:, 2] >> 48).astype(np.float64)
print("k_scanw cycles/query (s_memtime): total %.0f (p50 %.0f p99 %.0f max %.0f)  fill %.0f  tile phases %.0f  refine rounds %.0f  rest (prologue, epilogue) %.0f" % (
    total.mean(), np.percentile(total, 50), np.percentile(total, 99), total.max(), fill.mean(), tile.mean(), rounds.mean(), (total - fill - tile - rounds).mean()))
print("per query: tiles %.1f (looked up %.1f)  candidates below the tile's threshold %.1f  refine rounds %.1f  -> per tile %.0f cycles, per round %.0f cycles" % (
    ntile.mean(), nlive.mean(), ncand.mean(), nround.mean(), tile.mean() / max(ntile.mean(), 1), rounds.mean() / max(nround.mean(), 1)))
