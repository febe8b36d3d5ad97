"""Diagnostic: pipelined rate of every stage ALONE (option stage_mask: the other stages' launches are skipped, the workspaces keep
what the last full call wrote) on the bench's own protocol — 32 distinct batches, N streams, device-resident.  Tells which
stage the overlapped step time is made of:  1 / rate(all) against  sum_s 1 / rate(s).
  python tools/stage_rates.py [bench args] ; env STAGE_MASKS="15,8,4,2,1,12,3" STAGE_STREAMS="12" STAGE_STEPS=400"""
import os, sys, time
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
for kv in a.option:
    k, v = kv.split("=")
    idx.set_option(k, int(v))
NB = a.nbatches
q_all = mix.draw(NB * a.batch, 20260102).contiguous().view(NB, a.batch, a.dim)
masks = [int(m, 0) for m in os.environ.get("STAGE_MASKS", "15,8,4,2,1,12,3").split(",")]
steps = int(os.environ.get("STAGE_STEPS", "400"))
names = {1: "prep", 2: "rank", 4: "select", 8: "scan"}
for ns in [int(s) for s in os.environ.get("STAGE_STREAMS", str(a.streams)).split(",")]:
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    d_ids = torch.empty(2 * ns, a.batch, a.top_k, dtype=torch.int64, device=dev)
    d_sc = torch.empty(2 * ns, a.batch, a.top_k, dtype=torch.float32, device=dev)
    d_cnt = torch.empty(2 * ns, a.batch, dtype=torch.int32, device=dev)
    cnt = [0]
    def step():
        i = cnt[0]; cnt[0] += 1
        s, slot = i % ns, i % (2 * ns)
        idx.search_batch_device(q_all[i % NB].data_ptr(), a.batch, a.dim, a.top_k, a.nprobe, d_ids[slot].data_ptr(), d_sc[slot].data_ptr(),
                                d_cnt[slot].data_ptr(), stream=streams[s].cuda_stream)
    idx.set_option("stage_mask", 15)
    torch.cuda.synchronize(dev)
    for _ in range(4 * ns):
        step()
    torch.cuda.synchronize(dev)
    res = {}
    for m in masks:
        idx.set_option("stage_mask", m)
        for _ in range(2 * ns):
            step()
        torch.cuda.synchronize(dev)
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize(dev)
            ts.append((time.perf_counter() - t0) / steps)
        res[m] = sorted(ts)[1]
        label = "+".join(n for b, n in names.items() if m & b)
        print("streams %2d  mask %2d %-22s  %.4f ms/step  %.2f M queries/s" % (ns, m, label, res[m] * 1e3, a.batch / res[m] / 1e6), flush=True)
    idx.set_option("stage_mask", 15)
    if all(k in res for k in (1, 2, 4, 8, 15)):
        print("streams %2d  sum of the four alone %.4f ms  against all together %.4f ms" % (ns, sum(res[k] for k in (1, 2, 4, 8)) * 1e3, res[15] * 1e3), flush=True)
    for st in streams:
        idx.release_stream(st.cuda_stream)
