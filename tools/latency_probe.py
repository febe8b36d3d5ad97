"""Diagnostic (GPU box): where the time of ONE small rbq_search_batch call goes — host phases (host_trace) and the four
kernels alone (profile taps) — at nq in {1, 8, 64, 256} on the headline index.  Usage: python tools/latency_probe.py [bench args]"""
import os, sys, time, ctypes as C
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
lib = rq.index.lib()
for kv in a.option:
    k, v = kv.split("=")
    idx.set_option(k, int(v))
q = mix.draw(4096, 20260102).cpu().numpy()
for nq in [int(v) for v in os.environ.get('LAT_PROBE_NQ', '1,8,64,256').split(',')]:
    nsets = 16
    pin = []
    for j in range(nsets):
        p = [lib.rbq_host_alloc(nq * a.dim * 4), lib.rbq_host_alloc(nq * a.top_k * 8), lib.rbq_host_alloc(nq * a.top_k * 4), lib.rbq_host_alloc(nq * 4)]
        C.memmove(p[0], q[j * nq:(j + 1) * nq].ctypes.data, nq * a.dim * 4)
        pin.append(p)
    call = lambda j: lib.rbq_search_batch(idx._h, pin[j][0], nq, a.dim, a.top_k, a.nprobe, None, 0, pin[j][1], pin[j][2], pin[j][3], None)
    for j in range(nsets):
        call(j)
    ts = []
    for r in range(int(os.environ.get('LAT_PROBE_CALLS', '200'))):
        t0 = time.perf_counter(); call(r % nsets); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    idx.profile_begin()
    for r in range(50):
        call(r % nsets)
    idx.profile_end()
    st = {s: round(idx.profile_stage(s)[0] * 1e3, 1) for s in ("prep", "rank", "select", "scan")}
    print("nq %4d: call p50 %.1f us p99 %.1f us; kernels alone (us): %s sum %.1f" % (nq, np.percentile(ts, 50), np.percentile(ts, 99), st, sum(st.values())))
    if os.environ.get("RBQ_PROBE_TRACE"):
        idx.set_option("host_trace", 1)
        for r in range(3):
            call(r)
        idx.set_option("host_trace", 0)
