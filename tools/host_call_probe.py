"""Per-call time of rbq_search_batch (host buffers in and out, ONE caller thread) against the host-path switches:
zero-copy query reads, staging helper threads (pageable queries), sub-batch size x lanes.  Median / p10 of many calls (the box's host CPUs are shared:
single regions scatter), ids checked against the device entry.
python tools/host_call_probe.py [n] [nlist]        HOST_PROBE_SHAPES="256x4,512x2" HOST_PROBE_NQ="1024,4096" HOST_PROBE_TOPK=100 HOST_PROBE_OPTIONS="exact_heap=1"
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402
import bench  # noqa: E402
import rabitq_rs_amd as rq  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dim, top_k, nprobe = 960, int(os.environ.get("HOST_PROBE_TOPK", "10")), 128
dev = torch.device("cuda", 0)
mix = bench.Mixture(torch, dev, dim, nlist, "mixture_id32", False)
x = mix.draw(n, 20260105)
cent, assign = bench.kmeans_gpu(torch, x, nlist, 4, 20260103)
xs = mix.draw(8192, 99).cpu().numpy()
small = rq.builder.train_with_clusters(xs, cent.cpu().numpy(), (np.arange(8192) % nlist).astype(np.uint32), 7, 0, 1, 20260104, True)
idx = rq.IvfRabitqIndex.build_on_device(small.hdr_ptr, cent.cpu().numpy(), x.data_ptr(), assign.to(torch.int32).contiguous().data_ptr(), n,
                                        small.t_const)
del x
for kv in os.environ.get("HOST_PROBE_OPTIONS", "").split():
    k_, v_ = kv.split("=")
    idx.set_option(k_, int(v_))
lib = rq.index.lib()
NSETS = 8


def device_ids(q):
    nq = q.shape[0]
    d_i = torch.empty(nq, top_k, dtype=torch.int64, device=dev)
    d_s = torch.empty(nq, top_k, dtype=torch.float32, device=dev)
    d_c = torch.empty(nq, dtype=torch.int32, device=dev)
    qd = torch.from_numpy(q).to(dev)
    idx.search_batch_device(qd.data_ptr(), nq, dim, top_k, nprobe, d_i.data_ptr(), d_s.data_ptr(), d_c.data_ptr(), stream=None)
    torch.cuda.synchronize(dev)
    return d_i.cpu().numpy().view(np.uint64)


def run(batch, shapes, reps):
    qh = [mix.draw(batch, 500 + b).cpu().numpy() for b in range(NSETS)]
    want = device_ids(qh[0])
    nbytes = [batch * dim * 4, batch * top_k * 8, batch * top_k * 4, batch * 4]
    pin = [[lib.rbq_host_alloc(b) for b in nbytes] for _ in range(NSETS)]
    for j in range(NSETS):
        C.memmove(pin[j][0], qh[j].ctypes.data, nbytes[0])
    out = (np.empty((batch, top_k), np.uint64), np.empty((batch, top_k), np.float32), np.empty(batch, np.uint32))

    def call(j, pinned):
        if pinned:
            p = pin[j]
            rc = lib.rbq_search_batch(idx._h, p[0], batch, dim, top_k, nprobe, None, 0, p[1], p[2], p[3], None)
        else:
            rc = lib.rbq_search_batch(idx._h, qh[j].ctypes.data, batch, dim, top_k, nprobe, None, 0, out[0].ctypes.data, out[1].ctypes.data,
                                      out[2].ctypes.data, None)
        assert rc == 0, rc

    def measure(pinned):
        for r in range(8):
            call(r % NSETS, pinned)
        ts = np.empty(reps)
        for r in range(reps):
            t0 = time.perf_counter()
            call(r % NSETS, pinned)
            ts[r] = time.perf_counter() - t0
        if os.environ.get("HOST_PROBE_PER_SET") and pinned:  # per query set: median call time, tied queries replayed from the log
            rep = []
            for j in range(NSETS):
                a0 = idx.tie_log_stats(); call(j, pinned); a1 = idx.tie_log_stats()
                rep.append("set %d: %.0f us, %d replays (%d entries, %d heap ops)" % (j, np.median(ts[j::NSETS]) * 1e6, a1["replays"] - a0["replays"],
                           a1["entries"] - a0["entries"], a1["heap_ops"] - a0["heap_ops"]))
            print("    " + "; ".join(rep), flush=True)
        call(0, pinned)
        got = np.ctypeslib.as_array(C.cast(pin[0][1], C.POINTER(C.c_uint64)), shape=(batch, top_k)) if pinned else out[0]
        return np.median(ts) * 1e6, np.percentile(ts, 10) * 1e6, bool(np.array_equal(got, want))

    print(f"--- {batch} queries per call, {reps} calls per cell; us per call: median (p10) -> M queries/s at the median", flush=True)
    for zc in (0, 1):
        for hlp in (0, 1):
            idx.set_option("host_zero_copy", zc)
            idx.set_option("host_stage_helpers", hlp)
            for sub, lanes in shapes:
                idx.set_option("host_subbatch", sub)
                idx.set_option("host_lanes", lanes)
                mp, pp, okp = measure(True)
                mg, pg, okg = measure(False)
                print(f"zero_copy {zc} helpers {hlp} sub {sub:5d} x {lanes}: pinned {mp:7.1f} ({pp:7.1f}) -> {batch / mp:5.2f} M   "
                      f"pageable {mg:7.1f} ({pg:7.1f}) -> {batch / mg:5.2f} M   ids ok {okp} {okg}", flush=True)
    for j in range(NSETS):
        for p in pin[j]:
            lib.rbq_host_free(p)


def shapes_of(env, default):
    v = os.environ.get(env)
    return [tuple(int(t) for t in w.split("x")) for w in v.split(",")] if v else default


for nq_ in [int(t) for t in os.environ.get("HOST_PROBE_NQ", "1024,4096,256,64,1").split(",")]:
    if nq_ >= 4096:
        run(nq_, shapes_of("HOST_PROBE_SHAPES_BIG", [(1024, 4), (0, 0)]), 150)
    elif nq_ >= 1024:
        run(nq_, shapes_of("HOST_PROBE_SHAPES", [(512, 2), (256, 4), (0, 0)]), 300)
    else:
        run(nq_, [(0, 0)], 300)
print('tie log:', idx.tie_log_stats(), 'heap restarts', idx.heap_restarts())
