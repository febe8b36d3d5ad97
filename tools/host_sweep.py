"""PCIe-inclusive rate of rbq_search_batch for several pipeline shapes (sub-batch size x lanes), pageable and pinned
caller buffers, 1 and 4 caller threads.  python tools/host_sweep.py [n] [nlist]
HOST_SWEEP_BIG=1: calls of 8192 / 4096 queries against the number of lanes instead (round 3: no shape beats the default)."""
import ctypes as C
import os
import sys
import threading
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
dim, batch, top_k, nprobe = 960, 1024, 10, 128
dev = torch.device("cuda", 0)
mix = bench.Mixture(torch, dev, dim, nlist, "mixture_id32", False)
x = mix.draw(n, 20260105)
cent, assign = bench.kmeans_gpu(torch, x, nlist, 4, 20260103)
xs = mix.draw(8192, 99).cpu().numpy()
small = rq.builder.train_with_clusters(xs, cent.cpu().numpy(), (np.arange(8192) % nlist).astype(np.uint32), 7, 0, 1, 20260104, True)
idx = rq.IvfRabitqIndex.build_on_device(small.hdr_ptr, cent.cpu().numpy(), x.data_ptr(), assign.to(torch.int32).contiguous().data_ptr(), n,
                                        small.t_const)
del x
lib = rq.index.lib()
lib = rq.index.lib()


def run(batch, configs):
    qh = [mix.draw(batch, 500 + b).cpu().numpy() for b in range(8)]
    nbytes = [batch * dim * 4, batch * top_k * 8, batch * top_k * 4, batch * 4]
    pin = [[[lib.rbq_host_alloc(b) for b in nbytes] for _ in range(8)] for _ in range(4)]
    for t in range(4):
        for j in range(8):
            C.memmove(pin[t][j][0], qh[j].ctypes.data, nbytes[0])
    out = [(np.empty((batch, top_k), np.uint64), np.empty((batch, top_k), np.float32), np.empty(batch, np.uint32)) for _ in range(4)]

    def call(t, j, pinned):
        if pinned:
            p = pin[t][j]
            rc = lib.rbq_search_batch(idx._h, p[0], batch, dim, top_k, nprobe, None, 0, p[1], p[2], p[3], None)
        else:
            o = out[t]
            rc = lib.rbq_search_batch(idx._h, qh[j].ctypes.data, batch, dim, top_k, nprobe, None, 0, o[0].ctypes.data, o[1].ctypes.data,
                                      o[2].ctypes.data, None)
        assert rc == 0

    def timed(nthreads, nrep, pinned):
        def loop(t):
            for r in range(nrep):
                call(t, (t * 3 + r) % 8, pinned)
        th = [threading.Thread(target=loop, args=(t,)) for t in range(nthreads)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        return batch * nthreads * nrep / (time.perf_counter() - t0)

    call(0, 0, False); call(0, 0, True)
    same = np.array_equal(out[0][0], np.ctypeslib.as_array(C.cast(pin[0][0][1], C.POINTER(C.c_uint64)), shape=(batch, top_k)))
    print(f"--- {batch} queries per call (pageable == pinned ids: {same}); rows = queries/s")
    for sub, lanes in configs:
        idx.set_option("host_subbatch", sub)
        idx.set_option("host_lanes", lanes)
        timed(1, 3, False); timed(1, 3, True)
        idx.set_option("host_trace", 1)
        call(0, 0, False); call(0, 0, True)
        idx.set_option("host_trace", 0)
        reps = max(10, 40960 // batch)
        print(f"sub {sub:5d} lanes {lanes}: pageable 1t {timed(1, reps, False):9.0f}  2t {timed(2, reps, False):9.0f}  4t {timed(4, reps, False):9.0f}   "
              f"pinned 1t {timed(1, reps, True):9.0f}  2t {timed(2, reps, True):9.0f}  4t {timed(4, reps, True):9.0f}", flush=True)
    for t in range(4):
        for j in range(8):
            for p in pin[t][j]:
                lib.rbq_host_free(p)


if os.environ.get("HOST_SWEEP_BIG") == "1":  # large calls: how many lanes pay
    run(8192, ((1024, 4), (1024, 6), (1024, 8), (1024, 12), (512, 12), (2048, 4), (0, 0)))
    run(4096, ((1024, 4), (512, 6), (512, 8), (0, 0)))
    sys.exit(0)
run(1024, ((1024, 1), (512, 2), (256, 4), (0, 0)))
run(4096, ((4096, 1), (1024, 4), (1024, 2), (2048, 2), (512, 4), (0, 0)))
run(256, ((0, 0),))
