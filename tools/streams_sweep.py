"""Diagnostic: throughput of rbq_search_batch_device against the number of caller streams (no profiling taps)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import rabitq_rs_amd as rq
a = bench.parse()
dev = torch.device("cuda", 0)
mix = bench.Mixture(torch, dev, a.dim, a.nlist, 'mixture_id32', False)
x = mix.draw(a.n, 20260105)
cent, assign = bench.kmeans_gpu(torch, x, a.nlist, 6, 20260103)
built = rq.builder.train_with_clusters(x.cpu().numpy(), cent.cpu().numpy(), assign.cpu().numpy().astype(np.uint32), a.bits, 0, 1, 20260104, True)
idx = rq.IvfRabitqIndex.from_built(built)
qds = [mix.draw(a.batch, 20260102 + b).contiguous() for b in range(16)]  # every step a different batch
use_null = os.environ.get("USE_NULL", "0") == "1"
for ns in [int(v) for v in os.environ.get("NS", "1,2,3,4,5,6,8").split(",")]:
    streams = ([torch.cuda.current_stream(dev)] if use_null else []) + [torch.cuda.Stream(dev) for _ in range(ns - (1 if use_null else 0))]
    outs = [(torch.zeros(a.batch, a.top_k, dtype=torch.int64, device=dev), torch.zeros(a.batch, a.top_k, dtype=torch.float32, device=dev), torch.zeros(a.batch, dtype=torch.int32, device=dev)) for _ in range(ns)]
    def step(i):
        o = outs[i % ns]
        idx.search_batch_device(qds[i % 16].data_ptr(), a.batch, a.dim, a.top_k, a.nprobe, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), stream=streams[i % ns].cuda_stream)
    for i in range(3 * ns):
        step(i)
    torch.cuda.synchronize(dev)
    n = int(os.environ.get("N_STEPS", "96"))
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    print("streams %d (null stream %s): %.4f ms/step, %.0f q/s (host enqueue %.4f ms/step)" % (ns, "used" if use_null else "not used", dt / n * 1e3, a.batch * n / dt, t_enq / n * 1e3), flush=True)
