"""Diagnostic (GPU box): queries/s of the device entry on the headline index WITH an id filter (search_filtered, src/ivf.rs:1723-1730)
beside the unfiltered rate, same streams / batches protocol as bench.py's timed region.  python tools/filtered_rate.py [pass fractions ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import bench
import rabitq_rs_amd as rq

n, dim, nlist, nprobe, top_k, batch, NB, ns = 1_000_000, 960, 4096, 128, 10, 1024, 16, 12
dev = torch.device("cuda", 0)
mix = bench.Mixture(torch, dev, dim, nlist, "mixture_id32", False)
x = mix.draw(n, 20260105)
cent, assign = bench.kmeans_gpu(torch, x, nlist, 4, 20260103)
xs = mix.draw(8192, 99).cpu().numpy()
small = rq.builder.train_with_clusters(xs, cent.cpu().numpy(), (np.arange(8192) % nlist).astype(np.uint32), 7, 0, 1, 20260104, True)
idx = rq.IvfRabitqIndex.build_on_device(small.hdr_ptr, cent.cpu().numpy(), x.data_ptr(), assign.to(torch.int32).contiguous().data_ptr(), n, small.t_const)
del x
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("="); idx.set_option(k, int(v))
q = mix.draw(NB * batch, 20260102).contiguous().view(NB, batch, dim)
streams = [torch.cuda.Stream(dev) for _ in range(ns)]
outs = [(torch.empty(batch, top_k, dtype=torch.int64, device=dev), torch.empty(batch, top_k, dtype=torch.float32, device=dev), torch.empty(batch, dtype=torch.int32, device=dev)) for _ in range(ns)]


def rate(d_filter, nbits, steps=96):
    def step(i):
        o = outs[i % ns]
        idx.search_batch_device(q[i % NB].data_ptr(), batch, dim, top_k, nprobe, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                                stream=streams[i % ns].cuda_stream, d_filter=d_filter, filter_nbits=nbits)
    for i in range(2 * ns):
        step(i)
    torch.cuda.synchronize(dev)
    best = 0.0
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize(dev)
        best = max(best, batch * steps / (time.perf_counter() - t0))
    return best


print("unfiltered: %.2f M queries/s" % (rate(None, 0) / 1e6))
rng = np.random.default_rng(1)
for frac in (0.9, 0.5, 0.1, 0.01):
    bits = rng.random(n) < frac
    words = np.packbits(bits.reshape(-1, 8)[:, ::-1], axis=1).reshape(-1)[: (n + 7) // 8]
    w32 = np.zeros((n + 31) // 32, np.uint32)
    w32.view(np.uint8)[: len(words)] = words
    d_f = torch.from_numpy(w32.view(np.int32)).to(dev)
    print("filter passing %.0f %%: %.2f M queries/s" % (100 * frac, rate(d_f.data_ptr(), n) / 1e6), flush=True)
