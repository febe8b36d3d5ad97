"""Cross-check of the two clocks bench.py's roofline rests on: per-launch duration of k_scan (block bound off) from the HIP
event pair carried by its dispatch packet (rbq_profile_*) — printed here, one launch per profile window — against the
kernel trace of the same process (run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/event_vs_trace.py`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import bench
import rabitq_rs_amd as rq
n, dim, nlist, batch, top_k, nprobe = 1_000_000, 960, 4096, 1024, 10, 128
dev = torch.device("cuda", 0)
mix = bench.Mixture(torch, dev, dim, nlist, "mixture_id32", False)
x = mix.draw(n, 20260105)
cent, assign = bench.kmeans_gpu(torch, x, nlist, 4, 20260103)
xs = mix.draw(8192, 99).cpu().numpy()
small = rq.builder.train_with_clusters(xs, cent.cpu().numpy(), (np.arange(8192) % nlist).astype(np.uint32), 7, 0, 1, 20260104, True)
idx = rq.IvfRabitqIndex.build_on_device(small.hdr_ptr, cent.cpu().numpy(), x.data_ptr(), assign.to(torch.int32).contiguous().data_ptr(), n, small.t_const)
del x
idx.set_option("block_bound", 0)
qs = [mix.draw(batch, 700 + b).contiguous() for b in range(12)]
st = torch.cuda.Stream(dev)
o = (torch.empty(batch, top_k, dtype=torch.int64, device=dev), torch.empty(batch, top_k, dtype=torch.float32, device=dev), torch.empty(batch, dtype=torch.int32, device=dev))
def go(q):
    idx.search_batch_device(q.data_ptr(), batch, dim, top_k, nprobe, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), stream=st.cuda_stream)
for q in qs[:2]:
    go(q)
torch.cuda.synchronize(dev)
ev = []
for q in qs[2:]:
    idx.profile_begin(stages=("scan",), every=1)
    go(q)
    torch.cuda.synchronize(dev)
    idx.profile_end()
    ev.append(idx.profile_stage("scan")[0])
print("EVENT_MS", " ".join("%.4f" % e for e in ev))
