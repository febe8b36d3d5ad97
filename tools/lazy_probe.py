"""Diagnostic (GPU box): taps of the lazy probe selection on a test-sized index: T_ub, z0, shortlist size, lists scored, lists scanned."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import rabitq_rs_amd as rq
from conftest import build_index, make_dataset
n, dim, nlist, bits, metric, nq, top_k, nprobe = [int(v) for v in (sys.argv[1:9] if len(sys.argv) >= 9 else "40000 128 256 7 0 96 10 64".split())]
data, built = build_index(n=n, dim=dim, nlist=nlist, total_bits=bits, metric=metric, normalize=(metric == 1), seed=900 + dim + bits)
idx = rq.IvfRabitqIndex.from_built(built)
rng = np.random.default_rng(901)
q = data[rng.choice(n, nq, replace=False)] + 0.05 * rng.standard_normal((nq, dim)).astype(np.float32)
if metric == 1:
    q /= np.linalg.norm(q, axis=1, keepdims=True)
q = np.ascontiguousarray(q, dtype=np.float32)
dev = torch.device("cuda", 0)
qd = torch.from_numpy(q).to(dev)
o = (torch.zeros(nq, top_k, dtype=torch.int64, device=dev), torch.zeros(nq, top_k, dtype=torch.float32, device=dev), torch.zeros(nq, dtype=torch.int32, device=dev))
st = torch.cuda.Stream(dev)
torch.cuda.synchronize(dev)
idx.search_batch_device(qd.data_ptr(), nq, dim, top_k, nprobe, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), stream=st.cuda_stream)
torch.cuda.synchronize(dev)
ds = idx.debug_copy_workspace(st.cuda_stream, "dead_skipped", np.empty((4, nq), np.uint32))
consts = idx.debug_copy_workspace(st.cuda_stream, "consts", np.empty((nq, 12), np.float32))
sc = o[1].cpu().numpy()
tub = ds[2].view(np.float32)
for i in range(min(nq, 12)):
    z = int(ds[3, i])
    print("q%d: scanned %d  T_ub %.4g  final T %.4g  z0 %d shortlist %d scored-beyond-head %d  exlo %.3g exhi %.3g amin %.0f amax %.0f" % (
        i, ds[1, i], tub[i], sc[i, top_k - 1], z & 1023, (z >> 10) & 1023, z >> 20, consts[i, 7], consts[i, 8], consts[i, 10], consts[i, 11]))
print("mean scanned %.1f of %d; T_ub finite for %.0f %% of the queries" % (ds[1].mean(), min(nprobe, nlist), 100 * np.isfinite(tub).mean()))
