"""Diagnostic: which intermediate buffer of a victim stream differs between a quiet run and a run beside noise streams?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import rabitq_rs_amd as rq
a = bench.parse()
reps = int(os.environ.get("STRESS_REPS", "8"))
dev = torch.device("cuda", 0)
x = bench.mixture(torch, dev, a.n, a.dim, a.nlist, 20260105, False)
cent, assign = bench.kmeans_gpu(torch, x, a.nlist, 6, 20260103)
built = rq.builder.train_with_clusters(x.cpu().numpy(), cent.cpu().numpy(), assign.cpu().numpy().astype(np.uint32), a.bits, 0, 1, 20260104, True)
q = bench.mixture(torch, dev, a.batch, a.dim, a.nlist, 20260102, False).cpu().numpy()
def mk(opts):
    idx = rq.IvfRabitqIndex.from_built(built)
    for kv in filter(None, opts.split(',')):
        k, v = kv.split('=')
        idx.set_option(k, int(v))
    return idx
victim, noise = mk(os.environ.get("VICTIM_OPTS", "f32_rank=1")), mk(os.environ.get("NOISE_OPTS", ""))
qd = torch.from_numpy(q).to(dev)
nn = 2
sv = torch.cuda.Stream(dev); sn = [torch.cuda.Stream(dev) for _ in range(nn)]
out = lambda: (torch.zeros(a.batch, a.top_k, dtype=torch.int64, device=dev), torch.zeros(a.batch, a.top_k, dtype=torch.float32, device=dev), torch.zeros(a.batch, dtype=torch.int32, device=dev))
ov, on = out(), [out() for _ in range(nn)]
D = built.padded_dim
bufs = {"rot": (np.float32, a.batch * D), "lut": (np.uint8, a.batch * D * 4), "consts": (np.float32, a.batch * 12),
        "probe": (np.uint32, a.batch * a.nprobe * 4), "nstream": (np.uint32, a.batch), "scores": (np.float32, a.batch * a.nlist)}
def snap():
    torch.cuda.synchronize(dev)
    d = {k: victim.debug_copy_workspace(sv.cuda_stream, k, np.empty(n, t)) for k, (t, n) in bufs.items()}
    d["ids"] = ov[0].cpu().numpy().copy(); d["sc"] = ov[1].cpu().numpy().copy()
    return d
def run_victim():
    victim.search_batch_device(qd.data_ptr(), a.batch, a.dim, a.top_k, a.nprobe, ov[0].data_ptr(), ov[1].data_ptr(), ov[2].data_ptr(), stream=sv.cuda_stream)
run_victim(); ref = snap()
run_victim(); again = snap()
print("quiet rerun identical:", all(np.array_equal(ref[k].view(np.uint8), again[k].view(np.uint8)) for k in ref if k != "scores"))
for r in range(reps):
    for rr in range(3):
        for i in range(nn):
            noise.search_batch_device(qd.data_ptr(), a.batch, a.dim, a.top_k, a.nprobe, on[i][0].data_ptr(), on[i][1].data_ptr(), on[i][2].data_ptr(), stream=sn[i].cuda_stream)
        run_victim()
    cur = snap()
    rep = []
    for k in cur:
        va, vb = ref[k].view(np.uint8), cur[k].view(np.uint8)
        if not np.array_equal(va, vb):
            va = np.ascontiguousarray(va).reshape(-1); vb = np.ascontiguousarray(vb).reshape(-1)
            per = len(va) // a.batch
            rows = np.nonzero((va.reshape(a.batch, per) != vb.reshape(a.batch, per)).any(axis=1))[0]
            rep.append("%s: %d queries differ %s" % (k, len(rows), rows[:6]))
    print("rep", r, "; ".join(rep) if rep else "all identical", flush=True)
    if rep:
        ca, cb = ref["consts"].reshape(a.batch, 12), cur["consts"].reshape(a.batch, 12)
        rows = np.nonzero((ca.view(np.uint32) != cb.view(np.uint32)).any(axis=1))[0]
        for b in rows[:4]:
            print("  q", b, "consts ref", ca[b], "\n        cur", cb[b])
            la, lb = ref["lut"].reshape(a.batch, -1)[b], cur["lut"].reshape(a.batch, -1)[b]
            w = np.nonzero(la != lb)[0]
            print("   lut bytes differing", len(w), "of", len(la), "first", w[:10], la[w[:10]], lb[w[:10]])
