"""Diagnostic for the x16-MFMA / k_prep interaction (rank_mfma.hpp): victim stream with the workgroup-per-query k_prep
(option wg_prep=1) beside noise streams; compares the victim's intermediates with a quiet run.  Run once per library
variant (RBQ_LIB_PATH).  python tests/diag/x16_probe.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import bench
import rabitq_rs_amd as rq
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n, dim, nlist, batch, top_k, nprobe = 1_000_000, 960, 4096, 1024, 10, 128
dev = torch.device("cuda", 0)
mix = bench.Mixture(torch, dev, dim, nlist, "mixture_id32", False)
x = mix.draw(n, 20260105)
cent, assign = bench.kmeans_gpu(torch, x, nlist, 3, 20260103)
built = rq.builder.train_with_clusters(x.cpu().numpy(), cent.cpu().numpy(), assign.cpu().numpy().astype(np.uint32), 7, 0, 1, 20260104, True)
def mk(opts):
    idx = rq.IvfRabitqIndex.from_built(built)
    for kv in filter(None, opts.split(',')):
        k, v = kv.split('=')
        idx.set_option(k, int(v))
    return idx
victim, noise = mk(os.environ.get("VICTIM_OPTS", "wg_prep=1")), mk(os.environ.get("NOISE_OPTS", "wg_prep=1"))
qd = mix.draw(batch, 20260102).contiguous()
nn = 3
sv = torch.cuda.Stream(dev); sn = [torch.cuda.Stream(dev) for _ in range(nn)]
out = lambda: (torch.zeros(batch, top_k, dtype=torch.int64, device=dev), torch.zeros(batch, top_k, dtype=torch.float32, device=dev), torch.zeros(batch, dtype=torch.int32, device=dev))
ov, on = out(), [out() for _ in range(nn)]
D = built.padded_dim
bufs = {"rot": (np.float32, batch * D), "lut": (np.uint8, batch * D * 4), "consts": (np.float32, batch * 12), "probe": (np.uint32, batch * nprobe * 4)}
def snap():
    torch.cuda.synchronize(dev)
    d = {k: victim.debug_copy_workspace(sv.cuda_stream, k, np.empty(m, t)) for k, (t, m) in bufs.items()}
    d["ids"] = ov[0].cpu().numpy().copy()
    return d
def run_victim():
    victim.search_batch_device(qd.data_ptr(), batch, dim, top_k, nprobe, ov[0].data_ptr(), ov[1].data_ptr(), ov[2].data_ptr(), stream=sv.cuda_stream)
run_victim(); ref = snap()
bad = 0
for r in range(reps):
    for rr in range(4):
        for i in range(nn):
            noise.search_batch_device(qd.data_ptr(), batch, dim, top_k, nprobe, on[i][0].data_ptr(), on[i][1].data_ptr(), on[i][2].data_ptr(), stream=sn[i].cuda_stream)
        run_victim()
    cur = snap()
    rep = []
    for k in cur:
        va, vb = ref[k].view(np.uint8).reshape(batch, -1), cur[k].view(np.uint8).reshape(batch, -1)
        rows = np.nonzero((va != vb).any(axis=1))[0]
        if len(rows):
            rep.append("%s: %d queries %s" % (k, len(rows), rows[:4]))
    if rep:
        bad += 1
        print("rep", r, "; ".join(rep), flush=True)
        la, lb = ref["lut"].reshape(batch, -1), cur["lut"].reshape(batch, -1)
        rows = np.nonzero((la != lb).any(axis=1))[0]
        for b in rows[:2]:
            w = np.nonzero(la[b] != lb[b])[0]
            print("   q", b, "lut bytes differing", len(w), "positions", w[:24], "ref", la[b][w[:8]], "cur", lb[b][w[:8]])
print(os.environ.get("RBQ_LIB_PATH", "default").split("librbq_")[-1], "victim", os.environ.get("VICTIM_OPTS", "wg_prep=1"), "noise", os.environ.get("NOISE_OPTS", "wg_prep=1"),
      ": %d of %d rounds differ from the quiet run" % (bad, reps))
