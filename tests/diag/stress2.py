"""Diagnostic: a 'victim' index on one stream and a 'noise' index on other streams; which combination of options
on either side makes the victim's results deviate from the oracle?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import rabitq_rs_amd as rq
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle
a = bench.parse()
reps = int(os.environ.get("STRESS_REPS", "6"))
dev = torch.device("cuda", 0)
x = bench.mixture(torch, dev, a.n, a.dim, a.nlist, 20260105, False)
cent, assign = bench.kmeans_gpu(torch, x, a.nlist, 6, 20260103)
built = rq.builder.train_with_clusters(x.cpu().numpy(), cent.cpu().numpy(), assign.cpu().numpy().astype(np.uint32), a.bits, 0, 1, 20260104, True)
q = bench.mixture(torch, dev, a.batch, a.dim, a.nlist, 20260102, False).cpu().numpy()
rc, oids, osc, ocnt, odiag = oracle.search_batch(built, q, a.top_k, a.nprobe, want_diag=True)
def mk(opts):
    idx = rq.IvfRabitqIndex.from_built(built)
    for kv in filter(None, opts.split(',')):
        k, v = kv.split('=')
        idx.set_option(k, int(v))
    return idx
qd = torch.from_numpy(q).to(dev)
ta = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16); tb = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
nn = int(os.environ.get("NOISE_STREAMS", "2"))
for vopts, nopts in [tuple(c.split('|')) for c in os.environ.get("CASES", "f32_rank=1|").split(';')]:
    victim, noise = mk(vopts), mk(nopts)
    sv = torch.cuda.Stream(dev)
    sn = [torch.cuda.Stream(dev) for _ in range(nn)]
    out = lambda: (torch.zeros(a.batch, a.top_k, dtype=torch.int64, device=dev), torch.zeros(a.batch, a.top_k, dtype=torch.float32, device=dev), torch.zeros(a.batch, dtype=torch.int32, device=dev))
    ov, on = out(), [out() for _ in range(nn)]
    bad = 0; nbad = 0
    for r in range(reps):
        for rr in range(3):
            if os.environ.get("NOISE_TORCH"):
                for i in range(nn):
                    with torch.cuda.stream(sn[i]):
                        for _ in range(int(os.environ["NOISE_TORCH"])):
                            tc = ta @ tb
            for i in range(0 if os.environ.get("NOISE_TORCH") else nn):
                noise.search_batch_device(qd.data_ptr(), a.batch, a.dim, a.top_k, a.nprobe, on[i][0].data_ptr(), on[i][1].data_ptr(), on[i][2].data_ptr(), stream=sn[i].cuda_stream)
            victim.search_batch_device(qd.data_ptr(), a.batch, a.dim, a.top_k, a.nprobe, ov[0].data_ptr(), ov[1].data_ptr(), ov[2].data_ptr(), stream=sv.cuda_stream)
        torch.cuda.synchronize(dev)
        got = ov[0].cpu().numpy().view(np.uint64)
        m = int((got != oids).any(axis=1).sum())
        bad += m > 0
        if m and bad <= 2:
            w = np.nonzero((got != oids).any(axis=1))[0]
            print('  victim mismatching queries', len(w), w[:20])
            for b in w[:3]:
                print('   q', b, 'gpu', got[b], 'ref', oids[b], 'common', len(set(got[b].tolist()) & set(oids[b].tolist())))
        nbad += sum(int((o[0].cpu().numpy().view(np.uint64) != oids).any(axis=1).sum()) > 0 for o in on)
    print("victim[%s] noise[%s]: victim bad reps %d/%d, noise bad outputs %d/%d" % (vopts, nopts, bad, reps, nbad, reps * nn), flush=True)
    victim.close(); noise.close()
