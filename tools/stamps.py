"""Diagnostic: run one batch through a -DRBQ_STAMPS build and print where scanner wave 0 spends its cycles."""
import os, sys
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
if a.n <= 100_000:
    built = rq.builder.train_with_clusters(x.cpu().numpy(), cent.cpu().numpy(), assign.cpu().numpy().astype(np.uint32), a.bits, 0, 1, 20260104, True)
    idx = rq.IvfRabitqIndex.from_built(built)
else:  # large index: the device encoder (a small CPU build only supplies the header, rotator and t_const)
    xs = mix.draw(max(2 * a.nlist, 8192), 99).cpu().numpy()
    small = rq.builder.train_with_clusters(xs, cent.cpu().numpy(), (np.arange(len(xs)) % a.nlist).astype(np.uint32), a.bits, 0, 1, 20260104, True)
    idx = rq.IvfRabitqIndex.build_on_device(small.hdr_ptr, cent.cpu().numpy(), x.data_ptr(), assign.to(torch.int32).contiguous().data_ptr(), a.n, small.t_const)
    del x
qs = [mix.draw(a.batch, 20260102 + i).cpu().numpy() for i in range(3)]
idx.set_option('host_subbatch', 1 << 20)  # one sub-batch: the diag slots carry the stamps of one launch
for q in qs + ([qs[-1]] if os.environ.get('STAMPS_WARM') == '1' else []):  # the last batch is cold: nothing of it is in the caches (STAMPS_WARM=1: run it again, warm)
    ids, sc, cnt, diag = idx.batch_search_raw(q, rq.SearchParams(a.top_k, a.nprobe), want_diag=True)
d = diag.astype(np.uint64)
if a.batch < 16:  # small calls (the latency regime): the stamps of 32 separate calls, each on queries the caches have not seen
    rows = []
    for i in range(32):
        qi = mix.draw(a.batch, 20260200 + i).cpu().numpy()
        rows.append(idx.batch_search_raw(qi, rq.SearchParams(a.top_k, a.nprobe), want_diag=True)[3])
    d = np.concatenate(rows).astype(np.uint64)
if os.environ.get('RBQ_STAMPS_MODE') == '3':
    lo = lambda c: (d[:, c] & 0xffffffff).astype(np.float64).mean(); hi = lambda c: (d[:, c] >> 32).astype(np.float64).mean()
    print('scanner wave 0 cycles/query: lookups %.0f  waitA %.0f  live tiles %.1f  survivors %.0f  fill %.0f  heavy tiles %.1f' % (lo(0), hi(0), lo(1), hi(1), lo(2), hi(2)))
    sys.exit(0)
if os.environ.get('RBQ_STAMPS_MODE') == '4':
    lo = lambda c: (d[:, c] & 0xffffffff).astype(np.float64).mean(); hi = lambda c: (d[:, c] >> 32).astype(np.float64).mean()
    print('replay wave cycles/query: batch data from LDS %.0f  merge %.0f = candidate pass %.0f + proviso/serial decision %.0f + scatter/reload/checks %.0f;  waitA %.0f' % (lo(0), hi(0), lo(1), hi(1), lo(2), hi(2)))
    sys.exit(0)
if os.environ.get('RBQ_STAMPS_MODE') == '2':
    lo = lambda c: (d[:, c] & 0xffffffff).astype(np.float64).mean(); hi = lambda c: (d[:, c] >> 32).astype(np.float64).mean()
    print('replay wave cycles/query: collect %.0f  round0 refine %.0f  replay %.0f  waitC %.0f  light tiles %.0f  waitA %.0f' % (lo(0), hi(0), lo(1), hi(1), lo(2), hi(2)))
    sys.exit(0)
heavy = (d[:, 0] & 0xffffffff).astype(np.float64); rounds = (d[:, 0] >> 32).astype(np.float64); waitA = np.zeros(len(d)); ntile = dead = np.zeros(len(d))
print('heavy refine rounds/query %.1f' % rounds.mean())
total = (d[:, 1] & 0xffffffff).astype(np.float64); prolog = ((d[:, 1] >> 32) & 0xffff).astype(np.float64); ntiles = (d[:, 1] >> 48).astype(np.float64); nheavy = np.zeros(len(d))
tiles_t = (d[:, 2] & 0xffffffff).astype(np.float64); surv = np.zeros(len(d)); look = (d[:, 2] >> 32).astype(np.float64)
print('prologue %.0f  tile steps total %.0f (incl. heavy)  tiles %.1f  -> per non-heavy tile %.0f' % (prolog.mean(), tiles_t.mean(), ntiles.mean(), (tiles_t.mean() - heavy.mean()) / max(ntiles.mean(), 1)))
print("per-query means (cycles of s_memtime): total %.0f  fill %.0f (%.0f%%)  waitA %.0f (%.0f%%)  heavy %.0f (%.0f%%)" % (
    total.mean(), look.mean(), 100 * look.mean() / total.mean(), waitA.mean(), 100 * waitA.mean() / total.mean(), heavy.mean(), 100 * heavy.mean() / total.mean()))
print("tiles/query %.1f dead(wave0) %.1f" % (ntile.mean(), dead.mean()))
print("heavy tiles/query %.1f  survivors/query %.0f  total p50 %.0f p99 %.0f max %.0f" % (nheavy.mean(), surv.mean(), np.percentile(total, 50), np.percentile(total, 99), total.max()))
