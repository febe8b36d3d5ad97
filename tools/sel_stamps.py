"""Diagnostic: per-phase cycles of k_select_mfma from a -DRBQ_SEL_STAMPS build (RBQ_LIB_PATH=.../librbq_sel.so)."""
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
xs = mix.draw(max(2 * a.nlist, 8192), 99).cpu().numpy()  # the device encoder (a small CPU build only supplies the header, rotator and t_const)
small = rq.builder.train_with_clusters(xs, cent.cpu().numpy(), (np.arange(len(xs)) % a.nlist).astype(np.uint32), a.bits, 0, 1, 20260104, True)
idx = rq.IvfRabitqIndex.build_on_device(small.hdr_ptr, cent.cpu().numpy(), x.data_ptr(), assign.to(torch.int32).contiguous().data_ptr(), a.n, small.t_const)
del x
qds = [mix.draw(a.batch, 20260102 + i).contiguous() for i in range(3)]
s = torch.cuda.Stream(dev)
o = (torch.zeros(a.batch, a.top_k, dtype=torch.int64, device=dev), torch.zeros(a.batch, a.top_k, dtype=torch.float32, device=dev), torch.zeros(a.batch, dtype=torch.int32, device=dev))
for kv in a.option:
    idx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for qd in qds:
    idx.search_batch_device(qd.data_ptr(), a.batch, a.dim, a.top_k, a.nprobe, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), stream=s.cuda_stream)
torch.cuda.synchronize(dev)
v = idx.debug_copy_workspace(s.cuda_stream, "nvec", np.empty(a.batch, np.uint64))
if os.environ.get("SEL_MODE") == "prep":
    c = idx.debug_copy_workspace(s.cuda_stream, "consts", np.empty(a.batch * 12, np.float32)).reshape(a.batch, 12)
    print("k_prep ticks: rotate %.0f  sums %.0f  lut+consts %.0f" % (c[:, 7].mean(), c[:, 8].mean(), c[:, 9].mean()))
    sys.exit(0)
if os.environ.get("SEL_MODE") == "3":
    c = v.astype(np.int64)
    print("shortlist size: mean %.1f p50 %d p90 %d p99 %d max %d; > 128: %.1f %%" % (c.mean(), np.percentile(c, 50), np.percentile(c, 90), np.percentile(c, 99), c.max(), 100.0 * (c > 128).mean()))
    sys.exit(0)
if os.environ.get("SEL_MODE") == "2":
    st = (v >> np.uint64(32)).astype(np.int64); du = (v & np.uint64(0xffffffff)).astype(np.int64)
    st = ((st - st.min()) & 0xffffffff) * 16
    print("starts: min 0 p50 %d p90 %d max %d ; duration mean %d p99 %d max %d ; last end %d" % (np.percentile(st, 50), np.percentile(st, 90), st.max(), du.mean(), np.percentile(du, 99), du.max(), (st + du).max()))
    order = np.argsort(st)
    if a.batch >= 1024: print("start of WG #0,256,512,768,1023 in start order:", st[order][[0, 256, 512, 768, 1023]])
    sys.exit(0)
if os.environ.get("SEL_MODE") == "4":  # -DRBQ_SEL_STAMPS=4: sub-phases of the lazy branch, 128-cycle units
    for t, nme in enumerate(["approx sort", "certain members (z0)", "head scoring", "T_ub", "classification", "todo scoring", "membership+compaction+sort"]):
        if t >= 6:
            break
        c = ((v >> np.uint64(10 * t)) & np.uint64(0x3ff)).astype(np.float64) * 128
        print("%-28s mean %7.0f cycles  p99 %7.0f  max %7.0f" % (nme, c.mean(), np.percentile(c, 99), c.max()))
    sys.exit(0)
names = ["stage row+q", "radix select", "shortlist", "canonical", "sort", "probe+stream"]
tot = 0
for t, n in enumerate(names):
    c = ((v >> np.uint64(10 * t)) & np.uint64(0x3ff)).astype(np.float64) * 256
    tot += c.mean()
    print("%-14s mean %7.0f cycles  p99 %7.0f  max %7.0f" % (n, c.mean(), np.percentile(c, 99), c.max()))
print("total %.0f cycles; shortlist size mean (mod 16) %.1f" % (tot, (v >> np.uint64(60)).astype(np.float64).mean()))
