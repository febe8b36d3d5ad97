"""Diagnostic (GPU box): the kernels of SMALL rbq_search_batch calls, one after the other, under rocprofv3 --kernel-trace.
  run:    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lt -- python3 tools/lat_trace.py [bench args] [--option k=v]
  parse:  python3 tools/lat_trace.py --parse gpurun_out/lt/**/*_kernel_trace.csv
The parse step groups the trace into calls (a chain ends with a scan kernel) and prints, per query count (the scan kernel's grid),
the median duration of every kernel of the chain, the gaps between them and first-start -> last-end."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(path):
    import csv, collections
    rows = list(csv.DictReader(open(path)))
    ev = []
    for r in rows:
        n = r['Kernel_Name']
        if 'rbq::' not in n:
            continue
        short = n.split('rbq::')[1].split('(')[0].split('<')[0]
        wg = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1)
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short, int(r['Grid_Size_X' if 'Grid_Size_X' in r else 'Grid_Size']) // max(wg, 1)))
    ev.sort()
    chains, cur = collections.defaultdict(list), []
    for e in ev:
        cur.append(e)
        if e[2].startswith('k_scan') or e[2] == 'k_lat_scan':
            chains[(e[3], tuple(x[2] for x in cur))].append(cur)
            cur = []
    med = lambda v: sorted(v)[len(v) // 2]
    for (nwg, names), cs in sorted(chains.items()):
        if len(cs) < 8:
            continue
        cs = cs[len(cs) // 4:]  # (the first calls of a size are warm-up)
        line = []
        for i, nm in enumerate(names):
            line.append("%s %.1f" % (nm, med([(c[i][1] - c[i][0]) / 1e3 for c in cs])))
            if i + 1 < len(names):
                line.append("[gap %.1f]" % med([(c[i + 1][0] - c[i][1]) / 1e3 for c in cs]))
        print("scan workgroups %5d  n %3d  chain %.1f us:  %s" % (nwg, len(cs), med([(c[-1][1] - c[0][0]) / 1e3 for c in cs]), "  ".join(line)))


if len(sys.argv) > 2 and sys.argv[1] == '--parse':
    for p in sys.argv[2:]:
        parse(p)
    sys.exit(0)

import numpy as np, torch
import bench
import rabitq_rs_amd as rq

a = bench.parse()
dev = torch.device("cuda", 0)
mix = bench.Mixture(torch, dev, a.dim, a.nlist, a.dataset, a.metric == 1)
x = mix.draw(a.n, 20260105)
cent, assign = bench.kmeans_gpu(torch, x, a.nlist, a.kmeans_iters, 20260103)
xs = mix.draw(max(2 * a.nlist, 4096), 99).cpu().numpy()
small = rq.builder.train_with_clusters(xs, cent.cpu().numpy(), (np.arange(len(xs)) % a.nlist).astype(np.uint32), a.bits, a.metric, 1, 20260104, True)
idx = rq.IvfRabitqIndex.build_on_device(small.hdr_ptr, cent.cpu().numpy(), x.data_ptr(), assign.to(torch.int32).contiguous().data_ptr(), a.n, small.t_const)
del x
lib = rq.index.lib()
for kv in a.option:
    k, v = kv.split("=")
    idx.set_option(k, int(v))
q = mix.draw(4096, 20260102).cpu().numpy()
for nq in (1, 8, 64, 256):
    nsets = 16
    pin = []
    for j in range(nsets):
        p = [lib.rbq_host_alloc(nq * a.dim * 4), lib.rbq_host_alloc(nq * a.top_k * 8), lib.rbq_host_alloc(nq * a.top_k * 4), lib.rbq_host_alloc(nq * 4)]
        C.memmove(p[0], q[j * nq:(j + 1) * nq].ctypes.data, nq * a.dim * 4)
        pin.append(p)
    ts = []
    for r in range(64):
        t0 = time.perf_counter()
        lib.rbq_search_batch(idx._h, pin[r % nsets][0], nq, a.dim, a.top_k, a.nprobe, None, 0, pin[r % nsets][1], pin[r % nsets][2], pin[r % nsets][3], None)
        ts.append(time.perf_counter() - t0)
    print("nq %4d: call p50 %.1f us (traced)" % (nq, np.percentile(np.array(ts[16:]) * 1e6, 50)), flush=True)
