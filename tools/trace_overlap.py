"""Diagnostic: per-kernel durations and concurrency inside the multi-stream timed region of a rocprofv3
--kernel-trace of bench.py (argument: the *_kernel_trace.csv)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
names = ('k_prep', 'k_rank', 'k_select', 'k_scan')
ev = []
for r in rows:
    for k in names:
        if k in r['Kernel_Name']:
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k, r['Stream_Id']))
ev.sort()
cnt = collections.Counter(e[3] for e in ev)
mode = collections.Counter(cnt.values()).most_common(1)[0][0]   # the side streams of the timed region ran equally many
side = [s for s in cnt if cnt[s] == mode]
ref = [e for e in ev if e[3] == side[0]]
t0, t1 = ref[8][0], ref[-1][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
nb = len([e for e in win if e[2] == 'k_scan'])
print("window %.2f ms, %d batches, %.4f ms per batch, streams %s" % ((t1 - t0) / 1e6, nb, (t1 - t0) / 1e6 / nb, sorted(set(e[3] for e in win))))
d = collections.defaultdict(list)
for e in win:
    d[e[2]].append((e[1] - e[0]) / 1e3)
for k in names:
    v = d[k]
    if v:
        print("%-9s mean %.1f us  min %.1f  max %.1f  n %d" % (k, sum(v) / len(v), min(v), max(v), len(v)))
pts = []
for e in win:
    pts += [(e[0], 1), (e[1], -1)]
pts.sort()
c, last, hist = 0, t0, collections.Counter()
for t, dl in pts:
    hist[c] += t - last
    last = t
    c += dl
tot = sum(hist.values())
print("kernels in flight:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})

# per stream: idle time between the end of a kernel and the start of the next one of the same stream
bys = collections.defaultdict(list)
for e in win:
    bys[e[3]].append(e)
gaps = collections.defaultdict(list)
for s_, evs in bys.items():
    evs.sort()
    for x, y in zip(evs, evs[1:]):
        gaps[x[2] + '->' + y[2]].append((y[0] - x[1]) / 1e3)
for k, v in sorted(gaps.items()):
    v.sort()
    print("gap %-20s mean %.1f us  p50 %.1f  p90 %.1f  n %d" % (k, sum(v) / len(v), v[len(v) // 2], v[int(len(v) * 0.9)], len(v)))
busy = sum(e[1] - e[0] for e in win)
print("sum of kernel durations / window = %.2f (mean kernels in flight)" % (busy / (t1 - t0)))
