"""Summarise the passes of tools/profile_round.sh into markdown (stdout).  python tools/profile_summary.py DIR [bench args]"""
import collections
import csv
import glob
import json
import statistics
import sys

d = sys.argv[1]
args = " ".join(sys.argv[2:])


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("rbq::", "")
    return n[:60]


def bench_line(path):
    try:
        for line in open(path):
            if line.startswith("{"):
                return json.loads(line)
    except OSError:
        pass
    return None


print(f"# rocprofv3 profile set ({d.split('prof_')[-1]}), bench args: `{args or '(headline defaults)'}`\n")
b = bench_line(f"{d}/bench_stats.json")
if b:
    print(f"bench line of the stats run (under the tracer): {b['value']:.0f} queries/s, {b['ms_per_step']:.4f} ms/step, "
          f"recall {b.get('recall_at_10')}, roofline.frac {b['roofline']['frac']:.3f} "
          f"(k_scan bound off: {b['roofline']['avg_launch_ms']:.4f} ms/launch, {b['roofline']['algorithmic_bytes_per_launch'] / 1e9:.3f} GB algorithmic), "
          f"pruned: {b['pruned']['avg_launch_ms']:.4f} ms/launch overlapped, {b['pruned']['bytes_requested_per_launch'] / 1e9:.3f} GB requested, "
          f"skip {b['pruned']['block_skip_frac']:.4f}\n")

print("## `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --no-cpu --no-extras`\n")
ns = b["config"]["streams"] if b else "N"
print(f"One process: {ns} setup + 8 warm-up + 20 timed steps on {ns} streams (bound on, every step a different batch; traffic counters off), "
      f"{ns} setup + 32 launches of the counter pass (the same batches, counters on), 32 single-stream verification launches, then the "
      "roofline leg (bound off: 1 + 2 + 6 + 2 launches on one stream).  avg mixes those; min = the kernel alone on the chip (warm: a "
      "verification launch right after the same batch's counter-pass launch).\n")
print("| kernel | calls | avg ms | min ms | max ms | share |\n|---|---|---|---|---|---|")
rows = []
for f in glob.glob(f"{d}/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append(r)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if "rbq::" not in r["Name"]:
        continue
    print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.4f} | {float(r['MinNs']) / 1e6:.4f} | "
          f"{float(r['MaxNs']) / 1e6:.4f} | {float(r['Percentage']):.1f} % |")

# per-dispatch durations of k_scan from the kernel trace: the bound-off launches are the long ones
durs = collections.defaultdict(list)
for f in glob.glob(f"{d}/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rbq::" in r["Kernel_Name"]:
            durs[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in durs.items():
    if "k_scan" in k:
        v.sort()
        # the process ends with the roofline leg (bound off, one stream): 1 set-up + 2 warm-up + 6 TIMED launches, then 2
        # verification launches — the timed six are launches [-8:-2]
        tail = [x[1] / 1e6 for x in v[-8:-2]]
        print(f"\n`{k}`: the six timed launches of the roofline leg (bound off, one stream) in the kernel trace: "
              f"{', '.join('%.3f' % t for t in tail)} ms — median {statistics.median(tail):.4f}, mean {statistics.mean(tail):.4f}; "
              f"roofline.avg_launch_ms of the same run's bench line (HIP events carried by the dispatch packets): "
              f"{b['roofline']['avg_launch_ms']:.4f} ms.  (Every launch scans a different batch.)" if b else "")


def pmc(sub):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rbq::" in r["Kernel_Name"]:
                out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


print("\n## HBM traffic (`--pmc FETCH_SIZE`, separate passes; per launch = per 1024-query batch)\n")
print("FETCH_SIZE is reported in KB; on gfx950 it tallies 128-B requests at 64 B for wide coalesced reads "
      "(MI355X_MICROARCH.md, HBM section), so the corrected figure doubles it.  Median over the launches of the pass.\n")
print("| kernel | bound | launches | FETCH_SIZE raw (MB) | corrected x2 (MB) |\n|---|---|---|---|---|")
for sub, lab in (("fetch_on", "on"), ("fetch_off", "off")):
    for k, c in pmc(sub).items():
        if "FETCH_SIZE" in c:
            v = c["FETCH_SIZE"]
            med = statistics.median(v)
            print(f"| `{k}` | {lab} | {len(v)} | {med / 1e3:.1f} | {2 * med / 1e3:.1f} |")
# the bound-off HBM bytes per launch of the four-wave scan kernel, for bench.py's roofline.traffic (profiles/rN/pmc_traffic.json)
try:
    off = pmc("fetch_off")
    cand = [(k, c["FETCH_SIZE"]) for k, c in off.items() if "k_scan<" in k and "FETCH_SIZE" in c]
    if cand:
        k, v = max(cand, key=lambda kv: statistics.median(kv[1]))
        with open(f"{d}/pmc_traffic.json", "w") as f:
            json.dump({"bytes": 2.0 * statistics.median(v) * 1e3, "kernel": k, "launches": len(v),
                       "source": "rocprofv3 --pmc FETCH_SIZE (its own pass), x2 on gfx950 (MI355X_MICROARCH.md), median over the bound-off launches: "
                                 "profiles/r5/summary_headline.md"}, f)
except Exception as e:  # noqa: BLE001
    print(f"(pmc_traffic.json not written: {e})")
for sub in ("fetch_on", "fetch_off"):
    bb = bench_line(f"{d}/bench_{sub}.json")
    if bb:
        print(f"\nbench line of the {sub} pass: algorithmic {bb['roofline']['algorithmic_bytes_per_launch'] / 1e6:.1f} MB/launch, "
              f"kernel-counted requests (timed leg) {bb['pruned']['bytes_requested_per_launch'] / 1e6:.1f} MB/launch, "
              f"bound-off leg {bb['roofline']['bytes_requested_per_launch'] / 1e6:.1f} MB/launch")

print("\n## Issue-slot view (`--pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES`, `--streams 1`)\n")
print("Median per launch (wave-level instruction counts).\n")
print("| kernel | launches | VALU | LDS | SALU | VMEM rd | wave cycles | busy cycles |\n|---|---|---|---|---|---|---|---|")
for k, c in pmc("sq").items():
    g = lambda n: statistics.median(c[n]) if n in c else float("nan")  # noqa: E731
    print(f"| `{k}` | {len(c.get('SQ_INSTS_VALU', []))} | {g('SQ_INSTS_VALU') / 1e6:.2f} M | {g('SQ_INSTS_LDS') / 1e6:.2f} M | "
          f"{g('SQ_INSTS_SALU') / 1e6:.2f} M | {g('SQ_INSTS_VMEM_RD') / 1e6:.2f} M | {g('SQ_WAVE_CYCLES') / 1e6:.1f} M | {g('SQ_BUSY_CYCLES') / 1e6:.1f} M |")

w = pmc("sqwait")
if w:
    print("\n## Where the resident waves' cycles go (`--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS`, `--streams 1`)\n")
    print("Quad-cycles, median per launch.  WAIT_ANY = wave parked (s_waitcnt / barrier); WAIT_INST_ANY = issue stall; ACTIVE_INST_ANY = issuing; "
          "the three add up to about WAVE_CYCLES (MI355X_MICROARCH.md).\n")
    print("| kernel | launches | wave cycles | parked (WAIT_ANY) | issue-stalled | issuing | LDS issue stall | LDS active |\n|---|---|---|---|---|---|---|---|")
    for k, c in w.items():
        if not any(t in k for t in ("k_scan", "k_select", "k_rank", "k_prep")):
            continue
        g = lambda n: statistics.median(c[n]) if n in c else float("nan")  # noqa: E731
        wc = g("SQ_WAVE_CYCLES")
        pct = lambda n: f"{g(n) / 1e6:.1f} M ({100 * g(n) / wc:.0f} %)" if wc == wc and wc > 0 else "n/a"  # noqa: E731
        print(f"| `{k}` | {len(c.get('SQ_WAVE_CYCLES', []))} | {wc / 1e6:.1f} M | {pct('SQ_WAIT_ANY')} | {pct('SQ_WAIT_INST_ANY')} | {pct('SQ_ACTIVE_INST_ANY')} | "
              f"{pct('SQ_WAIT_INST_LDS')} | {pct('SQ_ACTIVE_INST_LDS')} |")
