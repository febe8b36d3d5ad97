"""Diagnostic: mean of every PMC counter per (kernel, workgroups) from a rocprofv3 --pmc counter_collection.csv.
python3 tools/pmc_by_kernel.py <counter_collection.csv> [max workgroups]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name']
    if 'rbq::' not in n:
        continue
    short = n.split('rbq::')[1].split('(')[0].split('<')[0]
    wg = int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1)
    if wg <= lim:
        acc[(short, wg)][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print("%-16s wg %5d  n %3d  " % (k[0], k[1], len(next(iter(acc[k].values())))) + "  ".join("%s %.0f" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
