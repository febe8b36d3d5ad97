#!/bin/bash
# tools/pmc.sh OUTDIR "COUNTERS..." : one rocprofv3 PMC pass over bench.py (GPU box). Counters in their own run.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=$1; shift
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $out -- python3 bench.py --steps 3 --warmup 1 --no-cpu ${BENCH_ARGS} > $out.log 2>&1
python3 - "$out" <<'PY'
import sys, glob, csv, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rbq::' in r['Kernel_Name']:
            d[r['Kernel_Name'].split('(')[0][-28:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in d.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
