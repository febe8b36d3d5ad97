#!/bin/bash
# tools/sq_pass.sh TAG [bench args] : one PMC pass of SQ instruction counters (median per k_scan launch), for the library
# selected by RBQ_LIB_PATH (default: in-tree)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/sq_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $out -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --min-seconds 0 --nbatches 8 --streams 1 "$@" > $out/bench.json 2> $out/log || exit 1
python3 - $out <<'PY'
import csv, glob, statistics, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rbq::k_scan' in r['Kernel_Name']: rows[r['Counter_Name']][r['Dispatch_Id']].append(float(r['Counter_Value']))
for c, d in rows.items():
    v = [sum(x) for x in d.values()]
    print(sys.argv[1], c, 'median per launch %.2f M over %d launches' % (statistics.median(v) / 1e6, len(v)))
PY
