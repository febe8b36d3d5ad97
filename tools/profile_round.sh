#!/bin/bash
# tools/profile_round.sh TAG [bench args...] : the profile set of a round, on the GPU box, written under gpurun_out/prof_TAG/:
#   stats/   rocprofv3 --kernel-trace --stats of a 20-step bench run (what the driver runs, without the CPU leg)
#   fetch_on/ fetch_off/   PMC pass FETCH_SIZE (its own run), block bound on / off
#   sq/      PMC pass SQ instruction counters (its own run)
# then tools/profile_summary.py turns them into gpurun_out/prof_TAG/summary.md (copy that into profiles/).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --no-cpu --no-extras --min-seconds 0 "$@" > $out/bench_stats.json 2> $out/stats.log || exit 1
echo "[profile] stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_on -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --min-seconds 0 --nbatches 8 "$@" > $out/bench_fetch_on.json 2> $out/fetch_on.log || exit 1
echo "[profile] FETCH_SIZE (bound on) done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_off -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --min-seconds 0 --nbatches 8 --option block_bound=0 "$@" > $out/bench_fetch_off.json 2> $out/fetch_off.log || exit 1
echo "[profile] FETCH_SIZE (bound off) done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --min-seconds 0 --nbatches 8 --streams 1 --no-wave-roofline "$@" > $out/bench_sq.json 2> $out/sq.log || exit 1
echo "[profile] SQ pass done"
# where the resident waves' cycles go: parked (s_waitcnt / barrier), issue-stalled, issuing; LDS-issue stalls (8 SQ counters per pass)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/sqwait -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --min-seconds 0 --nbatches 8 --streams 1 --no-wave-roofline "$@" > $out/bench_sqwait.json 2> $out/sqwait.log || echo "[profile] wait-counter pass FAILED (see $out/sqwait.log)"
echo "[profile] SQ wait pass done"
python3 tools/profile_summary.py $out "$@" > $out/summary.md
cat $out/summary.md
