"""Diagnostic soak: the seeded random configurations of tests/test_gpu_parity.py over a long seed range, in one
process (python tests/diag/soak.py FIRST LAST).  Prints the seeds whose result differs from the oracle.
Import paths are set up exactly as tests/conftest.py does (oracle/oracle.py must win over the oracle/ directory).
A harness error (the same non-assertion exception three times in a row) aborts the run: it says nothing about parity."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401  (inserts ROOT and ROOT/oracle into sys.path in the order the tests use)
import __graft_entry__ as g
g.build_cpu_libs()
import oracle
assert hasattr(oracle, "search_batch"), "wrong `oracle` module on sys.path: %r" % (oracle,)
import test_gpu_parity as t

first, last = int(sys.argv[1]), int(sys.argv[2])
bad, harness = [], []
t0 = time.time()
for seed in range(first, last):
    try:
        t.test_random_configurations_match_oracle(seed)
        harness = []
    except AssertionError:
        bad.append(seed)
        harness = []
        print("MISMATCH seed", seed, t._random_case(seed), traceback.format_exc().splitlines()[-1][:300], flush=True)
    except Exception:
        msg = traceback.format_exc().splitlines()[-1][:300]
        print("ERROR seed", seed, t._random_case(seed), msg, flush=True)
        harness.append(msg)
        if len(harness) >= 3 and len(set(harness[-3:])) == 1:
            print("aborting: harness error, no parity information in this run")
            sys.exit(2)
    if (seed - first) % 50 == 49:
        print("... %d seeds, %d mismatches, %.0f s" % (seed - first + 1, len(bad), time.time() - t0), flush=True)
print("done: %d seeds, mismatches: %s" % (last - first, bad))
sys.exit(1 if bad else 0)
