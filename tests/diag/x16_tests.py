"""Runs the two multi-stream parity tests with the workgroup-per-query k_prep forced (wg_prep=1) — under
RBQ_LIB_PATH=<x16 variant> this is the combination rank_mfma.hpp describes.  python tests/diag/x16_tests.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import __graft_entry__ as g
g.build_cpu_libs()
import rabitq_rs_amd as rq
orig = rq.IvfRabitqIndex.from_built.__func__
def forced(cls, built, device=None, devices=None):
    idx = orig(cls, built, device, devices)
    idx.set_option("wg_prep", 1)
    return idx
rq.IvfRabitqIndex.from_built = classmethod(forced)
import test_gpu_parity as t1, test_gpu_round2 as t2
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
fails = 0
for r in range(rounds):
    for f in (lambda: t1.test_concurrent_streams_match_oracle(1), t2.test_matrix_rotator_concurrent_streams):
        try:
            f()
        except AssertionError as e:
            fails += 1
            print("round", r, getattr(f, "__name__", "test"), "FAILED:", str(e)[:200], flush=True)
print(os.environ.get("RBQ_LIB_PATH", "default").split("librbq_")[-1], ": %d failures in %d rounds of both tests" % (fails, rounds))
