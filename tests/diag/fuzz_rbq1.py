"""Diagnostic (CPU only): rbq_index_load_rbq1 against mutated RBQ1 streams whose CRC is made VALID again, so that the
parser goes past the checksum into its size / count / offset checks.  Every mutation must come back as an error code
(or reach the device stage, which fails here for lack of a GPU) — never crash, never allocate without bound.  Run in
a child process per batch so that a crash is reported instead of ending the run:  python tests/diag/fuzz_rbq1.py N"""
import ctypes as C
import os, subprocess, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(first, last):
    import resource
    resource.setrlimit(resource.RLIMIT_AS, (24 << 30, 24 << 30))  # a runaway allocation fails instead of taking the box
    import numpy as np
    import conftest
    import rabitq_rs_amd as rq
    from rabitq_rs_amd import index as ix
    outcomes = {}
    for seed in range(first, last):
        rng = np.random.default_rng(seed)
        bits, metric, rot = int(rng.choice([1, 3, 7])), int(rng.integers(0, 2)), int(rng.integers(0, 2))
        dim = int(rng.choice([16, 48, 64, 100])) if rot == 1 else 32
        data, built = conftest.build_index(n=int(rng.integers(40, 400)), dim=dim, nlist=int(rng.integers(1, 6)),
                                           total_bits=bits, metric=metric, rotator=rot, seed=seed)
        blob = bytearray(built.save_rbq1())
        for m in range(40):
            bad = bytearray(blob)
            kind = int(rng.integers(0, 5))
            if kind == 0:    # header field overwritten with an extreme value
                off = int(rng.integers(4, min(64, len(bad) - 12)))
                vals = [0, 1, 0xff, 0xffff, 2**31, 2**32 - 1, 2**63, 2**64 - 1]
                bad[off:off + 8] = vals[int(rng.integers(0, len(vals)))].to_bytes(8, "little")
            elif kind == 1:  # random byte anywhere
                bad[int(rng.integers(0, len(bad) - 4))] = int(rng.integers(0, 256))
            elif kind == 2:  # a run of bytes replaced
                off = int(rng.integers(8, len(bad) - 20))
                bad[off:off + 8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8))
            elif kind == 3:  # truncated
                bad = bad[:int(rng.integers(8, len(bad) - 4))] + bad[-4:]
            else:            # bytes inserted (all later offsets shift)
                off = int(rng.integers(8, len(bad) - 4))
                bad[off:off] = bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
            if len(bad) >= 12:
                bad[-4:] = zlib.crc32(bytes(bad[8:-4])).to_bytes(4, "little")  # CRC32 over [8, len-4)
            h = C.c_void_p()
            buf = (C.c_uint8 * len(bad)).from_buffer_copy(bytes(bad))
            rc = ix.lib().rbq_index_load_rbq1(buf, len(bad), 1, None, C.byref(h))
            key = (rc, ix._detail()[:60] if rc else "loaded")
            outcomes[key] = outcomes.get(key, 0) + 1
            assert rc != 0 or h.value, "success without a handle"
            if rc == 0:
                ix.lib().rbq_index_destroy(h)
    for k, v in sorted(outcomes.items(), key=lambda kv: -kv[1]):
        print("   %6d  rc=%d  %s" % (v, k[0], k[1]))


if __name__ == "__main__":
    if len(sys.argv) > 2:
        child(int(sys.argv[1]), int(sys.argv[2]))
    else:
        n = int(sys.argv[1])
        for a in range(0, n, 10):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), str(a), str(min(n, a + 10))], capture_output=True, text=True)
            print("seeds %d-%d: exit %d" % (a, min(n, a + 10) - 1, out.returncode))
            print(out.stdout, end="")
            if out.returncode != 0:
                print(out.stderr[-1500:])
