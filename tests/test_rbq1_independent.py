"""RBQ1-v3 against an INDEPENDENT writer (tests/rbq1_writer.py: pure Python struct + zlib, written from the reference's
save_to_writer, src/ivf.rs:1317-1474): its bytes must equal the C++ writer's, the library's reader must accept them, and
every validation message of load_from_reader (src/ivf.rs:1484-1702) — the 1 M-vectors-per-cluster cap included — must come
back for the stream that provokes it.  CPU only: load_from_reader's checks run before any device work."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

import rabitq_rs_amd as rq
from conftest import build_index
from rabitq_rs_amd import index as ix
from rbq1_writer import from_built, write_rbq1

P = rq._abi.RBQ_INVALID_PERSISTENCE


def _load_err(blob):
    h = C.c_void_p()
    buf = (C.c_uint8 * len(blob)).from_buffer_copy(bytes(blob))
    rc = ix.lib().rbq_index_load_rbq1(buf, len(blob), 1, None, C.byref(h))
    assert not h.value
    return rc, ix._detail()


@pytest.mark.parametrize("bits,metric,rotator,dim,n,nlist", [(7, 0, 1, 64, 300, 4), (3, 1, 1, 100, 257, 3), (1, 0, 1, 48, 95, 2),
                                                              (7, 1, 0, 32, 130, 5), (3, 0, 0, 32, 64, 1), (7, 0, 1, 960, 70, 2)])
def test_independent_writer_reproduces_the_cpp_writer_byte_for_byte(bits, metric, rotator, dim, n, nlist):
    _, built = build_index(n=n, dim=dim, nlist=nlist, total_bits=bits, metric=metric, rotator=rotator, seed=bits * 100 + dim)
    mine = from_built(built)
    theirs = built.save_rbq1()
    assert len(mine) == len(theirs)
    assert mine == theirs
    # the facts the reference's own persistence tests state (src/tests.rs:471-517)
    assert mine[:4] == b"RBQ1" and struct.unpack_from("<I", mine, 4)[0] == 3
    assert struct.unpack_from("<Q", mine, 20)[0] == n                       # vector_count at offset 20
    assert struct.unpack_from("<I", mine, len(mine) - 4)[0] == zlib.crc32(mine[8:-4])  # CRC over [8, len - 4)


def _tiny():
    """a 2-cluster 7-bit stream described by hand (no builder involved): D = 64, clusters of 33 and 1 vectors"""
    D, ex = 64, 6
    rng = np.random.default_rng(5)
    clusters = []
    for n in (33, 1):
        nb = (n + 31) // 32
        clusters.append({"centroid": rng.standard_normal(D).astype(np.float32).tolist(), "ids": list(range(100, 100 + n)),
                         "batch_data": rng.integers(0, 256, nb * (D * 4 + 384), dtype=np.uint8).tobytes(),
                         "ex_codes": [rng.integers(0, 256, D * ex // 8, dtype=np.uint8).tobytes() for _ in range(n)],
                         "f_add_ex": [0.5] * n, "f_rescale_ex": [1.5] * n, "delta": [0.0] * n, "vl": [0.0] * n})
    return dict(dim=60, padded_dim=D, metric_tag=0, rotator_tag=1, ex_bits=ex, rotator_bytes=bytes(range(4 * D // 8)), clusters=clusters)


def _msg(**changes):
    kw = _tiny()
    kw.update(changes)
    return _load_err(write_rbq1(**kw))


def test_every_validation_message_of_load_from_reader():
    """each stream is written VALID except for the one field under test (CRC included), so the message is the check's own"""
    assert _msg(magic=b"RBQ2") == (P, "unrecognized file header")
    assert _msg(version=2) == (P, "unsupported index format version (expected V3 with unified memory layout)")
    assert _msg(version=4)[1].startswith("unsupported index format version")
    assert _msg(dim=0) == (P, "dimension must be positive")
    assert _msg(dim=65) == (P, "padded_dim must be >= dim")
    assert _msg(metric_tag=2) == (P, "unknown metric tag")
    assert _msg(rotator_tag=2) == (P, "unknown rotator type tag")
    assert _msg(ex_bits=17, total_bits=18) == (P, "ex_bits out of range")
    assert _msg(ex_bits=16, total_bits=17) == (P, "total_bits out of range")
    assert _msg(total_bits=0) == (P, "total_bits out of range")
    assert _msg(total_bits=6) == (P, "total_bits does not match ex_bits")
    assert _msg(rotator_bytes=bytes(31)) == (P, "FHT rotator flip bits length mismatch")      # src/rotation.rs:491-497
    assert _msg(rotator_tag=0) == (P, "rotator matrix length mismatch")                       # 32 bytes are no 64 x 64 f32 matrix (src/rotation.rs:213-219)
    assert _msg(vector_count=35) == (P, "vector count metadata mismatch")
    kw = _tiny(); kw["clusters"][0]["batch_data"] = kw["clusters"][0]["batch_data"][:-1]
    assert _load_err(write_rbq1(**kw)) == (P, "batch_data length mismatch - possible corruption or version incompatibility")
    kw = _tiny(); kw["clusters"][1]["ex_codes"][0] = kw["clusters"][1]["ex_codes"][0] + b"\0"
    assert _load_err(write_rbq1(**kw)) == (P, "ex_code_packed length mismatch - possible corruption or version incompatibility")
    kw = _tiny(); kw["ex_bits"] = 0  # total_bits 1: every packed ex code must then be EMPTY (src/ivf.rs:1603-1607)
    assert _load_err(write_rbq1(**kw)) == (P, "ex_code_packed length mismatch - possible corruption or version incompatibility")
    good = bytearray(write_rbq1(**_tiny()))
    bad = bytearray(good); bad[-1] ^= 0x01
    assert _load_err(bad) == (P, "checksum mismatch")
    bad = bytearray(good); bad[len(bad) // 2] ^= 0x10  # a flipped payload bit changes nothing the parser checks but the CRC
    assert _load_err(bad) == (P, "checksum mismatch")
    for cut in (3, 7, 11, 40, 200, len(good) - 5, len(good) - 1):  # read_exact on a short stream -> Io(UnexpectedEof)
        rc, msg = _load_err(good[:cut])
        assert rc == rq._abi.RBQ_IO and msg == "failed to fill whole buffer", (cut, rc, msg)


def test_cluster_size_cap_is_one_million_exactly():
    """MAX_CLUSTER_SIZE = 1_000_000 (src/ivf.rs:1561-1566): n = 1_000_001 is 'possible corruption'; n = 1_000_000 passes the cap
    (and then fails on the bytes that are not there)."""
    kw = _tiny()
    blob = bytearray(write_rbq1(**kw))
    off = 8 + 12 + 24 + len(kw["rotator_bytes"]) + 64 * 4  # header | counts | rotator | first centroid -> first num_vectors
    assert struct.unpack_from("<Q", blob, off)[0] == 33
    struct.pack_into("<Q", blob, off, 1_000_001)
    assert _load_err(blob) == (P, "cluster size exceeds reasonable limits - possible corruption")
    struct.pack_into("<Q", blob, off, 1_000_000)
    rc, msg = _load_err(blob)
    assert rc == rq._abi.RBQ_IO and msg == "failed to fill whole buffer"
    struct.pack_into("<Q", blob, off, 2**63)
    assert _load_err(blob) == (P, "cluster size exceeds reasonable limits - possible corruption")


def test_hand_written_stream_parses_to_its_own_arrays():
    """the sanitizer shim's parser view of a hand-described stream: list count, vector count, and the CRC the library computes"""
    kw = _tiny()
    blob = write_rbq1(**kw)
    assert ix.lib().rbq_abi_version() >> 16 == 2
    # rbq1_parse accepts it: the only failure left on a GPU-less host is the device step
    h = C.c_void_p()
    buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
    rc = ix.lib().rbq_index_load_rbq1(buf, len(blob), 1, None, C.byref(h))
    if rc == 0:
        assert ix.lib().rbq_index_len(h) == 34 and ix.lib().rbq_index_cluster_count(h) == 2
        ix.lib().rbq_index_destroy(h)
    else:
        assert rc == rq._abi.RBQ_DEVICE, (rc, ix._detail())
