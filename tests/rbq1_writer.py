"""An independent RBQ1-v3 writer: pure Python `struct` + `zlib.crc32`, written from the reference's
`IvfRabitqIndex::save_to_writer` (src/ivf.rs:1317-1474) alone — field by field, in its order — and sharing no code with the
repository's C++ writer (csrc/host/rbq_build.cpp: rbq_built_save_rbq1) or reader (csrc/host/rbq_host_logic.hpp).  TEST
INFRASTRUCTURE: it breaks the writer/reader symmetry of the RBQ1 tests (round-3 VERDICT: the reader had only ever read
bytes of the repository's own writer).

Stream (all little-endian; the CRC-32/IEEE covers every byte after the version word, src/tests.rs:503-506):
    "RBQ1" | u32 version = 3                                   (not hashed, src/ivf.rs:1319-1320)
    u32 dim | u32 padded_dim | u8 metric_tag | u8 rotator_type | u8 ex_bits | u8 total_bits = ex_bits + 1
    u64 vector_count | u64 cluster_count | u64 rotator_len | rotator bytes
    per cluster:  f32 centroid[padded_dim] | u64 num_vectors | u64 ids[n] | u64 batch_data_len | batch_data bytes |
                  n x (u64 ex_code_len | ex_code bytes) | f32 f_add_ex[n] | f32 f_rescale_ex[n] | f32 delta[n] | f32 vl[n]
    u32 crc32                                                  (not hashed)
"""
import struct
import zlib

MAGIC = b"RBQ1"
VERSION = 3


def write_rbq1(dim, padded_dim, metric_tag, rotator_tag, ex_bits, rotator_bytes, clusters, vector_count=None, total_bits=None,
               version=VERSION, magic=MAGIC):
    """clusters: iterable of dicts with centroid (padded_dim floats), ids (ints), batch_data (bytes), ex_codes (list of
    bytes, one per vector), f_add_ex, f_rescale_ex, delta, vl (floats, one per vector).  vector_count / total_bits /
    version / magic can be overridden to write deliberately inconsistent streams."""
    clusters = list(clusters)
    body = bytearray()
    body += struct.pack("<I", dim)
    body += struct.pack("<I", padded_dim)
    body += struct.pack("<B", metric_tag)
    body += struct.pack("<B", rotator_tag)
    body += struct.pack("<B", ex_bits)
    body += struct.pack("<B", ex_bits + 1 if total_bits is None else total_bits)
    n_total = sum(len(c["ids"]) for c in clusters)
    body += struct.pack("<Q", n_total if vector_count is None else vector_count)
    body += struct.pack("<Q", len(clusters))
    body += struct.pack("<Q", len(rotator_bytes))
    body += bytes(rotator_bytes)
    for c in clusters:
        n = len(c["ids"])
        for v in c["centroid"]:
            body += struct.pack("<f", v)
        body += struct.pack("<Q", n)
        for i in c["ids"]:
            body += struct.pack("<Q", int(i))
        bd = bytes(c["batch_data"])
        body += struct.pack("<Q", len(bd))
        body += bd
        for code in c["ex_codes"]:
            code = bytes(code)
            body += struct.pack("<Q", len(code))
            body += code
        for name in ("f_add_ex", "f_rescale_ex", "delta", "vl"):
            for v in c[name]:
                body += struct.pack("<f", v)
    crc = zlib.crc32(bytes(body)) & 0xFFFFFFFF
    return bytes(magic) + struct.pack("<I", version) + bytes(body) + struct.pack("<I", crc)


def from_built(built, **overrides):
    """The stream of a CPU-built index (rabitq_rs_amd.builder.BuiltIndex), every array taken from its ClusterData views."""
    h = built.header
    D, ex = int(h.padded_dim), int(h.ex_bits)
    clusters = []
    for c in range(int(h.n_lists)):
        a = built.list_arrays(c)
        n = len(a["ids"])
        clusters.append({"centroid": [float(v) for v in a["centroid"]], "ids": [int(i) for i in a["ids"]],
                         "batch_data": a["batch_data"].tobytes(),
                         "ex_codes": [a["ex_codes"][v].tobytes() if ex else b"" for v in range(n)],
                         "f_add_ex": a["f_add_ex"].tolist(), "f_rescale_ex": a["f_rescale_ex"].tolist(),
                         "delta": a["delta"].tolist(), "vl": a["vl"].tolist()})
        assert len(a["centroid"]) == D
    return write_rbq1(int(h.dim), D, int(h.metric), int(h.rotator), ex, built.rotator_blob(), clusters, **overrides)
