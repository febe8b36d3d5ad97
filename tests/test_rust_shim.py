"""The Rust side of the boundary (integration/gpu_ivf.rs) checked against include/rbq.h without a Rust
toolchain: every `extern "C"` signature (name, arity, integer widths, pointer constness, pointee), every #[repr(C)]
struct (field order and types), the error-code constants and the map_err table, and the detail strings the RBQ1 reader
can return.  CPU only."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "include", "rbq.h")).read()
RS = open(os.path.join(ROOT, "integration", "gpu_ivf.rs")).read()

C_SCALARS = {"uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "int": "c_int", "size_t": "usize", "float": "f32",
             "double": "f64", "void": "c_void", "char": "c_char", "rbq_index": "RbqIndex", "rbq_builder": "RbqBuilder",
             "rbq_header": "RbqHeader", "rbq_list_view": "RbqListView", "rbq_diag": "RbqDiag"}


def strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def c_type(tokens):
    """canonical form of a C type: nested ('ptr', 'const'|'mut', inner) around a scalar name"""
    toks = [t for t in re.findall(r"\w+|\*", tokens)]
    # leading type (with optional const either side), then a chain of '*' each optionally followed by const
    const = False
    base = None
    i = 0
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            const = True
        elif toks[i] in ("struct", "unsigned", "signed"):
            pass
        else:
            base = toks[i]
        i += 1
    assert base in C_SCALARS, (tokens, base)
    t = C_SCALARS[base]
    while i < len(toks):
        assert toks[i] == "*", tokens
        t = ("ptr", "const" if const else "mut", t)
        const = False
        i += 1
        if i < len(toks) and toks[i] == "const":
            const = True
            i += 1
    return t


def split_args(s):
    return [a.strip() for a in s.split(",") if a.strip()]


def c_functions():
    src = strip_c_comments(HDR)
    out = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(rbq_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        params = []
        if args and args != "void":
            for a in split_args(args):
                mm = re.match(r"(.*?)(\w+)$", a.strip())
                params.append((mm.group(2), c_type(mm.group(1))))
        out[name] = (None if ret == "void" else c_type(ret), params)
    return out


def c_structs():
    src = strip_c_comments(HDR)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(rbq_\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"(.*?)(\w+)$", decl)
            fields.append((mm.group(2), c_type(mm.group(1))))
        out[C_SCALARS[m.group(2)]] = fields
    return out


def rs_type(s):
    s = s.strip()
    m = re.match(r"\*(const|mut)\s+(.*)$", s)
    if m:
        return ("ptr", m.group(1), rs_type(m.group(2)))
    return s


def strip_rs_comments(s):
    return re.sub(r"//[^\n]*", "", s)


def rs_functions():
    src = strip_rs_comments(RS)
    m = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S)
    assert m, 'no extern "C" block'
    out = {}
    for f in re.finditer(r"fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", m.group(1), flags=re.S):
        name, args, ret = f.group(1), f.group(2), f.group(3)
        params = []
        for a in split_args(" ".join(args.split())):
            pn, pt = a.split(":", 1)
            params.append((pn.strip(), rs_type(pt)))
        out[name] = (rs_type(ret) if ret else None, params)
    return out


def rs_structs():
    src = strip_rs_comments(RS)
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*pub struct (\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(","):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"(?:pub\s+)?(\w+)\s*:\s*(.*)$", decl, flags=re.S)
            fields.append((mm.group(1), rs_type(mm.group(2))))
        out[m.group(1)] = fields
    return out


PRODUCT_API = ["rbq_index_create", "rbq_index_load_rbq1", "rbq_index_build_device", "rbq_build_stream_begin", "rbq_build_stream_push",
               "rbq_build_stream_finish", "rbq_build_stream_abort", "rbq_index_destroy", "rbq_index_len", "rbq_index_cluster_count",
               "rbq_index_dim", "rbq_index_padded_dim", "rbq_index_device_count", "rbq_search_batch", "rbq_posting_scan_batch",
               "rbq_search_batch_device", "rbq_release_stream", "rbq_host_alloc", "rbq_host_free", "rbq_index_set_rerank_vectors",
               "rbq_debug_set_option", "rbq_strerror", "rbq_last_error_detail", "rbq_abi_version"]


def test_every_extern_signature_matches_the_header():
    cf, rf = c_functions(), rs_functions()
    assert len(cf) >= 30, sorted(cf)
    for name, (rret, rparams) in rf.items():
        assert name in cf, f"{name} is not declared in include/rbq.h"
        cret, cparams = cf[name]
        assert rret == cret, f"{name}: return type {rret} vs {cret}"
        assert len(rparams) == len(cparams), f"{name}: arity {len(rparams)} vs {len(cparams)}"
        for (rn, rt), (cn, ct) in zip(rparams, cparams):
            assert rt == ct, f"{name}({cn}): rust {rt} vs C {ct}"
    missing = [n for n in PRODUCT_API if n not in rf]
    assert not missing, f"product entry points without a Rust binding: {missing}"


def test_repr_c_structs_match_the_header():
    cs, rs = c_structs(), rs_structs()
    for name in ("RbqHeader", "RbqListView", "RbqDiag"):
        assert name in rs, name
        assert [t for _, t in rs[name]] == [t for _, t in cs[name]], (name, rs[name], cs[name])
        assert [n for n, _ in rs[name]] == [n for n, _ in cs[name]], (name, rs[name], cs[name])
    for opaque in ("RbqIndex", "RbqBuilder"):
        assert rs[opaque] == [("_p", "[u8; 0]")]


def test_error_codes_and_map_err_cover_the_header():
    codes = dict(re.findall(r"#define\s+(RBQ_(?:OK|DIMENSION_MISMATCH|INVALID_CONFIG|EMPTY_INDEX|IO|INVALID_PERSISTENCE|DEVICE))\s+(\d+)", HDR))
    assert len(codes) == 7
    for k, v in codes.items():
        m = re.search(rf"pub const {k}: c_int = (\d+);", RS)
        assert m and m.group(1) == v, k
    body = RS[RS.index("fn map_err"):]
    body = body[:body.index("\n}\n")]
    arms = {"RBQ_OK": "Ok(())", "RBQ_DIMENSION_MISMATCH": "RabitqError::DimensionMismatch", "RBQ_INVALID_CONFIG": "RabitqError::InvalidConfig",
            "RBQ_EMPTY_INDEX": "RabitqError::EmptyIndex", "RBQ_IO": "RabitqError::Io", "RBQ_INVALID_PERSISTENCE": "RabitqError::InvalidPersistence",
            "RBQ_DEVICE": "RabitqError::Io"}
    for code, what in arms.items():
        m = re.search(rf"{code}\s*=>\s*(.*?)(?=\n        RBQ_|\n        other)", body, flags=re.S)
        assert m and what in m.group(1), (code, what)


def test_known_detail_strings_cover_the_rbq1_reader():
    """every InvalidPersistence string the RBQ1 parser (the reference's own, src/ivf.rs:1484-1702) can return has a
    static twin in the Rust table (InvalidPersistence carries &'static str)"""
    logic = open(os.path.join(ROOT, "rabitq-rs_amd", "csrc", "host", "rbq_host_logic.hpp")).read()
    strings = set(re.findall(r'RBQ_INVALID_PERSISTENCE,\s*(?:[^"\n]*\?\s*)?"([^"]+)"', logic))
    strings |= set(re.findall(r':\s*"([^"]+ mismatch)"\)', logic))
    assert len(strings) >= 15, strings
    table = RS[RS.index("const KNOWN_DETAILS"):RS.index("];", RS.index("const KNOWN_DETAILS"))]
    known = set(re.findall(r'"([^"]+)"', table))
    assert strings <= known, strings - known


def test_shim_uses_the_reference_api_names():
    for needle in ("pub fn search(&self, query: &[f32], params: SearchParams) -> Result<Vec<SearchResult>, RabitqError>",
                   "pub fn search_filtered(", "filter: &RoaringBitmap", "pub fn batch_search(&self, queries: &[&[f32]], params: SearchParams)",
                   "Vec<Result<Vec<SearchResult>, RabitqError>>", "index.save_to_writer(&mut buf)?", "impl Drop for GpuIvf"):
        assert needle in RS, needle
    build = open(os.path.join(ROOT, "integration", "build.rs")).read()
    assert "cargo:rustc-link-lib=dylib=rbq" in build and "RBQ_LIB_DIR" in build


@pytest.mark.parametrize("decl,want", [("const float*", ("ptr", "const", "f32")), ("rbq_index**", ("ptr", "mut", ("ptr", "mut", "RbqIndex"))),
                                        ("const rbq_index*", ("ptr", "const", "RbqIndex")), ("uint64_t", "u64"),
                                        ("const uint8_t*", ("ptr", "const", "u8")), ("void*", ("ptr", "mut", "c_void"))])
def test_c_type_parser(decl, want):
    assert c_type(decl) == want
