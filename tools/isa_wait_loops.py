"""Diagnostic (no GPU needed): loops that wait for every load.
hipcc batches the loads of a copy loop only when it unrolls it; a loop that carries state across iterations, or whose trip count it does
not know, is compiled as  load -> s_waitcnt vmcnt(0) -> use  per iteration: one dependent global round trip per iteration (over PCIe when
the pointer is page-locked host memory).  This script compiles the kernel translation units to ISA and lists every self-loop that holds
a few global / flat loads and an `s_waitcnt vmcnt(0)` — round 5 found the query's initial load (15 round trips per query at D = 960),
k_scanw's LUT + rotated-query prologue, the selection's staged centroid rows and the cfg5 prefilter (256 per pass) this way.
usage: python tools/isa_wait_loops.py [unit ...]      (default: k_query k_scan k_scanw)
Remainder loops of an unrolled copy and scalar tails show up too: read the loop body before acting on a line."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
units = sys.argv[1:] or ["k_query", "k_scan", "k_scanw"]
seen = collections.Counter()
for u in units:
    out = f"/tmp/isa_{u}.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-rdc",
                           "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out,
                           os.path.join(ROOT, "rabitq-rs_amd", "csrc", "device", u + ".hip")], stderr=subprocess.DEVNULL)
    for f in re.split(r"\n(?=_ZN3rbq[\w]+:\s)", open(out).read()):
        name = f.split(":")[0]
        if not name.startswith("_ZN3rbq"):
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void rbq::", "").replace("rbq::", "")
        lines = f.split("\n")
        lines = lines[:next((i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")), len(lines))]
        blocks, cur = [], None
        for l in lines:
            if re.match(r"^\.LBB\d+_\d+:", l):
                cur = {"label": l.split(":")[0], "hdr": l, "body": []}
                blocks.append(cur)
            elif cur is not None:
                cur["body"].append(l.strip())
        for b in blocks:
            body = b["body"]
            if "Loop Header" not in b["hdr"] or not any("s_cbranch" in x and b["label"] in x for x in body):
                continue
            nload = sum(1 for x in body if x.startswith(("global_load", "flat_load")))
            waits = sum(1 for x in body if x.startswith("s_waitcnt") and "vmcnt(0)" in x)
            if nload and waits and nload <= 4:
                mem = tuple(x.split()[0] for x in body if x.startswith(("global_load", "flat_load", "global_store", "ds_write", "ds_read")))
                seen[(dem, len(body), mem)] += 1
for (k, n, mem), c in sorted(seen.items()):
    print(f"{k:40s} {n:3d} instructions  x{c}  |  {' '.join(mem)[:110]}")
print(f"{len(seen)} distinct loops")
