"""What each compile-time numeric variant of the reference is worth: for cfg3 (GIST-1M shape, 128 queries) and
for the tests' seeded generator, the fraction of queries whose ids / score bits differ from the default variant
(target-cpu=native on an AVX-512 host).  Oracle only (CPU); cfg3's index is built with bench.py's machinery, so run that part on
the GPU box:   python tools/variant_table.py [--config cfg3] [--small-only]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
import rabitq_rs_amd as rq

small_only = "--small-only" in sys.argv
sys.argv = [a for a in sys.argv if a != "--small-only"]
out = {}
# the seeded generator of the parity tests: a few shapes, 7-bit / 3-bit, L2 / IP, in-distribution and out-of-distribution queries
from conftest import build_index, make_dataset
agg = {}
nq_total = 0
for seed in range(12):
    rng = np.random.default_rng(9000 + seed)
    dim = int(rng.choice([64, 128, 200, 384, 960])); bits = int(rng.choice([3, 7])); metric = int(rng.integers(0, 2))
    n = int(rng.integers(3000, 9000)); nlist = int(rng.integers(16, 64))
    data, built = build_index(n=n, dim=dim, nlist=nlist, total_bits=bits, metric=metric, normalize=(metric == 1), seed=9100 + seed)
    q = np.concatenate([make_dataset(96, dim, max(nlist // 4, 1), 9200 + seed, normalize=(metric == 1)), data[:32] + np.float32(1e-3)])
    tab = oracle.variant_diff_table(built, q, 10, max(4, nlist // 3))
    nq_total += len(q)
    for name, row in tab.items():
        a = agg.setdefault(name, {"ids": 0.0, "bits": 0.0, "rel": 0.0, "scale": 0.0})
        a["ids"] += row["ids_differ_frac"] * len(q); a["bits"] += row["score_bits_differ_frac"] * len(q)
        a["rel"] = max(a["rel"], row["max_rel_score_diff_same_ids"]); a["scale"] = max(a["scale"], row["max_score_diff_over_scale_same_ids"])
out["seeded_generator"] = {"queries": nq_total, "variants": {k: {"ids_differ_frac": v["ids"] / nq_total, "score_bits_differ_frac": v["bits"] / nq_total,
                           "max_rel_score_diff_same_ids": v["rel"], "max_score_diff_over_scale_same_ids": v["scale"]} for k, v in agg.items()}}
print(json.dumps({"seeded_generator": out["seeded_generator"]}), flush=True)
if not small_only:
    import torch, bench
    a = bench.parse()
    dev = torch.device("cuda", 0)
    mix = bench.Mixture(torch, dev, a.dim, a.nlist, a.dataset, a.metric == 1)
    x = mix.draw(a.n, 20260105)
    cent, assign = bench.kmeans_gpu(torch, x, a.nlist, 6, 20260103)
    built = rq.builder.train_with_clusters(x.cpu().numpy(), cent.cpu().numpy(), assign.cpu().numpy().astype(np.uint32), a.bits, a.metric, 1, 20260104, True)
    q = mix.draw(128, 20260102).cpu().numpy()
    out["cfg"] = {"n": a.n, "dim": a.dim, "nlist": a.nlist, "nprobe": a.nprobe, "bits": a.bits, "metric": a.metric, "top_k": a.top_k,
                  "variants": oracle.variant_diff_table(built, q, a.top_k, a.nprobe)}
    print(json.dumps({"cfg": out["cfg"]}), flush=True)
