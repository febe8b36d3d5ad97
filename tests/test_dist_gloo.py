"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): queries are sharded across ranks with the
index replicated; the only exchange is the all_gather of per-rank [batch][top_k] (id, score) blocks
(SURVEY.md §8e).  The per-rank search is played by the oracle here (no GPU in this container) — what is under
test is the sharding/gather logic bench.py uses: order preservation and equality with the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from conftest import build_index, make_dataset
    data, built = build_index(n=1500, dim=64, nlist=12, total_bits=7, seed=77)   # replicated: same seed on every rank
    q_all = make_dataset(32, 64, 3, 123)
    per = q_all.shape[0] // world
    q = q_all[rank * per:(rank + 1) * per]                                          # contiguous query shard
    rc, ids, sc, cnt, _ = oracle.search_batch(built, q, 10, 6)
    t_ids = torch.from_numpy(ids.view(np.int64).copy()); t_sc = torch.from_numpy(sc.copy())
    g_ids = [torch.empty_like(t_ids) for _ in range(world)]; g_sc = [torch.empty_like(t_sc) for _ in range(world)]
    dist.all_gather(g_ids, t_ids); dist.all_gather(g_sc, t_sc)                      # the path's only collective
    dist.barrier()
    if rank == 0:
        full_ids = torch.cat(g_ids).numpy().view(np.uint64); full_sc = torch.cat(g_sc).numpy()
        rc, rids, rsc, rcnt, _ = oracle.search_batch(built, q_all, 10, 6)
        ok = np.array_equal(full_ids, rids) and np.array_equal(full_sc.view(np.uint32), rsc.view(np.uint32))
        open(os.path.join(out_dir, "ok"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


def test_query_sharding_and_topk_gather_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok").read() == "1"
