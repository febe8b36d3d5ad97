"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): queries are sharded across ranks with the
index replicated; the only exchange is the all_gather of per-rank packed [batch][top_k] (id, score) blocks
(SURVEY.md §8e).  The per-rank search is played by the oracle here (no GPU in this container); the packing, the
gather and the unpacking are bench.py's own code (bench.TopkExchange — the object the timed loop uses on HBM buffers
over RCCL): order preservation and bit-equality with the unsharded result, with two batches in flight on two
exchange slots, per batch and per bucket of batches."""
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
    import bench
    import oracle
    from conftest import build_index, make_dataset
    data, built = build_index(n=1500, dim=64, nlist=12, total_bits=7, seed=77)   # replicated: same seed on every rank
    top_k, nprobe, batch = 10, 6, 16
    ok = True
    for bucket in (1, 2):  # one all_gather per batch / one per bucket of two batches (what the timed loop does, with ns batches)
        ex = bench.TopkExchange(torch, torch.device("cpu"), batch, top_k, 2, world, gather=True, bucket=bucket)
        for slot, seed in ((0, 123), (1, 124)):                                          # two batches, two result slots
            q_all = make_dataset(batch * world, 64, 3, seed)
            q = q_all[rank * batch:(rank + 1) * batch]                                   # contiguous query shard
            rc, ids, sc, cnt, _ = oracle.search_batch(built, q, top_k, nprobe)
            ex.ids[slot].copy_(torch.from_numpy(ids.view(np.int64).copy()))            # what rbq_search_batch_device writes
            ex.scores[slot].copy_(torch.from_numpy(sc.copy()))
            if bucket == 1 or slot == 1:
                ex.gather(dist, slot // bucket)                                          # the path's only collective
        dist.barrier()
        if rank == 0:
            for slot, seed in ((0, 123), (1, 124)):
                q_all = make_dataset(batch * world, 64, 3, seed)
                gid, gsc = ex.unpack(slot)
                rc, rids, rsc, rcnt, _ = oracle.search_batch(built, q_all, top_k, nprobe)
                ok &= np.array_equal(gid.numpy().view(np.uint64), rids) and np.array_equal(gsc.numpy().view(np.uint32), rsc.view(np.uint32))
    if rank == 0:
        open(os.path.join(out_dir, "ok"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_query_sharding_and_topk_gather(tmp_path, world):
    """world 2, and world 8 = the shape of the driver's 8-GPU run (eight contiguous query shards, one bucketed all_gather):
    the gathered blocks are the unsharded result, rank order = query order"""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / "ok").read() == "1"
