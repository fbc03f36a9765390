"""N > 1 path on CPU: two gloo ranks shard the chain axis (no data-path collective), run their shards and
gather the per-chain results to rank 0 -- the path's only collective.  Without a GPU the shard compute is
the oracle (tests may use it as a stand-in checker); what is under test is the partition + gather logic of
epidemicmodeling_amd.batch that bench.py uses over RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, H.ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from epidemicmodeling_amd import batch, synth
    w = synth.make_cfg4(4, 5, 30, 10)                  # 20 chains
    lo, hi = batch.shard_chains(w.B, rank, world)
    shard = w.select(np.arange(lo, hi))
    out = H.oracle_batch(shard, n_threads=1)
    per = (w.B + world - 1) // world
    res = torch.zeros((6, per), dtype=torch.float64)   # equal-shape shards (last one padded)
    res[:, : hi - lo] = torch.from_numpy(out["S_SMOOTH"][29])
    parts = batch.gather_to_root(res)
    if rank == 0:
        full = torch.cat(parts, dim=1)[:, : w.B].numpy()
        ref = H.oracle_batch(w, n_threads=1)["S_SMOOTH"][29]
        q.put(bool(np.array_equal(full, ref)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok, "gathered shards differ from the unsharded run"
