"""world_size-2 data-parallel plumbing on the CPU with gloo: the bucketed all-reduce the trainer uses (same class, CPU
tensors), the 1/N scaling, and the sharding rule of the inference path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vita_amd.parallel import BucketReducer, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        # --- a "model" whose per-rank gradient is the gradient of the rank's shard mean ----------------
        n, B = 1000, 12
        w = torch.randn(n)
        x = torch.randn(B, n)                                     # the global batch, identical on every rank
        lo, hi = shard_range(B, rank, world)
        # loss = mean_rows(0.5 * (x_r . w)^2)  ->  grad = mean_rows((x_r . w) x_r)
        def grad(rows):
            return ((rows @ w)[:, None] * rows).mean(0)
        flat = grad(x[lo:hi]).clone()
        ranges = [(0, 100), (100, 700), (700, 1000)]               # decoder | flow net | encoders
        red = BucketReducer(flat, ranges, max_bucket_elems=256)    # forces the 600-element segment to split
        assert red.world == world and len(red.segments[1]) == 3
        for i in range(3):
            red.reduce(i)
        scale = red.finish()
        mean_grad = flat * scale
        ref = grad(x)                                              # equal shards: mean of shard means == global mean
        ok = bool((mean_grad - ref).abs().max() <= 2e-6 * ref.abs().max())
        # --- inference sharding: no collective, shards tile the batch ---------------------------------
        rows = torch.arange(4096)
        mine = rows[slice(*shard_range(4096, rank, world))]
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)                             # test-only gather to check the tiling
        ok = ok and torch.equal(torch.cat(gathered), rows)
        # --- replicas stay identical after a broadcast + identical updates ------------------------------
        p = torch.full((10,), float(rank))
        dist.broadcast(p, src=0)
        ok = ok and bool((p == 0).all())
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_allreduce_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")   # never fork(): the parent may hold a HIP context when the whole suite runs in one process
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.start_processes(_worker, args=(world, _free_port(), out), nprocs=world, join=True, start_method="spawn")
    res = dict(out)
    mgr.shutdown()
    assert res == {0: True, 1: True}


def test_reducer_single_process_is_a_noop():
    flat = torch.arange(10, dtype=torch.float32)
    red = BucketReducer(flat, [(0, 4), (4, 10)])
    red.reduce(0)
    red.reduce(1)
    assert red.finish() == 1.0 and torch.equal(flat, torch.arange(10, dtype=torch.float32))
    with pytest.raises(ValueError):
        BucketReducer(flat, [(0, 11)])
