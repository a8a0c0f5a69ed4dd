"""N > 1 path on CPU: world_size-2 gloo processes.  Sharding covers every image exactly once; the gradient exchange
(reduce-scatter + all-gather buckets) equals the mean over ranks; bench-style max-over-ranks timing reduction works."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from anyedit_amd.parallel import shard_range, shard_batch, GradientExchange
    # inference sharding: 7 images over 2 ranks
    s, e = shard_range(7, rank, world)
    mine = torch.zeros(7)
    mine[s:e] = 1
    dist.all_reduce(mine)
    ok_shard = bool((mine == 1).all())
    batch = {"x": torch.arange(7 * 3).reshape(7, 3)}
    ok_slice = torch.equal(shard_batch(batch, rank, world)["x"], batch["x"][s:e])
    # training exchange
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (5, 1000, 33, 70000)]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1)) + torch.arange(p.numel(), dtype=torch.float32) * 1e-3 * (rank + 1)
    ex = GradientExchange(params, bucket_bytes=2048)
    nbytes = ex.reduce()
    ok_grad = True
    for p in params:
        expect = torch.full_like(p, 1.5) + torch.arange(p.numel(), dtype=torch.float32) * 1e-3 * 1.5
        ok_grad &= bool(torch.allclose(p.grad, expect, rtol=1e-6, atol=1e-6))
    # bench.py's timing reduction
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ret[rank] = (ok_shard, ok_slice, ok_grad, nbytes, float(t), len(ex.buckets))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_sharding_and_gradient_exchange():
    world = 2
    port = 29000 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        ok_shard, ok_slice, ok_grad, nbytes, tmax, nb = ret[r]
        assert ok_shard and ok_slice and ok_grad
        assert nbytes == (5 + 1000 + 33 + 70000) * 4 and tmax == 2.0 and nb >= 2


def test_shard_range_partitions():
    from anyedit_amd.parallel import shard_range, conditioning_dropout_masks
    for n in (1, 7, 64, 65):
        for w in (1, 2, 4, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in rs) - min(e - s for s, e in rs) <= 1
    pm, im = conditioning_dropout_masks(torch.tensor([0.01, 0.05, 0.0999, 0.1, 0.1499, 0.15, 0.9]), 0.05)
    assert pm.tolist() == [True, True, True, False, False, False, False] and im.tolist() == [1, 0, 0, 0, 0, 1, 1]
