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
    # DDP-shaped path: gradients written into the persistent buckets in backward order, buckets leave as soon as they are full
    named = {f"p{i}": torch.nn.Parameter(torch.randn(n)) for i, n in enumerate((5, 1000, 33, 70000))}
    ex2 = GradientExchange(named, bucket_bytes=2048)
    ex2.begin_step()
    for name in reversed(list(named)):          # backward produces the LAST registered tensor first
        buf = ex2.grad_buffer(name)
        buf.copy_(torch.full_like(buf, float(rank + 1)) + torch.arange(buf.numel(), dtype=torch.float32) * 1e-3 * (rank + 1))
        ex2.grad_ready(name)
    log = list(ex2.launch_log)
    out = ex2.finish()
    ok_overlap = True
    for name, pp in named.items():
        expect = torch.full_like(pp, 1.5) + torch.arange(pp.numel(), dtype=torch.float32) * 1e-3 * 1.5
        ok_overlap &= bool(torch.allclose(out[name], expect, rtol=1e-6, atol=1e-6))
        ok_overlap &= out[name].data_ptr() == ex2.grad_buffer(name).data_ptr()       # no copy-out: the views ARE the result
    first_rs = next(i for i, e in enumerate(log) if e[0] == "reduce_scatter")
    last_ready = max(i for i, e in enumerate(log) if e[0] == "ready")
    ok_overlap &= first_rs < last_ready                                              # a bucket left before the "backward" had finished
    ok_overlap &= [e for e in log if e[0] == "reduce_scatter"] == [("reduce_scatter", b) for b in range(len(ex2.buckets))]  # in bucket order
    # a second step reuses the same buffers (no reallocation) and a tensor that reports nothing averages to zero
    ptrs = [f.data_ptr() for f in ex2.flat]
    ex2.begin_step()
    ex2.grad_buffer("p3").fill_(float(rank))
    ex2.grad_ready("p3")
    out = ex2.finish()
    ok_overlap &= ptrs == [f.data_ptr() for f in ex2.flat]
    ok_overlap &= bool(torch.allclose(out["p3"], torch.full_like(out["p3"], 0.5))) and float(out["p1"].abs().max()) == 0.0
    # bench.py's timing reduction
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ret[rank] = (ok_shard, ok_slice, ok_grad and ok_overlap, nbytes, float(t), len(ex.buckets))
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


def _worker4(rank, world, port, ret):
    """World 4: uneven image shards (n = 7, 65), the exchange at its DEFAULT 25 MB bucket size over adapter-shaped tensors (three
    [11, 640, 768] fp32 expert stacks = 21.6 MB each + the small projection / embedding / router tensors), gradient accumulation over two
    micro-batches with no-sync on the first (the bucket protocol AnySDTrainer.backward drives), and bench.py's per-rank time gather."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from anyedit_amd.parallel import shard_range, shard_batch, GradientExchange
    failed = []

    def chk(i, cond):
        if not cond:
            failed.append(i)

    for n in (7, 65):
        s, e = shard_range(n, rank, world)
        cover = torch.zeros(n)
        cover[s:e] = 1
        dist.all_reduce(cover)
        chk(1, bool((cover == 1).all()))                                  # every image exactly once
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([e - s]))
        sz = [int(v) for v in sizes]
        chk(2, sum(sz) == n and max(sz) - min(sz) <= 1 and sz == sorted(sz, reverse=True))   # earlier ranks take the extra
        batch = {"x": torch.arange(n * 2).reshape(n, 2), "code": torch.arange(n) % 3}
        mine = shard_batch(batch, rank, world)
        chk(3, torch.equal(mine["x"], batch["x"][s:e]) and torch.equal(mine["code"], batch["code"][s:e]))
    torch.manual_seed(0)
    shapes = {"image_proj_model.proj.weight": (3072, 64), "image_proj_model.proj.bias": (3072,), "adapter_modules.0": (11, 640, 768),
              "adapter_modules.1": (11, 640, 768), "adapter_modules.2": (11, 640, 768), "task_embs": (25, 768), "gate.weight": (11, 768),
              "gate.bias": (11,)}
    named = {k: torch.nn.Parameter(torch.zeros(v)) for k, v in shapes.items()}
    ex = GradientExchange(named)                                        # default bucket_bytes = 25 MB
    chk(4, len(ex.buckets) >= 2 and all(f.numel() % world == 0 for f in ex.flat))

    def micro_grad(name, micro):                                        # deterministic per (rank, micro-batch, tensor) values
        g = torch.Generator().manual_seed(1000 * rank + 10 * micro + len(name))
        return torch.randn(shapes[name], generator=g)

    for step in range(2):                                               # two optimizer steps: buckets and handles are reused
        ex.begin_step()
        order = list(reversed(list(named)))                             # backward order
        for name in order:                                              # micro-batch 0 under no-sync: write, send nothing
            ex.grad_buffer(name).copy_(micro_grad(name, 2 * step))
        chk(5, not [e for e in ex.launch_log if e[0] == "reduce_scatter"])
        for name in order:                                              # last micro-batch: add, release buckets as they complete
            ex.grad_buffer(name).add_(micro_grad(name, 2 * step + 1))
            ex.grad_ready(name)
        log = list(ex.launch_log)
        out = ex.finish()
        first_rs = next(i for i, e in enumerate(log) if e[0] == "reduce_scatter")
        chk(6, first_rs < max(i for i, e in enumerate(log) if e[0] == "ready"))          # a bucket left before the "backward" ended
        chk(7, [e[1] for e in log if e[0] == "reduce_scatter"] == list(range(len(ex.buckets))))
        for name in named:
            expect = torch.zeros(shapes[name])
            for r in range(world):
                for micro in (2 * step, 2 * step + 1):
                    g = torch.Generator().manual_seed(1000 * r + 10 * micro + len(name))
                    expect += torch.randn(shapes[name], generator=g)
            chk(8, bool(torch.allclose(out[name], expect / world, rtol=1e-5, atol=1e-5)))
    # bench.py: every rank's wall time gathered, job time = max
    tt = torch.tensor([1.0 + 0.25 * rank], dtype=torch.float64)
    allt = torch.zeros(world, dtype=torch.float64)
    dist.all_gather_into_tensor(allt, tt)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    chk(9, allt.tolist() == [1.0 + 0.25 * r for r in range(world)] and float(tt) == 1.75)
    ret[rank] = (failed, ex.bytes_per_step, len(ex.buckets))
    dist.barrier()
    dist.destroy_process_group()


def test_world4_gloo_uneven_shards_default_buckets_and_accumulation():
    world = 4
    port = 31000 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker4, args=(world, port, ret), nprocs=world, join=True)
    n_params = 3072 * 64 + 3072 + 3 * 11 * 640 * 768 + 25 * 768 + 11 * 768 + 11
    for r in range(world):
        failed, nbytes, nb = ret[r]
        assert not failed, f"rank {r}: checks {failed} failed"
        assert nbytes == 4 * n_params and nb >= 2


def test_shard_range_partitions():
    from anyedit_amd.parallel import shard_range, conditioning_dropout_masks
    for n in (1, 7, 64, 65):
        for w in (1, 2, 4, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in rs) - min(e - s for s, e in rs) <= 1
    pm, im = conditioning_dropout_masks(torch.tensor([0.01, 0.05, 0.0999, 0.1, 0.1499, 0.15, 0.9]), 0.05)
    assert pm.tolist() == [True, True, True, False, False, False, False] and im.tolist() == [1, 0, 0, 0, 0, 1, 1]
