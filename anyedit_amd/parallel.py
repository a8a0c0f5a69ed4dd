"""Multi-GPU plumbing for the hot path: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).

Inference: editing pairs are independent for all DDIM steps -> contiguous image sharding, NO data-path collective (the
reference shards JSON work lists the same way: local_pipeline_tool.py:579-583 --start-idx/--end-idx).
Training (train.py:483-485, 536-538, 703): data parallel; the only exchange per optimiser step is the mean of the adapter
gradients.  MI355X's xGMI is a fully connected point-to-point mesh (7 links/GPU), so the exchange is done as
reduce-scatter + all-gather over flat buckets — every link carries 1/8 of a bucket concurrently — instead of a ring.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous [start, end) of `n_items` owned by `rank` (sizes differ by at most 1; earlier ranks take the extra)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_batch(tensors, rank, world):
    """Slice every tensor of a dict/list along dim 0 to this rank's images."""
    n = (next(iter(tensors.values())) if isinstance(tensors, dict) else tensors[0]).shape[0]
    s, e = shard_range(n, rank, world)
    if isinstance(tensors, dict):
        return {k: v[s:e] for k, v in tensors.items()}
    return [v[s:e] for v in tensors]


def conditioning_dropout_masks(random_p, p):
    """train.py:652-669: from ONE U(0,1) draw per sample.  prompt_mask: text -> null; image_mask multiplies the image latents."""
    prompt_mask = random_p < 2 * p
    image_mask = 1 - ((random_p >= p).to(torch.float32) * (random_p < 3 * p).to(torch.float32))
    return prompt_mask, image_mask


class GradientExchange:
    """Bucketed mean-all-reduce of a fixed parameter list as reduce-scatter + all-gather on flat fp32 buckets."""

    def __init__(self, params, bucket_bytes=64 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets, cur, cur_n = [], [], 0
        for p in self.params:
            cur.append(p)
            cur_n += p.numel()
            if cur_n * 4 >= bucket_bytes:
                self.buckets.append(cur)
                cur, cur_n = [], 0
        if cur:
            self.buckets.append(cur)
        self.bytes_per_step = sum(p.numel() for p in self.params) * 4

    @torch.no_grad()
    def reduce(self):
        """In place: p.grad <- mean over ranks of p.grad.  Returns the handles' total payload in bytes."""
        if self.world == 1:
            return 0
        for bucket in self.buckets:
            n = sum(p.numel() for p in bucket)
            pad = (-n) % self.world
            dev = bucket[0].grad.device
            flat = torch.zeros(n + pad, dtype=torch.float32, device=dev)
            off = 0
            for p in bucket:
                flat[off:off + p.numel()] = p.grad.reshape(-1).float()
                off += p.numel()
            shard = torch.empty((n + pad) // self.world, dtype=torch.float32, device=dev)
            dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=self.group)
            shard /= self.world
            dist.all_gather_into_tensor(flat, shard, group=self.group)
            off = 0
            for p in bucket:
                p.grad.copy_(flat[off:off + p.numel()].reshape(p.grad.shape))
                off += p.numel()
        return self.bytes_per_step
