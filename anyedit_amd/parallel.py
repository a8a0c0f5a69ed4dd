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
    """DDP-shaped exchange of a fixed set of trainable tensors (train.py:536-538 wraps the adapter in DistributedDataParallel;
    train.py:703 steps the optimiser on the averaged gradients): mean over ranks as reduce-scatter + all-gather on flat fp32
    buckets, OVERLAPPED with the backward pass.

    * The buckets are persistent flat fp32 buffers.  `grad_buffer(name)` hands out a view into its bucket: the backward pass
      writes (or accumulates) the gradient THERE — no per-step flat allocation, no copy-in / copy-out.
    * Buckets are filled in backward order (reverse registration order).  `grad_ready(name)` marks one tensor final; when the
      last tensor of a bucket lands, that bucket's reduce-scatter is issued at once on a side stream (`async_op`), while the
      backward of the earlier layers keeps running on the compute stream.
    * `finish()` flushes whatever was never marked, waits for the reduce-scatters, scales by 1 / world, all-gathers, and makes
      the compute stream wait for the side stream.  It returns {name: averaged gradient (the same views)}.
    MI355X's xGMI is a fully connected point-to-point mesh (7 links per GPU): reduce-scatter + all-gather keeps every link busy
    with 1/8 of a bucket at a time instead of funnelling a ring through one link pair.  `launch_log` records (event, bucket) in
    issue order — the gloo test asserts from it that buckets left before the backward had finished.
    """

    def __init__(self, params, bucket_bytes=25 << 20, group=None, force_collectives=False):
        if isinstance(params, dict):
            named = list(params.items())
        else:
            named = [(str(i), p) for i, p in enumerate(params)]
        named = [(n, p) for n, p in named if getattr(p, "requires_grad", True)]
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collectives: issue the reduce-scatter / all-gather even for ONE rank (tests drive the RCCL call sequence, streams and
        # handles on a single GPU; with one rank the collectives are identities)
        self.collect = self.world > 1 or (force_collectives and dist.is_initialized())
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.device = dev
        # backward order = reverse registration order
        self.buckets, cur, cur_n = [], [], 0
        for n, p in reversed(named):
            cur.append((n, p))
            cur_n += p.numel()
            if cur_n * 4 >= bucket_bytes:
                self.buckets.append(cur)
                cur, cur_n = [], 0
        if cur:
            self.buckets.append(cur)
        self.flat, self.shard, self.views, self.bucket_of = [], [], {}, {}
        ALIGN = 64  # every tensor's slot starts on a 256-byte boundary: kernels write gradients straight into the slots
        for bi, bucket in enumerate(self.buckets):
            n = sum((p.numel() + ALIGN - 1) // ALIGN * ALIGN for _, p in bucket)
            pad = (-n) % self.world
            flat = torch.zeros(n + pad, dtype=torch.float32, device=dev)
            self.flat.append(flat)
            self.shard.append(torch.empty((n + pad) // self.world, dtype=torch.float32, device=dev))
            off = 0
            for name, p in bucket:
                self.views[name] = flat[off:off + p.numel()].view(p.shape)
                self.bucket_of[name] = bi
                off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.bytes_per_step = sum(p.numel() for p in self.params) * 4
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.launch_log = []
        self._pending, self._handles, self._launched = [], {}, set()
        self.begin_step(zero=False)

    # ------------------------------------------------------------------ per step
    def begin_step(self, zero=True):
        """Start a backward pass: every bucket empty (zeroed: tensors that receive no gradient this step average to zero)."""
        if zero:
            for f in self.flat:
                f.zero_()
        self._pending = [set(n for n, _ in b) for b in self.buckets]
        self._handles, self._launched = {}, set()
        self.launch_log = []

    def grad_buffer(self, name):
        """fp32 view of `name`'s slot in its bucket (zero at the start of the step): write the gradient here."""
        return self.views[name]

    def _launch(self, bi):
        if bi in self._launched:
            return
        self._launched.add(bi)
        self.launch_log.append(("reduce_scatter", bi))
        if not self.collect:
            return
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                self._handles[bi] = dist.reduce_scatter_tensor(self.shard[bi], self.flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._handles[bi] = dist.reduce_scatter_tensor(self.shard[bi], self.flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    @torch.no_grad()
    def grad_ready(self, name, grad=None):
        """`name`'s gradient is final.  `grad` (optional): a tensor to copy into the bucket slot (when it was not produced there)."""
        if grad is not None and grad.data_ptr() != self.views[name].data_ptr():
            self.views[name].copy_(grad.reshape(self.views[name].shape))
        bi = self.bucket_of[name]
        self._pending[bi].discard(name)
        self.launch_log.append(("ready", name))
        if not self._pending[bi]:
            self._launch(bi)

    @torch.no_grad()
    def finish(self):
        """Complete the exchange; returns {name: mean gradient} (views into the buckets, valid until the next begin_step)."""
        for bi in range(len(self.buckets)):
            self._launch(bi)  # whatever never reported stays as written (zeros if untouched)
        self.launch_log.append(("finish", -1))
        if self.collect:
            ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null()
            with ctx:
                gathers = []
                for bi in range(len(self.buckets)):
                    self._handles[bi].wait()
                    self.shard[bi].div_(self.world)
                    gathers.append(dist.all_gather_into_tensor(self.flat[bi], self.shard[bi], group=self.group, async_op=True))
                for h in gathers:
                    h.wait()
            if self.stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return dict(self.views)

    # ------------------------------------------------------------------ one-shot form (gradients already sit in p.grad)
    @torch.no_grad()
    def reduce(self):
        """In place: p.grad <- mean over ranks of p.grad, through the same buckets.  Returns the payload in bytes."""
        if not self.collect:
            return 0
        self.begin_step(zero=False)
        for n, p in zip(reversed(self.names), reversed(self.params)):
            self.grad_ready(n, p.grad.float())
        out = self.finish()
        for n, p in zip(self.names, self.params):
            p.grad.copy_(out[n].reshape(p.grad.shape))
        return self.bytes_per_step


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
