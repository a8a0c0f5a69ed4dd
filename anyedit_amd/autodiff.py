"""Reverse-mode tape over the HIP operators, for the training step.

The reference's step (train.py:625-710) is `loss = mse(ip_adapter(noisy, t, ehs, ref, code), eps); accelerator.backward(loss)`
with the UNet FROZEN (train.py:437-442) and only `image_proj_model`, `adapter_modules`, `task_embs` trainable (train.py:483-485).
torch.autograd cannot see through C-ABI calls, so this module records one closure per operator call (`ops.gemm`, `ops.conv3x3`,
`ops.groupnorm`, `ops.layernorm`, `ops.attention`, `ops.geglu`, layout ops) while the ordinary forward code of the modules runs,
and replays them in reverse.  Every closure is a HIP kernel call again:

  Linear / 1x1 conv  dX = dY W            -> ae_gemm_bf16 on the transposed packed weight (cached per weight)
  3x3 conv           dX = conv(dY, rot W) -> ae_conv3x3_bf16 on the rotated packed weight; stride 2 -> zero-insert gather;
                                             nearest-x2 upsample -> + ae_sumpool2x2_bf16
  GroupNorm(+SiLU), LayerNorm, GEGLU, attention -> their ae_*_bwd kernels
  trainable Linear   dW = dY^T A, db      -> ae_gemm_bf16 on transposed operands (fp32 out), ae_colsum_bf16_f32

Activation gradients are bf16 (the reference trains under bf16/fp16 autocast, train.py:294-303), parameter gradients fp32.
Only tensors that (transitively) depend on a trainable leaf get a gradient: layers upstream of the first adapter are skipped.
"""
import contextlib
import os

import torch

from anyedit_amd import ops

BF16 = torch.bfloat16


def _base(t):
    return t._base if t._base is not None else t


# Frozen weights never change between steps: their transposed / rotated packed copies are built once per process, keyed by the
# packed tensor (kept alive here, so ids stay unique).  Trainable weights are re-packed from the fp32 masters every step and are
# NOT cached.
_FROZEN_WT = {}
_FROZEN_WROT = {}
# AE_TAPE_FUSE_ADD=0: every fan-out gradient goes through ae_add_bf16 again (A/B knob; round 5: 134 add launches per training step -> 30)
FUSE_ACCUMULATE = os.environ.get("AE_TAPE_FUSE_ADD", "1") != "0"


class Tape:
    def __init__(self):
        self.active = False
        self.nodes = []          # backward closures in forward order
        self.req = {}            # id(base tensor) -> base tensor: tensors that need a gradient
        self.grads = {}          # id(base tensor) -> gradient (shape of the base)
        self.owned = set()       # data_ptr of the gradient buffers THIS tape allocated: only those are ever written in place.  A first gradient is
                                 # stored by reference (it may be another node's dy, or an outer tape's gradient); the second contribution sums out of place
                                 # into a buffer of the tape's own, and from then on backward kernels may add to it directly (`into`)
        self.param_grads = {}    # name -> fp32 tensor
        self.trainable = {}      # id(tensor) -> name (packed weights / vectors registered as trainable leaves)
        self._wt = {}            # id(packed weight) -> transposed packed weight
        self._wrot = {}          # id(packed conv weight) -> rotated packed weight
        self.keep = []           # keeps every recorded tensor alive (ids stay unique)
        self.mutated = set()     # ids of attention outputs a later call accumulated into in place (they no longer hold ONE segment's output)

    # ------------------------------------------------------------------ bookkeeping
    @contextlib.contextmanager
    def recording(self):
        prev = ops._TAPE
        ops._TAPE, self.active = self, True
        try:
            yield self
        finally:
            self.active = False
            ops._TAPE = prev

    @contextlib.contextmanager
    def paused(self):
        was = self.active
        self.active = False
        try:
            yield
        finally:
            self.active = was

    def require(self, t):
        b = _base(t)
        self.req[id(b)] = b
        return t

    def needs(self, t):
        return t is not None and id(_base(t)) in self.req

    def mark_trainable(self, t, name):
        self.trainable[id(t)] = name
        self.keep.append(t)

    def grad(self, t):
        g = self.grads.get(id(_base(t)))
        return None if g is None else g.reshape(t.shape) if t.numel() == g.numel() else None

    def accumulate(self, t, g):
        """Add gradient g (same number of elements as t, bf16) to t's base tensor."""
        b = _base(t)
        if t.numel() != b.numel():
            raise RuntimeError("autodiff: gradient of a partial view must be produced for the whole base tensor")
        g = g.reshape(b.shape)
        cur = self.grads.get(id(b))
        if cur is None:
            self.grads[id(b)] = g if g.is_contiguous() else g.contiguous()
        elif cur.data_ptr() in self.owned:
            ops.add(cur, g, out=cur)
        else:   # the stored gradient is borrowed: never written — the sum goes to a buffer of this tape's own (same launch count, one allocation)
            new = ops.add(cur, g)
            self.owned.add(new.data_ptr())
            self.grads[id(b)] = new

    def into(self, t, *not_aliasing):
        """The gradient buffer `t` already has (viewed in t's shape), for backward kernels that can ADD their result to it in place instead of
        producing a tensor for `accumulate` (one ae_add_bf16 launch and one pass over the gradient less per fan-out); None when t has no gradient
        yet, is a partial view, or the buffer is one of `not_aliasing` (the kernel's own inputs)."""
        if t is None or not FUSE_ACCUMULATE:
            return None
        b = _base(t)
        cur = self.grads.get(id(b))
        if cur is None or t.numel() != b.numel() or cur.dtype != BF16 or not cur.is_contiguous() or not t.is_contiguous():
            return None
        if cur.data_ptr() not in self.owned:   # a borrowed first gradient (by-reference store): the caller produces a tensor and `accumulate` sums out of place
            return None
        if any(o is not None and o.data_ptr() == cur.data_ptr() for o in not_aliasing):
            return None
        return cur.view(t.shape)

    def add_param_grad(self, name, g):
        cur = self.param_grads.get(name)
        self.param_grads[name] = g if cur is None else cur + g  # fp32 accumulation of a handful of small tensors

    def _record(self, outs, ins, fn):
        if any(self.needs(i) for i in ins):
            for o in outs:
                self.require(o)
            self.keep.extend([*outs, *[i for i in ins if i is not None]])
            self.nodes.append(fn)

    def backward(self):
        with torch.no_grad(), self.paused():
            for fn in reversed(self.nodes):
                fn()
        # the closures hold this tape (and through it every saved activation): dropping them here breaks the reference cycle, so a
        # step's activations go back to the allocator when the caller drops the tape instead of waiting for a gen-2 collection
        # (measured: +6 GB of fresh segments per step until the collector ran, with hipMalloc stalls of 100-200 ms)
        self.nodes = []
        self.keep = []
        return self.param_grads

    # ------------------------------------------------------------------ activation checkpointing (util.py:102-143 CheckpointFunction)
    def _child(self):
        sub = Tape()
        sub.trainable = self.trainable          # same trainable leaves
        sub.req = dict(self.req)                # every tensor known to need a gradient so far (they are all alive: req holds them)
        sub._wt, sub._wrot = self._wt, self._wrot
        return sub

    def checkpoint(self, fn):
        """`fn()` -> tensor or tuple of tensors, closing over its inputs (ResBlock / BasicTransformerBlock bodies, the reference's
        `checkpoint(self._forward, ...)` sites openaimodel.py:250, attention.py:268).  Forward: run it on a throw-away child tape —
        the SAME recorded kernels as an un-checkpointed step, so values are bit-identical — and drop every intermediate.  Backward:
        run it again on a fresh child tape, seed the outputs' gradients, back-propagate inside, and hand the gradients of
        everything that lives outside the segment (inputs, conditioning, parameters) to this tape."""
        tmp = self._child()
        with tmp.recording():
            outs = fn()
        depends = bool(tmp.nodes)               # something inside touched a tensor that needs a gradient
        tmp.nodes, tmp.keep, tmp.req, tmp.grads = [], [], {}, {}   # drop the intermediates now (the closures form a cycle with tmp)
        del tmp
        outs_t = tuple(outs) if isinstance(outs, (tuple, list)) else (outs,)
        if not depends:
            return outs
        for o in outs_t:
            self.require(o)
        self.keep.extend(outs_t)

        def bwd():
            gs = [self.grad(o) for o in outs_t]
            if all(g is None for g in gs):
                return
            sub = self._child()
            with torch.enable_grad(), sub.recording():
                outs2 = fn()
            outs2_t = tuple(outs2) if isinstance(outs2, (tuple, list)) else (outs2,)
            for o2, g in zip(outs2_t, gs):
                if g is not None:
                    sub.accumulate(o2, g)
            sub.backward()
            for name, g in sub.param_grads.items():
                self.add_param_grad(name, g)
            for tid, g in sub.grads.items():
                t = self.req.get(tid)
                if t is not None and not any(t is o for o in outs2_t):   # a tensor from outside the segment
                    if g.dtype == torch.float32:
                        self.accumulate_f32(t, g)   # fp32 leaves (gate values)
                    else:
                        self.accumulate(t, g)

        self.nodes.append(bwd)
        return outs

    # ------------------------------------------------------------------ weights for the data-gradient passes
    def transposed(self, w):
        if id(w) in self.trainable:
            wt = self._wt.get(id(w))
            if wt is None:
                wt = self._wt[id(w)] = w.t().contiguous()
            return wt
        ent = _FROZEN_WT.get(id(w))
        if ent is None or ent[0] is not w:
            ent = _FROZEN_WT[id(w)] = (w, w.t().contiguous())
        return ent[1]

    def rotated(self, w, cin):
        """packed conv weight [Cout, 9*CinPad] (ky,kx,cin) -> packed weight of the adjoint conv [Cin, 9*CoutPad]:
        w'[ci][ky'][kx'][co] = w[co][2-ky'][2-kx'][ci]."""
        ent = _FROZEN_WROT.get(id(w))
        wr = ent[1] if ent is not None and ent[0] is w else None
        if wr is None:
            cout = w.shape[0]
            cin_pad = w.shape[1] // 9
            cout_pad = (cout + 63) // 64 * 64
            w4 = w.view(cout, 3, 3, cin_pad).flip(1, 2).permute(3, 1, 2, 0)[:cin]  # [Cin, 3, 3, Cout]
            wr = torch.zeros(cin, 3, 3, cout_pad, dtype=BF16, device=w.device)
            wr[..., :cout] = w4
            wr = wr.reshape(cin, 9 * cout_pad).contiguous()
            _FROZEN_WROT[id(w)] = (w, wr)
        return wr

    # ------------------------------------------------------------------ operators
    def gemm(self, a, w, bias=None, residual=None, addvec=None, rows_per_batch=0, epilogue=ops.EPI_NONE, out_f32=False, a2=None,
             out=None):
        with self.paused():
            y = ops.gemm(a, w, bias=bias, residual=residual, addvec=addvec, rows_per_batch=rows_per_batch, epilogue=epilogue,
                         out_f32=out_f32, a2=a2, out=out)
        wname = self.trainable.get(id(w))
        bname = self.trainable.get(id(bias)) if bias is not None else None
        ins = [a, a2, residual]
        if wname is not None or bname is not None:
            self.require(y)
            self.keep.extend([y, a])
        if not (self.needs(a) or self.needs(a2) or self.needs(residual) or wname or bname):
            return y
        if epilogue != ops.EPI_NONE or out_f32:
            raise RuntimeError("autodiff: fused activations / fp32 outputs are not differentiable here (use the un-fused ops)")
        K1 = a.shape[1]

        def bwd():
            dy = self.grad(y)
            if dy is None:
                return
            if self.needs(residual):
                self.accumulate(residual, dy)
            if self.needs(a) or self.needs(a2):
                wt = self.transposed(w)  # [K, N]
                if self.needs(a):
                    cur = self.into(a, dy)
                    if cur is not None and cur.dim() == 2:
                        ops.gemm(dy, wt[:K1], residual=cur, out=cur)   # dX added to the gradient a already has, in the GEMM's epilogue
                    else:
                        self.accumulate(a, ops.gemm(dy, wt[:K1]))
                if a2 is not None and self.needs(a2):
                    self.accumulate(a2, ops.gemm(dy, wt[K1:]))
            if wname is not None:  # dW[N, K] = dY^T A  (rows padded to a multiple of 8 for the K-contiguous operand layout)
                M = dy.shape[0]
                Mp = (M + 7) // 8 * 8
                dyt = torch.zeros(dy.shape[1], Mp, dtype=BF16, device=dy.device)
                dyt[:, :M] = dy.t()
                at = torch.zeros(a.shape[1], Mp, dtype=BF16, device=dy.device)
                at[:, :M] = a.t()
                self.add_param_grad(wname, ops.gemm(dyt, at, out_f32=True))
            if bname is not None:
                self.add_param_grad(bname, ops.colsum(dy))

        self.require(y)
        self.keep.extend([y, *[i for i in ins if i is not None]])
        self.nodes.append(bwd)
        return y

    def conv3x3(self, x, w, bias, B, H, W, addvec=None, residual=None, stride=1, upsample2x=False, out_f32=False, out=None):
        with self.paused():
            y, Ho, Wo = ops.conv3x3(x, w, bias, B, H, W, addvec=addvec, residual=residual, stride=stride, upsample2x=upsample2x,
                                    out_f32=out_f32, out=out)
        cin = x.shape[1]

        def bwd():
            dy = self.grad(y)
            if dy is None:
                return
            if dy.dtype != BF16:
                dy = dy.to(BF16)
            if self.needs(residual):
                self.accumulate(residual, dy)
            if not self.needs(x):
                return
            cout = dy.shape[1]
            if cout % 8:  # e.g. the 4-channel output conv: pad the gradient's channels for the 16-byte gather
                pad = torch.zeros(dy.shape[0], (cout + 7) // 8 * 8, dtype=BF16, device=dy.device)
                pad[:, :cout] = dy
                dy = pad
            wr = self.rotated(w, cin)
            if wr.shape[1] != 9 * ((dy.shape[1] + 63) // 64 * 64):
                raise RuntimeError("autodiff: rotated conv weight does not match the gradient's channel count")
            if stride == 2:      # adjoint of the stride-2 conv: zero-insert upsampling folded into the gather
                dx, _, _ = ops.conv3x3(dy.contiguous(), wr, None, B, Ho, Wo, upsample2x=2)
            elif upsample2x:     # forward was conv(nearest_x2(x)): adjoint conv at 2H x 2W, then 2x2 block sums
                dxv, _, _ = ops.conv3x3(dy.contiguous(), wr, None, B, Ho, Wo)
                dx = ops.sumpool2x2(dxv, B, H, W)
            else:
                dx, _, _ = ops.conv3x3(dy.contiguous(), wr, None, B, Ho, Wo)
            self.accumulate(x, dx)

        self._record([y], [x, residual], bwd)
        return y, Ho, Wo

    def groupnorm(self, x, gamma, beta, B, HW, eps, silu=False, groups=32, x2=None, out=None):
        stat = torch.empty(B, groups, 2, dtype=torch.float32, device=x.device)   # (mean, rstd) of the forward, reused by the backward
        with self.paused():
            y = ops.groupnorm(x, gamma, beta, B, HW, eps, silu=silu, groups=groups, x2=x2, out=out, stat_out=stat)

        def bwd():
            dy = self.grad(y)
            if dy is None:
                return
            tx = self.into(x, dy) if self.needs(x) else None
            tx2 = self.into(x2, dy) if (x2 is not None and self.needs(x2)) else None
            if tx is not None and tx2 is not None and tx.data_ptr() == tx2.data_ptr():
                tx2 = None
            dx, dx2 = ops.groupnorm_bwd(x, gamma, beta, dy, B, HW, eps, silu=silu, groups=groups, x2=x2, stat=stat, dx_into=tx, dx2_into=tx2)
            if self.needs(x) and tx is None:
                self.accumulate(x, dx)
            if x2 is not None and self.needs(x2) and tx2 is None:
                self.accumulate(x2, dx2)

        self._record([y], [x, x2], bwd)
        return y

    def layernorm(self, x, gamma, beta, eps=1e-5, out=None):
        with self.paused():
            y = ops.layernorm(x, gamma, beta, eps, out=out)
        gname, bname = self.trainable.get(id(gamma)), self.trainable.get(id(beta))

        def bwd():
            dy = self.grad(y)
            if dy is None:
                return
            tx = self.into(x, dy) if self.needs(x) else None
            dx, dg, db = ops.layernorm_bwd(x, gamma, dy, eps, want_param_grads=gname is not None or bname is not None, dx_into=tx)
            if self.needs(x) and tx is None:
                self.accumulate(x, dx)
            if gname is not None:
                self.add_param_grad(gname, dg)
            if bname is not None:
                self.add_param_grad(bname, db)

        if gname is not None or bname is not None:
            self.require(y)
            self.keep.extend([y, x])
            self.nodes.append(bwd)
        else:
            self._record([y], [x], bwd)
        return y

    def geglu(self, h):
        with self.paused():
            y = ops.geglu(h)

        def bwd():
            dy = self.grad(y)
            if dy is not None:
                self.accumulate(h, ops.geglu_bwd(h, dy))

        self._record([y], [h], bwd)
        return y

    def concat_channels(self, a, b):
        with self.paused():
            y = ops.concat_channels(a, b)
        ca = a.shape[1]

        def bwd():
            dy = self.grad(y)
            if dy is None:
                return
            na, nb = self.needs(a), self.needs(b)
            ba, bb = _base(a), _base(b)
            if not (na or nb) or ba is bb or a.numel() != ba.numel() or b.numel() != bb.numel() or not dy.is_contiguous():
                if na:
                    self.accumulate(a, dy[:, :ca].contiguous())
                if nb:
                    self.accumulate(b, dy[:, ca:].contiguous())
                return
            # one pass: each half is written to a fresh gradient buffer or added to the one its tensor already has
            ga, gb = (self.grads.get(id(ba)) if na else None), (self.grads.get(id(bb)) if nb else None)
            if (ga is not None and ga.data_ptr() not in self.owned) or (gb is not None and gb.data_ptr() not in self.owned):
                # a borrowed gradient is never written in place: the two-launch route sums out of place
                if na:
                    self.accumulate(a, dy[:, :ca].contiguous())
                if nb:
                    self.accumulate(b, dy[:, ca:].contiguous())
                return
            da, db = ops.split_channels(dy, ca, None if ga is None else ga.reshape(a.shape), None if gb is None else gb.reshape(b.shape), na, nb)
            if na and ga is None:
                self.grads[id(ba)] = da.reshape(ba.shape)
                self.owned.add(da.data_ptr())
            if nb and gb is None:
                self.grads[id(bb)] = db.reshape(bb.shape)
                self.owned.add(db.data_ptr())

        self._record([y], [a, b], bwd)
        return y

    def rows_to_nchw(self, x, B, H, W, out_dtype=torch.float32):
        with self.paused():
            y = ops.rows_to_nchw(x, B, H, W, out_dtype=out_dtype)
        C = x.shape[1]

        def bwd():
            dy = self.grad(y)
            if dy is not None:
                self.accumulate(x, ops.nchw_to_rows(dy.reshape(B, C, H, W).float()))

        self._record([y], [x], bwd)
        return y

    def attention(self, q, k, v, B, H, Nq, Nk, D, scale, q_strides, k_strides, v_strides, out=None, key_mask=None, out_scale=None,
                  accumulate=False, seg2=None, rel_h=None):
        if key_mask is not None or rel_h is not None or (seg2 is not None and (accumulate or out_scale is not None)):
            raise RuntimeError("autodiff: masked / biased attention is not differentiable here")
        if accumulate and out is None:
            raise RuntimeError("autodiff: accumulate=True needs the tensor to accumulate into")
        dev = q.device
        if accumulate:
            self.mutated.add(id(out))
        lse = torch.empty(B, H, Nq, dtype=torch.float32, device=dev)
        lse2 = torch.empty(B, H, Nq, dtype=torch.float32, device=dev) if seg2 is not None else None
        with self.paused():
            # accumulate=True: out += out_scale[b] * Attn(q, k, v) in place — `out` keeps its identity, so the gradient that reaches
            # it later is the gradient of BOTH contributions (the un-fused adapter path for head_dim > 96)
            y = ops.attention(q, k, v, B, H, Nq, Nk, D, scale, q_strides, k_strides, v_strides, out=out, seg2=seg2, lse=lse, lse2=lse2,
                              out_scale=out_scale, accumulate=accumulate)
        ins = [q, k, v]
        if out_scale is not None:
            ins.append(out_scale)
        if seg2 is not None:
            k2, v2, Nk2, k2_strides, v2_strides, gate = seg2
            ins += [k2, v2, gate]

        def grad_views(tensors):
            """One gradient buffer per distinct base; returns per-tensor views with the forward's offsets."""
            bufs, views = {}, []
            for t in tensors:
                b = _base(t)
                if id(b) not in bufs:
                    bufs[id(b)] = (b, torch.zeros_like(b) if not b.is_contiguous() else torch.empty_like(b))
                gb = bufs[id(b)][1]
                views.append(gb.as_strided(t.shape, t.stride(), t.storage_offset() - b.storage_offset()))
            return bufs, views

        def bwd():
            dy = self.grad(y)
            if dy is None:
                return
            dy = dy.contiguous()
            need_kv = self.needs(k) or self.needs(v)
            bufs, (dq, dk, dv) = grad_views([q, k, v])
            # a buffer shared by q, k, v (fused qkv) is fully covered by the three views; otherwise uncovered columns must be 0
            for bid, (b, gb) in bufs.items():
                covered = sum(t.numel() for t in (q, k, v) if id(_base(t)) == bid and (t is q or need_kv))
                if covered < b.numel():
                    gb.zero_()
            delta = ops.attention_bwd(q, k, v, dy, lse, B, H, Nq, Nk, D, scale, q_strides, k_strides, v_strides, dq, dk if need_kv else None,
                                      dv if need_kv else None, q_strides, k_strides, v_strides, out_scale=out_scale,
                                      out=y if (seg2 is None and out_scale is None and not accumulate and y.is_contiguous() and id(y) not in self.mutated) else None)
            if out_scale is not None and self.needs(out_scale):
                self.accumulate_f32(out_scale, ops.rowsum_f32(delta.reshape(B, -1)))
            if seg2 is not None:
                need_kv2 = self.needs(k2) or self.needs(v2)
                bufs2, (dk2, dv2) = grad_views([k2, v2])
                for bid, (b, gb) in bufs2.items():
                    if sum(t.numel() for t in (k2, v2) if id(_base(t)) == bid) < b.numel() or not need_kv2:
                        gb.zero_()
                delta2 = ops.attention_bwd(q, k2, v2, dy, lse2, B, H, Nq, Nk2, D, scale, q_strides, k2_strides, v2_strides, dq,
                                           dk2 if need_kv2 else None, dv2 if need_kv2 else None, q_strides, k2_strides, v2_strides,
                                           out_scale=gate, accumulate_dq=True)
                if need_kv2:
                    for bid, (b, gb) in bufs2.items():
                        self.accumulate(b, gb)
                if self.needs(gate):  # d out / d gate_b = Attn(q, K2, V2): its inner product with dy is sum(delta2)
                    self.accumulate_f32(gate, ops.rowsum_f32(delta2.reshape(B, -1)))
            for bid, (b, gb) in bufs.items():
                if self.needs(b):
                    self.accumulate(b, gb)

        self._record([y], ins, bwd)
        return y

    # fp32 leaves (gate values): tiny vectors, accumulated outside the bf16 path
    def accumulate_f32(self, t, g):
        b = _base(t)
        cur = self.grads.get(id(b))
        self.grads[id(b)] = g.reshape(b.shape) if cur is None else cur + g.reshape(b.shape)
