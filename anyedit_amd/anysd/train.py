"""AnySD training step on MI355X — SURVEY.md §8a row A11 (specification: train.py:625-710; the eps-MSE objective is the in-tree
LatentDiffusion.p_losses / get_loss restatement, ddpm.py:889-932, 367-380).

    noise ~ N(0, I), t ~ U{0..999}                                  train.py:636-641
    noisy = sqrt(acp_t) latents + sqrt(1 - acp_t) noise             train.py:643-645  (ddpm.py:356-359)  -> ae_q_sample_f32
    conditioning dropout from ONE uniform draw per sample           train.py:652-669  -> parallel.conditioning_dropout_masks
    x = cat([noisy, image_cond_latents], 1)                         train.py:672
    eps_hat = ip_adapter(x, t, ehs, ref_embeds, edit_code)          train.py:694-695  -> MoE forward on the autodiff tape
    loss = mean((eps_hat.float() - noise.float())^2)                train.py:696      -> ae_mse_f32 / ae_mse_grad_f32
    backward; all-reduce adapter grads; AdamW on the adapters       train.py:483-485, 536-541, 703-706

Only `image_proj_model`, `adapter_modules`, `task_embs` are trained; the UNet is frozen and differentiated w.r.t. activations.
The AnySD source is absent from the reference (row A9: parity unpinned), so gradients are checked against torch.autograd of OUR
forward specification (oracle/anysd_ref.py), not against reference numbers.
"""
import torch

from anyedit_amd import ops
from anyedit_amd.autodiff import Tape
from anyedit_amd.parallel import GradientExchange, conditioning_dropout_masks

BF16 = torch.bfloat16


class AnySDTrainer:
    def __init__(self, moe, sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod, lr=1e-4, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=1e-2, process_group=None, bucket_bytes=25 << 20, always_exchange=False, force_collectives=False):
        self.moe = moe
        self.sqrt_ac = sqrt_alphas_cumprod.float()
        self.sqrt_1mac = sqrt_one_minus_alphas_cumprod.float()
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.step_count = 0
        # fp32 master parameters, in the order train.py:483-485 chains them
        self.params = {}
        for n, p in moe.image_proj_model.named_parameters():
            self.params["image_proj_model." + n] = p
        for i, p in enumerate(moe.adapter_modules):
            self.params[f"adapter_modules.{i}"] = p
        self.params["task_embs"] = moe.task_embs
        # the router (our spec): trained with the adapter group — a frozen random projection would make top-1
        # routing and the gate scale g_b arbitrary for the whole run
        self.params["gate.weight"] = moe.gate.weight
        self.params["gate.bias"] = moe.gate.bias
        self.state = {}
        self._micro = 0        # backward passes since the last optimizer step (gradient accumulation, train.py --gradient_accumulation_steps)
        self._acc = None       # single-process path: the accumulated gradients
        self._sync = True      # DDP path: False inside a no-sync micro-batch (nothing leaves the rank)
        self._reduced = None   # DDP path: the averaged gradients once the exchange has finished
        import torch.distributed as dist
        self.exchange = None
        # always_exchange: build the DDP buckets even for one rank (tests run the bucket-resident gradient path on a single GPU)
        if always_exchange or (dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1):
            self.exchange = GradientExchange(self.params, bucket_bytes, group=process_group, force_collectives=force_collectives)

    # ------------------------------------------------------------------------------------------------ forward on the tape
    def forward_loss(self, latents, image_cond, encoder_hidden_states, ref_embeds, edit_code, noise, timesteps, null_ehs=None,
                     dropout_u=None, dropout_p=0.05):
        """Returns (loss scalar tensor, tape, leaves) with the forward recorded; call `backward` next."""
        moe, dev = self.moe, latents.device
        B = latents.shape[0]
        sa, s1 = self.sqrt_ac.to(dev)[timesteps], self.sqrt_1mac.to(dev)[timesteps]
        noisy = ops.q_sample(latents.float(), noise.float(), sa, s1)
        ehs = encoder_hidden_states
        if dropout_u is not None:  # train.py:652-669: one draw u per sample decides prompt / image dropout
            prompt_mask, image_mask = conditioning_dropout_masks(dropout_u, dropout_p)
            if null_ehs is not None:
                ehs = torch.where(prompt_mask.view(B, 1, 1), null_ehs.to(ehs.dtype), ehs)
            image_cond = image_cond * image_mask.view(B, 1, 1, 1).to(image_cond.dtype)
        x = torch.cat([noisy, image_cond.float()], 1)  # channel concat of two 4-channel latents (plumbing, train.py:672)

        tape = Tape()
        L, Dc = ehs.shape[1], ehs.shape[2]
        T_ip = moe.image_proj_model.tokens
        with torch.no_grad():
            code = edit_code.long()
            te = moe.task_embs.detach()[code]                                                   # [B, Dc] lookup
            context_rows = torch.cat([ehs.float(), te[:, None, :].float()], 1).reshape(B * (L + 1), Dc).to(BF16).contiguous()
            probs, top1, top1p = moe.route(edit_code)
            gate = top1p.float().contiguous()
            tape.require(context_rows)
            tape.require(gate)
            # trainable leaves: packed bf16 copies of the fp32 masters
            ipm = moe.image_proj_model
            Wp, bp = ops.pack_linear(ipm.proj.weight), ipm.proj.bias.detach().float().contiguous()
            gam, bet = ipm.norm.weight.detach().float().contiguous(), ipm.norm.bias.detach().float().contiguous()
            for t, n in ((Wp, "image_proj_model.proj.weight"), (bp, "image_proj_model.proj.bias"),
                         (gam, "image_proj_model.norm.weight"), (bet, "image_proj_model.norm.bias")):
                tape.mark_trainable(t, n)
            cls = ref_embeds[:, 0].to(BF16).contiguous()
            with tape.recording():
                y = ops.gemm(cls, Wp, bp)
                ip_rows = ops.layernorm(y.reshape(B * T_ip, Dc), gam, bet, ipm.norm.eps)
                kv_cache = {}
                for li, (blk, W) in enumerate(zip(moe._blocks, moe.adapter_modules)):
                    attn = blk.attn2
                    kv_cache[id(attn)] = attn.project_kv(context_rows)
                    # The expert K|V projection of layer li is recorded LAZILY, by the block that consumes it (BasicTransformerBlock._rows
                    # calls the entry): its tape node then sits right in front of that block's cross-attention node, so in the backward
                    # pass layer li's weight gradient is final — and its exchange bucket may leave — as soon as block li's attention
                    # backward has run, with the backward of all earlier blocks still ahead (recorded up front, every
                    # reduce-scatter was issued only after the whole UNet backward).  Under activation checkpointing the entry is
                    # simply called again on the child tape.
                    kv_cache[("adapter", id(attn))] = (lambda W=W, li=li: (self._expert_kv(ip_rows, W, top1, T_ip, f"adapter_modules.{li}"), gate))
                eps_hat = moe.unet.forward_rows(x, timesteps, context_rows, kv_cache=kv_cache)
            loss = ops.mse(eps_hat, noise.float())
        leaves = {"context_rows": context_rows, "gate": gate, "probs": probs, "top1": top1, "code": code, "eps_hat": eps_hat,
                  "noise": noise.float(), "L": L, "B": B, "Dc": Dc}
        return loss, tape, leaves

    def _expert_kv(self, ip_rows, W, top1, T_ip, name):
        """kv_ip[b] = ip_rows[b] @ bf16(W[expert_b])^T for every sample; recorded as ONE node of the tape that is recording right now
        (the step's tape, or a checkpoint segment's child tape).  Forward, data gradient and weight gradient are one grouped launch
        each (csrc/expert_kv.hip): they read the fp32 masters directly, so a layer needs no per-expert bf16 copies, transposes or
        zero fills (round 1 looped over samples: ~40 launches per adapter layer)."""
        tape = ops._TAPE
        Wd = W.detach()
        with tape.paused():
            kv_ip = ops.expert_kv(ip_rows, Wd, top1, T_ip)

        def bwd():
            dkv = tape.grad(kv_ip)
            if dkv is None:
                return
            dkv = dkv.contiguous()
            d_ip = ops.expert_kv_dgrad(dkv, Wd, top1, T_ip)
            # dense gradient (AdamW decays the un-routed experts too).  Under DDP the first micro-batch writes it straight into its
            # exchange bucket slot; later micro-batches add to the slot.
            slot = self.exchange.grad_buffer(name) if self.exchange is not None else None
            first = self._micro == 0
            dW = ops.expert_kv_wgrad(dkv, ip_rows, top1, T_ip, W.shape[0], out=slot if first else None)
            tape.accumulate(ip_rows, d_ip)
            if slot is not None and not first:
                slot.add_(dW.reshape(slot.shape))
                dW = slot
            tape.add_param_grad(name, dW)
            if self.exchange is not None and self._sync:  # this layer's expert weights are final: their bucket may leave while backward goes on
                self.exchange.grad_ready(name)

        tape.require(kv_ip)
        tape.keep.extend([kv_ip, ip_rows])
        tape.nodes.append(bwd)
        return kv_ip

    # ------------------------------------------------------------------------------------------------ backward
    def backward(self, tape, leaves, loss_scale=1.0, sync=True):
        """Returns {parameter name: fp32 gradient}.  Gradients ACCUMULATE across calls until `optimizer_step` (micro-batches:
        the reference's --gradient_accumulation_steps, train.py:536-541 under accelerator.accumulate), identically with and without
        DDP: single process -> summed in this trainer's buffers; DDP -> summed in the exchange buckets.  `sync=False` is DDP's
        no_sync for all but the last micro-batch: nothing is sent; the last call (`sync=True`) releases every bucket as its last
        tensor becomes final, overlapped with the rest of the backward pass.  (the bucket path used to zero the buckets in
        every call and step on the last micro-batch only.)"""
        with torch.no_grad():
            first = self._micro == 0
            self._sync = bool(sync)
            if self.exchange is not None and first:
                self.exchange.begin_step()
                self._reduced = None
            tape.accumulate(leaves["eps_hat"], ops.mse_grad(leaves["eps_hat"], leaves["noise"], loss_scale))
            grads = dict(tape.backward())
            B, L, Dc = leaves["B"], leaves["L"], leaves["Dc"]
            # task embeddings: the (L+1)-th context token of every sample + the router gate
            dctx = tape.grad(leaves["context_rows"])
            dte = torch.zeros(B, Dc, dtype=torch.float32, device=leaves["gate"].device)
            if dctx is not None:
                dte = dctx.reshape(B, L + 1, Dc)[:, L].float().contiguous()
            dtask = torch.zeros(self.moe.task_embs.shape, dtype=torch.float32, device=dte.device)
            ops.scatter_add_rows(dte, leaves["code"], dtask)
            dgate = tape.grads.get(id(leaves["gate"]))
            if dgate is not None:
                ops.scatter_add_rows(ops.task_gate_bwd(leaves["probs"], leaves["top1"], dgate, self.moe.gate.weight.detach()),
                                     leaves["code"], dtask)
                grads["gate.weight"], grads["gate.bias"] = ops.task_gate_wgrad(leaves["probs"], leaves["top1"], dgate,
                                                                               self.moe.task_embs.detach(), leaves["code"])
            grads["task_embs"] = dtask
            for n, p in self.params.items():  # parameters that received nothing this micro-batch (experts not routed to)
                if n not in grads:
                    grads[n] = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
                grads[n] = grads[n].reshape(p.shape).contiguous()
            if self.exchange is not None:
                ex = self.exchange
                for n in self.params:
                    view = ex.views[n]
                    if grads[n].data_ptr() != view.data_ptr():   # produced outside its slot (projection, task embeddings, router, un-routed layers)
                        if first:
                            view.copy_(grads[n])
                        else:
                            view.add_(grads[n])
                    grads[n] = view                                  # the caller sees the accumulated, bucket-resident gradient
                if self._sync:  # the small tensors close the last buckets
                    for n in self.params:
                        if n in ex._pending[ex.bucket_of[n]]:
                            ex.grad_ready(n)
            else:
                if first:
                    self._acc = grads
                else:
                    for n in self.params:
                        self._acc[n].add_(grads[n])
                grads = self._acc
            self._micro += 1
        return grads

    def zero_grad(self):
        """Drop the gradients accumulated since the last optimizer step (the next backward starts a fresh accumulation)."""
        self._micro, self._acc, self._reduced, self._sync = 0, None, None, True

    def reduce_gradients(self):
        """DDP: complete the exchange (reduce-scatters issued during the last backward, all-gather) and return the gradients
        averaged over ranks — views into the buckets.  Clip / unscale THESE, in place, before `optimizer_step`; the dict returned by
        `backward` holds the same storage, so after this call it shows the averaged values too.  Single process: the accumulated
        gradients."""
        if self.exchange is None:
            return self._acc
        if self._reduced is None:
            if not self._sync:
                raise RuntimeError("reduce_gradients: the last backward ran with sync=False (nothing was exchanged); run the final micro-batch with sync=True")
            self._reduced = self.exchange.finish()
        return self._reduced

    # ------------------------------------------------------------------------------------------------ optimiser
    def optimizer_step(self, grads=None, grad_scale=1.0):
        """AdamW on the fp32 masters with the gradients accumulated since the last step (`grads=None`), or with `grads` as modified by
        the caller.  Under DDP `grads` must be bucket-resident (what `backward` / `reduce_gradients` returned, modified in place):
        a dict of other tensors would be un-averaged local values, and is rejected instead of being silently ignored."""
        with torch.no_grad():
            if self.exchange is not None:
                reduced = self.reduce_gradients()
                if grads is not None:
                    for n in self.params:
                        if grads[n].data_ptr() != reduced[n].data_ptr():
                            raise ValueError(f"optimizer_step: under DDP the gradient of {n} must be the bucket-resident tensor returned by "
                                             "backward() / reduce_gradients() (modify it in place after reduce_gradients())")
                grads = reduced
            elif grads is None:
                grads = self._acc
            if grads is None:
                raise RuntimeError("optimizer_step: no backward pass since the last step")
            self.step_count += 1
            for n, p in self.params.items():
                st = self.state.get(n)
                if st is None:
                    st = self.state[n] = (torch.zeros_like(p.data, dtype=torch.float32), torch.zeros_like(p.data, dtype=torch.float32))
                if p.data.dtype != torch.float32 or not p.data.is_contiguous():
                    raise TypeError(f"{n}: trainable parameters are fp32 contiguous masters")
                ops.adamw_step(p.data, grads[n].reshape(p.shape), st[0], st[1], self.step_count, self.lr, self.betas, self.eps, self.wd, grad_scale)
                # the kernel wrote through a raw pointer: bump the parameter's version so that every packed-weight cache keyed on it
                # (ops.weights_token) is rebuilt instead of serving the pre-step values
                torch.autograd.graph.increment_version(p)
            self._micro, self._acc, self._reduced = 0, None, None
        return grads

    def train_step(self, latents, image_cond, encoder_hidden_states, ref_embeds, edit_code, noise, timesteps, **kw):
        """One optimisation step; returns the fp32 loss (device scalar tensor)."""
        loss, tape, leaves = self.forward_loss(latents, image_cond, encoder_hidden_states, ref_embeds, edit_code, noise, timesteps, **kw)
        grads = self.backward(tape, leaves)
        self.optimizer_step(grads)
        return loss
