"""InstructPix2Pix / AnySD-style edit loop on the HIP path: 3-branch classifier-free guidance
(text+image, image-only, unconditional — AnyEdit_Collection/adaptive_editing_pipelines/tools/global_tool.py:166-177,
290-304; train.py:61-68 validation settings) driven by the ldm DDIM schedule/update (ldm/models/diffusion/ddim.py:23-52,
223-250), with the masked-latent blend of global_tool.py:183-184.

One UNet evaluation per step on the 3B batch, captured once as a HIP graph (torch.cuda.CUDAGraph = hipGraph) and replayed;
the CFG combine + DDIM update is a single fused kernel per step; text/task/expert K|V are projected once per edit.
"""
import numpy as np
import torch

from anyedit_amd import ops
from anyedit_amd.ldm.models.diffusion.ddim import DDIMSampler, _f32


class EditPipeline:
    def __init__(self, moe, schedule_model, use_graph=True):
        """moe: anysd.MoE (or any object with prepare_conditioning/denoise); schedule_model: the DDPM-duck-typed owner of
        betas/alphas_cumprod (anyedit_amd.ldm.models.diffusion.ddpm.DDPM or a reference LatentDiffusion)."""
        self.moe = moe
        self.sampler = DDIMSampler(schedule_model)
        self.schedule_model = schedule_model
        self.use_graph = use_graph
        self._graph = None
        self._graph_key = None
        self._emb_pack = None
        self.randn = torch.randn

    # ------------------------------------------------------------------------------------------------------------
    def _denoise_static(self):
        if self._emb_pack is not None:
            return self.moe.denoise(self._x_in, self._t, self._ctx_rows, self._kv_cache, emb_pack=self._emb_pack)
        return self.moe.denoise(self._x_in, self._t, self._ctx_rows, self._kv_cache)

    def _hoist_time_embedding(self, steps_desc, nb, dev):
        """Round 5: the time-embedding chain (timestep_embedding -> time_embed -> all ResBlock emb_layers: three M = 3B GEMMs, ~65 us per step at
        batch 12) depends on t alone — computed here ONCE for the whole schedule (M = number of steps), and per step one row is broadcast into the
        static buffer the captured graph reads.  Returns the table (fp32 [steps, sum Cout]) or None when the model does not offer the split
        (AE_HOIST_TEMB=0 turns it off: the A/B knob)."""
        import os
        unet = getattr(self.moe, "unet", None)
        if os.environ.get("AE_HOIST_TEMB", "1") == "0" or unet is None or not hasattr(unet, "time_embedding_rows") or getattr(unet, "num_classes", None) is not None:
            if self._emb_pack is not None:
                self._emb_pack, self._graph = None, None
            return None
        pack = unet.time_embedding_rows(torch.as_tensor(np.ascontiguousarray(steps_desc), dtype=torch.long, device=dev))
        if self._emb_pack is None or tuple(self._emb_pack.all.shape) != (nb, pack.all.shape[1]) or self._emb_pack.offsets != pack.offsets:
            from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import EmbPack
            self._emb_pack = EmbPack(None, torch.empty(nb, pack.all.shape[1], dtype=torch.float32, device=dev), pack.offsets)
            self._graph = None   # the graph reads this buffer: recapture
        return pack.all

    def set_step(self, t, temb_row=None):
        """Points the static buffers the UNet evaluation (and its captured graph) reads at DDIM timestep `t`: the timestep vector AND — when the
        time-embedding chain is hoisted — the embedding rows, which is what the evaluation actually consumes then (callers that only filled
        `_t` evaluated the last step's embedding).  temb_row: the row of `_hoist_time_embedding`'s table for this step; None recomputes it."""
        self._t.fill_(int(t))
        if self._emb_pack is not None:
            if temb_row is None:
                temb_row = self.moe.unet.time_embedding_rows(torch.full((1,), int(t), dtype=torch.long, device=self._t.device)).all[0]
            self._emb_pack.all.copy_(temb_row.unsqueeze(0).expand_as(self._emb_pack.all))

    def _ensure_graph(self, key):
        if self._graph is not None and self._graph_key == key:
            return
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up outside capture (module loading, allocator pools)
                self._denoise_static()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._eps = self._denoise_static()
        self._graph, self._graph_key = g, key

    @torch.no_grad()
    def prepare(self, img_lat, ehs, null_ehs, ref_embeds, edit_code):
        """Step-invariant conditioning for the 3B batch; returns nothing, state is kept for `edit`."""
        B = img_lat.shape[0]
        if null_ehs.shape[0] == 1:
            null_ehs = null_ehs.expand(B, -1, -1)
        ehs3 = torch.cat([ehs, null_ehs, null_ehs], 0)
        ref3 = torch.cat([ref_embeds, ref_embeds, torch.zeros_like(ref_embeds)], 0)
        code3 = torch.cat([edit_code] * 3, 0)
        ctx_rows, kv_cache = self.moe.prepare_conditioning(ehs3, ref3, code3)
        img_cond3 = torch.cat([img_lat, img_lat, torch.zeros_like(img_lat)], 0).float()      # global_tool.py:290-304
        C = img_lat.shape[1]
        shape_in = (3 * B, 2 * C, *img_lat.shape[2:])
        if self._same_layout(ctx_rows, kv_cache, shape_in):
            # same shapes as the captured edit: refresh the graph's static buffers in place (no re-capture: a capture costs
            # three eager UNet evaluations, ~6 % of a 50-step edit)
            self._ctx_rows.copy_(ctx_rows)
            for k, v in kv_cache.items():
                cur = self._kv_cache[k]
                if isinstance(v, tuple):
                    for c, n in zip(cur, v):
                        c.copy_(n)
                else:
                    cur.copy_(v)
        else:
            self._ctx_rows, self._kv_cache = ctx_rows, kv_cache
            self._x_in = torch.empty(shape_in, dtype=torch.float32, device=img_lat.device)
            self._t = torch.zeros(3 * B, dtype=torch.long, device=img_lat.device)
            self._graph = None  # buffers changed identity -> recapture
        self._x_in[:, C:] = img_cond3

    def _same_layout(self, ctx_rows, kv_cache, shape_in):
        if getattr(self, "_kv_cache", None) is None or getattr(self, "_x_in", None) is None:
            return False
        if tuple(self._x_in.shape) != tuple(shape_in) or self._ctx_rows.shape != ctx_rows.shape or self._kv_cache.keys() != kv_cache.keys():
            return False
        for k, v in kv_cache.items():
            cur = self._kv_cache[k]
            a, b = (cur, v) if isinstance(v, tuple) else ((cur,), (v,))
            if any(x.shape != y.shape or x.dtype != y.dtype for x, y in zip(a, b)):
                return False
        return True

    @torch.no_grad()
    def edit(self, x_T, img_lat, ehs, null_ehs, ref_embeds, edit_code, steps=50, s_txt=7.5, s_img=1.5, eta=0.0, mask=None,
             x0=None, prepared=False, step_callback=None):
        """Returns the edited latents [B,4,h,w] fp32.  mask/x0: optional masked-latent blend (global_tool.py:183-184)."""
        B, C = x_T.shape[0], x_T.shape[1]
        dev = x_T.device
        if not prepared:
            self.prepare(img_lat, ehs, null_ehs, ref_embeds, edit_code)
        s = self.sampler
        s.make_schedule(ddim_num_steps=steps, ddim_eta=eta, verbose=False)
        timesteps = s.ddim_timesteps
        total = timesteps.shape[0]
        img = x_T.float().contiguous()
        blend_noise = self.randn(x0.shape, device=dev) if mask is not None else None         # one noise, as global_tool.py:161
        temb = self._hoist_time_embedding(np.flip(timesteps), 3 * B, dev)
        if temb is not None:
            self._emb_pack.all.copy_(temb[0].unsqueeze(0).expand_as(self._emb_pack.all))     # valid contents for the capture's warm-up runs
        if self.use_graph:
            self._ensure_graph((tuple(x_T.shape), steps))
        x_view = self._x_in[:, :C].view(3, B, C, *x_T.shape[2:])
        for i, step in enumerate(np.flip(timesteps)):
            index = total - i - 1                                                            # ddim.py:151 bookkeeping
            x_view.copy_(img.unsqueeze(0))
            self.set_step(step, temb[i] if temb is not None else None)
            if self.use_graph:
                self._graph.replay()
                eps = self._eps
            else:
                eps = self._denoise_static()
            noise = self.randn(img.shape, device=dev) if eta != 0.0 else None
            img, _ = ops.ddim_step(img, eps, s._coeffs(index, False), 3, s0=float(s_txt), s1=float(s_img), noise=noise,
                                   want_pred_x0=False)
            if mask is not None:
                sa = float(_f32(self.schedule_model.sqrt_alphas_cumprod[int(step)]))
                s1 = float(_f32(self.schedule_model.sqrt_one_minus_alphas_cumprod[int(step)]))
                img = ops.mask_blend(img, x0, blend_noise, mask, sa, s1, ip2p_order=True)
            if step_callback is not None:
                step_callback(i, img)
        return img
