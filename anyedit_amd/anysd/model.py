"""AnySD task-aware routing / learnable task embeddings — row A9 of SURVEY.md §8a.

PARITY UNPINNED: the reference's `AnySD/` package is an empty, un-pinned git submodule (.gitmodules:1-4); only its call
contract is visible in train.py:25-28 (imports), :410-424 (construction: `MoE(unet, image_encoder, expert_num=11, ckpt_path)`),
:483-485 (trainables: `image_proj_model`, `adapter_modules`, `task_embs`) and :694-695 (call:
`ip_adapter(x[B,8,h,w], timesteps[B], ehs[B,77,768], ref_embeds[B,L,Dclip], edit_code[B]) -> eps[B,4,h,w]`).
This module keeps exactly that surface and implements OUR documented spec behind it (DESIGN.md "AnySD task router"):

  task token      te_b   = task_embs[edit_code_b]                               (learnable [n_tasks, Dc])
  text context    ctx'_b = concat(ehs_b, te_b)                                  (77 + 1 tokens)
  router          p_b    = softmax(W_g te_b + b_g) over E experts; e*_b = argmax (lowest index on ties); g_b = p_b[e*_b]
  visual tokens   ip_b   = LayerNorm(reshape(W_p cls(ref_embeds_b) + b_p, [T_ip, Dc]))   (image_proj_model)
  each cross-attn layer l:  out = Attn(q, K_l ctx', V_l ctx') + g_b * Attn(q, Wk_{l,e*_b} ip_b, Wv_{l,e*_b} ip_b)

the last line being IP-Adapter's decoupled cross-attention (other_modules/ip_adapter/attention_processor.py:141-173) with a
task-routed expert — the closest readable analogue in the reference tree.  Everything prompt-/task-dependent is
step-invariant, so it is computed ONCE per edit (`prepare_conditioning`) and reused by all DDIM steps.
"""
import torch
import torch.nn as nn

from anyedit_amd import ops
from anyedit_amd.cldm.model import trusted_torch_load
from anyedit_amd.ldm.modules.attention import BasicTransformerBlock

BF16 = torch.bfloat16

# AnySD.dataset.expert_name_list is not visible; the README's edit types stand in (README.md:49-88)
EDIT_TYPES = ["add", "remove", "replace", "color_alter", "appearance_alter", "material_alter", "action_change",
              "textual_change", "background_change", "tone_transfer", "style_change", "movement", "outpaint", "rotation_change",
              "resize", "implicit_change", "relation_change", "counting", "visual_reference", "visual_bbox", "visual_depth",
              "visual_scribble", "visual_segment", "visual_sketch", "visual_material_transfer"]


class ImageProjModel(nn.Module):
    """CLS token of the CLIP penultimate hidden states -> T_ip context tokens (IP-Adapter ImageProjModel shape)."""

    def __init__(self, clip_dim=1280, cross_dim=768, tokens=4):
        super().__init__()
        self.tokens, self.cross_dim = tokens, cross_dim
        self.proj = nn.Linear(clip_dim, tokens * cross_dim)
        self.norm = nn.LayerNorm(cross_dim)

    def rows(self, ref_embeds):
        """ref_embeds [B, L, clip_dim] -> bf16 rows [B*T_ip, cross_dim].  (Once per edit; clip_dim % 8 == 0.)"""
        B = ref_embeds.shape[0]
        cls = ref_embeds[:, 0].to(BF16).contiguous()
        y = ops.gemm(cls, ops.pack_linear(self.proj.weight), self.proj.bias.detach().float().contiguous())
        y = y.reshape(B * self.tokens, self.cross_dim)
        return ops.layernorm(y, self.norm.weight.detach().float().contiguous(), self.norm.bias.detach().float().contiguous(),
                             self.norm.eps)


class MoE(nn.Module):
    """`MoE(unet, image_encoder, expert_num=11, ckpt_path=None)` (train.py:420-424)."""

    def __init__(self, unet, image_encoder=None, expert_num=11, ckpt_path=None, n_tasks=len(EDIT_TYPES), context_dim=768,
                 clip_dim=1280, ip_tokens=4):
        super().__init__()
        self.unet = unet
        self.image_encoder = image_encoder  # CLIP vision tower: outside the hot path, the caller passes its hidden states
        self.expert_num = expert_num
        self.context_dim = context_dim
        self.image_proj_model = ImageProjModel(clip_dim, context_dim, ip_tokens)
        self.task_embs = nn.Parameter(torch.randn(n_tasks, context_dim) * 0.02)
        self.gate = nn.Linear(context_dim, expert_num)
        self._blocks = [m for m in unet.modules() if isinstance(m, BasicTransformerBlock)]
        # adapter_modules[l]: [E, 2*inner_l, context_dim] = per-expert (to_k_ip | to_v_ip) stacked
        self.adapter_modules = nn.ParameterList(
            [nn.Parameter(torch.randn(expert_num, 2 * b.attn2.heads * b.attn2.dim_head, context_dim) * context_dim ** -0.5)
             for b in self._blocks])
        if ckpt_path is not None:
            self.load_state_dict(trusted_torch_load(ckpt_path, "cpu"), strict=False)

    def save_pretrained(self, path):
        sd = {k: v for k, v in self.state_dict().items() if not k.startswith("unet.")}
        torch.save(sd, path)

    @torch.no_grad()
    def route(self, edit_code):
        """Task-router gate (ae_task_gate): returns (probs [B,E], top1 [B] int32, top1_prob [B])."""
        return ops.task_gate(self.task_embs.detach(), edit_code, self.gate.weight.detach(), self.gate.bias.detach())

    @torch.no_grad()
    def prepare_conditioning(self, encoder_hidden_states, ref_embeds, edit_code, ip_scale=None):
        """Everything step-invariant, once per edit.  Returns (context_rows [B*(L+1), Dc], kv_cache dict)."""
        B, L, Dc = encoder_hidden_states.shape
        dev = encoder_hidden_states.device
        te = self.task_embs.detach()[edit_code.long()]                                   # [B, Dc] lookup (indexing = plumbing)
        ctx = torch.cat([encoder_hidden_states.float(), te[:, None, :].float()], dim=1)   # [B, L+1, Dc]
        context_rows = ctx.reshape(B * (L + 1), Dc).to(BF16).contiguous()
        probs, top1, top1p = self.route(edit_code)
        gate = top1p if ip_scale is None else top1p * ip_scale
        ip_rows = self.image_proj_model.rows(ref_embeds)                                  # [B*T, Dc]
        T_ip = self.image_proj_model.tokens
        kv_cache = {}
        for blk, W in zip(self._blocks, self.adapter_modules):
            attn = blk.attn2
            kv_cache[id(attn)] = attn.project_kv(context_rows)
            # every sample's routed expert in ONE grouped launch reading the fp32 masters (csrc/expert_kv.hip; weights rounded to bf16 in
            # registers = the packed weights' values): no host sync on the routing, no per-expert slices and copies
            Wm = W.detach()
            if Wm.dtype != torch.float32 or not Wm.is_contiguous():
                Wm = Wm.float().contiguous()
            kv_ip = ops.expert_kv(ip_rows, Wm, top1, T_ip)
            kv_cache[("adapter", id(attn))] = (kv_ip, gate.float().contiguous())
            # round 6: the same K | V as the LDS images of the fused cross-attention launch (the 64x64-level layers; None elsewhere) — step-invariant like them
            img = attn.fused_kv_images(kv_cache[id(attn)], kv_cache[("adapter", id(attn))], B)
            if img is not None:
                kv_cache[("xattn_img", id(attn))] = img
        return context_rows, kv_cache

    @torch.no_grad()
    def denoise(self, x, timesteps, context_rows, kv_cache, emb_pack=None):
        """One UNet evaluation with the prepared conditioning (the per-step call of the DDIM loop).  emb_pack: the step's time-embedding rows when the
        sampler computed them ahead for its whole schedule (`UNetModel.time_embedding_rows`)."""
        return self.unet.forward_rows(x, timesteps, context_rows, kv_cache=kv_cache, emb_pack=emb_pack)

    def forward(self, noisy_latents, timesteps, encoder_hidden_states, image_embeds, edit_code):
        """train.py:694-695 call contract."""
        context_rows, kv_cache = self.prepare_conditioning(encoder_hidden_states, image_embeds, edit_code)
        return self.denoise(noisy_latents, timesteps, context_rows, kv_cache)
