"""Mirror of segment_anything/segment_anything/modeling/transformer.py on HIP kernels — SURVEY.md §8(f) N3.

Class names, constructor arguments and state-dict keys follow the reference (TwoWayTransformer :16-108, TwoWayAttentionBlock
:111-182, Attention :185-240).  Tokens and image embeddings move as channels-last bf16 rows; every projection is ae_gemm_bf16 (the
residual adds fused into the out_proj / lin2 epilogues, ReLU into lin1's), the softmax core is the fused attention kernel at head
dims 32 / 16 with ragged 5+N-token sides, and `keys + key_pe` is formed once per block for the two projections that consume it.
"""
import math
from typing import Tuple, Type

import torch
import torch.nn as nn
from torch import Tensor

from anyedit_amd import ops
from anyedit_amd.ldm.modules.diffusionmodules.util import Linear, LayerNorm

BF16 = torch.bfloat16


def _rows(x: Tensor) -> Tensor:
    return x.reshape(-1, x.shape[-1]).to(BF16).contiguous()


class ReLUMLPBlock(nn.Module):
    """common.py:13-27 with the decoder's activation (transformer.py:122: nn.ReLU) fused into lin1's epilogue."""

    def __init__(self, embedding_dim: int, mlp_dim: int, act: Type[nn.Module] = nn.ReLU) -> None:
        super().__init__()
        if act not in (nn.ReLU, nn.GELU):
            raise NotImplementedError("MLPBlock: only nn.ReLU / nn.GELU are implemented")
        self.lin1 = Linear(embedding_dim, mlp_dim)
        self.lin2 = Linear(mlp_dim, embedding_dim)
        self.act = act()
        self._epi = ops.EPI_RELU if act is nn.ReLU else ops.EPI_GELU

    def rows(self, x, residual=None):
        return self.lin2.rows(self.lin1.rows(x, epilogue=self._epi), residual=residual)

    def forward(self, x):
        return self.rows(_rows(x)).reshape(x.shape).to(x.dtype)


class Attention(nn.Module):
    """transformer.py:185-240: attention whose projections may narrow the embedding by `downsample_rate`."""

    def __init__(self, embedding_dim: int, num_heads: int, downsample_rate: int = 1) -> None:
        super().__init__()
        self.embedding_dim = embedding_dim
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        assert self.internal_dim % num_heads == 0, "num_heads must divide embedding_dim."
        self.q_proj = Linear(embedding_dim, self.internal_dim)
        self.k_proj = Linear(embedding_dim, self.internal_dim)
        self.v_proj = Linear(embedding_dim, self.internal_dim)
        self.out_proj = Linear(self.internal_dim, embedding_dim)

    def core(self, qp, kp, vp, B, Nq, Nk, residual=None):
        """qp / kp / vp: projected rows (row stride free, e.g. column slices of a merged projection) -> out_proj(attn) (+ residual)."""
        h, I = self.num_heads, self.internal_dim
        d = I // h
        o = ops.attention(qp, kp, vp, B, h, Nq, Nk, d, 1.0 / math.sqrt(d), (Nq * qp.stride(0), d, qp.stride(0)),
                          (Nk * kp.stride(0), d, kp.stride(0)), (Nk * vp.stride(0), d, vp.stride(0)))
        return self.out_proj.rows(o.reshape(B * Nq, I), residual=residual)

    def rows(self, q, k, v, B, Nq, Nk, residual=None):
        return self.core(self.q_proj.rows(q), self.k_proj.rows(k), self.v_proj.rows(v), B, Nq, Nk, residual=residual)

    def forward(self, q: Tensor, k: Tensor, v: Tensor) -> Tensor:
        B, Nq, Nk = q.shape[0], q.shape[1], k.shape[1]
        return self.rows(_rows(q), _rows(k), _rows(v), B, Nq, Nk).reshape(B, Nq, -1).to(q.dtype)


class TwoWayAttentionBlock(nn.Module):
    """transformer.py:111-182: (1) token self-attention, (2) tokens -> image, (3) MLP on tokens, (4) image -> tokens."""

    def __init__(self, embedding_dim: int, num_heads: int, mlp_dim: int = 2048, activation: Type[nn.Module] = nn.ReLU,
                 attention_downsample_rate: int = 2, skip_first_layer_pe: bool = False) -> None:
        super().__init__()
        self.self_attn = Attention(embedding_dim, num_heads)
        self.norm1 = LayerNorm(embedding_dim)
        self.cross_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm2 = LayerNorm(embedding_dim)
        self.mlp = ReLUMLPBlock(embedding_dim, mlp_dim, activation)
        self.norm3 = LayerNorm(embedding_dim)
        self.norm4 = LayerNorm(embedding_dim)
        self.cross_attn_image_to_token = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.skip_first_layer_pe = skip_first_layer_pe
        self._kq = None

    def repack(self):
        self._kq = None

    def _image_side_projection(self):
        """`keys + key_pe` feeds k_proj of (2) and q_proj of (4): one GEMM over the concatenated weights."""
        t2i, i2t = self.cross_attn_token_to_image, self.cross_attn_image_to_token
        if ops.cache_stale(self, "_kq", t2i.k_proj.weight, t2i.k_proj.bias, i2t.q_proj.weight, i2t.q_proj.bias):
            w = torch.cat([t2i.k_proj.weight.detach(), i2t.q_proj.weight.detach()], dim=0)
            b = torch.cat([t2i.k_proj.bias.detach(), i2t.q_proj.bias.detach()], dim=0)
            self._kq = (ops.pack_linear(w), b.float().contiguous())
        return self._kq

    def rows(self, queries, keys, query_pe, key_pe, B, T, N):
        """queries / query_pe: [B*T, C]; keys: [B*N, C]; key_pe: [N, C] shared by the batch.  Returns (queries, keys)."""
        if self.skip_first_layer_pe:
            queries = self.self_attn.rows(queries, queries, queries, B, T, T)
        else:
            q = ops.add(queries, query_pe)
            queries = self.self_attn.rows(q, q, queries, B, T, T, residual=queries)
        queries = self.norm1.rows(queries)

        t2i, i2t = self.cross_attn_token_to_image, self.cross_attn_image_to_token
        I = t2i.internal_dim
        w, b = self._image_side_projection()
        kq = ops.gemm(ops.add_bcast(keys, key_pe), w, b)                     # [B*N, 2I]: k of (2) | q of (4)
        q = ops.add(queries, query_pe)
        queries = t2i.core(t2i.q_proj.rows(q), kq[:, :I], t2i.v_proj.rows(keys), B, T, N, residual=queries)
        queries = self.norm2.rows(queries)

        queries = self.norm3.rows(self.mlp.rows(queries, residual=queries))

        q = ops.add(queries, query_pe)
        keys = i2t.core(kq[:, I:], i2t.k_proj.rows(q), i2t.v_proj.rows(queries), B, N, T, residual=keys)
        return queries, self.norm4.rows(keys)

    def forward(self, queries: Tensor, keys: Tensor, query_pe: Tensor, key_pe: Tensor) -> Tuple[Tensor, Tensor]:
        B, T, N = queries.shape[0], queries.shape[1], keys.shape[1]
        if key_pe.shape[0] != 1 and not bool((key_pe == key_pe[:1]).all()):
            raise NotImplementedError("TwoWayAttentionBlock: key_pe must be the same for every batch entry (it is the dense grid PE)")
        q, k = self.rows(_rows(queries), _rows(keys), _rows(query_pe), _rows(key_pe[0]).reshape(-1), B, T, N)
        return q.reshape(B, T, -1).to(queries.dtype), k.reshape(B, N, -1).to(keys.dtype)


class TwoWayTransformer(nn.Module):
    """transformer.py:16-108."""

    def __init__(self, depth: int, embedding_dim: int, num_heads: int, mlp_dim: int, activation: Type[nn.Module] = nn.ReLU,
                 attention_downsample_rate: int = 2) -> None:
        super().__init__()
        self.depth = depth
        self.embedding_dim = embedding_dim
        self.num_heads = num_heads
        self.mlp_dim = mlp_dim
        self.layers = nn.ModuleList()
        for i in range(depth):
            self.layers.append(TwoWayAttentionBlock(embedding_dim=embedding_dim, num_heads=num_heads, mlp_dim=mlp_dim,
                                                    activation=activation, attention_downsample_rate=attention_downsample_rate,
                                                    skip_first_layer_pe=(i == 0)))
        self.final_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm_final_attn = LayerNorm(embedding_dim)

    def rows(self, keys, key_pe, tokens, B, T, N):
        """keys [B*N, C], key_pe [N*C] (flat, shared), tokens [B*T, C] -> (queries [B*T, C], keys [B*N, C])."""
        queries = tokens
        for layer in self.layers:
            queries, keys = layer.rows(queries, keys, tokens, key_pe, B, T, N)
        q = ops.add(queries, tokens)
        k = ops.add_bcast(keys, key_pe)
        queries = self.final_attn_token_to_image.rows(q, k, keys, B, T, N, residual=queries)
        return self.norm_final_attn.rows(queries), keys

    def forward(self, image_embedding: Tensor, image_pe: Tensor, point_embedding: Tensor) -> Tuple[Tensor, Tensor]:
        bs, c, h, w = image_embedding.shape
        if image_pe.shape[0] != 1 and not bool((image_pe == image_pe[:1]).all()):
            raise NotImplementedError("TwoWayTransformer: image_pe must be the same for every batch entry (it is the dense grid PE)")
        keys = ops.nchw_to_rows(image_embedding.float().contiguous())
        pe = ops.nchw_to_rows(image_pe[:1].float().contiguous()).reshape(-1)
        T = point_embedding.shape[1]
        q, k = self.rows(keys, pe, _rows(point_embedding), bs, T, h * w)
        return q.reshape(bs, T, c).to(point_embedding.dtype), k.reshape(bs, h * w, c).to(image_embedding.dtype)
