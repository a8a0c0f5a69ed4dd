"""ldm/models/autoencoder.py:13-101 — AutoencoderKL inference surface on the HIP path.

`AutoencoderKL(ddconfig, lossconfig, embed_dim, ...)` keeps the reference constructor and state-dict layout (`encoder.*`,
`decoder.*`, `quant_conv`, `post_quant_conv`), so kl-f8 checkpoints load unchanged; `encode(x) -> DiagonalGaussianDistribution`,
`decode(z)`, `forward(input, sample_posterior)`.  Training of the first stage (loss / EMA / Lightning hooks) is outside the
AnyEdit path: the reference only ever runs it frozen (ddpm.py:563-568).
"""
import torch
import torch.nn as nn

from anyedit_amd import ops
from anyedit_amd.ldm.modules.diffusionmodules.model import Encoder, Decoder, _conv1x1_packed, _pad8
from anyedit_amd.ldm.modules.distributions.distributions import DiagonalGaussianDistribution


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=[], image_key="image", colorize_nlabels=None,
                 monitor=None, ema_decay=None, learn_logvar=False):
        super().__init__()
        self.learn_logvar = learn_logvar
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        assert ddconfig["double_z"]
        self.quant_conv = torch.nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = torch.nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        self._pk = {}
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    @property
    def device(self):
        return next(self.parameters()).device

    def init_from_ckpt(self, path, ignore_keys=list()):
        """autoencoder.py:52-61."""
        from anyedit_amd.cldm.model import trusted_torch_load
        sd = trusted_torch_load(path, "cpu")["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        self.load_state_dict(sd, strict=False)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._pk = {}
        for m in self.modules():
            if m is not self and hasattr(m, "repack"):
                m.repack()
        return r

    def _cache(self):
        own = (self.quant_conv.weight, self.quant_conv.bias, self.post_quant_conv.weight, self.post_quant_conv.bias)
        if self.__dict__.get("_pk_tok") != ops.weights_token(*own) or self._pk.get("dev") != self.device:
            self.__dict__["_pk_tok"] = ops.weights_token(*own)
            self._pk = {"dev": self.device}
        return self._pk

    @torch.no_grad()
    def encode(self, x):
        """autoencoder.py:83-87."""
        B, C, H, W = x.shape
        r = ops.nchw_to_rows(x, _pad8(C))
        h, Ho, Wo, _ = self.encoder.rows(r, B, H, W)                       # fp32 rows [B*Ho*Wo, 2z]
        w, b, cout = _conv1x1_packed(self._cache(), "quant_conv", self.quant_conv)
        m = ops.gemm(h.to(torch.bfloat16), w, b, out_f32=True)
        moments = ops.rows_to_nchw(m, B, Ho, Wo, out_dtype=torch.float32)[:, :cout].contiguous()
        return DiagonalGaussianDistribution(moments)

    @torch.no_grad()
    def decode(self, z):
        """autoencoder.py:88-91."""
        B, C, H, W = z.shape
        r = ops.nchw_to_rows(z, _pad8(C))
        w, b, _ = _conv1x1_packed(self._cache(), "post_quant_conv", self.post_quant_conv)
        h = ops.gemm(r, w, b)
        y, Ho, Wo, cout = self.decoder.rows(h, B, H, W)
        return ops.rows_to_nchw(y, B, Ho, Wo, out_dtype=torch.float32)[:, :cout].contiguous()

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior


class IdentityFirstStage(torch.nn.Module):
    """autoencoder.py:201-218."""

    def __init__(self, *args, vq_interface=False, **kwargs):
        self.vq_interface = vq_interface
        super().__init__()

    def encode(self, x, *args, **kwargs):
        return x

    def decode(self, x, *args, **kwargs):
        return x

    def quantize(self, x, *args, **kwargs):
        if self.vq_interface:
            return x, None, [None, None, None]
        return x

    def forward(self, x, *args, **kwargs):
        return x
