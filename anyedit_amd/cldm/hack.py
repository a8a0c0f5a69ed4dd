"""Mirror of AnyEdit_Collection/other_modules/cldm/hack.py (:11-28) — the run-time swap points the AnyDoor tool calls before building
its model (`disable_verbosity()`, `enable_sliced_attention()`).

The reference's `enable_sliced_attention` monkey-patches `CrossAttention.forward` with a per-head loop so the [B*h, N, N] logits never
exist at once.  On this path CrossAttention is already the fused flash-style HIP kernel (no logits tensor at all), so the call is
accepted and changes nothing.  `hack_everything` patches the CLIP text tower, which is outside this package's scope: it raises.
"""


def disable_verbosity():
    try:
        from transformers import logging
        logging.set_verbosity_error()
    except ImportError:  # transformers is only needed by the (out-of-scope) text tower
        pass
    print('logging improved.')
    return


def enable_sliced_attention():
    print('Enabled sliced_attention. (no-op: attention is the fused HIP kernel, the logits tensor is never materialised)')
    return


def hack_everything(clip_skip=0):
    raise NotImplementedError("cldm.hack.hack_everything patches FrozenCLIPEmbedder; the CLIP text tower is outside the anyedit_amd "
                              "hot path (the caller passes encoder hidden states)")
