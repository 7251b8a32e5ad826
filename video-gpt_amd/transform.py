"""Operator seam of the reference: `replace_attention(model)` (LVM/transform/sdpa_transform.py:162-169).

The reference walks the model, and for every attention module sets `module.local_attn`
(= F.scaled_dot_product_attention on (B,heads,S,d) with the additive mask) and
`module.dist_attn` (DeepSpeed-Ulysses wrapper; identity at sequence-parallel size 1).
Here `local_attn` becomes the block-masked HIP attention (`ops.sdpa`, same call signature:
`(q, k, v, attn_mask=, dropout_p=, is_causal=)`); `dist_attn` stays None because this build runs
data-parallel only (SP>1 / Ulysses all-to-all is a "next" row).

`hip_sdpa` can also be installed on a stock `transformers` Phi3 attention module that follows the
reference's `new_forward` protocol, which is how a maintainer of the reference would bind this
library.
"""
from __future__ import annotations

from . import ops
from .model import Phi3Attention

hip_sdpa = ops.sdpa


def replace_attention(model, variant: int = 0):
    for module in model.modules():
        if isinstance(module, Phi3Attention):
            module.local_attn = ops.sdpa
            module.dist_attn = None
    return model


# the NPU twin's name in the reference (LVM/transform/fa_transform.py) — same seam, no all-to-all
replace_simple_attention = replace_attention
