"""Operator seam of the reference: `replace_attention(model)` (LVM/transform/sdpa_transform.py:162-169).

The reference walks the model, and for every attention module sets `module.local_attn`
(= F.scaled_dot_product_attention on (B,heads,S,d) with the additive mask) and
`module.dist_attn` (DeepSpeed-Ulysses wrapper; identity at sequence-parallel size 1).
Here `local_attn` becomes the block-masked HIP attention (`ops.sdpa`, same call signature:
`(q, k, v, attn_mask=, dropout_p=, is_causal=)`); `dist_attn` is None at sequence-parallel size 1 and
`sequence_parallel.DistributedAttention` (Ulysses all-to-all over RCCL) once a sequence-parallel group of more than one
rank has been initialised (`initialize_sequence_parallel_state`, LVM/parallel_states.py:40-53).

`hip_sdpa` can also be installed on a stock `transformers` Phi3 attention module that follows the
reference's `new_forward` protocol, which is how a maintainer of the reference would bind this
library.
"""
from __future__ import annotations

from . import ops
from . import sequence_parallel as SP
from .model import Phi3Attention

hip_sdpa = ops.sdpa


def replace_attention(model, variant: int = 0):
    for module in model.modules():
        if isinstance(module, Phi3Attention):
            module.local_attn = ops.sdpa
            group = SP.get_sequence_parallel_group()
            module.dist_attn = SP.DistributedAttention(module.local_attn, group) if SP.sp_world(group) > 1 else None
    return model


# the NPU twin's name in the reference (LVM/transform/fa_transform.py) — same seam, no all-to-all
replace_simple_attention = replace_attention
