"""Ulysses sequence parallelism behind the reference's seam.

Reference: `LVM.frame_block_forward` gives every rank of the sequence-parallel group a contiguous L/P slice of the
token sequence (LVM/model.py:457-463) and all-gathers the last hidden state (:466-473); every attention module's
`dist_attn` (DeepSpeed `DistributedAttention`, rebound to `new_attn_forward`, LVM/transform/sdpa_transform.py:94-159,
168) turns (B, L/P, heads, d) into (B, L, heads/P, d) with one all-to-all per q / k / v, runs `local_attn` on the full
sequence for its heads and returns with a fourth all-to-all.

Here the exchange is `torch.distributed.all_to_all_single` over RCCL (one process per GPU, xGMI links); the data
re-layout around it is tensor copies, the attention itself is the block-masked HIP kernel on the full-length mask.
Transports without an all-to-all for device tensors (gloo, used by the 2-rank test on a single GPU) fall back to an
all-gather through host memory -- same result, test-only bandwidth.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

_GROUP = None


def initialize_sequence_parallel_state(sequence_parallel_size: Optional[int] = None, group=None):
    """LVM/parallel_states.py:40-53: the sequence-parallel group (default: all ranks)."""
    global _GROUP
    if not dist.is_initialized():
        raise RuntimeError("initialize torch.distributed before the sequence-parallel state")
    world = dist.get_world_size()
    if group is not None:
        _GROUP = group
    elif sequence_parallel_size in (None, world):
        _GROUP = dist.group.WORLD
    else:
        if world % sequence_parallel_size:
            raise ValueError("world size must be a multiple of the sequence-parallel size")
        rank = dist.get_rank()
        for g0 in range(0, world, sequence_parallel_size):
            ranks = list(range(g0, g0 + sequence_parallel_size))
            g = dist.new_group(ranks)
            if rank in ranks:
                _GROUP = g
    return _GROUP


def get_sequence_parallel_group():
    return _GROUP


def sp_world(group=None) -> int:
    group = group if group is not None else _GROUP
    return 1 if group is None else dist.get_world_size(group)


def sp_rank(group=None) -> int:
    group = group if group is not None else _GROUP
    return 0 if group is None else dist.get_rank(group)


def _exchange(chunks: torch.Tensor, group) -> torch.Tensor:
    """chunks (P, ...): chunk j goes to rank j; returns (P, ...) with chunk i received from rank i."""
    if dist.get_backend(group) == "nccl":
        out = torch.empty_like(chunks)
        dist.all_to_all_single(out, chunks.contiguous(), group=group)
        return out
    P, me = dist.get_world_size(group), dist.get_rank(group)
    host = chunks.contiguous().cpu()
    gathered = [torch.empty_like(host) for _ in range(P)]
    dist.all_gather(gathered, host, group=group)
    return torch.stack([gathered[src][me] for src in range(P)]).to(chunks.device)


def seq_all_to_all(x: torch.Tensor, scatter_idx: int, gather_idx: int, group) -> torch.Tensor:
    """DeepSpeed `_SeqAllToAll` on a (B, S, heads, d) tensor: split dim `scatter_idx` over the P ranks, concatenate the
    received pieces along dim `gather_idx` (rank order)."""
    P = dist.get_world_size(group)
    if P == 1:
        return x
    if x.shape[scatter_idx] % P:
        raise ValueError(f"dimension {scatter_idx} ({x.shape[scatter_idx]}) is not divisible by the group size {P}")
    parts = torch.stack(torch.chunk(x, P, dim=scatter_idx))      # (P, ..., scatter/P, ...)
    recv = _exchange(parts, group)                               # recv[i] = what rank i held for me
    return torch.cat(list(recv.unbind(0)), dim=gather_idx)


class DistributedAttention:
    """`module.dist_attn` of the reference: called as dist_attn(q, k, v, batch_dim_idx, **kw) on (B, L/P, heads, d)
    tensors, returns (B, L/P, heads, d).  `local_attn` gets (B, heads/P, L, d) like SDPA."""

    def __init__(self, local_attn, group=None, scatter_idx: int = 2, gather_idx: int = 1):
        self.local_attn, self.spg = local_attn, group
        self.scatter_idx, self.gather_idx = scatter_idx, gather_idx

    def __call__(self, query, key, value, batch_dim_idx: int = 0, *args, **kwargs):
        g = self.spg if self.spg is not None else _GROUP
        if batch_dim_idx != 0:
            raise ValueError("batch must be dimension 0")
        q = seq_all_to_all(query, self.scatter_idx, self.gather_idx, g).transpose(1, 2)
        k = seq_all_to_all(key, self.scatter_idx, self.gather_idx, g).transpose(1, 2)
        v = seq_all_to_all(value, self.scatter_idx, self.gather_idx, g).transpose(1, 2)
        ctx = self.local_attn(q, k, v, *args, **kwargs).transpose(1, 2)          # (B, L, heads/P, d)
        return seq_all_to_all(ctx.contiguous(), self.gather_idx, self.scatter_idx, g)

    forward = __call__


def shard_sequence(input_emb: torch.Tensor, position_ids: torch.Tensor, group=None):
    """LVM/model.py:457-463: this rank's contiguous L/P slice of the embeddings and positions."""
    P, r = sp_world(group), sp_rank(group)
    L = input_emb.shape[1]
    assert L % P == 0, "sequence length must be divisible by the sequence-parallel size"
    c = L // P
    return input_emb[:, r * c:(r + 1) * c].contiguous(), position_ids[:, r * c:(r + 1) * c].contiguous()


def gather_sequence(hidden: torch.Tensor, group=None) -> torch.Tensor:
    """LVM/model.py:466-473: all-gather of the last hidden state along the sequence."""
    group = group if group is not None else _GROUP
    P = sp_world(group)
    if P == 1:
        return hidden
    if dist.get_backend(group) == "nccl":
        parts = [torch.empty_like(hidden) for _ in range(P)]
        dist.all_gather(parts, hidden.contiguous(), group=group)
    else:
        host = hidden.contiguous().cpu()
        hp = [torch.empty_like(host) for _ in range(P)]
        dist.all_gather(hp, host, group=group)
        parts = [t.to(hidden.device) for t in hp]
    return torch.cat(parts, dim=1)
