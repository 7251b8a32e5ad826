"""Process-group state of the reference's `LVM/acceleration/parallel_states.py` (:18-80) on torch.distributed / RCCL.

`hccl_info` keeps the reference's name and fields (`group`, `world_size`, `rank` of the SEQUENCE-parallel group; the
reference initialises them to None / 0 / -1 and fills them in `initialize_sequence_parallel_group`).  One process per
GPU; the backend is "nccl" (= RCCL over xGMI on ROCm) whenever a GPU is visible, "gloo" otherwise (CPU tests).
The Ascend LCCL branch (`lccl_info`, :55-60) does not exist here."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import sequence_parallel as SP


class COMM_INFO:
    def __init__(self):
        self.group = None
        self.world_size = 0
        self.rank = -1


hccl_info = COMM_INFO()
_SEQUENCE_PARALLEL_STATE = False


def initialize_sequence_parallel_state(sequence_parallel_size):
    """LVM/acceleration/parallel_states.py:27-31."""
    global _SEQUENCE_PARALLEL_STATE
    if sequence_parallel_size >= 1:
        _SEQUENCE_PARALLEL_STATE = True
        initialize_sequence_parallel_group(sequence_parallel_size)


def set_sequence_parallel_state(state):
    global _SEQUENCE_PARALLEL_STATE
    _SEQUENCE_PARALLEL_STATE = state


def get_sequence_parallel_state():
    return _SEQUENCE_PARALLEL_STATE


def initialize_sequence_parallel_group(sequence_parallel_size):
    """Groups of `sequence_parallel_size` consecutive ranks (:40-53); also installs the group in sequence_parallel.py,
    which `replace_attention` and `LVM.frame_block_forward` read."""
    rank = int(os.getenv("RANK", "0"))
    world_size = int(os.getenv("WORLD_SIZE", "1"))
    assert world_size % sequence_parallel_size == 0, "world_size must be divisible by sequence_parallel_size"
    hccl_info.world_size = sequence_parallel_size
    hccl_info.rank = rank % sequence_parallel_size
    if not dist.is_initialized():
        if world_size != 1:
            raise RuntimeError("initialize torch.distributed (init_npu_env) before the sequence-parallel groups")
        return          # single process: world-size-1 state without a process group
    hccl_info.group = SP.initialize_sequence_parallel_state(sequence_parallel_size)


def destroy_sequence_parallel_group():
    if dist.is_initialized():
        dist.destroy_process_group()


def init_npu_env(args):
    """:66-80 — the reference's entry point despite its name: sets the device of this rank, initialises the default
    process group (DeepSpeed's `init_distributed` there, plain torch.distributed here) and the sequence-parallel groups."""
    local_rank = int(os.getenv("RANK", 0))
    world_size = int(os.getenv("WORLD_SIZE", 1))
    args.local_rank = local_rank
    args.world_size = world_size
    gpu = torch.cuda.is_available()
    if gpu:
        torch.cuda.set_device(int(os.getenv("LOCAL_RANK", local_rank)) % max(torch.cuda.device_count(), 1))
    if world_size > 1 and not dist.is_initialized():
        dist.init_process_group("nccl" if gpu else "gloo", rank=local_rank, world_size=world_size)
    initialize_sequence_parallel_state(args.sequence_parallel_size)
    return args
