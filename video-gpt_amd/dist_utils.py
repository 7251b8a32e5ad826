"""One-process-per-GPU plumbing for the multi-GPU bench and replica sampling.

The inference path shards by clip ("replicas only": independent clips/seeds per GPU, no data-path
collective — DESIGN.md §Multi-GPU); the only communication is the barrier and the max-over-ranks
reduction of the timed region.  On the GPU box the backend is "nccl" (= RCCL over xGMI); the same
code runs under "gloo" on CPU, which is how tests cover world_size > 1.
"""
from __future__ import annotations

import os
import time
from typing import Callable, List, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_from_env(backend: str, device=None) -> Tuple[int, int]:
    rank, _, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """Clip indices owned by `rank` (round-robin; every unit owned exactly once)."""
    return list(range(rank, n_units, world))


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def timed_region(fn: Callable[[], None], sync: Callable[[], None], device="cpu") -> float:
    """barrier + sync, run fn, sync + barrier; returns the MAX elapsed seconds over ranks."""
    sync()
    barrier()
    t0 = time.perf_counter()
    fn()
    sync()
    barrier()
    return max_over_ranks(time.perf_counter() - t0, device)
