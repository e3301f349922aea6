"""Multi-GPU plumbing for one-process-per-GPU runs (bench.py, user scripts).

MPC instances are independent, so the data path needs no collective at all: each rank solves its own
shard.  The only exchange is ONE broadcast of the kinematic constants (``oh_chain``, 2952 bytes) from
rank 0, done over RCCL through ``torch.distributed`` (backend "nccl" is RCCL on ROCm; "gloo" is used by
the CPU tests).  torch is imported lazily and only here -- liboptas_hip itself has no torch dependency.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _lib


def shard(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous partition [lo, hi) of n_total instances for this rank."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_process_group(backend: str, local_rank: int = 0):
    import torch
    import torch.distributed as dist

    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def broadcast_chain(chain: _lib.oh_chain, device: str, src: int = 0):
    """Broadcast the constants block from ``src``; returns (torch uint8 tensor on ``device``, oh_chain copy).
    On ranks != src the content of ``chain`` is ignored."""
    import torch
    import torch.distributed as dist

    nbytes = C.sizeof(_lib.oh_chain)
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    if dist.get_rank() == src:
        buf.copy_(torch.frombuffer(bytearray(bytes(chain)), dtype=torch.uint8))
    dist.broadcast(buf, src=src)
    out = _lib.oh_chain.from_buffer_copy(buf.cpu().numpy().tobytes())
    return buf, out


def max_over_ranks(value: float, device: str) -> float:
    import torch
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device: str) -> float:
    import torch
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
