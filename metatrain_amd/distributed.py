"""One process per GPU over torch.distributed ("nccl" is RCCL on ROCm; "gloo" for the CPU tests).

The inference / force path shards independent structures over ranks and needs NO data-path
collective; the only communication is the benchmark's barrier and the
max-over-ranks of the elapsed time. The training row (a16/a19) adds ONE collective per step:
the mean all-reduce of the flat 11.6 MB gradient bucket (torch DDP's role in the reference,
utils/distributed/distributed_data_parallel.py:7-15, pet/trainer.py:344-345).
"""
import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the launcher's environment (torch.distributed.run)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device: torch.device = None) -> None:
    if dist.is_initialized():
        return
    kwargs = {}
    if backend == "nccl" and device is not None:
        kwargs["device_id"] = device
    dist.init_process_group(backend, **kwargs)


def shard_structures(n_total: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of structure ids to ranks (identical synthetic boxes are balanced by
    construction; the reference's DistributedSampler does the same without shuffling for
    evaluation, pet/trainer.py:227-247)."""
    return list(range(rank, n_total, world))


def box_seeds(boxes_per_rank: int, rank: int) -> List[int]:
    """Weak scaling: every rank owns `boxes_per_rank` boxes with globally unique seeds."""
    return [rank * boxes_per_rank + b for b in range(boxes_per_rank)]


def barrier(device: torch.device = None) -> None:
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device: torch.device) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device: torch.device) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_reduce_gradients(model) -> None:
    """Mean of the gradient slots over ranks, as ONE collective on the flat bucket (2 903 298 fp32 =
    11.6 MB: a single ring all-reduce is per-link bound on xGMI, and one bucket keeps it at one
    launch). ``model`` exposes ``flat_grad()`` / ``set_flat_grad(t)`` (runtime.HipModel)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    flat = model.flat_grad()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    model.set_flat_grad(flat)
