"""One process per GPU over torch.distributed ("nccl" is RCCL on ROCm; "gloo" for the CPU tests).

The inference / force path shards independent structures over ranks and needs NO data-path
collective; the only communication is the benchmark's barrier and the
max-over-ranks of the elapsed time. The training row (a16/a19) adds ONE collective per step:
the mean all-reduce of the flat 11.6 MB gradient bucket (torch DDP's role in the reference,
utils/distributed/distributed_data_parallel.py:7-15, pet/trainer.py:344-345).
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the launcher's environment (torch.distributed.run)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def expand_hostlist(nodelist: str) -> List[str]:
    """SLURM's compressed node list (``nid[001-003,007],login1``) as host names -- what ``hostlist.expand_hostlist`` gives
    the reference (``utils/distributed/slurm.py:65``); python-hostlist is not a dependency here."""
    hosts: List[str] = []
    depth, start = 0, 0
    parts = []
    for k, ch in enumerate(nodelist):  # split on commas outside brackets
        depth += ch == "["
        depth -= ch == "]"
        if ch == "," and depth == 0:
            parts.append(nodelist[start:k])
            start = k + 1
    parts.append(nodelist[start:])
    def expand(part: str) -> List[str]:  # every bracket group of the part: cartesian product, left to right
        if "[" not in part:
            return [part]
        head, rest = part.split("[", 1)
        body, tail = rest.split("]", 1)
        tails = expand(tail)
        out: List[str] = []
        for item in body.split(","):
            if "-" in item:
                lo, hi = item.split("-")
                mids = [f"{v:0{len(lo)}d}" for v in range(int(lo), int(hi) + 1)]
            else:
                mids = [item]
            out += [f"{head}{mid}{t}" for mid in mids for t in tails]
        return out

    for part in parts:
        if part:
            hosts += expand(part)
    return hosts


def is_slurm() -> bool:
    """``utils/distributed/slurm.py:10-16``."""
    return ("SLURM_JOB_ID" in os.environ) and ("SLURM_PROCID" in os.environ)


def resolve_distributed(distributed) -> bool:
    """``utils/distributed/slurm.py:28-41``: unset = on exactly when inside a SLURM job with more than one task."""
    if distributed is None:
        return is_slurm() and int(os.environ.get("SLURM_NTASKS", "1")) > 1
    return bool(distributed)


def slurm_environment(port: int) -> Tuple[int, int, int]:
    """SLURM variables -> ``MASTER_ADDR / MASTER_PORT / WORLD_SIZE / RANK / LOCAL_RANK`` (``DistributedEnvironment``,
    ``utils/distributed/slurm.py:44-79``): first node of the job is the master. Returns (rank, local_rank, world)."""
    hostnames = expand_hostlist(os.environ["SLURM_JOB_NODELIST"])
    os.environ["MASTER_ADDR"] = hostnames[0]
    os.environ["MASTER_PORT"] = str(port)
    os.environ["WORLD_SIZE"] = os.environ["SLURM_NTASKS"]
    os.environ["RANK"] = os.environ["SLURM_PROCID"]
    os.environ["LOCAL_RANK"] = os.environ["SLURM_LOCALID"]
    return env_rank()


def initialize_slurm_nccl_process_group(port: int) -> Tuple[torch.device, int, int]:
    """``utils/distributed/slurm.py:82-102``: one process per GPU under ``srun``; device = local rank modulo the
    visible device count; ``backend="nccl"`` is RCCL on ROCm. Returns (device, world_size, rank)."""
    _, local_rank, _ = slurm_environment(port)
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    dist.init_process_group(backend="nccl", device_id=device)
    return device, dist.get_world_size(), dist.get_rank()


def init(backend: str, device: torch.device = None) -> None:
    if dist.is_initialized():
        return
    kwargs = {}
    if backend == "nccl" and device is not None:
        kwargs["device_id"] = device
    dist.init_process_group(backend, **kwargs)


def selftest(expected_world: int, device: torch.device = None) -> None:
    """Fail loudly, before any timed work, if the job is not what the launcher line promised: the process group must
    hold ``expected_world`` ranks and one all-reduce over it must work (sum of ones = world size on every rank). A bench
    that silently ran on fewer ranks -- a rank that died at start-up, a launcher that fell back to one process -- would
    report a number for the wrong job. Called by every ``bench*.py`` right after ``init`` when ``--gpus`` > 1."""
    if not dist.is_initialized():
        if expected_world > 1:
            raise SystemExit(f"distributed self-test: --gpus {expected_world} but no process group was initialised")
        return
    world = dist.get_world_size()
    if world != expected_world:
        raise SystemExit(f"distributed self-test: {world} rank(s) joined, {expected_world} expected")
    on_gpu = device is not None and device.type == "cuda"
    t = torch.ones(1 << 16, dtype=torch.float32, device=device if on_gpu else "cpu")
    dist.all_reduce(t)
    if on_gpu:
        torch.cuda.synchronize(device)
    if float(t[0]) != float(world) or float(t[-1]) != float(world):
        raise SystemExit(f"distributed self-test: all-reduce of ones over {world} rank(s) returned {float(t[0])}")


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


class GradientReduce:
    """An in-flight mean all-reduce of the flat gradient bucket (``all_reduce_gradients_async``). ``wait()`` makes the
    CURRENT stream wait for the collective (RCCL: a stream dependency, the host does not block; gloo: the host waits),
    finishes the mean and writes the bucket back into the model's gradient slots."""

    def __init__(self, model, flat=None, work=None, divide: int = 1, events=None, timing=None):
        self.model, self.flat, self.work, self.divide, self.events, self.timing = model, flat, work, divide, events, timing

    def wait(self) -> None:
        if self.flat is None:
            return
        if self.work is not None:
            self.work.wait()
        if self.divide > 1:
            self.flat /= self.divide
        self.model.set_flat_grad(self.flat)
        if self.events is not None:
            self.events[1].record()
            self.timing.append(self.events)
        self.flat = self.work = None


def all_reduce_gradients_async(model, timing: Optional[list] = None) -> GradientReduce:
    """Start the mean of the gradient slots over ranks as ONE collective on the flat bucket (2 903 298 fp32 = 11.6 MB: a
    single ring all-reduce is per-link bound on xGMI, and one bucket keeps it at one launch) and return at once: the
    bucket is copied out on the current stream, the collective is issued with ``async_op=True`` -- ProcessGroupNCCL (= RCCL)
    runs it on ITS OWN stream behind an event of the current one --, and the caller's stream stays free for whatever does
    not need the reduced gradients (the next batch's neighbour lists and graph build: ``pet/trainer.py TrainStep.begin`` /
    ``.end``) until ``wait()``. ``model`` exposes ``flat_grad()`` / ``set_flat_grad(t)`` (runtime.HipModel). RCCL takes the mean
    itself (``ReduceOp.AVG``); gloo (CPU tests, debugging) sums and ``wait()`` divides. ``timing``: a list that receives a pair
    of recorded events around copy-out ... copy-in (the benches read ``elapsed_time`` after the step)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return GradientReduce(model)
    ev = None
    if timing is not None and torch.cuda.is_available():
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    flat = model.flat_grad()
    if dist.get_backend() == "nccl":
        work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True)
        return GradientReduce(model, flat, work, 1, ev, timing)
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    return GradientReduce(model, flat, work, dist.get_world_size(), ev, timing)


def all_reduce_gradients(model, timing: Optional[list] = None) -> None:
    """``all_reduce_gradients_async(model, timing).wait()``: the blocking form (same collective, same result)."""
    all_reduce_gradients_async(model, timing).wait()
