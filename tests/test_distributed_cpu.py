"""world_size-2 gloo test of the N>1 path used by bench.py (sharding + timing reduction)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from metatrain_amd import distributed as d

    assert d.env_rank() == (rank, rank, world)
    d.init("gloo")
    dev = torch.device("cpu")
    d.selftest(world, dev)                               # the benches' preamble: right rank count, a working collective
    try:
        d.selftest(world + 1, dev)                       # "--gpus 3" with two ranks in the group: refuse to measure
        raise AssertionError("selftest accepted a smaller job than the launch line named")
    except SystemExit as exc:
        assert "2 rank(s) joined, 3 expected" in str(exc)
    d.barrier(dev)
    elapsed = d.max_over_ranks(1.0 + rank, dev)          # slowest rank defines the step time
    atoms = d.sum_over_ranks(10000.0 * len(d.box_seeds(2, rank)), dev)
    out.put((rank, elapsed, atoms, d.box_seeds(2, rank), d.shard_structures(7, rank, world)))
    d.barrier(dev)
    torch.distributed.destroy_process_group()


def test_two_rank_sharding_and_timing_reduce():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] == 2.0 for r in res)                  # max over ranks
    assert all(r[2] == 40000.0 for r in res)              # whole-job atom count
    seeds = res[0][3] + res[1][3]
    assert sorted(seeds) == [0, 1, 2, 3]                  # globally unique boxes
    shards = res[0][4] + res[1][4]
    assert sorted(shards) == list(range(7))               # a partition, nothing dropped or doubled


class _FakeModel:
    """Stands in for runtime.HipModel's flat gradient bucket (no GPU in the CPU suite)."""

    def __init__(self, flat):
        self.flat = flat

    def flat_grad(self):
        return self.flat.clone()

    def set_flat_grad(self, t):
        self.flat = t.clone()


def _grad_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from metatrain_amd import distributed as d

    d.init("gloo")
    n = 2903298  # the PET default parameter count: one 11.6 MB bucket
    g = torch.full((n,), float(rank + 1))
    g[rank] = 100.0
    model = _FakeModel(g)
    d.all_reduce_gradients(model)
    out.put((rank, model.flat[:3].tolist(), float(model.flat[5])))
    d.barrier(torch.device("cpu"))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_all_reduce_is_the_mean_of_one_flat_bucket():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, head, mid in res:  # identical on both ranks: mean over ranks (DDP semantics)
        assert head == [(100.0 + 2.0) / 2, (1.0 + 100.0) / 2, 1.5]
        assert mid == 1.5


def _grad_async_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from metatrain_amd import distributed as d

    d.init("gloo")
    n = 2903298
    gen = torch.Generator().manual_seed(7 + rank)
    g = torch.randn(n, generator=gen)
    blocking = _FakeModel(g.clone())
    d.all_reduce_gradients(blocking)
    model = _FakeModel(g.clone())
    handle = d.all_reduce_gradients_async(model)
    # work that does not need the reduced gradients runs between the start of the collective and wait() (the next batch's
    # graph build in TrainStep.begin / .end); the model's slots still hold the local gradient until wait()
    busy = float((torch.arange(1000.0) ** 0.5).sum())
    untouched = bool(torch.equal(model.flat, g))
    handle.wait()
    handle.wait()  # idempotent
    out.put((rank, untouched, bool(torch.equal(model.flat, blocking.flat)), float(model.flat[:16].double().sum()), busy > 0))
    d.barrier(torch.device("cpu"))
    torch.distributed.destroy_process_group()


def test_async_gradient_all_reduce_equals_the_blocking_one():
    """the gradient all-reduce is issued as soon as the flat bucket is final and waited for just before the
    optimizer step (``distributed.all_reduce_gradients_async`` -> ``GradientReduce.wait``; under RCCL the collective runs on
    the process group's own stream): same collective, bit-identical result to the blocking call, on both ranks."""
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_async_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] and r[4] for r in res)
    assert res[0][3] == res[1][3]  # the same mean on both ranks


def test_lr_schedule_matches_reference_formula():
    """Linear warm-up then cosine (pet/trainer.py:56-86): known points of the closed form."""
    from metatrain_amd.pet.trainer import lr_lambda

    total, wf = 1000, 0.01  # 10 warm-up steps
    assert lr_lambda(0, total, wf) == 0.0
    assert lr_lambda(5, total, wf) == 0.5
    assert lr_lambda(10, total, wf) == 1.0
    assert abs(lr_lambda(505, total, wf) - 0.5) < 1e-12
    assert abs(lr_lambda(1000, total, wf)) < 1e-12


def test_slurm_environment_and_process_group_call_sequence(monkeypatch):
    """tests/utils/test_slurm.py:81-117 of the reference, on this package's rendezvous: SLURM variables become the
    env:// variables (first node of the compressed node list is the master), the device is the local rank modulo
    the device count, and the NCCL (= RCCL) group is initialised on that device -- all calls faked, no communication."""
    from metatrain_amd import distributed as d

    assert d.expand_hostlist("nid[001-003,007],login1") == ["nid001", "nid002", "nid003", "nid007", "login1"]
    assert d.expand_hostlist("node12") == ["node12"]
    assert d.expand_hostlist("rack[1-2]n[01-02],x") == ["rack1n01", "rack1n02", "rack2n01", "rack2n02", "x"]
    for k, v in {"SLURM_JOB_ID": "7", "SLURM_JOB_NODELIST": "gpu[05-06]", "SLURM_NTASKS": "16", "SLURM_PROCID": "11",
                 "SLURM_LOCALID": "3"}.items():
        monkeypatch.setenv(k, v)
    assert d.is_slurm() and d.resolve_distributed(None) and not d.resolve_distributed(False)
    calls = []
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(torch.cuda, "set_device", lambda dev: calls.append(("set_device", str(dev))))
    monkeypatch.setattr(torch.distributed, "init_process_group",
                        lambda backend, device_id=None: calls.append(("init", backend, str(device_id))))
    monkeypatch.setattr(torch.distributed, "get_world_size", lambda: 16)
    monkeypatch.setattr(torch.distributed, "get_rank", lambda: 11)
    dev, world, rank = d.initialize_slurm_nccl_process_group(39591)
    assert (str(dev), world, rank) == ("cuda:1", 16, 11)          # 3 % 2
    assert calls == [("set_device", "cuda:1"), ("init", "nccl", "cuda:1")]
    assert os.environ["MASTER_ADDR"] == "gpu05" and os.environ["MASTER_PORT"] == "39591"
    assert (os.environ["WORLD_SIZE"], os.environ["RANK"], os.environ["LOCAL_RANK"]) == ("16", "11", "3")
    for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
