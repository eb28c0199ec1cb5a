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
