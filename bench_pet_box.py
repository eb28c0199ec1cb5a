"""Secondary benchmark: PET energy + forces of ONE large box over all GPUs (strong scaling).

  python bench_pet_box.py --gpus N --steps K --warmup W [--atoms 100000] [--hops H]

One "step" = slab + halo selection, device neighbour list, graph build, forward and dE/dR of this rank's sub-system, ONE
all-reduce(sum) of [gradient | energy] (metatrain_amd/pet/partition.py). The exact halo is (num_gnn_layers + 1) cutoffs;
`--hops 2` uses the range the reference declares (pet/model.py:1004), ~1e-4 accurate. At --gpus 1 the sub-system is the
whole box. Prints ONE JSON line; `bench.py` (independent boxes, weak scaling) stays the headline metric.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL at N > 1)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--atoms", type=int, default=100000)
    ap.add_argument("--hops", type=int, default=None, help="halo thickness in cutoffs (default: exact)")
    ap.add_argument("--exchange", action="store_true",
                    help="ONE-cutoff halos and an all-to-all of the foreign centres' edge tokens once per GNN layer "
                         "(and of their adjoints in the reverse pass) instead of (layers + 1)-cutoff halos")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="on ONE GPU: time every rank's share of a W-rank partition one after the other and report the "
                         "busiest rank (what a W-GPU step would take without its 1.2 MB all-reduce)")
    args = ap.parse_args()

    from metatrain_amd import distributed as pdist

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher: start the ranks ourselves (as bench.py does)
        import socket
        import subprocess

        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        raise SystemExit(subprocess.call(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
             "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]))
    rank, local_rank, world = pdist.env_rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench_pet_box.py needs MI355X GPUs"
    backend = os.environ.get("PET_BENCH_BACKEND", "nccl")  # "gloo": debugging aid, ranks may share a GPU (bench.py)
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        pdist.init(backend, dev)
        pdist.selftest(args.gpus, dev)  # the ranks the launch line names joined, and a collective works

    from metatrain_amd import runtime as rt
    from metatrain_amd.pet import default_hypers, partition
    from metatrain_amd.synthetic import random_box, synthetic_params

    hypers = default_hypers()
    model = rt.HipModel(hypers, [1, 6, 7, 8])
    model.load({k: v.to(dev) for k, v in synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0).items()}, "energy")
    pos, z, cell = random_box(args.atoms, seed=0)  # the same box on every rank
    posd, zd = pos.to(dev), z.to(dev)
    reduce = (lambda t: torch.distributed.all_reduce(t)) if world > 1 else None
    if args.emulate_world > 1:
        return emulate(args, model, hypers, posd, zd, cell, dev)

    def all_to_all(out, inp, out_splits, in_splits):
        if world == 1:
            return
        if backend == "nccl":  # RCCL on the current stream: ordered after the gather kernel the library just launched
            torch.distributed.all_to_all_single(out, inp, out_splits, in_splits)
        else:                  # gloo debugging aid: staged through the host
            recv = torch.empty(out.shape, dtype=out.dtype)
            torch.distributed.all_to_all_single(recv, inp.cpu(), out_splits, in_splits)
            out.copy_(recv)

    ghost = [0, 0]

    def step():
        if args.exchange:
            e, grad, n_sub, n_owned, ghost[0], ghost[1] = partition.energy_and_gradient_exchange(
                model, posd, zd, cell, [True] * 3, world, rank, all_to_all, all_reduce=reduce)
            return e, grad, n_sub, n_owned
        return partition.energy_and_gradient(model, posd, zd, cell, [True] * 3, world, rank, all_reduce=reduce,
                                             hops=args.hops)

    for _ in range(args.warmup):
        step()
    pdist.barrier(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e, grad, n_sub, n_owned = step()
    pdist.barrier(dev)
    torch.cuda.synchronize()
    elapsed = pdist.max_over_ranks(time.perf_counter() - t0, dev)
    n_sub_max = int(pdist.max_over_ranks(float(n_sub), dev))
    hops = 1 if args.exchange else args.hops if args.hops is not None else hypers["num_gnn_layers"] + 1
    exchange = (f", all-to-all of {ghost[1]} of {ghost[0]} edge-token rows ({ghost[1] * hypers['d_pet'] * 4 / 1e6:.1f} MB) "
                f"per GNN layer and direction on rank 0" if args.exchange else "")
    if rank == 0:
        assert torch.isfinite(grad).all()
        print(json.dumps({
            "metric": "atom-steps/sec (energy+forces) PET, one box over all GPUs",
            "value": args.atoms * args.steps / elapsed, "unit": "atom-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic random periodic box (0.05 atoms/A^3), random-init weights",
            "config": {"workload": f"PET forward + dE/dR of ONE {args.atoms}-atom box, centres partitioned into {world} "
                                   f"slab(s) + {hops} x {hypers['cutoff']} A halos, device neighbour list per step, one "
                                   f"all-reduce of [gradient | energy] ({(3 * args.atoms + 1) * 4 / 1e6:.1f} MB)" + exchange,
                       "arithmetic": "fp32 results; GEMM stages as f16x3 split-operand MFMA products",
                       "atoms_on_the_busiest_rank": n_sub_max, "atoms_owned_rank0": n_owned,
                       "total_energy": float(e)},
        }), flush=True)
    if world > 1:
        pdist.barrier(dev)
        torch.distributed.destroy_process_group()


def emulate(args, model, hypers, posd, zd, cell, dev):
    from metatrain_amd.pet import partition

    W = args.emulate_world
    per_rank, subs = [], []
    ghost = [0, 0]
    for r in range(W):
        def step():
            if args.exchange:  # no peers here: the hook leaves the (zeroed) ghost rows alone, the compute is the same
                e, grad, n_sub, n_owned, ghost[0], g1 = partition.energy_and_gradient_exchange(
                    model, posd, zd, cell, [True] * 3, W, r, lambda *a: None)
                ghost[1] = max(ghost[1], g1)
                return e, grad, n_sub, n_owned
            return partition.energy_and_gradient(model, posd, zd, cell, [True] * 3, W, r, hops=args.hops)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            e, grad, n_sub, n_owned = step()
        torch.cuda.synchronize()
        per_rank.append((time.perf_counter() - t0) / args.steps * 1e3)
        subs.append(n_sub)
    whole = None
    try:
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            partition.energy_and_gradient(model, posd, zd, cell, [True] * 3, 1, 0)
            torch.cuda.synchronize()
            whole = (time.perf_counter() - t0) * 1e3
    except torch.OutOfMemoryError:  # a box that only fits partitioned (1 M atoms: 560 GB of workspace in one piece)
        whole = None
    hops = 1 if args.exchange else args.hops if args.hops is not None else hypers["num_gnn_layers"] + 1
    print(json.dumps({
        "metric": "ms per step of the busiest rank, PET one box partitioned (ranks emulated one after the other on 1 GPU)",
        "value": max(per_rank), "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "higher_is_better": False, "data": "synthetic random periodic box (0.05 atoms/A^3), random-init weights",
        "config": {"workload": f"ONE {args.atoms}-atom box cut into {W} slabs + {hops} x {hypers['cutoff']} A halos",
                   "ms_per_rank": [round(t, 2) for t in per_rank], "atoms_per_rank": subs,
                   "whole_box_on_one_gpu_ms": round(whole, 2) if whole else None,
                   "projected_speedup_over_one_gpu": round(whole / max(per_rank), 2) if whole else None,
                   "not_included": "the all-reduce of [gradient | energy] (1.2 MB per 100 k atoms)" + (
                       f"; {2 * hypers['num_gnn_layers']} all-to-alls of up to {ghost[1]} edge-token rows "
                       f"({ghost[1] * hypers['d_pet'] * 4 / 1e6:.1f} MB) each" if args.exchange else "")},
    }), flush=True)


if __name__ == "__main__":
    main()
