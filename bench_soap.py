"""Secondary benchmark (BASELINE.json configs[4]): SOAP-BPNN energy + forces on one large synthetic box.

  python bench_soap.py --gpus N --steps K --warmup W [--atoms 100000] [--alchemical]

One "step" = graph build (edge vectors, CSR, ij->ji) + spherical expansion + power spectrum + LayerNorm/MLP
tail + the reverse pass to dE/dR for one random periodic box per GPU (rho = 0.05 / A^3, 5 A cutoff, default
SOAP-BPNN hypers: max_angular 6, max_radial 7 -> 4544 power-spectrum features per atom). Boxes are
independent per rank (weak scaling, no data-path collective). Prints ONE JSON line with the roofline of the
dominant kernel and the CPU oracle timed beside it. Parity of this row is unpinned against torch-spex
(oracle/soap.py header); the GPU path is checked against the oracle in tests/test_gpu_soap.py.
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

HBM_PEAK_GBS = 8000.0


def cpu_baseline(hypers, n=10000):
    from oracle import nl as onl
    from oracle import pet as opet
    from oracle import soap as osoap

    types = [1, 6, 7, 8]
    n_per_l = osoap.basis(hypers)[0]
    params = osoap.synthetic_params(hypers, 4, n_per_l, 0, torch.float32)
    pos, z, cell = opet.random_box(n, seed=0)
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, 5.0)
    args = (params, hypers, types, pos, cell[None], torch.tensor(i), torch.tensor(j), torch.tensor(s).long(), z,
            torch.zeros(n, dtype=torch.long))
    nt = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(nt)
    # one 10 000-atom box (the baseline used to be a 500-atom box next to a 100 000-atom GPU number); the
    # oracle's per-atom cost does not depend on the box size beyond this (cell list, ~26 neighbours per atom)
    t0 = time.perf_counter()
    reps = 1
    osoap.energy_and_gradient(*args)
    dt = (time.perf_counter() - t0) / reps
    return {"value": n / dt, "unit": "atom-steps/s", "cores": nt, "kind": "port",
            "sample": f"{reps} x (forward + dE/dR) of one {n}-atom box with the torch-CPU oracle, {dt:.2f} s each"}



SOAP_STAGE_KERNELS = {"soap_expand": "k_soap_expand_w", "soap_ps": "k_soap_ps_m", "soap_tail": "k_soap_tail_fwd_set",
                      "soap_tail_bwd": "k_soap_tail_bwd_set", "soap_ps_bwd": "k_soap_ps_bwd_m",
                      "soap_expand_bwd": "k_soap_expand_bwd_p"}


def pmc_traffic(stage, n_pairs):
    """HBM bytes per launch of the stage's kernel and of a whole step from the committed rocprofv3 PMC passes of this bench
    (profiles/r0N_soap_traffic.json of the newest round: --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs, (2 FETCH + WRITE) x 1024); None
    unless the profiled run had the same number of pairs."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_soap_traffic.json") for r in (6, 5, 4)) if os.path.exists(q)),
                os.path.join(ROOT, "profiles", "r04_soap_traffic.json"))
    if not os.path.exists(path):
        return None, None
    with open(path) as fh:
        data = json.load(fh)
    if data.get("workload_edges") != n_pairs:
        return None, None
    recs = [v for k, v in data["kernels"].items() if k.startswith(SOAP_STAGE_KERNELS.get(stage, "-"))]
    return (recs[0]["hbm_bytes_per_launch"] if recs else None), data.get("step_hbm_bytes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--atoms", type=int, default=100000)
    ap.add_argument("--alchemical", action="store_true", help="non-legacy: 4 pseudo-species + centre encoding")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--partition", action="store_true",
                    help="strong scaling: ONE box for all ranks, cut into slabs + halos, one all-reduce of energy and "
                         "gradient per step (metatrain_amd/soap_bpnn/partition.py; BASELINE configs[4] at 8 GPUs)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="with --partition on ONE GPU: time every rank's share of a W-rank partition one after the other "
                         "and report the busiest rank (a W-GPU step without its 1.2 MB all-reduce)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=INT",
                    help="library switch for A/B runs (pet_config_set), e.g. --set soap_packed=0")
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
    assert torch.cuda.is_available(), "bench_soap.py needs MI355X GPUs"
    backend = os.environ.get("PET_BENCH_BACKEND", "nccl")  # "gloo": debugging aid, ranks may share a GPU (bench.py)
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        pdist.init(backend, dev)
        pdist.selftest(args.gpus, dev)  # the ranks the launch line names joined, and a collective works

    from metatrain_amd import runtime as rt
    from metatrain_amd.soap_bpnn import SoapBpnnHip, default_hypers
    from metatrain_amd.synthetic import random_box

    for kv in args.set:
        rt.config_set(kv.split("=")[0], int(kv.split("=")[1]))
    hypers = default_hypers()
    hypers["legacy"] = not args.alchemical
    model = SoapBpnnHip(hypers, [1, 6, 7, 8])
    S, H = model.feature_size, hypers["bpnn"]["num_neurons_per_layer"]
    gen = torch.Generator().manual_seed(0)
    params = {}
    if args.alchemical:
        params["species_embedding.weight"] = torch.randn(4, 4, generator=gen)
        params["center_encoding.weight"] = torch.randn(4, S, generator=gen)
    for s in range(1 if args.alchemical else 4):
        params[f"layernorm.{s}.weight"] = 1 + 0.1 * torch.randn(S, generator=gen)
        params[f"layernorm.{s}.bias"] = 0.1 * torch.randn(S, generator=gen)
        params[f"bpnn.{s}.0.weight"] = torch.randn(H, S, generator=gen) / S**0.5
        params[f"bpnn.{s}.2.weight"] = torch.randn(H, H, generator=gen) / H**0.5
        params[f"last_layers.energy.{s}.weight"] = torch.randn(1, H, generator=gen) / H**0.5
    model.load({k: v.to(dev) for k, v in params.items()})

    if args.partition:
        return partitioned(args, model, S, rank, world, dev)
    pos, z, cell = random_box(args.atoms, seed=pdist.box_seeds(1, rank)[0])
    posd = pos.to(dev)
    pairs, _ = rt.neighbor_list(posd, cell, [True] * 3, model.cutoff)
    gargs = (posd, cell[None].to(dev), pairs[:, 0].contiguous(), pairs[:, 1].contiguous(), pairs[:, 2:5].contiguous(),
             z.to(dev), torch.zeros(args.atoms, dtype=torch.int32, device=dev))
    ones = torch.ones(args.atoms, device=dev)

    def step():
        g = model.graph(*gargs)
        a = model.forward(g)
        return a, model.backward(g, ones), g

    for _ in range(args.warmup):
        step()
    rt.profile(True)
    step()
    torch.cuda.synchronize()
    table = rt.profile_report()
    rt.profile(False)
    dominant = max(table, key=lambda r: r["total_ms"])
    pdist.barrier(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        atomic, grad, g = step()
    pdist.barrier(dev)
    elapsed = pdist.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        assert torch.isfinite(grad).all()
        ach = dominant["bytes"] / (dominant["total_ms"] * 1e-3) / 1e9
        out = {
            "metric": "atom-steps/sec (energy+forces) SOAP-BPNN",
            "value": args.atoms * world * args.steps / elapsed,
            "unit": "atom-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic random periodic box, random weights (parity unpinned vs torch-spex)",
            "config": {
                "workload": f"SOAP-BPNN forward + dE/dR, one {args.atoms}-atom box per GPU per step, 5 A cutoff, "
                            f"max_angular 6 / max_radial 7 ({S} features/atom), "
                            f"{'alchemical (4 pseudo-species)' if args.alchemical else 'legacy (per-species heads)'}",
                "pairs_per_gpu_per_step": int(g.n_edges),
                "total_energy_rank0": float(atomic.double().sum()),
                "power_spectrum_storage": ("full [N][S]" if args.alchemical or ("soap_packed=0" in args.set) else
                                           "upper triangle of every l block (p_l[a][b] = p_l[b][a]): "
                                           f"{sum((k * 4) * (k * 4 + 1) // 2 for k in model.n_per_l)} of {S} floats per atom"),
            },
            "roofline": {"bound": "hbm", "kernel": dominant["name"], "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": dominant["total_ms"],
                         "algorithmic_bytes_per_launch": dominant["bytes"], "traffic": None,
                         "stages_ms": {r["name"]: round(r["total_ms"], 3) for r in table}},
        }
        out["roofline"]["traffic"], out["roofline"]["step_traffic_bytes"] = pmc_traffic(dominant["name"], int(g.n_edges))
        if dominant["name"] in ("soap_expand", "soap_expand_bwd"):
            # the descriptor kernels are VALU work (spherical-harmonic recurrences per pair, fp64 adjoint accumulators,
            # no matrix product): neither the HBM nor the MFMA roof describes them; the HBM figure is kept as the
            # lower of the two fractions, the stage that IS an HBM stream is soap_tail (feature matrix read once)
            tail = [r for r in table if r["name"] == "soap_tail"]
            out["roofline"]["note"] = "VALU-bound stage (no matrix product); HBM-streaming stage for comparison: soap_tail"
            if tail:
                out["roofline"]["soap_tail_hbm_frac"] = tail[0]["bytes"] / (tail[0]["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(hypers)
        print(json.dumps(out), flush=True)
    if world > 1:
        pdist.barrier(dev)
        torch.distributed.destroy_process_group()


def partitioned(args, model, S, rank, world, dev):
    """One box, all ranks: positions replicated, centres partitioned, ONE all-reduce(sum) of [gradient | energy]."""
    from metatrain_amd import distributed as pdist
    from metatrain_amd.soap_bpnn import partition
    from metatrain_amd.synthetic import random_box

    pos, z, cell = random_box(args.atoms, seed=0)  # the same box on every rank
    posd, zd = pos.to(dev), z.to(dev)
    reduce = (lambda t: torch.distributed.all_reduce(t)) if world > 1 else None
    if args.emulate_world > 1 and world == 1:
        W, per_rank, subs = args.emulate_world, [], []
        for r in range(W):
            for _ in range(args.warmup):
                partition.energy_and_gradient(model, posd, zd, cell, [True] * 3, W, r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                e, grad, n_sub, n_owned = partition.energy_and_gradient(model, posd, zd, cell, [True] * 3, W, r)
            torch.cuda.synchronize()
            per_rank.append((time.perf_counter() - t0) / args.steps * 1e3)
            subs.append(n_sub)
        partition.energy_and_gradient(model, posd, zd, cell, [True] * 3, 1, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        partition.energy_and_gradient(model, posd, zd, cell, [True] * 3, 1, 0)
        torch.cuda.synchronize()
        whole = (time.perf_counter() - t0) * 1e3
        print(json.dumps({
            "metric": "ms per step of the busiest rank, SOAP-BPNN one box partitioned (ranks emulated one after the other on 1 GPU)",
            "value": max(per_rank), "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": False,
            "config": {"workload": f"ONE {args.atoms}-atom box cut into {W} slabs + 5 A halos",
                       "ms_per_rank": [round(t, 2) for t in per_rank], "atoms_per_rank": subs,
                       "whole_box_on_one_gpu_ms": round(whole, 2),
                       "projected_speedup_over_one_gpu": round(whole / max(per_rank), 2),
                       "not_included": "the all-reduce of [gradient | energy]"}}), flush=True)
        return

    def step():
        return partition.energy_and_gradient(model, posd, zd, cell, [True] * 3, world, rank, all_reduce=reduce)

    for _ in range(args.warmup):
        step()
    pdist.barrier(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e, grad, n_sub, n_owned = step()
    pdist.barrier(dev)
    elapsed = pdist.max_over_ranks(time.perf_counter() - t0, dev)
    n_sub_max = int(pdist.max_over_ranks(float(n_sub), dev))
    if rank == 0:
        assert torch.isfinite(grad).all()
        print(json.dumps({
            "metric": "atom-steps/sec (energy+forces) SOAP-BPNN, one box over all GPUs",
            "value": args.atoms * args.steps / elapsed, "unit": "atom-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic random periodic box, random weights (parity unpinned vs torch-spex)",
            "config": {"workload": f"SOAP-BPNN forward + dE/dR of ONE {args.atoms}-atom box, centres partitioned into "
                                   f"{world} slab(s) + 5 A halos, device neighbour list per step, one all-reduce of "
                                   f"[gradient | energy] ({(3 * args.atoms + 1) * 4 / 1e6:.1f} MB)",
                       "atoms_on_the_busiest_rank": n_sub_max, "atoms_owned_rank0": n_owned,
                       "total_energy": float(e)},
        }), flush=True)
    if world > 1:
        pdist.barrier(dev)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
