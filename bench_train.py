"""Secondary benchmark (BASELINE.json configs[2] / configs[3]): one PET TRAINING step on synthetic boxes.

  python bench_train.py --gpus N --steps K --warmup W [--boxes 64 --atoms 1000]

One "step" = the body of the reference's training loop (pet/trainer.py:417-467) on one batch per GPU:
zero_grad -> forward -> dE/dR (create_graph in the reference) -> MSE(E/atom) + MSE(dE/dR) ->
loss.backward() (second order) -> [N > 1: mean all-reduce of the flat 11.6 MB gradient bucket over RCCL]
-> clip_grad_norm_(1.0) -> Adam -> weights re-packed. Neighbour lists are built before the clock starts.
bench.py (inference, energy + forces) stays the headline metric; this script prints ONE JSON line of the
same shape for the training row.
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


def cpu_baseline(hypers, params, n=1000):
    """The CPU oracle's training step (fp32, autograd double backward + torch Adam) on ONE n-atom box."""
    from oracle import nl as onl
    from oracle import pet as opet

    pos, z, cell = opet.random_box(n, seed=0)
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"])
    p = {k: (v if k == "species_to_species_index" else v.clone().requires_grad_(True)) for k, v in params.items()}
    leaves = [v for k, v in p.items() if k != "species_to_species_index"]
    opt = torch.optim.Adam(leaves, lr=1e-4)
    sysidx = torch.zeros(n, dtype=torch.long)
    nt = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(nt)

    def step():
        opt.zero_grad()
        r = pos.clone().requires_grad_(True)
        atomic = opet.pet_atomic_energies(p, hypers, r, cell[None], torch.tensor(i), torch.tensor(j),
                                          torch.tensor(s).long(), z, sysidx, "energy")[:, 0]
        e = atomic.sum()
        (g,) = torch.autograd.grad(e, r, create_graph=True)
        loss = (e / n) ** 2 + (g ** 2).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(leaves, 1.0)
        opt.step()

    step()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        step()
    dt = (time.perf_counter() - t0) / reps
    return {"value": n / dt, "unit": "atom-steps/s", "cores": nt, "kind": "port",
            "sample": f"{reps} x training step (fwd, dE/dR with create_graph, backward, clip, Adam) of one {n}-atom "
                      f"box, {dt:.2f} s each"}



TRAIN_STAGE_KERNELS = {"so_gemm": ("k_rowgemm_s", "k_rowgemm_n128", "k_rowgemm_k128", "k_gemm_h"), "wgrad": ("k_wgrad_b", "k_wgrad<"),
                       "so_attn_rev": ("k_attn_rev_p",), "so_attn_jvp": ("k_attn_jvp_p",), "emlp": ("k_emlp_p2", "k_emlp_s"),
                       "emlp_bwd": ("k_emlp_bwd_p2",), "attn_blk_bwd": ("k_ablk_bwd",), "attn_blk": ("k_ablk_fwd",)}


def pmc_traffic(stage, n_edges):
    """HBM bytes PER STEP of the stage group's kernels and of the whole step from the committed rocprofv3 PMC passes of this
    bench (profiles/r0N_train_traffic.json of the newest round: --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs, (2 FETCH + WRITE) x 1024);
    None unless the profiled run had the same number of edges."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_train_traffic.json") for r in (6, 5, 4)) if os.path.exists(q)),
                os.path.join(ROOT, "profiles", "r04_train_traffic.json"))
    if not os.path.exists(path):
        return None, None
    with open(path) as fh:
        data = json.load(fh)
    if data.get("workload_edges") != n_edges:
        return None, None
    steps = max(data.get("steps_profiled", 1), 1)
    recs = [v for k, v in data["kernels"].items() if k.startswith(TRAIN_STAGE_KERNELS.get(stage, ("-",)))]
    per_step = sum(v["hbm_bytes_per_launch"] * v["calls"] for v in recs) / steps if recs else None
    return per_step, data.get("step_hbm_bytes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--boxes", type=int, default=64, help="boxes per GPU per step")
    ap.add_argument("--atoms", type=int, default=1000, help="atoms per box")
    ap.add_argument("--micro", type=int, default=0,
                    help="boxes per micro-batch (gradient accumulation, TrainStep.microbatched); 0 = the whole batch at once. "
                         "BASELINE configs[3] per GPU: --boxes 64 --atoms 10000 --micro 8")
    ap.add_argument("--total-boxes", type=int, default=0,
                    help="strong-scaling mode: boxes per step in the WHOLE job, split over the ranks (0 = weak: --boxes "
                         "per GPU)")
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE configs[3]: 512 x 10 000-atom boxes per step in the whole job (64 per GPU at N = 8), "
                         "micro-batches of 4 = --total-boxes 512 --atoms 10000 --micro 4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs that are not `value` (single-term mode, two micro-batches): profiling runs")
    ap.add_argument("--no-two-micro", action="store_true",
                    help="skip the two-micro-batch leg (it allocates a second workspace); keeps the train_bf16 leg")
    ap.add_argument("--train-bf16", action="store_true",
                    help="pet_config_set('train_bf16', 1): ONE 16-bit MFMA term per product in the second-order and "
                         "weight-gradient GEMMs (BASELINE configs[2]'s 'bf16 MFMA MLPs'; gradients to ~1e-3, not the parity "
                         "mode; the default run reports it next to `value` as `train_bf16`)")
    ap.add_argument("--normalization", default="RMSNorm", choices=["RMSNorm", "LayerNorm"],
                    help="LayerNorm: the legacy-checkpoint norm")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=INT",
                    help="library switch for A/B runs (pet_config_set), as in bench.py")
    args = ap.parse_args()
    if args.config4:
        args.total_boxes, args.atoms = args.total_boxes or 512, 10000
        args.micro = args.micro or 4

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
    assert torch.cuda.is_available(), "bench_train.py needs MI355X GPUs"
    backend = os.environ.get("PET_BENCH_BACKEND", "nccl")  # "gloo": debugging aid, ranks may share a GPU (bench.py)
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        pdist.init(backend, dev)
        pdist.selftest(args.gpus, dev)  # the ranks the launch line names joined, and a collective works

    from metatrain_amd import runtime as rt
    from metatrain_amd.pet import default_hypers
    from metatrain_amd.pet.trainer import TrainStep
    from metatrain_amd.synthetic import random_box, synthetic_params

    hypers = dict(default_hypers(), normalization=args.normalization)
    params = synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
    model = rt.HipModel(hypers, [1, 6, 7, 8])
    model.load({k: v.to(dev) for k, v in params.items()}, "energy")

    for kv in args.set:
        rt.config_set(kv.split("=")[0], int(kv.split("=")[1]))
    if args.train_bf16:
        rt.config_set("train_bf16", 1)
    gen = torch.Generator().manual_seed(1234 + rank)
    micro = args.micro if args.micro > 0 else args.boxes
    strong = args.total_boxes > 0
    if strong:  # the global batch is fixed; rank r takes boxes r, r + world, ...
        if args.total_boxes % world:
            raise SystemExit(f"--total-boxes {args.total_boxes} is not a multiple of {world} ranks (DDP semantics: equal shards)")
        seeds = list(range(rank, args.total_boxes, world))
        args.boxes = len(seeds)
        micro = min(args.micro, args.boxes) if args.micro > 0 else args.boxes
    else:
        seeds = pdist.box_seeds(args.boxes, rank)

    def make_batches(micro):
        """The step's boxes in micro-batches of `micro`, and ONE training workspace, sized for the largest micro-batch, that
        all of them walk."""
        batches = []
        for m0 in range(0, args.boxes, micro):
            pos_l, z_l, cell_l, pair_l, sys_l = [], [], [], [], []
            for b, seed in enumerate(seeds[m0:m0 + micro]):
                pos, z, cell = random_box(args.atoms, seed=seed)
                posd = pos.to(dev)
                pairs, _ = rt.neighbor_list(posd, cell, [True] * 3, hypers["cutoff"])
                pairs = pairs.clone()
                pairs[:, 0:2] += b * args.atoms
                pos_l.append(posd); z_l.append(z.to(dev)); cell_l.append(cell.to(dev)); pair_l.append(pairs)
                sys_l.append(torch.full((args.atoms,), b, dtype=torch.int32, device=dev))
            nb = len(pos_l)
            pairs = torch.cat(pair_l)
            graph = rt.HipGraph(model, torch.cat(pos_l), torch.stack(cell_l), pairs[:, 0].contiguous(), pairs[:, 1].contiguous(),
                                pairs[:, 2:5].contiguous(), torch.cat(z_l), torch.cat(sys_l))
            per_box = torch.full((nb,), float(args.atoms), device=dev)
            batches.append(dict(graph=graph, target_energies=(torch.randn(nb, generator=gen) * 0.1).to(dev) * per_box,
                                n_atoms=per_box, target_gradients=(torch.randn(nb * args.atoms, 3, generator=gen) * 0.1).to(dev)))
        fw = rt.HipForward(model, max(batches, key=lambda b: b["graph"].n_edges)["graph"], train=True)
        for b in batches:
            b["fw"] = fw
        return batches, fw

    batches, fw = make_batches(micro)
    n_edges = sum(int(b["graph"].n_edges) for b in batches)
    n_atoms = args.boxes * args.atoms
    train = TrainStep(model, {"warmup_fraction": 0.0, "num_epochs": 10**6})
    comm_events = []

    def step(*_):
        return train.microbatched(batches) if len(batches) > 1 else train(
            batches[0]["graph"], fw, batches[0]["target_energies"], batches[0]["n_atoms"], batches[0]["target_gradients"])

    target_e = per_box = target_g = None
    graph = batches[0]["graph"]

    losses = []
    for _ in range(args.warmup):
        step(graph, fw, target_e, per_box, target_g)
    # per-stage durations of one untimed step (HIP events around every instrumented stage of the three sweeps): the stage
    # with the largest share is the roofline's kernel; the generic GEMMs of the second-order pass report FLOPs and bytes
    rt.profile(True)
    step(graph, fw, target_e, per_box, target_g)
    torch.cuda.synchronize()
    table = rt.profile_report()
    rt.profile(False)
    pdist.barrier(dev)
    train.comm_events = comm_events   # (start, end) events around each step's gradient all-reduce (N > 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step(graph, fw, target_e, per_box, target_g)["loss"])
    pdist.barrier(dev)
    elapsed = pdist.max_over_ranks(time.perf_counter() - t0, dev)
    train.comm_events = None
    one_term = None
    if world == 1 and not args.train_bf16 and not args.no_extras:  # not `value`: the same step in the single-term mode (a few more Adam steps)
        rt.config_set("train_bf16", 1)
        step(graph, fw, target_e, per_box, target_g)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            step(graph, fw, target_e, per_box, target_g)
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / 3
        rt.config_set("train_bf16", 0)
        one_term = {"value": n_atoms / dt1, "unit": "atom-steps/s", "ms_per_step": dt1 * 1e3,
                    "what": "pet_config_set('train_bf16', 1): ONE 16-bit MFMA term per product in the second-order and "
                            "weight-gradient GEMMs (gradients to ~1e-3: 20-step loss curve in tests/test_gpu_train.py); "
                            "forward and force pass unchanged"}
    two_micro = None
    workspace_gb = (fw.nbytes + fw.workspace2.numel()) / 1e9
    if world == 1 and len(batches) == 1 and args.boxes >= 2 and not args.no_extras and not args.train_bf16 and not args.no_two_micro:
        # not `value`: the same step (one Adam step over the same boxes) as TWO micro-batches with gradient accumulation
        # (TrainStep.microbatched) on a workspace half the size -- what a user short of memory runs
        gen.manual_seed(1234 + rank)  # (the whole-batch workspace stays allocated: 123 + 62 GB of the 288)
        mb, fw2 = make_batches((args.boxes + 1) // 2)
        train.microbatched(mb)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            train.microbatched(mb)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t1) / 3
        two_micro = {"value": n_atoms / dt2, "unit": "atom-steps/s", "ms_per_step": dt2 * 1e3,
                     "workspace_gb": (fw2.nbytes + fw2.workspace2.numel()) / 1e9,
                     "what": f"the same step as 2 micro-batches of {(args.boxes + 1) // 2} boxes (gradient accumulation, one Adam "
                             "step; --micro): the workspace is sized for one micro-batch"}
        del mb, fw2
    comm_ms = (sum(a.elapsed_time(b) for a, b in comm_events) / len(comm_events)) if comm_events else 0.0
    if rank == 0:
        ls = [float(x) for x in losses]
        assert all(l == l for l in ls), "training diverged to NaN"
        out = {
            "metric": "atom-steps/sec (training step: energy+forces loss, double backward, Adam) PET",
            "value": n_atoms * world * args.steps / elapsed,
            "unit": "atom-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32 (single-term 16-bit MFMA training GEMMs: --train-bf16)" if args.train_bf16 else "f32",
            "data": "synthetic random periodic boxes and random targets, weights from a seeded generator",
            "config": {
                "workload": f"PET training step, "
                            + (f"{args.total_boxes} x {args.atoms}-atom boxes per step in the whole job ({args.boxes} per GPU)"
                               if strong else f"{args.boxes} x {args.atoms}-atom boxes per GPU per step") +
                            f"{f' in micro-batches of {micro} (gradient accumulation)' if len(batches) > 1 else ''}, default PET "
                            f"hypers{'' if args.normalization == 'RMSNorm' else ' with normalization=' + args.normalization} (2.9M params), MSE(E/atom)+MSE(dE/dR), clip 1.0, Adam lr 1e-4",
                "atoms_per_gpu_per_step": n_atoms,
                "edges_per_gpu_per_step": n_edges,
                "micro_batches": len(batches),
                "parallelism": f"boxes sharded over {world} rank(s); one 11.6 MB gradient all-reduce per step"
                               if world > 1 else "single GPU",
                "gradient_all_reduce_ms_per_step": comm_ms,
                "gradient_all_reduce_backend": backend if world > 1 else None,
                "loss_first_last": [ls[0], ls[-1]],
                "workspace_gb": workspace_gb,
            },
        }
        # roofline: the stage group with the largest share of the step (instrumented stages only), its algorithmic FLOPs
        # against the pipe it runs on (f16x3: 2500 / 3 TFLOP/s fp32-equivalent) and its bytes against HBM; plus the whole
        # step against SURVEY 8(d)'s training FLOPs (6 x the forward's GEMM FLOPs)
        if table:
            dom = max(table, key=lambda r: r["total_ms"])
            tf = dom["flops"] / max(dom["total_ms"], 1e-9) / 1e9
            gb = dom["bytes"] / max(dom["total_ms"], 1e-9) / 1e6
            hbm_bound = gb / 8000.0 > tf / (2500.0 / 3.0)
            out["roofline"] = {
                "bound": "hbm" if hbm_bound else "mfma", "kernel": dom["name"],
                "achieved": gb if hbm_bound else tf, "peak": 8000.0 if hbm_bound else 2500.0 / 3.0,
                "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": (gb / 8000.0) if hbm_bound else tf / (2500.0 / 3.0),
                "stage_ms_per_step": round(dom["total_ms"], 3), "stage_calls_per_step": dom["calls"], "traffic": None,
                "frac_of_hbm_peak": gb / 8000.0, "frac_of_f16x3_mfma_peak": tf / (2500.0 / 3.0),
                "stages_ms": {r["name"]: round(r["total_ms"], 3) for r in sorted(table, key=lambda r: -r["total_ms"])[:10]},
            }
            # PMC bytes of the stage group per step (same unit as stage_ms_per_step) and of every kernel of the step
            out["roofline"]["traffic"], out["roofline"]["step_traffic_bytes"] = pmc_traffic(dom["name"], n_edges) \
                if len(batches) == 1 else (None, None)
            g0 = batches[0]["graph"]
            rowptr = g0.csr()["rowptr"].double()
            t2 = float((((rowptr[1:] - rowptr[:-1]) + 1) ** 2).sum())
            fwd = (2001152.0 * g0.n_edges + 4292864.0 * g0.n_nodes + 2048.0 * t2) * (n_edges / max(g0.n_edges, 1))
            out["roofline"]["whole_step_algorithmic_tflops"] = 6.0 * fwd / (elapsed / args.steps) / 1e12
        if one_term:
            out["train_bf16"] = one_term
        if two_micro:
            out["two_micro_batches"] = two_micro
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(hypers, params)
        print(json.dumps(out), flush=True)
    if world > 1:
        pdist.barrier(dev)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
