"""Headline benchmark: PET energy + forces on synthetic 10 000-atom periodic boxes.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (PETBackend.preprocess -> calculate_features ->
predict -> dE/dR, i.e. what the reference's ``mtt eval`` times at cli/eval.py:246-256)
over one batch of ``--boxes`` random periodic boxes per GPU (cubic,
rho = 0.05 / A^3, U[0,L)^3 positions, species uniform over {1,6,7,8}, default PET hypers,
fp32). The neighbour list is built once, before the clock starts, exactly like the
reference (CPU collate, outside its timed region); its GPU build time is reported
separately. Boxes are sharded over ranks with no data-path collective. Default: ``--boxes`` per
GPU per step whatever N is ("scaling": "weak"). ``--total-boxes B`` fixes the GLOBAL batch instead: the
B boxes of a step are split over the N ranks ("scaling": "strong"; north_star's 8-GPU target is a
strong-scaling one) and every rank walks its share in chunks of at most ``--boxes`` boxes that reuse one
activation workspace (64 boxes = 352 GB of activations do not fit one GPU at once).
Prints ONE JSON line on rank 0.
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

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
# The default GEMM stages compute fp32 products as three fp16 MFMA terms (f16x3): their FLOP/s
# are fp32-equivalent (2 M N K / time). `peak` stays the fp32 MFMA figure (the roof of the arithmetic the path
# delivers); the 16-bit pipe's own ceiling for this scheme, 2.5 PFLOP/s / 3 terms, is reported next to it.
MFMA_SPLIT_EQUIV_PEAK_TFLOPS = 2500.0 / 3.0
ATOMS_PER_BOX = 10000


def synthetic_params(hypers):
    """Weights of the reference state-dict schema from the documented per-key generator
    (random init; there are no checkpoints offline) -- the same weights the golden vectors use."""
    from metatrain_amd.synthetic import synthetic_params as gen

    return gen(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)


# ProfScope stage -> the kernels it launches (base names, template arguments stripped). The attention stages
# launch one kernel per neighbour-count bucket (NT = 1, 2, 3 ...): their traffic is the sum over the buckets.
STAGE_KERNELS = {"attn_blk": ("k_ablk_fwd", "k_ablk_fwd4"), "attn_blk_bwd": ("k_ablk_bwd",),
                 "attn_bwd": ("k_attn_bwd_a", "k_attn_bwd_l"), "attn_fwd": ("k_attn_fwd_p",),
                 "emlp": ("k_emlp_p2", "k_emlp_s"), "emlp_bwd": ("k_emlp_bwd_p2", "k_emlp_bwd_s"), "qkv": ("k_qkv_s",), "qkv_bwd": ("k_qkv_bwd_h",),
                 "comb": ("k_comb_p2", "k_comb_s"), "comb_bwd": ("k_comb_bwd_p2",)}


SPLIT_MFMA_STAGES = {"attn_blk", "attn_blk_bwd", "emlp", "emlp_bwd", "qkv", "qkv_bwd", "oproj", "oproj_bwd", "comb", "comb_bwd", "compress",
                 "compress_bwd", "head_edge", "head_edge_bwd", "node", "center", "head_node"}
ARITHMETIC = ("f32 results: every dense stage computes its fp32 products as three fp16 MFMA terms on 2-way split "
              "operands (f16x3, fp32 accumulate, 1.7e-7 product error vs fp64) -- the attention products of the fused "
              "per-atom block included; soft-max, norms and geometry in fp32")
SURVEY_8D_BYTES_PER_ATOM = 150e3  # forward + forces with activations recomputed in-tile


def _traffic_file():
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):   # newest round first
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as fh:
                return json.load(fh)
    return None


def pmc_traffic(stage, n_edges):
    """HBM bytes per launch of the stage's kernel from the committed rocprofv3 PMC passes
    (profiles/r0N_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this
    same bench, corrected as MI355X_MICROARCH.md prescribes). Only quoted when the profiled run
    had the same number of edges per launch; otherwise null."""
    data = _traffic_file()
    if not data or stage not in STAGE_KERNELS:
        return None
    if data.get("workload_edges") != n_edges:
        return None
    recs = [rec for name, rec in data["kernels"].items() if name.split("<")[0] in STAGE_KERNELS[stage]]
    if not recs:
        return None
    if stage.startswith("attn"):   # every bucket / tile-size kernel runs once per stage call
        return sum(r["hbm_bytes_per_launch"] for r in recs)
    calls = sum(r.get("calls", 1) for r in recs)   # template variants are alternatives (first / later GNN layer)
    return sum(r["hbm_bytes_per_launch"] * r.get("calls", 1) for r in recs) / calls


def pmc_mfma_busy(stage):
    """SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE of the stage's largest kernel from the committed SQ pass of this bench
    (profiles/r0N_sq_counters_table.txt of the newest round, column mfma_util), or None."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_sq_counters_table.txt") for r in (6, 5, 4)) if os.path.exists(q)), None)
    if stage not in STAGE_KERNELS or path is None:
        return None
    best = None
    with open(path) as fh:
        for line in fh:
            if line.split("<")[0].strip() in STAGE_KERNELS[stage]:
                parts = line.split()
                try:
                    us, util = float(parts[-7]), float(parts[-6])
                except (ValueError, IndexError):
                    continue
                if best is None or us > best[0]:
                    best = (us, util)
    return best[1] if best else None


def pmc_step_traffic(n_edges):
    """Total HBM bytes of one step over ALL kernels from the committed PMC passes, or None (other workload)."""
    data = _traffic_file()
    if not data or data.get("workload_edges") != n_edges:
        return None
    return data.get("step_hbm_bytes")


def cpu_baseline(hypers, params, seconds_budget=12.0):
    """The CPU oracle (a torch-CPU restatement of the reference path, kind="port") timed on this host on a bounded
    sample of the same workload: forward + dE/dR of ONE 10 000-atom box (the metric's box size; one repeat after
    the thread count was chosen on a 1000-atom box, BASELINE config 2's shape, whose rate is reported beside it)."""
    from oracle import nl as onl
    from oracle import pet as opet

    def box_args(n):
        pos, z, cell = opet.random_box(n, seed=0)
        i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"])
        return (params, hypers, pos, cell[None], torch.tensor(i), torch.tensor(j), torch.tensor(s).long(), z,
                torch.zeros(n, dtype=torch.long))

    small = box_args(1000)
    # pick the thread count that serves this host best (many-core hosts oversubscribe on the
    # small per-bucket ops); report the count actually used for the quoted number
    ncpu = os.cpu_count() or 1
    best = None
    for nt in sorted({min(ncpu, t) for t in (8, 16, 32)}):
        torch.set_num_threads(nt)
        opet.energy_and_gradient(*small)  # warm-up
        t0 = time.perf_counter()
        reps = 0
        while reps < 2 or (time.perf_counter() - t0 < seconds_budget / 3 and reps < 6):
            opet.energy_and_gradient(*small)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        if best is None or dt < best[0]:
            best = (dt, nt, reps)
    dt1k, nt, reps1k = best
    torch.set_num_threads(nt)
    big = box_args(ATOMS_PER_BOX)
    t0 = time.perf_counter()
    opet.energy_and_gradient(*big)
    dt = time.perf_counter() - t0
    return {
        "value": ATOMS_PER_BOX / dt,
        "unit": "atom-steps/s",
        "cores": nt,
        "kind": "port",
        "sample": f"1 x (forward + dE/dR) of one {ATOMS_PER_BOX}-atom box (rho=0.05/A^3, 4.5 A cutoff, fp32, default "
                  f"hypers), {dt:.1f} s; NL excluded",
        "box1000": {"value": 1000 / dt1k, "unit": "atom-steps/s",
                    "sample": f"{reps1k} x one 1000-atom box (BASELINE config 2 shape), {dt1k:.3f} s each"},
    }


def gpu_box1000(model, hypers, dev, reps=200):
    """BASELINE configs[1] on the GPU (not `value`): ONE 1000-atom box per step -- graph build + forward + dE/dR from a
    resident neighbour list, and the same step with the device neighbour list rebuilt every step (the MD regime). The CPU
    oracle's rate on this box is `cpu_baseline.box1000`."""
    from metatrain_amd import runtime as rt
    from metatrain_amd.synthetic import random_box
    pos, z, cell = random_box(1000, 0)
    pos, z, cell = pos.to(dev), z.to(dev), cell.to(dev)
    sysidx = torch.zeros(1000, dtype=torch.int32, device=dev)
    ones = torch.ones(1000, device=dev)
    st = {}

    def step(with_nl):
        if with_nl or "pairs" not in st:
            pr, _ = rt.neighbor_list(pos, cell, [True] * 3, hypers["cutoff"])
            st["pairs"] = (pr[:, 0].contiguous(), pr[:, 1].contiguous(), pr[:, 2:5].contiguous())
        i, j, s = st["pairs"]
        g = rt.HipGraph(model, pos, cell[None], i, j, s, z, sysidx)
        fw = st["fw"].rebind(g) if "fw" in st else st.setdefault("fw", rt.HipForward(model, g))
        fw.forward()
        return fw.backward(ones)

    res = {}
    for key, with_nl in (("graph+forward+backward_ms", False), ("nl+graph+forward+backward_ms", True)):
        for _ in range(20):
            step(with_nl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step(with_nl)
        torch.cuda.synchronize()
        res[key] = (time.perf_counter() - t0) / reps * 1e3
    res["value"] = 1000.0 / (res["graph+forward+backward_ms"] * 1e-3)
    res["unit"] = "atom-steps/s"
    res["sample"] = f"{reps} x one 1000-atom box (BASELINE config 2 shape: rho=0.05/A^3, 4.5 A cutoff), one box per step"
    return res


def _child_bench(script, argv, timeout=600):
    """Run one of the sibling benches in a child process and return its JSON line (or the reason there is none)."""
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, script)] + argv
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": f"{script} timed out after {timeout} s"}
    for line in reversed(res.stdout.strip().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return {"error": f"{script} exit {res.returncode}: {res.stderr.strip()[-400:]}"}


def leg_train64x1000(no_cpu):
    """BASELINE configs[2]: PET training step (forward, dE/dR with its double backward, clip, Adam) on 64 x 1000-atom boxes --
    `bench_train.py`'s own line, reduced; its parity mode is `value`, the single-term 16-bit mode is `train_bf16`."""
    r = _child_bench("bench_train.py", ["--steps", "5", "--warmup", "2", "--no-two-micro"] + (["--no-cpu-baseline"] if no_cpu else []))
    if "error" in r:
        return r
    roof = r.get("roofline", {})
    out = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
           "workspace_gb": r["config"]["workspace_gb"], "workload": r["config"]["workload"],
           "train_bf16": {k: r["train_bf16"][k] for k in ("value", "ms_per_step")} if "train_bf16" in r else None,
           "roofline": {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "stage_ms_per_step",
                                                 "whole_step_algorithmic_tflops", "step_traffic_bytes")}}
    if "cpu_baseline" in r:
        out["cpu_baseline"] = r["cpu_baseline"]
    return out


def leg_soap100k(no_cpu):
    """BASELINE configs[4] on one GPU: SOAP-BPNN forward + dE/dR of one 100 000-atom box -- `bench_soap.py`'s own line, reduced."""
    r = _child_bench("bench_soap.py", ["--steps", "10", "--warmup", "3"] + (["--no-cpu-baseline"] if no_cpu else []))
    if "error" in r:
        return r
    roof = r.get("roofline", {})
    out = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
           "workload": r["config"]["workload"],
           "roofline": {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                                 "stages_ms", "step_traffic_bytes", "soap_tail_hbm_frac")}}
    if "cpu_baseline" in r:
        out["cpu_baseline"] = r["cpu_baseline"]
    return out


def respawn_under_launcher(n_gpus):
    """`python bench.py --gpus N` without a launcher: start N ranks of this same script under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1) and hand its exit code back."""
    import socket
    import subprocess

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--boxes", type=int, default=8,
                    help="10k-atom boxes per GPU per step (larger batches amortise wave quantisation and launch gaps; DESIGN.md 5); "
                         "with --total-boxes: the chunk size a rank walks its share in")
    ap.add_argument("--total-boxes", type=int, default=0,
                    help="strong-scaling mode: this many boxes per step in the WHOLE job, split over the ranks and "
                         "walked in chunks of --boxes (0 = weak scaling, --boxes per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs that are not `value` (box1000, host_resident_inputs): what the rocprofv3 passes run")
    ap.add_argument("--profile-all", action="store_true", help="print a per-stage table to stderr")
    ap.add_argument("--normalization", default="RMSNorm", choices=["RMSNorm", "LayerNorm"],
                    help="side runs only: the legacy-checkpoint norm; the headline metric is the default (RMSNorm)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=INT",
                    help="library switch for A/B runs, e.g. --set side_stream=0 (pet_config_set)")
    args = ap.parse_args()

    from metatrain_amd import distributed as pdist

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_launcher(args.gpus)
    rank, local_rank, world = pdist.env_rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # PET_BENCH_BACKEND=gloo: a debugging aid for boxes with fewer GPUs than ranks (the ranks then share devices and
    # the barrier / max-over-ranks go through gloo); the measured configuration is one rank per GPU over RCCL
    backend = os.environ.get("PET_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        pdist.init(backend, dev)  # "nccl" is RCCL on ROCm
        pdist.selftest(args.gpus, dev)  # the ranks the launch line names joined, and a collective works

    from metatrain_amd import runtime as rt
    from metatrain_amd.pet import default_hypers
    from metatrain_amd.synthetic import random_box

    side_stream = 1   # the production configuration (csrc default); `--set side_stream=0` keeps the step on one stream
    for kv in args.set:
        if kv.split("=")[0] == "side_stream":
            side_stream = int(kv.split("=")[1])
        key, val = kv.split("=")
        rt.config_set(key, int(val))
    hypers = dict(default_hypers(), normalization=args.normalization)
    params = synthetic_params(hypers)
    model = rt.HipModel(hypers, [1, 6, 7, 8])
    model.load({k: v.to(dev) for k, v in params.items()}, "energy")

    # ---- inputs resident in HBM before the clock starts ---------------------------------
    strong = args.total_boxes > 0
    if strong:  # rank r takes boxes r, r + world, ... of the global batch
        my_ids = list(range(rank, args.total_boxes, world))
        if args.total_boxes < world:
            raise SystemExit(f"--total-boxes {args.total_boxes} < {world} ranks")
    else:
        my_ids = pdist.box_seeds(args.boxes, rank)
    chunks = []   # per chunk of <= --boxes boxes: the concatenated inputs, resident in HBM
    nl_ms = 0.0
    pos_l, z_l, cell_l = [], [], []
    for c0 in range(0, len(my_ids), args.boxes):
        ids = my_ids[c0:c0 + args.boxes]
        cp, cz, cc, cpair, csys = [], [], [], [], []
        for b, seed in enumerate(ids):
            pos, z, cell = random_box(ATOMS_PER_BOX, seed=seed)
            posd = pos.to(dev)
            rt.neighbor_list(posd[:64].contiguous(), cell, [True] * 3, hypers["cutoff"])  # warm the kernels
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pairs, _ = rt.neighbor_list(posd, cell, [True] * 3, hypers["cutoff"])
            torch.cuda.synchronize()
            nl_ms += (time.perf_counter() - t0) * 1e3
            pairs = pairs.clone()
            pairs[:, 0:2] += b * ATOMS_PER_BOX
            cp.append(posd); cz.append(z.to(dev)); cc.append(cell.to(dev)); cpair.append(pairs)
            csys.append(torch.full((ATOMS_PER_BOX,), b, dtype=torch.int32, device=dev))
        if c0 == 0:
            pos_l, z_l, cell_l = cp, cz, cc
        pairs = torch.cat(cpair)
        chunks.append({"positions": torch.cat(cp), "species": torch.cat(cz), "cells": torch.stack(cc),
                       "centers": pairs[:, 0].contiguous(), "neighbors": pairs[:, 1].contiguous(),
                       "shifts": pairs[:, 2:5].contiguous(), "sysidx": torch.cat(csys),
                       "ones": torch.ones(len(ids) * ATOMS_PER_BOX, dtype=torch.float32, device=dev)})
    boxes = len(chunks[0]["ones"]) // ATOMS_PER_BOX   # boxes of the first (largest) chunk
    n_atoms = len(my_ids) * ATOMS_PER_BOX             # atoms this rank processes per step
    ones = chunks[0]["ones"]

    state = {}

    def step():
        out = None
        for ch in chunks:
            graph = rt.HipGraph(model, ch["positions"], ch["cells"], ch["centers"], ch["neighbors"], ch["shifts"],
                                ch["species"], ch["sysidx"])
            fw = state.get("fw")
            try:   # activation workspace: allocated once (for the largest chunk seen), reused by every chunk and step
                fw = fw.rebind(graph) if fw is not None else rt.HipForward(model, graph)
            except rt.PetHipError:
                state["fw"] = fw = None
                fw = rt.HipForward(model, graph)
            state["fw"] = fw
            atomic = fw.forward()
            grad = fw.backward(ch["ones"])
            out = out or (atomic, grad, graph)
        return out

    def barrier():
        pdist.barrier(dev)

    for _ in range(args.warmup):
        step()
    # ---- find the dominant stage: one untimed step on a single stream (clean per-stage durations),
    #      then time K steps in the production configuration with HIP events on that stage only ----
    rt.config_set("side_stream", 0)
    rt.profile(True)
    step()
    torch.cuda.synchronize()
    table = rt.profile_report()
    rt.profile(False)
    rt.config_set("side_stream", side_stream)
    dominant = max(table, key=lambda r: r["total_ms"])["name"]
    if args.profile_all and rank == 0:
        for r in sorted(table, key=lambda r: -r["total_ms"]):
            tf = r["flops"] / max(r["total_ms"], 1e-9) / 1e9
            gb = r["bytes"] / max(r["total_ms"], 1e-9) / 1e6
            print(f"  {r['name']:16s} {r['total_ms']:8.3f} ms x{r['calls']:<3d} {tf:8.1f} TFLOP/s {gb:8.0f} GB/s(alg)",
                  file=sys.stderr)
    step()  # back on two streams before the clock starts

    # `value`: K steps with NO instrumentation in the timed region ...
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        atomic, grad, graph = step()
    barrier()
    elapsed = time.perf_counter() - t0
    # ... and the dominant stage's launch durations in a SEPARATE pass of the same steps in the same (two-stream)
    # configuration, HIP events on the stream the stage is launched on
    rt.profile(True, stage=dominant)
    for _ in range(min(args.steps, 10)):
        step()
    torch.cuda.synchronize()
    dom = [r for r in rt.profile_report() if r["name"] == dominant][0]
    rt.profile(False)

    elapsed = pdist.max_over_ranks(elapsed, dev)
    total_atoms = args.total_boxes * ATOMS_PER_BOX if strong else n_atoms * world
    ms_per_step = elapsed / args.steps * 1e3
    value = total_atoms * args.steps / elapsed

    if rank == 0:
        e_total = float(atomic.double().sum())
        assert torch.isfinite(grad).all()
        avg_ms = dom["total_ms"] / dom["calls"]
        flops_per_launch = dom["flops"] / dom["calls"]
        bytes_per_launch = dom["bytes"] / dom["calls"]
        # which roof bounds this stage: algorithmic bytes at HBM peak vs algorithmic FLOPs at the peak of the matrix pipe
        # the stage RUNS on -- the fp32 MFMA for attention, the 16-bit MFMA at three terms per product (f16x3:
        # 2500 / 3 TFLOP/s fp32-equivalent) for the GEMM stages
        split = dominant in SPLIT_MFMA_STAGES
        mfma_peak = MFMA_SPLIT_EQUIV_PEAK_TFLOPS if split else MFMA_F32_PEAK_TFLOPS
        hbm_bound = bytes_per_launch / (HBM_PEAK_GBS * 1e9) > flops_per_launch / (mfma_peak * 1e12)
        if hbm_bound:
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "avg_launch_ms": avg_ms,
                    "algorithmic_bytes_per_launch": bytes_per_launch, "traffic": None}
        else:
            achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": dominant, "achieved": achieved, "peak": mfma_peak,
                    "unit": "TFLOP/s", "frac": achieved / mfma_peak, "avg_launch_ms": avg_ms,
                    "algorithmic_flops_per_launch": flops_per_launch, "traffic": None}
        roof["traffic"] = pmc_traffic(dominant, int(graph.n_edges))
        roof["mfma_busy_pmc"] = pmc_mfma_busy(dominant)  # what the matrix pipe was busy with, recompute and padding included
        roof["arithmetic"] = ("fp32 MFMA soft-max attention kernel; the surrounding projections: " + ARITHMETIC
                              if dominant in ("attn_fwd", "attn_bwd") else ARITHMETIC)
        if split:  # both roofs of a split-operand GEMM stage, whichever one "bound" names
            tf = flops_per_launch / (avg_ms * 1e-3) / 1e12
            roof["fp32_equivalent_tflops"] = tf
            roof["frac_of_f16x3_mfma_peak"] = tf / MFMA_SPLIT_EQUIV_PEAK_TFLOPS
            roof["frac_of_hbm_peak"] = bytes_per_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof["whole_step_algorithmic_tflops"] = None
        roof["stages_single_stream_ms"] = {r["name"]: round(r["total_ms"], 3)
                                           for r in sorted(table, key=lambda r: -r["total_ms"])[:8]}
        # both roofs of every large stage, from the untimed single-stream step (per launch; the matrix roof is the one of
        # the pipe the stage runs on): the dominant stage changes hands between attn_bwd and emlp_bwd from run to run
        roof["stage_roofs_single_stream"] = {
            r["name"]: {
                "ms_per_launch": round(r["total_ms"] / r["calls"], 4),
                "frac_of_hbm_peak": round(r["bytes"] / (r["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if r["bytes"] else None,
                "frac_of_mfma_peak": round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12 /
                                           (MFMA_SPLIT_EQUIV_PEAK_TFLOPS if r["name"] in SPLIT_MFMA_STAGES
                                            else MFMA_F32_PEAK_TFLOPS), 4) if r["flops"] else None,
            } for r in sorted(table, key=lambda r: -r["total_ms"])[:8]}
        out = {
            "metric": "atom-steps/sec (energy+forces) PET 10k-atom box",
            "value": value,
            "unit": "atom-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic random periodic boxes (rho=0.05/A^3, 4 species), weights from a seeded generator",
            "config": {
                "workload": f"PET forward + dE/dR (preprocess+features+predict+backward), "
                            + (f"{args.total_boxes} x {ATOMS_PER_BOX}-atom boxes per step in the whole job "
                               f"({len(my_ids)} per GPU, chunks of {boxes})" if strong else
                               f"{boxes} x {ATOMS_PER_BOX}-atom boxes per GPU per step") + ", default PET hypers"
                            f"{'' if args.normalization == 'RMSNorm' else ' with normalization=' + args.normalization} (2.9M params), "
                            f"4.5 A cutoff, {graph.n_edges // boxes} edges/box",
                "atoms_per_gpu_per_step": n_atoms,
                "edges_per_gpu_per_step": int(graph.n_edges),
                "parallelism": f"boxes sharded over {world} rank(s), no data-path collective",
                "chunks_per_step_per_gpu": len(chunks),
                "arithmetic": ARITHMETIC,
                "neighbor_list_gpu_ms_per_box": nl_ms / boxes,
                "total_energy_rank0": e_total,
            },
            "roofline": roof,
        }
        # whole-step view: SURVEY §8(d) algorithmic GEMM FLOPs, forward x2 for forces
        # (`graph` is the step's first chunk; the other chunks are boxes of the same size and density)
        chunk_atoms = boxes * ATOMS_PER_BOX
        e, n = graph.n_edges, chunk_atoms
        rowptr = graph.csr()["rowptr"].double()
        t2 = float((((rowptr[1:] - rowptr[:-1]) + 1) ** 2).sum())
        fwd_flops = (2001152.0 * e + 4292864.0 * n + 2048.0 * t2) * (n_atoms / chunk_atoms)
        out["roofline"]["whole_step_algorithmic_tflops"] = 2 * fwd_flops / (ms_per_step * 1e-3) / 1e12
        # step-level traffic: what this design moves per step (sum of the stages' own byte counts; the PMC total
        # of the committed profile when it is the same workload) against SURVEY 8(d)'s 150 KB/atom
        step_alg = SURVEY_8D_BYTES_PER_ATOM * n_atoms
        design = sum(r["bytes"] for r in table)   # the untimed profiled step walks every chunk
        pmc_step = pmc_step_traffic(int(graph.n_edges)) if len(chunks) == 1 else None
        out["roofline"]["step_traffic"] = {
            "survey_8d_algorithmic_bytes": step_alg, "design_bytes_counted_by_stages": design or None,
            "pmc_bytes": pmc_step, "ratio_to_survey_8d": (pmc_step or design or 0.0) / step_alg or None,
            "hbm_frac_whole_step": (pmc_step or design or 0.0) / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9) or None}
        if world == 1 and not strong and not args.no_extras:
            # not `value`: the same step started from HOST-resident systems (what an MD driver or a DataLoader hands
            # over) -- H2D of positions / species / cells, device neighbour lists + collate (metatrain_amd.data), graph
            # build, forward, dE/dR, D2H of per-atom energies and gradients. DESIGN.md section 5 quotes it.
            from metatrain_amd import data as pdata
            host = [(pos_l[b].cpu().pin_memory(), z_l[b].cpu().pin_memory(), cell_l[b].cpu(), [True] * 3)
                    for b in range(boxes)]

            def host_step():
                systems = [(p.to(dev, non_blocking=True), z.to(dev, non_blocking=True), c.to(dev), pbc)
                           for p, z, c, pbc in host]
                batch = pdata.collate(systems, hypers["cutoff"])
                g = pdata.graph_of(model, batch)
                try:
                    fw = state["fw"].rebind(g)
                except rt.PetHipError:
                    fw = state["fw"] = rt.HipForward(model, g)
                a = fw.forward()
                return a.cpu(), fw.backward(ones).cpu()

            host_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                host_step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            out["host_resident_inputs"] = {
                "value": n_atoms / dt, "unit": "atom-steps/s", "ms_per_step": dt * 1e3,
                "includes": "H2D positions/species/cells (pinned), device neighbour lists + collate, graph build, "
                            "forward, dE/dR, D2H per-atom energies + gradients"}
        if world == 1 and not strong and not args.no_extras:
            out["box1000"] = gpu_box1000(model, hypers, dev)
        if world == 1 and not strong and not args.no_extras:
            # BASELINE configs[2] and configs[4] on this GPU, from their own benches (bench_train.py / bench_soap.py, each a
            # child process started once this process has released its workspaces): not `value`
            state.clear()
            del atomic, grad, graph
            torch.cuda.empty_cache()
            out["train64x1000"] = leg_train64x1000(args.no_cpu_baseline)
            out["soap100k"] = leg_soap100k(args.no_cpu_baseline)
        if not args.no_cpu_baseline and world == 1:  # the reported CPU leg runs at N = 1 only
            out["cpu_baseline"] = cpu_baseline(hypers, params)
        print(json.dumps(out), flush=True)
    if world > 1:
        pdist.barrier(dev)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
