"""RCCL with more than one rank (``-m gpu``; skipped on a box with one GPU): the N > 1 path of ``bench.py`` and the
training step's ONE collective, run for real -- two processes, one per GPU, ``backend="nccl"`` (= RCCL on ROCm) over
xGMI, rendezvous on 127.0.0.1.

* the flat 11.6 MB gradient bucket after ``distributed.all_reduce_gradients`` equals, on both ranks, the mean of the
  two ranks' gradients -- and therefore (the loss is a fixed-weight sum over structures) half the 1-rank gradient of
  the concatenated batch, which rank 0 computes as well;
* ``bench.py --gpus 2`` started WITHOUT a launcher spawns its own ranks and prints one JSON line with n_gpus = 2.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _need_two_gpus():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the round-end driver runs it on the 8-GPU node)")


WORKER = r'''
import os, sys, json
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
sys.path.insert(0, sys.argv[1])
from metatrain_amd import distributed as D, runtime as rt, data
from metatrain_amd.pet import default_hypers
from metatrain_amd.pet.trainer import energy_loss_and_seeds, force_loss_and_seeds
from metatrain_amd.synthetic import random_box, synthetic_params

rank, local, world = D.env_rank()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
D.init("nccl", dev)
hypers = default_hypers()
params = synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
model = rt.HipModel(hypers, [1, 6, 7, 8]); model.load({k: v.to(dev) for k, v in params.items()}, "energy")

def flat_grad_of(box_ids, n_total):
    """flat gradient of loss = mean over ALL n_total boxes of (E/n)^2-type terms, restricted to `box_ids`"""
    boxes = [random_box(400, seed=50 + b) for b in box_ids]
    batch = data.collate([(p.to(dev), z.to(dev), c.to(dev), (True,) * 3) for p, z, c in boxes], hypers["cutoff"])
    g = data.graph_of(model, batch)
    fw = rt.HipForward(model, g, train=True)
    atomic = fw.forward()
    e = fw.sum_over_atoms(atomic)
    n_atoms = torch.full((len(box_ids),), 400.0, device=dev)
    gen = torch.Generator().manual_seed(3)
    e_t = torch.randn(n_total, generator=gen)[box_ids].to(dev)
    g_t = (0.3 * torch.randn(n_total, 400, 3, generator=gen))[box_ids].reshape(-1, 3).to(dev)
    ones = torch.ones_like(atomic)
    gpos = fw.backward(ones)
    _, seeds = energy_loss_and_seeds(e, e_t, n_atoms, g.system_of_atom())
    _, u = force_loss_and_seeds(gpos, g_t)
    model.zero_grad()
    fw.backward_train2(ones, seeds, u)
    return model.flat_grad().clone()

mine = flat_grad_of([2 * rank, 2 * rank + 1], 4)   # each rank: its own two boxes, per-rank mean (DDP semantics)
model.set_flat_grad(mine)
D.all_reduce_gradients(model)
reduced = model.flat_grad()
out = {"rank": rank, "norm": float(reduced.double().norm())}
if rank == 0:
    whole = flat_grad_of([0, 1, 2, 3], 4)          # 1-rank gradient of the concatenated batch (mean over 4 boxes)
    # per-rank losses are means over 2 boxes and DDP averages over ranks: equals the mean over all 4
    out["rel_err_vs_single_rank"] = float((reduced - whole).abs().max() / whole.abs().max())
gather = [None] * world
torch.distributed.all_gather_object(gather, out)
if rank == 0:
    print("RESULT " + json.dumps(gather), flush=True)
D.barrier(dev)
torch.distributed.destroy_process_group()
'''


def test_two_rank_rccl_gradient_all_reduce_equals_single_rank_gradient(tmp_path):
    _need_two_gpus()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script), ROOT]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    got = json.loads(line[len("RESULT "):])
    assert len(got) == 2 and abs(got[0]["norm"] - got[1]["norm"]) <= 1e-6 * got[0]["norm"]  # same bucket on both ranks
    assert got[0]["rel_err_vs_single_rank"] < 2e-5


def test_bench_spawns_its_own_ranks():
    _need_two_gpus()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--boxes", "2"], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["atoms_per_gpu_per_step"] == 20000
