"""CPU tests of the single-box partition (metatrain_amd/soap_bpnn/partition.py): ownership is a
partition, every neighbour of an owned atom is in the rank's sub-system (checked against the oracle neighbour list),
and the one-exchange reduction gives the whole box's energy and gradient -- with a torch pair potential standing in
for the HIP model (same interaction range: one cutoff), single process and world_size 2 over gloo."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from metatrain_amd.soap_bpnn import partition
from oracle import nl as onl
from oracle import pet as opet

CUTOFF = 5.0


def _boxes():
    pos, z, cell = opet.random_box(600, seed=3)
    tri = cell.clone()
    tri[1, 0], tri[2, 0], tri[2, 1] = 4.0, -3.0, 5.0
    slab = torch.diag(torch.tensor([40.0, 12.0, 12.0]))
    return [
        ("cubic", pos, z, cell, [True] * 3),
        ("triclinic", pos @ torch.linalg.inv(cell) @ tri, z, tri, [True] * 3),
        ("mixed pbc", pos @ torch.linalg.inv(cell) @ slab, z, slab, [False, True, True]),
        ("open cluster", pos, z, torch.zeros(3, 3), [False] * 3),
        # metatomic's surface convention: zero lattice vector along the open direction; the longest direction is periodic
        ("surface, zero row", pos @ torch.linalg.inv(cell) @ slab, z, torch.diag(torch.tensor([40.0, 12.0, 0.0])),
         [True, True, False]),
    ]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ownership_is_a_partition_and_halos_hold_every_neighbour(world):
    for name, pos, z, cell, pbc in _boxes():
        i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cell.double().numpy(), pbc, CUTOFF)
        owner = np.full(len(pos), -1)
        for rank in range(world):
            index, owned, _ = partition.slab_partition(pos, cell, pbc, CUTOFF, world, rank)
            index, owned = index.numpy(), owned.numpy()
            assert np.all(np.diff(index) > 0), name
            assert np.all(owner[index[owned]] == -1), f"{name}: an atom owned twice"
            owner[index[owned]] = rank
            mine = np.isin(i, index[owned])
            assert np.all(np.isin(j[mine], index)), f"{name}: a neighbour of an owned atom is missing (rank {rank})"
        assert np.all(owner >= 0), f"{name}: an atom owned by nobody"


class _PairModel:
    """E_i = sum_j (1 - d_ij / rc)^3 z_j over the neighbours within rc: the same interaction range as SOAP-BPNN."""

    cutoff = CUTOFF

    def graph(self, pos, cells, centers, neighbors, shifts, z, sysidx):
        return {"pos": pos.double(), "cell": cells[0].double(), "i": centers.long(), "j": neighbors.long(),
                "s": shifts.double(), "z": z.double()}

    def _atomic(self, g, pos):
        d = (pos[g["j"]] - pos[g["i"]] + g["s"] @ g["cell"]).norm(dim=1)
        e = (1.0 - d / CUTOFF).clamp(min=0) ** 3 * g["z"][g["j"]]
        return torch.zeros(len(pos), dtype=torch.float64).index_add(0, g["i"], e)

    def forward(self, g):
        return self._atomic(g, g["pos"]).float()

    def backward(self, g, seeds):
        pos = g["pos"].clone().requires_grad_(True)
        (grad,) = torch.autograd.grad((self._atomic(g, pos) * seeds.double()).sum(), pos)
        return grad.float()


def _cpu_nl(pos, cell, pbc, cutoff):
    i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cell.double().numpy(), list(pbc), cutoff)
    return torch.tensor(np.concatenate([i[:, None], j[:, None], s], axis=1), dtype=torch.int32), None


def _whole(pos, z, cell, pbc):
    m = _PairModel()
    pairs, _ = _cpu_nl(pos, cell, pbc, CUTOFF)
    g = m.graph(pos, cell[None], pairs[:, 0], pairs[:, 1], pairs[:, 2:5], z, None)
    return m.forward(g).double().sum(), m.backward(g, torch.ones(len(pos)))


@pytest.mark.parametrize("world", [2, 5])
def test_partial_results_add_up_to_the_whole_box(world):
    for name, pos, z, cell, pbc in _boxes():
        e_ref, g_ref = _whole(pos, z, cell, pbc)
        e, grad, owned_total = 0.0, torch.zeros(len(pos), 3), 0
        for rank in range(world):
            er, gr, n_sub, n_owned = partition.energy_and_gradient(_PairModel(), pos, z, cell, pbc, world, rank,
                                                                   neighbor_list=_cpu_nl)
            e, grad, owned_total = e + float(er), grad + gr, owned_total + n_owned
            assert n_owned <= n_sub <= len(pos)
        assert owned_total == len(pos)
        assert abs(e - float(e_ref)) < 1e-5 * abs(float(e_ref)), name
        assert float((grad - g_ref).abs().max()) < 1e-5 * float(g_ref.abs().max()), name


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    name, pos, z, cell, pbc = _boxes()[1]
    e, grad, _, _ = partition.energy_and_gradient(_PairModel(), pos, z, cell, pbc, world, rank,
                                                  all_reduce=lambda t: dist.all_reduce(t), neighbor_list=_cpu_nl)
    if rank == 0:
        torch.save((e.clone(), grad.clone()), out)
    dist.destroy_process_group()


def test_one_all_reduce_over_two_ranks(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, 29500 + os.getpid() % 2000, out), nprocs=2, join=True)
    e, grad = torch.load(out)
    name, pos, z, cell, pbc = _boxes()[1]
    e_ref, g_ref = _whole(pos, z, cell, pbc)
    assert abs(float(e) - float(e_ref)) < 1e-5 * abs(float(e_ref))
    assert float((grad - g_ref).abs().max()) < 1e-5 * float(g_ref.abs().max())
