"""CPU tests of the single-box PET partition (metatrain_amd/pet/partition.py; SURVEY §8(e): PET's interaction range is
num_gnn_layers cutoffs, one more with the adaptive cutoff): a torch message-passing toy with exactly PET's dependence
structure stands in for the HIP runtime -- G rounds of `h_i <- sum_j w_ij (h_j + 1)` over the neighbour list and one
more gather of the neighbours' final features (the reversed-edge term of the last combination), optionally
with pair weights that read both atoms' own neighbour counts (what the symmetrised adaptive cutoffs do) -- and the
one-exchange reduction must give the whole box's energy and gradient, single process and world_size 2 over gloo. With a
halo one hop too short the same check must FAIL, i.e. the test can tell."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from metatrain_amd import partition as generic
from metatrain_amd.pet import partition
from oracle import nl as onl
from oracle import pet as opet

CUTOFF = 2.5


class _ToyModel:
    def __init__(self, layers, adaptive):
        self.hypers = {"cutoff": CUTOFF, "num_gnn_layers": layers, "num_neighbors_adaptive": 4.0 if adaptive else None}


class _ToyRuntime:
    """Stand-in for metatrain_amd.runtime: HipGraph / HipForward / neighbor_list with the same call signatures."""

    @staticmethod
    def neighbor_list(pos, cell, pbc, cutoff):
        i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cell.double().numpy(), list(pbc), cutoff)
        return torch.tensor(np.concatenate([i[:, None], j[:, None], s], axis=1), dtype=torch.int32), None

    class HipGraph:
        def __init__(self, model, pos, cells, centers, neighbors, shifts, z, sysidx):
            self.model, self.pos, self.cell = model, pos.double(), cells[0].double()
            self.i, self.j, self.s, self.z = centers.long(), neighbors.long(), shifts.double(), z.double()

    class HipForward:
        def __init__(self, model, graph):
            self.g = graph

        def _atomic(self, pos):
            g = self.g
            n = len(pos)
            d = (pos[g.j] - pos[g.i] + g.s @ g.cell).norm(dim=1)
            w = (1.0 - d / CUTOFF).clamp(min=0)  # linear: the last hop must stay visible to the checks below
            if g.model.hypers["num_neighbors_adaptive"] is not None:
                count = torch.zeros(n, dtype=torch.float64).index_add(0, g.i, w)
                w = w * 0.5 * (torch.tanh(count[g.i]) + torch.tanh(count[g.j]))
            h = g.z / 8.0
            for _ in range(g.model.hypers["num_gnn_layers"]):
                h = torch.zeros(n, dtype=torch.float64).index_add(0, g.i, w * (h[g.j] + 1.0))
            # the last layer's features also gather the reversed edges' outputs (backend.py:559-575): one more hop
            return h + torch.zeros(n, dtype=torch.float64).index_add(0, g.i, w * torch.tanh(h[g.j]))

        def forward(self):
            return self._atomic(self.g.pos).float()

        def backward(self, seeds):
            pos = self.g.pos.clone().requires_grad_(True)
            (grad,) = torch.autograd.grad((self._atomic(pos) * seeds.double()).sum(), pos)
            return grad.float()


def _boxes():
    pos, z, cell = opet.random_box(700, seed=5)
    long_cell = torch.diag(torch.tensor([60.0, 14.0, 16.0]))
    tri = long_cell.clone()
    tri[1, 0], tri[2, 0], tri[2, 1] = 6.0, -4.0, 3.0
    frac = pos @ torch.linalg.inv(cell)
    return [
        ("long cell", frac @ long_cell, z, long_cell, [True] * 3),
        ("triclinic", frac @ tri, z, tri, [True] * 3),
        ("mixed pbc", frac @ long_cell, z, long_cell, [False, True, True]),
        # metatomic convention for a surface: ZERO lattice vector along the open direction (such a cell
        # was treated as fully open and cut along a periodic axis without a wrap-around halo)
        ("surface, zero row", frac @ long_cell, z, torch.diag(torch.tensor([60.0, 14.0, 0.0])), [True, True, False]),
        ("wire, two zero rows", frac @ long_cell, z, torch.diag(torch.tensor([60.0, 0.0, 0.0])), [True, False, False]),
    ]


def _whole(model, pos, z, cell, pbc):
    pairs, _ = _ToyRuntime.neighbor_list(pos, cell, pbc, CUTOFF)
    g = _ToyRuntime.HipGraph(model, pos, cell[None], pairs[:, 0], pairs[:, 1], pairs[:, 2:5], z, None)
    fw = _ToyRuntime.HipForward(model, g)
    return fw.forward().double().sum(), fw.backward(torch.ones(len(pos)))


def _summed(model, pos, z, cell, pbc, world):
    e, grad, owned_total, sub_max = 0.0, torch.zeros(len(pos), 3), 0, 0
    for rank in range(world):
        er, gr, n_sub, n_owned = partition.energy_and_gradient(model, pos, z, cell, pbc, world, rank, runtime=_ToyRuntime)
        e, grad, owned_total, sub_max = e + float(er), grad + gr, owned_total + n_owned, max(sub_max, n_sub)
    return e, grad, owned_total, sub_max


@pytest.mark.parametrize("layers,adaptive", [(1, False), (2, False), (3, False), (2, True)])
@pytest.mark.parametrize("world", [2, 3])
def test_partial_results_add_up_to_the_whole_box(layers, adaptive, world):
    model = _ToyModel(layers, adaptive)
    for name, pos, z, cell, pbc in _boxes():
        e_ref, g_ref = _whole(model, pos, z, cell, pbc)
        e, grad, owned_total, sub_max = _summed(model, pos, z, cell, pbc, world)
        assert owned_total == len(pos)
        if world == 2 and layers <= 2 and not adaptive:
            assert sub_max < 0.95 * len(pos), name  # 30 A slabs + two halos of at most 7.5 A in a 60 A cell
        assert abs(e - float(e_ref)) < 1e-5 * abs(float(e_ref)), name
        assert float((grad - g_ref).abs().max()) < 1e-5 * float(g_ref.abs().max()), name


def test_a_halo_one_hop_short_is_detected():
    """The check above has teeth: the same sum with a halo one hop short (the range the reference declares,
    num_gnn_layers cutoffs) differs from the whole box."""
    deep = _ToyModel(1, False)
    name, pos, z, cell, pbc = _boxes()[0]
    e_ref, g_ref = _whole(deep, pos, z, cell, pbc)
    grad = torch.zeros(len(pos), 3)
    for rank in range(2):
        index, owned, _ = generic.slab_partition(pos, cell, pbc, CUTOFF * deep.hypers["num_gnn_layers"], 2, rank)
        pairs, _ = _ToyRuntime.neighbor_list(pos[index], cell, pbc, CUTOFF)
        g = _ToyRuntime.HipGraph(deep, pos[index], cell[None], pairs[:, 0], pairs[:, 1], pairs[:, 2:5], z[index], None)
        grad[index] += _ToyRuntime.HipForward(deep, g).backward(owned.float())
    assert float((grad - g_ref).abs().max()) > 1e-4 * float(g_ref.abs().max())  # ten times the pass bar


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    name, pos, z, cell, pbc = _boxes()[1]
    e, grad, _, _ = partition.energy_and_gradient(_ToyModel(2, True), pos, z, cell, pbc, world, rank,
                                                  all_reduce=lambda t: dist.all_reduce(t), runtime=_ToyRuntime)
    if rank == 0:
        torch.save((e.clone(), grad.clone()), out)
    dist.destroy_process_group()


def test_one_all_reduce_over_two_ranks(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, 31500 + os.getpid() % 2000, out), nprocs=2, join=True)
    e, grad = torch.load(out)
    name, pos, z, cell, pbc = _boxes()[1]
    e_ref, g_ref = _whole(_ToyModel(2, True), pos, z, cell, pbc)
    assert abs(float(e) - float(e_ref)) < 1e-5 * abs(float(e_ref))
    assert float((grad - g_ref).abs().max()) < 1e-5 * float(g_ref.abs().max())


class _CsrGraph:
    """What ExchangePlan reads of a HipGraph: the CSR rows (sorted by centre) of the kept pairs."""

    def __init__(self, pairs):
        order = torch.argsort(pairs[:, 0].long(), stable=True)
        self.ctr, self.nbr = pairs[order, 0].int(), pairs[order, 1].int()

    def csr(self):
        return {"ctr": self.ctr, "nbr": self.nbr}


@pytest.mark.parametrize("world", [2, 3, 4])
def test_exchange_plans_of_the_ranks_agree_without_negotiation(world):
    """Per-layer exchange (energy_and_gradient_exchange): what rank a exports to rank b is, row for row and in the same
    order, what rank b holds as ghost rows from a; every edge of the box is computed by exactly one rank."""
    gen = torch.Generator().manual_seed(3)
    cell = torch.diag(torch.tensor([30.0, 11.0, 12.0]))
    cell[1, 0] = 3.0
    n = 700
    pos = torch.rand(n, 3, generator=gen) @ cell
    pbc = [True] * 3
    owner = generic.slab_owner(pos, cell, pbc, world)
    whole, _ = _ToyRuntime.neighbor_list(pos, cell, pbc, CUTOFF)
    plans, keys, own_edges = [], [], 0
    for rank in range(world):
        index, owned, _ = generic.slab_partition(pos, cell, pbc, CUTOFF, world, rank)
        assert bool((owner[index][owned] == rank).all()) and int(owned.sum()) == int((owner == rank).sum())
        pairs, _ = _ToyRuntime.neighbor_list(pos[index], cell, pbc, CUTOFF)
        pairs = pairs[owned[pairs[:, 0].long()] | owned[pairs[:, 1].long()]]
        own_edges += int(owned[pairs[:, 0].long()].sum())
        g = _CsrGraph(pairs)
        plan = partition.ExchangePlan(g, index, owned, owner, world, 4)
        gi, gj = index[g.ctr.long()], index[g.nbr.long()]
        plans.append(plan)
        keys.append((torch.stack([gi[plan.export_rows.long()], gj[plan.export_rows.long()]], 1),
                     torch.stack([gi[plan.ghost_rows.long()], gj[plan.ghost_rows.long()]], 1)))
        assert plan.send_splits[rank] == 0 and plan.recv_splits[rank] == 0
    assert own_edges == whole.shape[0]
    for a in range(world):
        for b in range(world):
            assert plans[a].send_splits[b] == plans[b].recv_splits[a]
            sa, sb = sum(plans[a].send_splits[:b]), sum(plans[b].recv_splits[:a])
            cnt = plans[a].send_splits[b]
            assert torch.equal(keys[a][0][sa:sa + cnt], keys[b][1][sb:sb + cnt])


def _clustered():
    """All atoms between x = 20 and 25 A of a periodic 60 A cell: with four slabs only rank 1 owns (or sees) atoms."""
    gen = torch.Generator().manual_seed(11)
    cell = torch.diag(torch.tensor([60.0, 9.0, 9.0]))
    pos = torch.rand(80, 3, generator=gen) * torch.tensor([5.0, 9.0, 9.0]) + torch.tensor([20.0, 0.0, 0.0])
    return pos, torch.ones(80, dtype=torch.int32), cell, [True, True, True]


class _ExchangeModel(_ToyModel):
    def __init__(self, layers):
        super().__init__(layers, False)
        self.hypers["d_pet"] = 4


@pytest.mark.parametrize("rank", [0, 2, 3])
def test_exchange_rank_with_an_empty_slab_still_enters_every_collective(rank):
    """a rank whose slab + halo holds no atom launches nothing (the library returns before any layer), but its
    peers sit in 2 x num_gnn_layers all-to-alls and one all-reduce: it has to enter all of them, with zero rows."""
    pos, z, cell, pbc = _clustered()
    layers, log = 3, []

    def all_to_all(out, inp, out_splits, in_splits):
        log.append(("a2a", tuple(out.shape), tuple(inp.shape), list(out_splits), list(in_splits)))

    def all_reduce(t):
        log.append(("ar", t.numel()))

    e, grad, n_sub, n_owned, n_rows, n_ghost = partition.energy_and_gradient_exchange(
        _ExchangeModel(layers), pos, z, cell, pbc, 4, rank, all_to_all, all_reduce, runtime=_ToyRuntime)
    assert (n_sub, n_owned, n_rows, n_ghost) == (0, 0, 0, 0) and float(e) == 0.0 and float(grad.abs().max()) == 0.0
    assert [c[0] for c in log] == ["ar"] + ["a2a"] * (2 * layers) + ["ar"]
    # the final message is [gradient | energy | run-phase error flag] (a rank that failed while it ran says so
    # in the reduction every rank enters anyway)
    assert log[0] == ("ar", 1) and log[-1] == ("ar", 3 * len(pos) + 2)
    assert all(c[1] == (0, 4) and c[2] == (0, 4) and c[3] == [0] * 4 and c[4] == [0] * 4 for c in log[1:-1])


def test_exchange_setup_failure_reaches_every_rank_before_the_layer_loop():
    """A rank that fails while it sets up (here: its neighbour list raises) says so in the first all-reduce and raises its
    own error; a rank that was fine but reads a non-zero flag raises too, and neither enters an all-to-all."""
    pos, z, cell, pbc = _clustered()
    calls = []

    class _Broken(_ToyRuntime):
        @staticmethod
        def neighbor_list(pos, cell, pbc, cutoff):
            raise MemoryError("pair buffer")

    def all_to_all(*a):
        calls.append("a2a")

    def flag_sum(extra):
        def all_reduce(t):
            calls.append(("ar", float(t.sum())))
            t += extra
        return all_reduce

    with pytest.raises(MemoryError):  # rank 1 holds the atoms and fails
        partition.energy_and_gradient_exchange(_ExchangeModel(2), pos, z, cell, pbc, 4, 1, all_to_all, flag_sum(0.0),
                                               runtime=_Broken)
    assert calls == [("ar", 1.0)]
    calls.clear()
    with pytest.raises(RuntimeError, match="another rank failed"):  # rank 2 is fine, but the summed flag says rank 1 is not
        partition.energy_and_gradient_exchange(_ExchangeModel(2), pos, z, cell, pbc, 4, 2, all_to_all, flag_sum(1.0),
                                               runtime=_ToyRuntime)
    assert calls == [("ar", 0.0)]


def test_exchange_run_phase_failure_of_a_peer_stops_every_rank():
    """a rank that fails AFTER the set-up agreement tops its collectives up and used to add a partly filled buffer
    to the final all-reduce: its peers returned a finite but wrong energy. The buffer now carries an error flag; a rank
    that finds it raised by somebody else stops too."""
    pos, z, cell, pbc = _clustered()
    layers = 3

    def all_to_all(out, inp, out_splits, in_splits):
        pass

    def all_reduce_with_failed_peer(t):
        if t.numel() > 1:   # the final [gradient | energy | flag] message: a peer raised its flag
            t[-1] += 1.0

    with pytest.raises(RuntimeError, match="another rank failed during the per-layer exchange"):
        partition.energy_and_gradient_exchange(_ExchangeModel(layers), pos, z, cell, pbc, 4, 0, all_to_all,
                                               all_reduce_with_failed_peer, runtime=_ToyRuntime)
