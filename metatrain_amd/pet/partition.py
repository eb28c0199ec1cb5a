"""ONE large PET box on several GPUs: slab + halo centre partition with a single exchange.

How far does the energy of atom ``i`` read? The features of ``i`` after the LAST GNN layer are
``input + output + combination([output ; output[reversed]])`` (``backend.py:559-575``): they gather the layer-G outputs of
the reversed edges ``j -> i`` from the transformers of its neighbours ``j``, which read the positions of THEIR
neighbours and, through the layer-(G-1) messages, one hop further per layer below. That is ``(G + 1) r_c`` in positions,
one hop MORE than the ``num_gnn_layers x r_c`` the reference declares as its interaction range (``pet/model.py:1004``; the
last hop's weight is the product of two cutoff functions and small: leaving it out changes dE/dR by ~1e-4 of its
largest entry on the 30 000-atom test box, which is why it goes unnoticed at the reference's usual tolerances but
not at this repository's 1e-5 bar). With the adaptive cutoff (``structures.py:225-263``) the cutoff of a pair is the
mean of its two atoms' cutoffs, each a function of that atom's own neighbourhood: one more hop again.

SURVEY §8(e) defers a per-layer halo exchange of edge messages; what is built here needs none: rank ``r`` OWNS the
atoms of slab ``r`` and works on the slab plus every atom within ``hops x r_c`` of it (positions are replicated: 16 B
per atom), with the ordinary kernels and the cell unchanged (periodic images stay what they are). On that sub-system

* every transformer output that an owned energy reads belongs to an atom with a complete neighbourhood, so it is the
  one the whole box would give; the outermost shell has truncated neighbourhoods and wrong features, which no owned
  energy reads;
* the reverse pass is seeded with 1 on owned atoms and 0 on halo atoms: ``d(sum of owned energies)/dR`` for every atom
  of the sub-system;
* ONE all-reduce(sum) of ``[gradient | energy]`` (12 B per atom + 4 B) combines the ranks.

The price is redundant work in the halo: a slab of thickness ``L / world`` computes ``L / world + 2 hops r_c``, so the
scheme pays off for boxes whose slabs are thick against 27 A (a 1 M-atom box at 0.05 atoms / A^3 on 8 GPUs: 34 A
slabs, 56 % efficiency; the 100 k-atom box: 16 A slabs, 37 %; ``hops = num_gnn_layers`` reproduces the reference's
declared range at ~1e-4 accuracy and 47 %).

The per-layer exchange cuts the halo to ONE cutoff (:func:`energy_and_gradient_exchange`, round 3). A rank then runs the
transformers only on the centres it owns; what it lacks is, per GNN layer, the edge tokens of the edges ``j -> i`` whose
centre ``j`` is foreign and whose neighbour ``i`` is its own (the reversed-edge operand of its combination stage). Those
rows exist in its CSR layout as GHOST rows (the halo atom's edges towards owned atoms; halo-halo edges are dropped), the
owner of ``j`` holds the same edges as EXPORT rows, and the library calls the host between the edge transformer and the
combination stage of every layer (``pet_graph_set_exchange``, include/pet_hip.h): export rows are gathered into one send
buffer ordered by (peer, i, j), the host runs ONE all-to-all (``all_to_all_single`` on RCCL: one message per peer and
xGMI link), the received rows are scattered over the ghost rows. The reverse pass sends the ghost rows' adjoints back
the same way and ADDS them to the export rows' adjoints. Needs a cell wider than two cutoffs in every periodic
direction (a pair connected by two images has no unique (i, j) key), the fixed cutoff and the compiled default size.
"""
from typing import Callable, Optional, Sequence

import torch

from ..partition import slab_partition


def energy_and_gradient(model, positions: torch.Tensor, species: torch.Tensor, cell: torch.Tensor, pbc: Sequence[bool],
                        world: int, rank: int, all_reduce: Optional[Callable[[torch.Tensor], None]] = None,
                        neighbor_list: Optional[Callable] = None, runtime=None, hops: Optional[int] = None):
    """Energy and dE/dR ``[N, 3]`` of one box, rank ``rank``'s share computed here and summed over ranks by
    ``all_reduce(tensor)`` (in place; ``None``: return the partial results -- the caller, or a single-process test,
    adds them). ``model``: a loaded :class:`metatrain_amd.runtime.HipModel` (single energy target); ``neighbor_list``:
    the device neighbour list by default; ``hops``: halo thickness in cutoffs (default: exact, ``num_gnn_layers + 1``, one
    more with the adaptive cutoff). Returns ``(energy [1], gradient [N, 3], n_sub, n_owned)``."""
    if runtime is None:
        from .. import runtime
    if neighbor_list is None:
        neighbor_list = runtime.neighbor_list
    cutoff = float(model.hypers["cutoff"])
    if hops is None:
        hops = int(model.hypers["num_gnn_layers"]) + 1 + (1 if model.hypers.get("num_neighbors_adaptive") is not None else 0)
    halo = cutoff * hops
    index, owned, _ = slab_partition(positions, cell, pbc, halo, world, rank)
    dev = positions.device
    n = positions.shape[0]
    buf = torch.zeros(3 * n + 1, dtype=torch.float32, device=dev)  # [gradient | energy]: one message
    if index.numel():
        sub_pos = positions.detach()[index].to(torch.float32).contiguous()
        sub_z = species[index].to(torch.int32).contiguous()
        pairs, _ = neighbor_list(sub_pos, cell, pbc, cutoff)
        graph = runtime.HipGraph(model, sub_pos, cell.reshape(1, 3, 3).to(dev, torch.float32), pairs[:, 0].contiguous(),
                                 pairs[:, 1].contiguous(), pairs[:, 2:5].contiguous(), sub_z,
                                 torch.zeros(index.numel(), dtype=torch.int32, device=dev))
        fw = runtime.HipForward(model, graph)
        seeds = owned.to(torch.float32)
        atomic = fw.forward()
        grad_sub = fw.backward(seeds)
        buf[: 3 * n].view(n, 3)[index] = grad_sub
        buf[3 * n] = (atomic.reshape(-1) * seeds).sum()
    if all_reduce is not None:
        all_reduce(buf)
    return buf[3 * n:], buf[: 3 * n].view(n, 3), int(index.numel()), int(owned.sum())


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: ONE box over several GPUs with a PER-LAYER exchange of edge tokens (one-cutoff halos)
# ---------------------------------------------------------------------------------------------------------------------
class ExchangePlan:
    """Who sends which edge rows to whom (``pet_graph_set_exchange``). Built from the rank's sub-system alone: positions
    are replicated, so every rank knows the owner of every atom (``slab_owner``); the rows a rank exports to rank ``q`` --
    edges ``(i -> j)`` with ``i`` its own and ``j`` owned by ``q`` -- are exactly the rows rank ``q`` holds as ghosts, and both
    sides order them by the global pair ``(i, j)``, so no negotiation is needed."""

    def __init__(self, graph, index: torch.Tensor, owned: torch.Tensor, owner: torch.Tensor, world: int, d_pet: int):
        csr = graph.csr()
        ctr, nbr = csr["ctr"].long(), csr["nbr"].long()
        gi, gj = index[ctr], index[nbr]                      # global atom numbers of every CSR row's centre / neighbour
        own_c, own_n = owned[ctr], owned[nbr]
        n_glob = int(owner.numel())

        def ordered(mask, peer_of, first, second):
            rows = torch.nonzero(mask).squeeze(1)
            peer = peer_of[rows]
            key = (peer * n_glob + first[rows]) * n_glob + second[rows]
            order = torch.argsort(key)
            rows, peer = rows[order], peer[order]
            if rows.numel() > 1 and bool((key[order][1:] == key[order][:-1]).any()):
                raise ValueError("a pair of atoms is connected by more than one image: the per-layer exchange needs a cell "
                                 "wider than two cutoffs in every periodic direction")
            return rows.to(torch.int32).contiguous(), torch.bincount(peer, minlength=world).tolist()

        # export: own centre, foreign neighbour -> to the neighbour's owner; ghost: foreign centre -> from the centre's owner
        self.export_rows, self.send_splits = ordered(own_c & ~own_n, owner[gj], gi, gj)
        self.ghost_rows, self.recv_splits = ordered(~own_c, owner[gi], gi, gj)
        dev = ctr.device
        self.export_buf = torch.zeros((self.export_rows.numel(), d_pet), dtype=torch.float32, device=dev)
        self.ghost_buf = torch.zeros((self.ghost_rows.numel(), d_pet), dtype=torch.float32, device=dev)


def energy_and_gradient_exchange(model, positions: torch.Tensor, species: torch.Tensor, cell: torch.Tensor,
                                 pbc: Sequence[bool], world: int, rank: int, all_to_all: Callable,
                                 all_reduce: Optional[Callable[[torch.Tensor], None]] = None,
                                 neighbor_list: Optional[Callable] = None, runtime=None):
    """Energy and dE/dR of one box with ONE-cutoff halos: rank ``rank`` runs the transformer layers on the atoms it owns; the
    edge tokens its combination stage needs from foreign centres arrive by ``all_to_all(out, inp, out_splits, in_splits)``
    once per GNN layer (and their adjoints go back once per layer in the reverse pass) -- ``torch.distributed.
    all_to_all_single`` on RCCL: one message per peer over its xGMI link. Then ONE all-reduce(sum) of ``[gradient | energy]``
    as in :func:`energy_and_gradient`. Default model size, PreLN + feedforward, fixed cutoff.
    Returns ``(energy [1], gradient [N, 3], n_sub, n_owned, n_rows, n_ghost_rows)``."""
    import ctypes

    from ..partition import slab_owner

    if runtime is None:
        from .. import runtime
    if neighbor_list is None:
        neighbor_list = runtime.neighbor_list
    if model.hypers.get("num_neighbors_adaptive") is not None:
        raise ValueError("the per-layer exchange is built for the fixed cutoff")
    cutoff = float(model.hypers["cutoff"])
    dev = positions.device
    n = positions.shape[0]
    n_layers = int(model.hypers["num_gnn_layers"])
    buf = torch.zeros(3 * n + 2, dtype=torch.float32, device=dev)  # [gradient | energy | run-phase error flag]
    n_rows = n_ghost = 0
    # Every rank must enter the same collectives: 2 x num_gnn_layers all-to-alls and one all-reduce. A rank whose slab
    # (plus halo) is empty has nothing to launch -- the library returns before any layer -- and a rank that fails while
    # it sets up (neighbour list, graph, plan; an atom of more than 127 neighbours, which the exchange does not serve)
    # would leave its peers waiting. So: set up under a guard, agree on an "everybody is fine" flag FIRST (one tiny
    # all-reduce; every rank raises if any rank failed), then run, and top the all-to-all count up with zero-row calls.
    index = owned = graph = plan = None
    setup_error: Optional[BaseException] = None
    try:
        index, owned, _ = slab_partition(positions, cell, pbc, cutoff, world, rank)
        owner = slab_owner(positions, cell, pbc, world)
        if index.numel():
            sub_pos = positions.detach()[index].to(torch.float32).contiguous()
            sub_z = species[index].to(torch.int32).contiguous()
            pairs, _ = neighbor_list(sub_pos, cell, pbc, cutoff)
            if pairs.numel():   # halo-halo edges are nobody's business here: every kept edge touches an owned atom
                keep = owned[pairs[:, 0].long()] | owned[pairs[:, 1].long()]
                pairs = pairs[keep]
            graph = runtime.HipGraph(model, sub_pos, cell.reshape(1, 3, 3).to(dev, torch.float32), pairs[:, 0].contiguous(),
                                     pairs[:, 1].contiguous(), pairs[:, 2:5].contiguous(), sub_z,
                                     torch.zeros(index.numel(), dtype=torch.int32, device=dev))
            if int(getattr(graph, "max_neighbors", 0)) > 127:
                raise ValueError("an atom has more than 127 neighbours: the per-layer exchange is built for the tuned kernels")
            plan = ExchangePlan(graph, index, owned, owner, world, int(model.hypers["d_pet"]))
            n_rows, n_ghost = int(graph.n_edges), int(plan.ghost_rows.numel())
    except Exception as exc:  # noqa: BLE001 -- reported to every rank below
        setup_error = exc
    if all_reduce is not None:
        flag = torch.tensor([0.0 if setup_error is None else 1.0], dtype=torch.float32, device=dev)
        all_reduce(flag)
        if float(flag) > 0 and setup_error is None:
            raise RuntimeError("another rank failed while setting up the per-layer exchange; this rank stops with it")
    if setup_error is not None:
        raise setup_error
    empty = torch.zeros((0, int(model.hypers["d_pet"])), dtype=torch.float32, device=dev)
    zeros = [0] * world
    calls = [0, 0]  # all-to-alls issued so far, per direction

    def exchange(direction):
        if plan is None:
            all_to_all(empty, empty, zeros, zeros)
        elif direction == 0:
            all_to_all(plan.ghost_buf, plan.export_buf, plan.recv_splits, plan.send_splits)
        else:
            all_to_all(plan.export_buf, plan.ghost_buf, plan.send_splits, plan.recv_splits)
        calls[direction] += 1

    run_error: Optional[BaseException] = None
    if plan is not None:
        def hook(_user, direction, _layer):
            try:
                exchange(direction)
                return 0
            except Exception as exc:  # a Python exception must not unwind through the C frames
                plan.error = exc
                return 1

        from .._lib import EXCHANGE_FN

        cb = EXCHANGE_FN(hook)
        rt_check = runtime.check
        rt_check(model.lib.pet_graph_set_exchange(graph.handle, runtime._ptr(plan.export_rows), plan.export_rows.numel(),
                                                  runtime._ptr(plan.ghost_rows), plan.ghost_rows.numel(),
                                                  runtime._ptr(plan.export_buf), runtime._ptr(plan.ghost_buf), cb, None))
        try:
            fw = runtime.HipForward(model, graph)
            seeds = owned.to(torch.float32)
            atomic = fw.forward()
            while calls[0] < n_layers:  # a sub-system without edges: the library had nothing to exchange
                exchange(0)
            grad_sub = fw.backward(seeds)
            buf[: 3 * n].view(n, 3)[index] = grad_sub
            buf[3 * n] = (atomic.reshape(-1) * seeds).sum()
        except Exception as exc:  # noqa: BLE001 -- the peers still get their collectives below
            run_error = getattr(plan, "error", None) or exc
        finally:
            model.lib.pet_graph_set_exchange(graph.handle, None, 0, None, 0, None, None, EXCHANGE_FN(), None)
    if plan is not None and getattr(plan, "error", None) is not None:
        raise plan.error  # the collective itself failed: nothing to keep in step with
    for direction in (0, 1):  # keep in step with the peers: an empty slab, or a local failure after the agreement
        while calls[direction] < n_layers:
            exchange(direction)
    if run_error is not None:
        buf[3 * n + 1] = 1.0  # every rank learns of it with the reduction it enters anyway
    if all_reduce is not None:
        all_reduce(buf)
    if run_error is not None:
        raise run_error
    if float(buf[3 * n + 1]) > 0:
        raise RuntimeError("another rank failed during the per-layer exchange step: its contribution to the energy and the "
                           "gradient is missing; this rank stops with it")
    return (buf[3 * n:3 * n + 1], buf[: 3 * n].view(n, 3), 0 if index is None else int(index.numel()),
            0 if owned is None else int(owned.sum()), n_rows, n_ghost)

