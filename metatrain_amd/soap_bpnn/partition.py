"""ONE large box on several GPUs (BASELINE ``configs[4]``: the 100 000-atom SOAP-BPNN box on 8 GPUs).

SOAP-BPNN's interaction range is one cutoff (``soap_bpnn/model.py:994-1000``): the energy of atom ``i`` reads the
positions of ``i`` and of its neighbours within ``r_c``, nothing else. So the box is cut into ``world`` slabs along
the lattice direction with the largest plane spacing; rank ``r`` OWNS the atoms of slab ``r`` and additionally reads
the halo atoms within ``r_c`` of the slab (positions are replicated: 100 k atoms x 16 B = 1.6 MB). It builds the
neighbour list, the descriptor, the networks and the reverse pass for that sub-system with the ordinary kernels --
energies of owned atoms are complete, halo atoms are only neighbours (their own, incomplete energies are masked out
and seeded with zero) -- and the partial results are combined by ONE exchange: an all-reduce(sum) of the energy and of
the ``[N, 3]`` gradient (1.2 MB), as SURVEY §8(e) prescribes. No kernel changes, no halo exchange of features.

PET needs ``num_gnn_layers x r_c`` = 9 A halos and an exchange of edge messages per GNN layer; for PET a single box
stays "replicas only" (DESIGN.md §6).
"""
from typing import Callable, Optional, Sequence, Tuple

import torch


def slab_partition(positions: torch.Tensor, cell: torch.Tensor, pbc: Sequence[bool], cutoff: float, world: int,
                   rank: int) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """``(index [n_sub] int64, owned [n_sub] bool, axis)``: the atoms rank ``rank`` of ``world`` works on (slab + halo,
    ascending global index) and which of them it owns. Every atom is owned by exactly one rank. Element-wise tensor
    work on the device (plumbing of the exchange, not the hot path)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world size {world}")
    dev = positions.device
    n = positions.shape[0]
    if world == 1:
        return torch.arange(n, device=dev), torch.ones(n, dtype=torch.bool, device=dev), 0
    c = cell.detach().to("cpu", torch.float64).reshape(3, 3)
    periodic_cell = bool(abs(torch.det(c)) > 1e-12)
    pos = positions.detach()
    if periodic_cell:
        # plane spacing of lattice direction a: V / |b x c|; cut along the direction with the largest one
        vol = abs(float(torch.det(c)))
        heights = []
        for a in range(3):
            b1, b2 = c[(a + 1) % 3], c[(a + 2) % 3]
            heights.append(vol / float(torch.linalg.norm(torch.linalg.cross(b1, b2))))
        axis = max(range(3), key=lambda a: heights[a])
        inv = torch.linalg.inv(c).to(dev, pos.dtype)
        f = (pos @ inv)[:, axis]
        wrap = bool(pbc[axis])
        if wrap:
            f = f - torch.floor(f)
            f = torch.where(f >= 1.0, f - 1.0, f)  # guard the rounding of values just below an integer
            lo_all, width = 0.0, 1.0
        else:
            lo_all, width = float(f.min()), max(float(f.max() - f.min()), 1e-12) * (1.0 + 1e-6)
        h = cutoff / heights[axis] * 1.0001
    else:  # open system without a cell: slabs of the bounding box along the longest Cartesian extent
        ext = pos.max(0).values - pos.min(0).values
        axis = int(torch.argmax(ext))
        f = pos[:, axis]
        wrap = False
        lo_all, width = float(f.min()), max(float(ext[axis]), 1e-12) * (1.0 + 1e-6)
        h = cutoff * 1.0001
    lo, hi = lo_all + width * rank / world, lo_all + width * (rank + 1) / world
    owned = (f >= lo) & (f < hi)
    below, above = lo - f, f - hi  # > 0 on the respective outside
    if wrap:
        below, above = torch.remainder(below, 1.0), torch.remainder(above, 1.0)
        near = torch.minimum(below, above) < h
    else:
        near = ((below > 0) & (below < h)) | ((above >= 0) & (above < h))
    index = torch.nonzero(owned | near).squeeze(1)
    return index, owned[index], axis


def energy_and_gradient(model, positions: torch.Tensor, species: torch.Tensor, cell: torch.Tensor, pbc: Sequence[bool],
                        world: int, rank: int, all_reduce: Optional[Callable[[torch.Tensor], None]] = None,
                        neighbor_list: Optional[Callable] = None):
    """Energy and dE/dR ``[N, 3]`` of one box, rank ``rank``'s share computed here and summed over ranks by
    ``all_reduce(tensor)`` (in place; ``None``: return the partial results -- the caller, or a single-process test,
    adds them). ``model``: a loaded :class:`SoapBpnnHip` (anything with ``cutoff``, ``graph``, ``forward``, ``backward``);
    ``neighbor_list``: the device neighbour list by default. Returns ``(energy [1], gradient [N, 3], n_sub, n_owned)``."""
    if neighbor_list is None:
        from .. import runtime as rt

        neighbor_list = rt.neighbor_list
    cutoff = float(model.cutoff)
    index, owned, _ = slab_partition(positions, cell, pbc, cutoff, world, rank)
    dev = positions.device
    n = positions.shape[0]
    buf = torch.zeros(3 * n + 1, dtype=torch.float32, device=dev)  # [gradient | energy]: one message
    if index.numel():
        sub_pos = positions.detach()[index].to(torch.float32).contiguous()
        sub_z = species[index].to(torch.int32).contiguous()
        pairs, _ = neighbor_list(sub_pos, cell, pbc, cutoff)
        g = model.graph(sub_pos, cell.reshape(1, 3, 3).to(dev, torch.float32), pairs[:, 0].contiguous(),
                        pairs[:, 1].contiguous(), pairs[:, 2:5].contiguous(), sub_z,
                        torch.zeros(index.numel(), dtype=torch.int32, device=dev))
        seeds = owned.to(torch.float32)
        atomic = model.forward(g)
        grad_sub = model.backward(g, seeds)
        buf[: 3 * n].view(n, 3)[index] = grad_sub
        buf[3 * n] = (atomic * seeds).sum()
    if all_reduce is not None:
        all_reduce(buf)
    return buf[3 * n:], buf[: 3 * n].view(n, 3), int(index.numel()), int(owned.sum())
