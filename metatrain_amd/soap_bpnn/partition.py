"""ONE large box on several GPUs (BASELINE ``configs[4]``: the 100 000-atom SOAP-BPNN box on 8 GPUs).

SOAP-BPNN's interaction range is one cutoff (``soap_bpnn/model.py:994-1000``): the energy of atom ``i`` reads the
positions of ``i`` and of its neighbours within ``r_c``, nothing else. So the box is cut into ``world`` slabs along
the lattice direction with the largest plane spacing; rank ``r`` OWNS the atoms of slab ``r`` and additionally reads
the halo atoms within ``r_c`` of the slab (positions are replicated: 100 k atoms x 16 B = 1.6 MB). It builds the
neighbour list, the descriptor, the networks and the reverse pass for that sub-system with the ordinary kernels --
energies of owned atoms are complete, halo atoms are only neighbours (their own, incomplete energies are masked out
and seeded with zero) -- and the partial results are combined by ONE exchange: an all-reduce(sum) of the energy and of
the ``[N, 3]`` gradient (1.2 MB), as SURVEY §8(e) prescribes. No kernel changes, no halo exchange of features.

PET uses the same scheme with ``num_gnn_layers x r_c`` halos (``metatrain_amd/pet/partition.py``).
"""
from typing import Callable, Optional, Sequence

import torch

from ..partition import slab_partition  # noqa: F401  (re-exported: the partition itself is shared with PET)


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
