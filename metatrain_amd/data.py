"""On-device data path: neighbour lists and batching built on the GPU, so that a training or MD
loop never round-trips systems through CPU collate workers.

Replaces, for the PET / SOAP-BPNN hot path, ``utils/neighbor_lists.py:100-135`` (vesin on the CPU, one ASE round
trip per system) and the pickled-blob ``CollateFn`` of ``utils/data/dataset.py:381-445``: the batch is the seven
plain tensors ``systems_to_batch`` / ``PETBackend.preprocess`` take (``pet/modules/structures.py:20-95``), with atom
indices offset per system exactly as ``concatenate_structures`` does.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import runtime as rt

System = Tuple[torch.Tensor, torch.Tensor, torch.Tensor, Sequence[bool]]  # positions, species, cell, pbc


def collate(systems: List[System], cutoff: float, targets: Optional[Dict[str, List[torch.Tensor]]] = None
            ) -> Dict[str, torch.Tensor]:
    """Device-side collate of ``(positions [n,3], atomic numbers [n], cell [3,3], pbc)`` systems (device tensors).

    Returns ``positions [N,3] f32, cells [S,3,3] f32, centers / neighbors [E] i32, cell_shifts [E,3] i32,
    species [N] i32, system_indices [N] i32`` (all on the systems' device) plus, for every entry of ``targets``,
    the per-system tensors concatenated along dim 0 (energies ``[S, ...]``, forces ``[N, 3]``)."""
    if not systems:
        raise ValueError("collate needs at least one system")
    dev = systems[0][0].device
    pos_l, z_l, cell_l, sys_l, first = [], [], [], [], [0]
    for k, (pos, z, cell, pbc) in enumerate(systems):
        pos = pos.detach().to(dev, torch.float32)
        pos_l.append(pos)
        z_l.append(z.to(dev, torch.int32))
        cell_l.append(cell.detach().to(torch.float32))
        sys_l.append(torch.full((int(pos.shape[0]),), k, dtype=torch.int32, device=dev))
        first.append(first[-1] + int(pos.shape[0]))
    positions = torch.cat(pos_l)
    cells = torch.stack([c.to(dev) for c in cell_l])
    # every system's neighbour list in ONE set of launches, global atom indices (pet_nl_build_batch)
    pairs, _ = rt.neighbor_list_batch(positions, torch.stack([c.cpu() for c in cell_l]), [s[3] for s in systems], first,
                                      cutoff, want_vectors=False)
    batch = {
        "positions": positions,
        "cells": cells,
        "centers": pairs[:, 0].contiguous(),
        "neighbors": pairs[:, 1].contiguous(),
        "cell_shifts": pairs[:, 2:5].contiguous(),
        "species": torch.cat(z_l),
        "system_indices": torch.cat(sys_l),
    }
    for name, values in (targets or {}).items():
        if len(values) != len(systems):
            raise ValueError(f"target '{name}': {len(values)} entries for {len(systems)} systems")
        batch[name] = torch.cat([v.to(dev).reshape((-1,) + tuple(v.shape[1:])) if v.dim() else v.to(dev).reshape(1)
                                 for v in values])
    return batch


def graph_of(model: rt.HipModel, batch: Dict[str, torch.Tensor]) -> rt.HipGraph:
    """``preprocess`` of a collated batch (CSR / NEF indices, geometry, cutoff factors) on the device."""
    return rt.HipGraph(model, batch["positions"], batch["cells"], batch["centers"], batch["neighbors"],
                       batch["cell_shifts"], batch["species"], batch["system_indices"])
