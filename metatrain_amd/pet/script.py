"""TorchScript surface of the PET hot path: ``torch.classes.pet_hip.PetHipModule`` from
``lib/libpet_hip_torch.so`` (csrc/torch_ops.cpp) wrapped in a scriptable ``torch.nn.Module``.

An exported model (``mtt export`` -> ``torch.jit.save``, ``cli/export.py:235-266``) needs every op reachable from
``forward`` to be TorchScript-visible; this module is what ``PET.forward`` (``pet/model.py:416-537``) would hold in
place of the eager ``PETBackend`` when it is scripted: per-atom energies that are differentiable w.r.t. positions and
cells inside TorchScript, weights pickled with the module. MD engines load it after
``torch.ops.load_library("libpet_hip_torch.so")``.
"""
import os
from typing import Dict, List, Optional

import torch

from ..build import TORCH_LIB
from ..runtime import hypers_struct
from .._lib import PetHipError

_loaded = False


def load_ops() -> None:
    """Register ``torch.classes.pet_hip.*`` (idempotent). Raises if the library has not been built."""
    global _loaded
    if _loaded:
        return
    if not os.path.exists(TORCH_LIB):
        raise PetHipError(f"{TORCH_LIB} not found: build it with `python -m metatrain_amd.build`")
    from .. import _lib

    _lib.load()  # libpet_hip.so first (and torch before both): one HIP runtime
    torch.ops.load_library(TORCH_LIB)
    _loaded = True


def make_core(hypers: dict, atomic_types: List[int], state_dict: Dict[str, torch.Tensor], target: str,
              block: Optional[str] = None):
    """``torch.classes.pet_hip.PetHipModule`` for one target of a reference-schema state dict."""
    load_ops()
    block = block or target
    h = hypers_struct(hypers, atomic_types)
    numbers = [float(getattr(h, name)) for name, _ in h._fields_]
    keys, tensors = [], []
    for key, t in state_dict.items():
        parts = key.split(".")
        if parts[0] in ("node_heads", "edge_heads", "node_last_layers", "edge_last_layers"):
            if parts[1] != target:
                continue
            parts[1] = "@"
            if parts[0].endswith("last_layers"):
                if parts[3] != block:
                    continue
                parts[3] = "@"
        keys.append(".".join(parts))
        tensors.append(t.detach().cpu().contiguous())
    return torch.classes.pet_hip.PetHipModule(numbers, [int(z) for z in atomic_types], keys, tensors)


class PETScriptModule(torch.nn.Module):
    """Scriptable: ``forward(positions, cells, centers, neighbors, cell_shifts, species, system_indices)`` ->
    per-atom predictions ``[N, 1]`` of one target (sum them per system for energies; autograd for forces)."""

    def __init__(self, core):
        super().__init__()
        self.core = core

    def forward(self, positions: torch.Tensor, cells: torch.Tensor, centers: torch.Tensor, neighbors: torch.Tensor,
                cell_shifts: torch.Tensor, species: torch.Tensor, system_indices: torch.Tensor) -> torch.Tensor:
        return self.core.atomic_energies(positions, cells, centers, neighbors, cell_shifts, species, system_indices)


class EnergyAndForces(torch.nn.Module):
    """What an exported model does around the core: total energies per system and forces by autograd, all inside
    TorchScript."""

    def __init__(self, core, n_systems_hint: int = 1):
        super().__init__()
        self.pet = PETScriptModule(core)

    def forward(self, positions: torch.Tensor, cells: torch.Tensor, centers: torch.Tensor, neighbors: torch.Tensor,
                cell_shifts: torch.Tensor, species: torch.Tensor, system_indices: torch.Tensor):
        positions = positions.detach().requires_grad_(True)
        atomic = self.pet(positions, cells, centers, neighbors, cell_shifts, species, system_indices)
        energies = torch.zeros(cells.shape[0], dtype=atomic.dtype, device=atomic.device).index_add(
            0, system_indices.to(torch.long), atomic[:, 0])
        grads = torch.autograd.grad([energies.sum()], [positions])
        g = grads[0]
        assert g is not None
        return energies, -g
