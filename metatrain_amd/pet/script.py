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
        t = t.detach().cpu()
        if hypers["activation"] == "SiLU" and ".w_in." in key:
            t = torch.cat([t, t], dim=0)  # silu(W x + b) on the SwiGLU stage: value half = gate half (runtime.py)
        tensors.append(t.contiguous())
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


class ExportedEnergyModel(torch.nn.Module):
    """Tensor-level equivalent of what ``PET.forward`` does around the backbone at evaluation time
    (``pet/model.py:592-660``), scriptable end to end:

    * ``scale``: the scaler's per-target factor (``utils/scaler``: prediction * scale),
    * ``composition``: the additive composition model's per-species energies, indexed by atomic number
      (``utils/additive/composition.py``: + sum_i w[Z_i]); not differentiated (it does not depend on positions),
    * ``selected_atoms``: optional bool / index mask of the atoms that contribute to the per-system sums and that
      are returned per atom (metatomic's ``selected_atoms``); forces are still returned for every atom,
    * forces ``-dE/dR`` and, on request, the stress ``(1/V) dE/d(strain)`` assembled from ``dE/dR`` and
      ``dE/dcell`` of the HIP backward (``utils/evaluate_model.py`` strain trick, done analytically:
      ``dE/deps = R^T dE/dR + h^T dE/dh``).

    ``forward`` returns ``(energies [S], forces [N, 3], stress [S, 3, 3] or empty, per_atom [n_selected])``."""

    def __init__(self, core, scale: float = 1.0, composition: Optional[torch.Tensor] = None):
        super().__init__()
        self.pet = PETScriptModule(core)
        self.scale = float(scale)
        self.register_buffer("composition", composition.detach().clone().to(torch.float32)
                             if composition is not None else torch.zeros(0, dtype=torch.float32))

    def forward(self, positions: torch.Tensor, cells: torch.Tensor, centers: torch.Tensor, neighbors: torch.Tensor,
                cell_shifts: torch.Tensor, species: torch.Tensor, system_indices: torch.Tensor,
                selected_atoms: Optional[torch.Tensor] = None, with_stress: bool = False):
        positions = positions.detach().requires_grad_(True)
        cells = cells.detach().requires_grad_(with_stress)
        atomic = self.pet(positions, cells, centers, neighbors, cell_shifts, species, system_indices)[:, 0]
        atomic = atomic * self.scale
        keep = torch.ones(positions.shape[0], dtype=torch.bool, device=positions.device)
        if selected_atoms is not None:
            if selected_atoms.dtype == torch.bool:
                keep = selected_atoms.to(positions.device)
            else:
                keep = torch.zeros_like(keep).index_fill(0, selected_atoms.to(positions.device, torch.long), True)
        masked = torch.where(keep, atomic, torch.zeros_like(atomic))
        sysl = system_indices.to(torch.long)
        energies = torch.zeros(cells.shape[0], dtype=atomic.dtype, device=atomic.device).index_add(0, sysl, masked)
        wrt = [positions, cells] if with_stress else [positions]
        grads = torch.autograd.grad([energies.sum()], wrt)
        g_pos = grads[0]
        assert g_pos is not None
        stress = torch.zeros((0, 3, 3), dtype=atomic.dtype, device=atomic.device)
        if with_stress:
            g_cell = grads[1]
            assert g_cell is not None
            outer = positions.detach().unsqueeze(2) * g_pos.unsqueeze(1)  # [N, 3, 3]: R_a dE/dR_b per atom
            virial = torch.zeros((cells.shape[0], 3, 3), dtype=atomic.dtype, device=atomic.device).index_add(
                0, sysl, outer)
            virial = virial + torch.matmul(cells.detach().transpose(1, 2), g_cell)
            volume = torch.abs(torch.linalg.det(cells.detach()))
            stress = virial / volume.clamp_min(1e-30).reshape(-1, 1, 1)
        per_atom = atomic.detach()
        energies = energies.detach()
        if self.composition.numel() > 0:
            base = self.composition[species.to(torch.long)].to(atomic.dtype)
            per_atom = per_atom + base
            energies = energies + torch.zeros_like(energies).index_add(
                0, sysl, torch.where(keep, base, torch.zeros_like(base)))
        return energies, -g_pos, stress, per_atom[keep]
