"""Slab + halo partition of ONE box over several ranks: the host-side plumbing shared by the
SOAP-BPNN (`soap_bpnn/partition.py`, halo = one cutoff) and PET (`pet/partition.py`, halo = num_gnn_layers + 1
cutoffs) single-box paths. Positions are replicated, centres are cut into slabs along the lattice direction with the largest
plane spacing, a rank works on its slab plus every atom within `halo` of it."""
from typing import Sequence, Tuple

import torch


def complete_lattice(cell: torch.Tensor, pbc: Sequence[bool]) -> torch.Tensor:
    """The 3 x 3 fp64 lattice with the rows of non-periodic directions replaced by unit vectors orthogonal to the
    periodic ones (metatomic stores zero vectors there). Raises if the periodic rows are linearly dependent."""
    c = cell.detach().to("cpu", torch.float64).reshape(3, 3).clone()
    basis = []

    def residual(v):
        r = v.clone()
        for b in basis:
            r = r - torch.dot(r, b) * b
        return r

    for a in range(3):
        if pbc[a]:
            r = residual(c[a])
            n = float(torch.linalg.norm(r))
            if n <= 1e-9:
                raise ValueError("singular cell: the lattice vectors of the periodic directions are linearly dependent")
            basis.append(r / n)
    for a in range(3):
        if pbc[a]:
            continue
        cand = [residual(torch.eye(3, dtype=torch.float64)[e]) for e in range(3)]
        best = max(cand, key=lambda r: float(torch.linalg.norm(r)))
        best = best / torch.linalg.norm(best)
        c[a] = best
        basis.append(best)
    if abs(float(torch.det(c))) <= 1e-12:
        raise ValueError("singular cell")
    return c


def _slab_coordinate(positions: torch.Tensor, cell: torch.Tensor, pbc: Sequence[bool]):
    """``(f [N], wrap, lo_all, width, unit, axis)``: the coordinate the slabs cut (fractional along the lattice direction
    with the largest extent, or Cartesian for an open system), whether it wraps, its range and the length of one unit."""
    dev = positions.device
    pos = positions.detach()
    periodic = [bool(p) for p in pbc]
    if any(periodic):
        # lattice completed for the non-periodic rows (metatomic: zero vectors there) exactly as the neighbour list does
        # (csrc/nl.hip::lattice_params): a surface / wire keeps its wrap-around halo along the periodic directions
        c = complete_lattice(cell, periodic)
        vol = abs(float(torch.det(c)))
        heights = []
        for a in range(3):
            b1, b2 = c[(a + 1) % 3], c[(a + 2) % 3]
            heights.append(vol / float(torch.linalg.norm(torch.linalg.cross(b1, b2))))
        inv = torch.linalg.inv(c).to(dev, pos.dtype)
        frac = pos @ inv
        span = [heights[a] if periodic[a] else float(frac[:, a].max() - frac[:, a].min()) * heights[a] for a in range(3)]
        axis = max(range(3), key=lambda a: span[a])
        f = frac[:, axis]
        wrap = periodic[axis]
        if wrap:
            f = f - torch.floor(f)
            f = torch.where(f >= 1.0, f - 1.0, f)  # guard the rounding of values just below an integer
            lo_all, width = 0.0, 1.0
        else:
            lo_all, width = float(f.min()), max(float(f.max() - f.min()), 1e-12) * (1.0 + 1e-6)
        unit = heights[axis]
    else:  # open system: slabs of the bounding box along the longest Cartesian extent
        ext = pos.max(0).values - pos.min(0).values
        axis = int(torch.argmax(ext))
        f = pos[:, axis]
        wrap = False
        lo_all, width = float(f.min()), max(float(ext[axis]), 1e-12) * (1.0 + 1e-6)
        unit = 1.0
    return f, wrap, lo_all, width, unit, axis


def slab_owner(positions: torch.Tensor, cell: torch.Tensor, pbc: Sequence[bool], world: int) -> torch.Tensor:
    """``[N]`` int64: the rank that owns every atom -- the same intervals, bound for bound, as :func:`slab_partition`."""
    f, _, lo_all, width, _, _ = _slab_coordinate(positions, cell, pbc)
    return _owner_of(f, lo_all, width, world)


def _owner_of(f: torch.Tensor, lo_all: float, width: float, world: int) -> torch.Tensor:
    """Slab number of every coordinate: ``floor((f - lo) / width * world)`` clamped to ``[0, world)``, so that a value that
    rounds onto the upper bound of the last slab still has an owner and NaN / inf positions are refused here
    instead of surfacing as an opaque ``bincount`` error in the exchange plan."""
    if not bool(torch.isfinite(f).all()):
        raise ValueError("non-finite atom positions: the slab partition cannot place them")
    return torch.clamp(torch.floor((f - lo_all) / width * world), 0, world - 1).to(torch.long)


def slab_partition(positions: torch.Tensor, cell: torch.Tensor, pbc: Sequence[bool], halo: float, world: int,
                   rank: int) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """``(index [n_sub] int64, owned [n_sub] bool, axis)``: the atoms rank ``rank`` of ``world`` works on (slab + halo,
    ascending global index; the halo holds every atom within ``halo`` of the slab) and which of them it owns. Every atom
    is owned by exactly one rank. Element-wise tensor work on the device (plumbing of the exchange, not the hot
    path)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world size {world}")
    dev = positions.device
    n = positions.shape[0]
    if world == 1:
        return torch.arange(n, device=dev), torch.ones(n, dtype=torch.bool, device=dev), 0
    f, wrap, lo_all, width, unit, axis = _slab_coordinate(positions, cell, pbc)
    h = halo / unit * 1.0001
    lo, hi = lo_all + width * rank / world, lo_all + width * (rank + 1) / world
    owned = _owner_of(f, lo_all, width, world) == rank  # the same expression slab_owner uses: every atom has one owner
    below, above = lo - f, f - hi  # > 0 on the respective outside
    if wrap:
        # (an atom within rounding of a slab bound may be owned by the neighbouring rank while lo - f is -1e-17: shifted by
        # the same 1e-12 as the open branch so that it wraps to ~0 and not to ~1 and lands in this rank's halo)
        below, above = torch.remainder(below + 1e-12, 1.0) - 1e-12, torch.remainder(above + 1e-12, 1.0) - 1e-12
        near = torch.minimum(below, above) < h
    else:
        near = ((below > -1e-12) & (below < h)) | ((above > -1e-12) & (above < h))
    index = torch.nonzero(owned | near).squeeze(1)
    return index, owned[index], axis
