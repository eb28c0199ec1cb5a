"""CPU oracle of the SOAP-BPNN hot path. TEST INFRASTRUCTURE ONLY.

**Parity unpinned.** The descriptor arithmetic of the reference lives in torch-spex
(``>=0.1,<0.2``), sphericart-torch (``>=2.0.3``) and scipy, none of which ships under
``/root/reference`` (``pyproject.toml:69-73``); the reference's own regression energies
(``soap_bpnn/tests/test_regression.py:23-66``) depend on spex's RNG-order-dependent init and cannot be
regenerated here. This file restates the *configuration* the reference passes to spex
(``soap_bpnn/model.py:251-264``):

    SphericalExpansion(cutoff, max_angular,
                       radial   = LaplacianEigenstates(max_radial),      # trimmed by eigenvalue
                       angular  = "SphericalHarmonics",                  # orthonormal real Y_lm
                       cutoff_function = ShiftedCosine(width),
                       species  = Orthogonal(species) | Alchemical(4, n_species))

from the published definition of those components, and restates exactly the parts that are plain
torch in the reference: the power-spectrum contraction ``soap_bpnn/modules/power_spectrum.py:125-136``
and the dense tail ``soap_bpnn/model.py:553-595, 1204-1219`` (centre encoding, LayerNorm, 2 x 32 SiLU
MLP without biases, bias-free linear last layer; per centre species when ``legacy``).

Published definitions used (Laplacian-eigenstate basis, Bigi et al., J. Chem. Phys. 157, 234101 (2022)):
  * R_nl(r) = N_nl j_l(z_nl r / r_c), z_nl the n-th positive zero of the spherical Bessel function j_l,
    N_nl = [ r_c^3 / 2 * j_{l+1}(z_nl)^2 ]^(-1/2)  (orthonormal on [0, r_c] with weight r^2);
  * trimming: keep (n, l) with z_nl^2 <= z_{max_radial, 0}^2 = ((max_radial + 1) pi)^2, l <= max_angular,
    so n_per_l[0] = max_radial + 1;
  * the power spectrum sums over m, hence it does not depend on the sign / ordering convention of the
    real spherical harmonics, only on their orthonormality.
"""
import math
from typing import Dict, List, Tuple

import numpy as np
import torch


DEFAULT_HYPERS = {  # soap_bpnn/documentation.py:53-120
    "soap": {"max_angular": 6, "max_radial": 7, "cutoff": {"radius": 5.0, "width": 0.5}},
    "legacy": True,
    "bpnn": {"num_hidden_layers": 2, "num_neurons_per_layer": 32, "layernorm": True},
}


# ---------------------------------------------------------------------------------------------
# radial basis
# ---------------------------------------------------------------------------------------------
def _sph_jn(l: int, x: torch.Tensor) -> torch.Tensor:
    """Spherical Bessel function j_l(x), differentiable, accurate for all x >= 0 in fp64:
    ascending series below x = l + 1.5, upward recurrence above."""
    xs = torch.where(x < l + 1.5, x, torch.full_like(x, l + 1.5))  # keep the unused branch finite
    # series: j_l(x) = x^l / (2l+1)!! * sum_k (-x^2/2)^k / (k! (2l+3)(2l+5)...(2l+2k+1))
    dfact = 1.0
    for k in range(1, l + 1):
        dfact *= 2 * k + 1
    term = torch.ones_like(xs)
    total = torch.ones_like(xs)
    for k in range(1, 40):
        term = term * (-0.5 * xs * xs) / (k * (2 * l + 2 * k + 1))
        total = total + term
    series = xs**l / dfact * total
    xl = torch.where(x < l + 1.5, torch.full_like(x, l + 1.5), x)
    j0 = torch.sin(xl) / xl
    if l == 0:
        rec = j0
    else:
        j1 = torch.sin(xl) / xl**2 - torch.cos(xl) / xl
        jm, jc = j0, j1
        for k in range(1, l):
            jm, jc = jc, (2 * k + 1) / xl * jc - jm
        rec = jc
    return torch.where(x < l + 1.5, series, rec)


def bessel_zeros(max_l: int, n_zeros: int) -> np.ndarray:
    """``z[l, n]``: the first zeros of j_l, by bisection between the zeros of j_{l-1} (interlacing)."""
    z = np.zeros((max_l + 1, n_zeros + max_l), dtype=np.float64)
    z[0] = np.arange(1, n_zeros + max_l + 1) * math.pi
    for l in range(1, max_l + 1):
        for n in range(n_zeros + max_l - l):
            lo, hi = z[l - 1, n], z[l - 1, n + 1]
            flo = float(_sph_jn(l, torch.tensor(lo, dtype=torch.float64)))
            for _ in range(200):
                mid = 0.5 * (lo + hi)
                fm = float(_sph_jn(l, torch.tensor(mid, dtype=torch.float64)))
                if (fm > 0) == (flo > 0):
                    lo, flo = mid, fm
                else:
                    hi = mid
            z[l, n] = 0.5 * (lo + hi)
    return z[:, :n_zeros]


def laplacian_eigenstates(cutoff: float, max_radial: int, max_angular: int):
    """``(n_per_l, zeros[l][n], norms[l][n])`` of the trimmed Laplacian-eigenstate basis."""
    z = bessel_zeros(max_angular, max_radial + 1)
    threshold = z[0, max_radial] ** 2 * (1 + 1e-12)
    n_per_l = [int((z[l] ** 2 <= threshold).sum()) for l in range(max_angular + 1)]
    zeros, norms = [], []
    for l, n in enumerate(n_per_l):
        zl = z[l, :n]
        jl1 = _sph_jn(l + 1, torch.tensor(zl, dtype=torch.float64)).numpy()
        zeros.append(zl)
        norms.append(1.0 / np.sqrt(cutoff**3 / 2.0 * jl1**2))
    return n_per_l, zeros, norms


def radial_basis(r: torch.Tensor, cutoff: float, zeros, norms) -> List[torch.Tensor]:
    """Per l: ``[pair, n_l]`` values R_nl(r) (no cutoff function)."""
    out = []
    for l, (zl, nl) in enumerate(zip(zeros, norms)):
        cols = [float(nl[n]) * _sph_jn(l, r * (float(zl[n]) / cutoff)) for n in range(len(zl))]
        # an l without radial functions (small max_radial, large max_angular) contributes no features
        out.append(torch.stack(cols, dim=1) if cols else r.new_zeros((r.shape[0], 0)))
    return out


def shifted_cosine(r: torch.Tensor, cutoff: float, width: float) -> torch.Tensor:
    s = ((r - (cutoff - width)) / width).clamp(0.0, 1.0)
    return torch.where(r < cutoff, 0.5 * (1.0 + torch.cos(math.pi * s)), torch.zeros_like(r))


# ---------------------------------------------------------------------------------------------
# angular basis: orthonormal real spherical harmonics of the unit vector, per l [pair, 2l+1]
# ---------------------------------------------------------------------------------------------
def spherical_harmonics(u: torch.Tensor, max_l: int) -> List[torch.Tensor]:
    x, y, z = u[:, 0], u[:, 1], u[:, 2]
    c = [torch.ones_like(x)]
    s = [torch.zeros_like(x)]
    for m in range(1, max_l + 1):
        c.append(x * c[m - 1] - y * s[m - 1])
        s.append(x * s[m - 1] + y * c[m - 1])
    q: Dict[Tuple[int, int], torch.Tensor] = {}
    for m in range(0, max_l + 1):
        q[(m, m)] = torch.ones_like(x) if m == 0 else -(2 * m - 1) * q[(m - 1, m - 1)]
        if m + 1 <= max_l:
            q[(m + 1, m)] = (2 * m + 1) * z * q[(m, m)]
        for l in range(m + 2, max_l + 1):
            q[(l, m)] = ((2 * l - 1) * z * q[(l - 1, m)] - (l + m - 1) * q[(l - 2, m)]) / (l - m)
    out = []
    for l in range(max_l + 1):
        cols = []
        for m in range(-l, l + 1):
            am = abs(m)
            f = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - am) / math.factorial(l + am))
            if m == 0:
                cols.append(f * q[(l, 0)])
            elif m > 0:
                cols.append(math.sqrt(2.0) * f * q[(l, am)] * c[am])
            else:
                cols.append(math.sqrt(2.0) * f * q[(l, am)] * s[am])
        # an l without radial functions (small max_radial, large max_angular) contributes no features
        out.append(torch.stack(cols, dim=1) if cols else r.new_zeros((r.shape[0], 0)))
    return out


# ---------------------------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------------------------
def soap_size(n_per_l: List[int], n_channels: int) -> int:
    return sum((n * n_channels) ** 2 for n in n_per_l)


def state_dict_schema(hypers: dict, n_species: int, n_per_l: List[int]):
    """Ordered ``(key, shape, kind)`` of the plain-tensor parameters of the path (our own key names: the
    reference wraps these modules in metatensor ``ModuleMap``s whose key names are not recoverable here)."""
    legacy = bool(hypers["legacy"])
    c = n_species if legacy else 4
    size = soap_size(n_per_l, c)
    nn, nh = hypers["bpnn"]["num_neurons_per_layer"], hypers["bpnn"]["num_hidden_layers"]
    n_sets = n_species if legacy else 1
    out = []
    if not legacy:
        out.append(("species_embedding.weight", (n_species, 4), "embedding"))
        out.append(("center_encoding.weight", (n_species, size), "embedding"))
    for s in range(n_sets):
        if hypers["bpnn"]["layernorm"]:
            out.append((f"layernorm.{s}.weight", (size,), "norm_w"))
            out.append((f"layernorm.{s}.bias", (size,), "norm_b"))
        for k in range(nh):
            out.append((f"bpnn.{s}.{2 * k}.weight", (nn, size if k == 0 else nn), "linear_w"))
        if (hypers.get("heads") or {}).get("energy") == "mlp":  # MLPHeadMap (soap_bpnn/model.py:117-135): Linear(nn, nn, bias=False) + SiLU
            out.append((f"heads.energy.{s}.0.weight", (nn, nn), "linear_w"))
        out.append((f"last_layers.energy.{s}.weight", (1, nn if nh > 0 else size), "linear_w"))
    return out


def synthetic_params(hypers: dict, n_species: int, n_per_l: List[int], seed: int = 0, dtype=torch.float32):
    """Per-key seeded generator: linear ~ U(-1,1)/sqrt(fan_in), embeddings ~ N(0,1)-like U sqrt(3),
    norm weights 1 + 0.1 U, norm biases 0.1 U (drawn in float64, cast to ``dtype``)."""
    params = {}
    for n, (key, shape, kind) in enumerate(state_dict_schema(hypers, n_species, n_per_l)):
        g = torch.Generator().manual_seed(seed * 100003 + 7000 + n)
        u = torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1
        if kind == "linear_w":
            v = u / math.sqrt(shape[-1])
        elif kind == "embedding":
            v = u * math.sqrt(3.0)
        elif kind == "norm_w":
            v = 1.0 + 0.1 * u
        else:
            v = 0.1 * u
        params[key] = v.to(dtype)
    return params


# ---------------------------------------------------------------------------------------------
# the model
# ---------------------------------------------------------------------------------------------
_BASIS_CACHE: Dict[Tuple[float, int, int], tuple] = {}


def basis(hypers: dict):
    so = hypers["soap"]
    key = (float(so["cutoff"]["radius"]), int(so["max_radial"]), int(so["max_angular"]))
    if key not in _BASIS_CACHE:
        _BASIS_CACHE[key] = laplacian_eigenstates(*key)
    return _BASIS_CACHE[key]


def spherical_expansion(v, centers, sp_index, n_nodes, hypers, species_weights):
    """``spex.SphericalExpansion.forward``: per l ``[N, 2l+1, n_l, c]``."""
    so = hypers["soap"]
    rc, width = float(so["cutoff"]["radius"]), float(so["cutoff"]["width"])
    n_per_l, zeros, norms = basis(hypers)
    r = torch.sqrt((v * v).sum(-1))
    fc = shifted_cosine(r, rc, width)
    rad = radial_basis(r, rc, zeros, norms)
    ang = spherical_harmonics(v / r[:, None], so["max_angular"])
    w = species_weights[sp_index]  # [pair, c] weights of the NEIGHBOUR species
    out = []
    for l in range(so["max_angular"] + 1):
        full = torch.einsum("pm,pn,pc->pmnc", ang[l], rad[l] * fc[:, None], w)
        acc = torch.zeros((n_nodes,) + full.shape[1:], dtype=v.dtype)
        out.append(acc.index_add(0, centers, full))
    return out


def power_spectrum(expansion: List[torch.Tensor]) -> torch.Tensor:
    """``soap_bpnn/modules/power_spectrum.py:125-136``."""
    blocks = []
    for t in expansion:
        t = t.reshape(t.shape[0], t.shape[1], t.shape[2] * t.shape[3])
        blocks.append(torch.einsum("smn,smN->snN", t, t).reshape(t.shape[0], -1))
    return torch.cat(blocks, dim=1)


def soap_bpnn_atomic_energies(params, hypers, atomic_types, positions, cells, centers, neighbors, cell_shifts,
                              species, system_indices, return_features=False):
    """Per-atom energies ``[N]`` of SOAP-BPNN for a scalar target: the default (linear = Identity) head, or with
    ``hypers["heads"] = {"energy": "mlp"}`` the reference's MLP head between the BPNN and the last layer
    (soap_bpnn/model.py:117-135 ``MLPHeadMap``: one bias-free Linear(H, H) + SiLU per centre species; applied at
    model.py:671-672, selected at model.py:1110-1133)."""
    legacy = bool(hypers["legacy"])
    ns = len(atomic_types)
    table = torch.full((max(atomic_types) + 1,), -1, dtype=torch.long)
    table[torch.tensor(atomic_types)] = torch.arange(ns)
    sp = table[species.long()]
    centers, neighbors = centers.long(), neighbors.long()
    shifts = cell_shifts.to(positions.dtype)
    v = positions[neighbors] - positions[centers] + torch.einsum(
        "ab,abc->ac", shifts, cells[system_indices.long()[centers]])
    n = positions.shape[0]
    if legacy:
        species_weights = torch.eye(ns, dtype=positions.dtype)
    else:
        species_weights = params["species_embedding.weight"].to(positions.dtype)
    feats = power_spectrum(spherical_expansion(v, centers, sp[neighbors], n, hypers, species_weights))
    if not legacy:
        feats = feats * params["center_encoding.weight"][sp]
    sets = sp if legacy else torch.zeros_like(sp)
    energies = torch.zeros(n, dtype=positions.dtype)
    nh = hypers["bpnn"]["num_hidden_layers"]
    for s in torch.unique(sets).tolist():
        idx = torch.nonzero(sets == s).squeeze(-1)
        x = feats[idx]
        if hypers["bpnn"]["layernorm"]:
            x = torch.nn.functional.layer_norm(x, (x.shape[1],), params[f"layernorm.{s}.weight"],
                                               params[f"layernorm.{s}.bias"], 1e-5)
        for k in range(nh):
            x = torch.nn.functional.silu(x @ params[f"bpnn.{s}.{2 * k}.weight"].T)
        if (hypers.get("heads") or {}).get("energy") == "mlp":
            x = torch.nn.functional.silu(x @ params[f"heads.energy.{s}.0.weight"].T)
        e = (x @ params[f"last_layers.energy.{s}.weight"].T)[:, 0]
        energies = energies.index_add(0, idx, e)
    if return_features:
        return energies, feats
    return energies


def energy_and_gradient(params, hypers, atomic_types, positions, cells, centers, neighbors, cell_shifts, species,
                        system_indices):
    pos = positions.detach().clone().requires_grad_(True)
    atomic = soap_bpnn_atomic_energies(params, hypers, atomic_types, pos, cells, centers, neighbors, cell_shifts,
                                       species, system_indices)
    n_sys = cells.shape[0]
    energies = torch.zeros(n_sys, dtype=atomic.dtype).index_add(0, system_indices.long(), atomic)
    (grad,) = torch.autograd.grad(energies.sum(), pos)
    return energies.detach(), grad, atomic.detach()
