"""Host-side construction of the radial basis, as the reference does it (torch-spex builds the
Laplacian-eigenstate basis with scipy and hands a spline to the device code; ``soap_bpnn/model.py:251-264``
selects ``LaplacianEigenstates(max_radial)``). The device evaluates a cubic Hermite spline on a uniform grid (node values, node derivatives and fp64-made chord slopes).

Published definition (Bigi et al., J. Chem. Phys. 157, 234101 (2022)):
    R_nl(r) = N_nl j_l(z_nl r / r_c),   N_nl = [ r_c^3 / 2 * j_{l+1}(z_nl)^2 ]^(-1/2),
    (n, l) kept while z_nl^2 <= z_{max_radial, 0}^2 and l <= max_angular.
"""
from typing import List, Tuple

import numpy as np
from scipy import optimize, special


def bessel_zeros(max_l: int, n_zeros: int) -> np.ndarray:
    """``z[l, n]``: first zeros of the spherical Bessel functions j_l (Brent between the zeros of j_{l-1})."""
    z = np.zeros((max_l + 1, n_zeros + max_l))
    z[0] = np.arange(1, n_zeros + max_l + 1) * np.pi
    for l in range(1, max_l + 1):
        for n in range(n_zeros + max_l - l):
            z[l, n] = optimize.brentq(lambda x: special.spherical_jn(l, x), z[l - 1, n], z[l - 1, n + 1], xtol=1e-14)
    return z[:, :n_zeros]


def laplacian_eigenstates(cutoff: float, max_radial: int, max_angular: int) -> Tuple[List[int], list, list]:
    z = bessel_zeros(max_angular, max_radial + 1)
    threshold = z[0, max_radial] ** 2 * (1 + 1e-12)
    n_per_l = [int((z[l] ** 2 <= threshold).sum()) for l in range(max_angular + 1)]
    zeros = [z[l, :n] for l, n in enumerate(n_per_l)]
    norms = [1.0 / np.sqrt(cutoff**3 / 2.0 * special.spherical_jn(l + 1, zl) ** 2) for l, zl in enumerate(zeros)]
    return n_per_l, zeros, norms


def spline_table(cutoff: float, zeros, norms, n_grid: int = 2049) -> np.ndarray:
    """``[n_grid, F, 4]`` float32: (R(r_k), dR/dr(r_k), chord slope (R(r_k+1) - R(r_k)) / h, 0) on a uniform grid over
    [0, cutoff], functions l-major. The chord slope is formed here in fp64: from the fp32 node values it would carry
    4e-5 |R| of rounding noise (two numbers 2e-3 apart, times 1 / h)."""
    r = np.linspace(0.0, cutoff, n_grid)
    h = cutoff / (n_grid - 1)
    cols = []
    for l, (zl, nl) in enumerate(zip(zeros, norms)):
        for zn, nn in zip(zl, nl):
            k = zn / cutoff
            val = nn * special.spherical_jn(l, k * r)
            chord = np.concatenate([(val[1:] - val[:-1]) / h, [0.0]])
            cols.append(np.stack([val, nn * k * special.spherical_jn(l, k * r, derivative=True), chord,
                                  np.zeros_like(val)], axis=1))
    return np.stack(cols, axis=1).astype(np.float32)
