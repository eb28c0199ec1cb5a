"""CPU tests of the SOAP-BPNN row (a17/a18): the oracle's basis against its published definition and
scipy, and the product's host-side radial tables (scipy) against the oracle's independent implementation.
Parity against torch-spex itself is UNPINNED (oracle/soap.py header)."""
import numpy as np
import pytest
import scipy.special as sps
import torch

from oracle import soap as osoap


@pytest.fixture(scope="module")
def basis():
    return osoap.basis(osoap.DEFAULT_HYPERS)


def test_trimmed_laplacian_eigenstate_basis(basis):
    n_per_l, zeros, norms = basis
    assert n_per_l == [8, 7, 7, 6, 6, 5, 5]                       # max_radial 7, max_angular 6
    assert osoap.soap_size(n_per_l, 4) == 4544
    for l, zl in enumerate(zeros):
        assert np.abs(sps.spherical_jn(l, zl)).max() < 1e-12       # they are zeros of j_l
    r = torch.linspace(0, 5, 20001, dtype=torch.float64)[1:]
    rad = osoap.radial_basis(r, 5.0, zeros, norms)
    for l in (0, 3, 6):                                           # orthonormal with weight r^2 on [0, rc]
        gram = (rad[l].T * (r * r)) @ rad[l] * (5 / 20000)
        assert np.abs(gram.numpy() - np.eye(gram.shape[0])).max() < 1e-6


def test_spherical_harmonics_addition_theorem():
    gen = torch.Generator().manual_seed(0)
    u = torch.randn(16, 3, generator=gen, dtype=torch.float64)
    u = u / u.norm(dim=1, keepdim=True)
    w = torch.randn(16, 3, generator=gen, dtype=torch.float64)
    w = w / w.norm(dim=1, keepdim=True)
    yu, yw = osoap.spherical_harmonics(u, 6), osoap.spherical_harmonics(w, 6)
    cosg = (u * w).sum(1).numpy()
    for l in range(7):
        lhs = (yu[l] * yw[l]).sum(1).numpy()
        rhs = (2 * l + 1) / (4 * np.pi) * sps.eval_legendre(l, cosg)
        np.testing.assert_allclose(lhs, rhs, atol=1e-12)


def test_product_radial_tables_match_the_oracle(basis):
    """metatrain_amd/soap_bpnn/radial.py (scipy, what the device spline is built from) and the oracle's
    torch implementation are independent statements of the same basis."""
    from metatrain_amd.soap_bpnn import radial

    n_per_l, zeros, norms = radial.laplacian_eigenstates(5.0, 7, 6)
    assert n_per_l == basis[0]
    for a, b in zip(zeros, basis[1]):
        np.testing.assert_allclose(a, b, rtol=1e-12)
    for a, b in zip(norms, basis[2]):
        np.testing.assert_allclose(a, b, rtol=1e-10)
    table = radial.spline_table(5.0, zeros, norms, 513)
    assert table.shape == (513, 44, 4) and table.dtype == np.float32
    # third component: chord slope of the interval that starts at the node (made in fp64 on the host)
    np.testing.assert_allclose(table[:-1, :, 2], np.diff(table[:, :, 0].astype(np.float64), axis=0) / (5.0 / 512), atol=2e-4)
    assert np.all(table[-1, :, 2] == 0) and np.all(table[:, :, 3] == 0)
    r = torch.linspace(0, 5, 513, dtype=torch.float64).requires_grad_(True)
    rad = torch.cat(osoap.radial_basis(r, 5.0, basis[1], basis[2]), dim=1)
    np.testing.assert_allclose(table[:, :, 0], rad.detach().numpy(), atol=2e-6)
    (d,) = torch.autograd.grad(rad[:, 17].sum(), r)
    np.testing.assert_allclose(table[:, 17, 1], d.numpy(), atol=2e-5)


def test_oracle_energy_is_rotation_and_permutation_invariant():
    from oracle import nl as onl
    from oracle import pet as opet

    hypers = dict(osoap.DEFAULT_HYPERS)
    types = [1, 6, 7, 8]
    n_per_l = osoap.basis(hypers)[0]
    params = osoap.synthetic_params(hypers, 4, n_per_l, 0, torch.float64)
    pos, z, cell = opet.random_box(24, seed=3, dtype=torch.float64)
    pos = pos * 0.6  # a cluster (no periodicity): rotations are exact symmetries
    i, j, s, _ = onl.neighbor_list(pos.numpy(), np.eye(3) * 100.0, [False] * 3, 5.0)
    args = (torch.eye(3, dtype=torch.float64)[None] * 100.0, torch.tensor(i), torch.tensor(j), torch.tensor(s).long(), z,
            torch.zeros(24, dtype=torch.long))
    e0 = osoap.soap_bpnn_atomic_energies(params, hypers, types, pos, *args)
    q, _ = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1)))
    e1 = osoap.soap_bpnn_atomic_energies(params, hypers, types, pos @ q.T, *args)
    np.testing.assert_allclose(e0.numpy(), e1.numpy(), rtol=1e-9, atol=1e-12)


def test_power_spectrum_against_the_reference_module():
    """``tests/golden/soap_ps_box24.npz`` (``make_golden.py --soap-ps``): the reference's ``soap_bpnn/modules/power_spectrum.py``
    imported unchanged and run on the oracle's spherical expansion (torch-spex itself is not installable: parity of the
    EXPANSION stays unpinned). What this pins on reference-run data: the per-l contraction, the (l, n, channel, n',
    channel') feature order and the ``center_type`` block split of the legacy model."""
    import os

    from oracle import nl as onl  # noqa: F401  (the fixture stores its own pair list)

    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "soap_ps_box24.npz")))
    pos, z, cell = torch.tensor(g["in_positions"]).double(), torch.tensor(g["in_species"]), torch.tensor(g["in_cell"]).double()
    table = torch.full((9,), -1, dtype=torch.long)
    table[torch.tensor([1, 6, 7, 8])] = torch.arange(4)
    sp = table[z.long()]
    for tag, legacy in (("legacy", True), ("alchemical", False)):
        hypers = dict(osoap.DEFAULT_HYPERS, legacy=legacy)
        pairs = torch.tensor(g[f"{tag}_pairs"]).long()
        i, j, sh = pairs[:, 0], pairs[:, 1], pairs[:, 2:5].double()
        v = pos[j] - pos[i] + sh @ cell
        w = torch.eye(4, dtype=torch.float64) if legacy else torch.tensor(g["alchemical_species_embedding"])
        feats = osoap.power_spectrum(osoap.spherical_expansion(v, i, sp[j], len(z), hypers, w))
        ref = g[f"{tag}_power_spectrum"].astype(np.float64)
        assert feats.shape == ref.shape == (len(z), 4544)
        assert np.abs(feats.numpy() - ref).max() < 2e-7 * np.abs(ref).max()  # the fixture is stored in fp32
