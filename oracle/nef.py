"""
Integer index oracle (numpy) -- test infrastructure, never imported by the product.

Restates, on numpy arrays, the integer kernels of the reference PET preprocessing:

* ``get_nef_indices``                 ``pet/modules/nef.py:34-85``
* ``get_corresponding_edges``         ``pet/modules/nef.py:88-166``
* ``compute_reversed_neighbor_list``  ``pet/modules/nef.py:221-251``
* the integer tail of ``compute_batch_tensors``
  ``pet/modules/structures.py:265-294, 320-363``

The restatement is deliberately *not* the reference's algorithm where the
contract allows a plainer one (a Python dict instead of the int64 hash + two
argsorts for the ij -> ji pairing): outputs are identical for every valid full
neighbour list, which is what the golden vectors in ``tests/golden`` check.
All outputs are bit-exact quantities (int64 / bool).
"""

from typing import Dict

import numpy as np


def nef_indices(centers: np.ndarray, n_nodes: int):
    """``nef.py:34-85``: stable sort of edges by centre -> padded [N, M] edge ids.

    :return: ``(nef_indices [N,M] i64 (0 on pads), nef_to_edges_neighbor [E] i64,
        nef_mask [N,M] bool, num_neighbors [N] i64)``
    """
    centers = np.asarray(centers, dtype=np.int64)
    n_edges = centers.shape[0]
    num_neighbors = np.bincount(centers, minlength=n_nodes).astype(np.int64)
    m = int(num_neighbors.max()) if n_nodes > 0 else 0
    order = np.argsort(centers, kind="stable")
    starts = np.cumsum(num_neighbors) - num_neighbors
    sorted_centers = centers[order]
    position_within = np.arange(n_edges, dtype=np.int64) - starts[sorted_centers]
    nef = np.zeros((n_nodes, m), dtype=np.int64)
    nef[sorted_centers, position_within] = order
    slot = np.empty(n_edges, dtype=np.int64)
    slot[order] = position_within
    mask = np.arange(m)[None, :] < num_neighbors[:, None]
    return nef, slot, mask, num_neighbors


def corresponding_edges(
    centers: np.ndarray, neighbors: np.ndarray, cell_shifts: np.ndarray
) -> np.ndarray:
    """``nef.py:88-166``: index of the edge (j, i, -S) for every edge (i, j, S)."""
    table: Dict[tuple, int] = {}
    for e in range(len(centers)):
        key = (int(centers[e]), int(neighbors[e])) + tuple(int(x) for x in cell_shifts[e])
        table[key] = e
    out = np.empty(len(centers), dtype=np.int64)
    for e in range(len(centers)):
        key = (int(neighbors[e]), int(centers[e])) + tuple(
            -int(x) for x in cell_shifts[e]
        )
        out[e] = table[key]
    return out


def reverse_neighbor_index(
    centers: np.ndarray,
    neighbors: np.ndarray,
    cell_shifts: np.ndarray,
    n_nodes: int,
) -> Dict[str, np.ndarray]:
    """Integer tail of ``structures.py:320-363``.

    :return: dict with ``nef_indices, nef_to_edges_neighbor, padding_mask,
        reverse_neighbor_index`` exactly as found in the reference ``batch_data``.
    """
    nef, slot, mask, _ = nef_indices(centers, n_nodes)
    m = nef.shape[1]
    if len(centers) == 0:
        return {
            "nef_indices": nef,
            "nef_to_edges_neighbor": slot,
            "padding_mask": mask,
            "reverse_neighbor_index": np.zeros((n_nodes, m), dtype=np.int64),
        }
    corr = corresponding_edges(centers, neighbors, cell_shifts)
    # nef.py:245-249
    reversed_nl = slot[corr[nef]]
    reversed_nl = np.where(mask, reversed_nl, 0)
    # structures.py:344-363
    neighbors_index = np.asarray(neighbors, dtype=np.int64)[nef]
    rni = neighbors_index * m + reversed_nl
    flat_mask = mask.reshape(-1)
    padded_unique = np.cumsum(~flat_mask) - 1
    rni = np.where(flat_mask, rni.reshape(-1), padded_unique).reshape(n_nodes, m)
    return {
        "nef_indices": nef,
        "nef_to_edges_neighbor": slot,
        "padding_mask": mask,
        "reverse_neighbor_index": rni.astype(np.int64),
    }
