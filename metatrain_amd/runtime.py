"""Thin host layer over the C ABI: torch owns device memory and streams, libpet_hip
does the work. Nothing here computes on the CPU and nothing falls back to torch ops.
"""
import ctypes
from ctypes import byref, c_double, c_float, c_int, c_int32, c_int64, c_void_p
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import PetHipError, PetHypers, check


def _stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> c_void_p:
    return c_void_p(0 if t is None else t.data_ptr())


def _require_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t.device.type != "cuda":
            raise PetHipError(
                "metatrain_amd runs on MI355X only: got a tensor on "
                f"'{t.device}'. There is no CPU path in this package."
            )


def hypers_struct(hypers: dict, atomic_types: List[int]) -> PetHypers:
    fn = hypers["cutoff_function"].lower()
    if fn not in ("bump", "cosine"):
        raise ValueError(f"Unknown cutoff function type: {hypers['cutoff_function']}")
    variants = {}
    for key, choices in (("normalization", ("RMSNorm", "LayerNorm")), ("transformer_type", ("PreLN", "PostLN")),
                         ("featurizer_type", ("feedforward", "residual"))):
        if hypers[key] not in choices:  # transformer.py:170-176, :336-341, backend.py:76-81
            raise ValueError(f"Unknown {key}: {hypers[key]!r} (expected one of {choices})")
        variants[key] = choices.index(hypers[key])
    if hypers["activation"] not in ("SwiGLU", "SiLU"):
        raise ValueError(f"Unknown activation flag: {hypers['activation']}")  # transformer.py:342-346
    method = str(hypers.get("adaptive_cutoff_method", "solver")).lower()
    if method not in ("solver", "grid"):  # structures.py:229-251
        raise ValueError(f"adaptive_cutoff_method must be 'grid' or 'solver', got {hypers['adaptive_cutoff_method']}")
    variants["adaptive_cutoff_method"] = 1 if method == "grid" else 0
    if hypers.get("system_conditioning", False):
        variants.update(system_conditioning=1, max_charge=int(hypers["max_charge"]),
                        max_spin_multiplicity=int(hypers["max_spin_multiplicity"]))
    return PetHypers(
        cutoff=float(hypers["cutoff"]),
        cutoff_width=float(hypers["cutoff_width"]),
        cutoff_function=_lib.PET_CUTOFF_BUMP if fn == "bump" else _lib.PET_CUTOFF_COSINE,
        d_pet=int(hypers["d_pet"]),
        d_head=int(hypers["d_head"]),
        d_node=int(hypers["d_node"]),
        d_feedforward=int(hypers["d_feedforward"]),
        num_heads=int(hypers["num_heads"]),
        num_attention_layers=int(hypers["num_attention_layers"]),
        num_gnn_layers=int(hypers["num_gnn_layers"]),
        attention_temperature=float(hypers["attention_temperature"]),
        nl_is_strict=int(bool(hypers["long_range"]["enable"])),
        n_species=len(atomic_types),
        max_atomic_number=max(atomic_types),
        num_neighbors_adaptive=float(hypers["num_neighbors_adaptive"] or 0.0),
        cutoff_width_adaptive=float(hypers.get("cutoff_width_adaptive", 1.0)),
        **variants,
    )


class HipModel:
    """Device-resident packed weights (``pet_model_t``)."""

    def __init__(self, hypers: dict, atomic_types: List[int]):
        self.lib = _lib.load()
        self.hypers = dict(hypers)
        self.atomic_types = list(atomic_types)
        self._h = hypers_struct(hypers, atomic_types)
        self._handle = c_void_p()
        check(self.lib.pet_model_create(byref(self._h), byref(self._handle)))
        self.target: Optional[str] = None

    def __del__(self):
        h = getattr(self, "_handle", None)
        try:
            if h is not None and h.value:
                self.lib.pet_model_destroy(h)
                self._handle = c_void_p()
        except Exception:  # interpreter shutdown: ctypes may already be torn down
            pass

    @property
    def handle(self) -> c_void_p:
        return self._handle

    def load(self, params: Dict[str, torch.Tensor], target: Optional[str], block: Optional[str] = None) -> None:
        """Upload a reference-schema state dict and pack it. ``target`` / ``block`` name the FUSED head
        (what ``pet_forward`` and the native training step evaluate; its last layers must have one property): its keys
        are uploaded with both names replaced by "@". Every other head / last layer of the state dict (other targets,
        blocks with several properties, further readout layers) is uploaded under its own name and served by
        :meth:`HipForward.predict`. ``target=None``: no fused head (features + ``predict`` only)."""
        block = block or target
        self.target = target
        self._fused_block = block
        self._ckeys: Dict[str, tuple] = {}
        self._tied = set()  # SiLU variant: w_in parameters uploaded twice (value half = gate half)
        last_w = params.get(f"node_last_layers.{target}.0.{block}.weight") if target is not None else None
        # the fused target: one property. Its heads and last layers go up under the name "@" -- of readout layer 0, and with
        # the residual featuriser (one readout per GNN layer, backend.py:589-649) of EVERY readout layer, which is what the
        # native training step reads (the fused inference entry points serve one readout layer only)
        fused = last_w is not None and last_w.shape[0] == 1
        self._fused_all_layers = fused and self.hypers["featurizer_type"] == "residual"
        if not fused:
            self.target = self._fused_block = None
        for key, t in params.items():
            _require_cuda(t)
            parts = key.split(".")
            layer_ok = len(parts) > 2 and (parts[2] == "0" or self._fused_all_layers)
            if fused and parts[0] in ("node_heads", "edge_heads") and parts[1] == target and layer_ok:
                parts[1] = "@"
            elif fused and parts[0] in ("node_last_layers", "edge_last_layers") and parts[1] == target \
                    and layer_ok and ".".join(parts[3:-1]) == block:
                parts[1] = "@"
                parts[3:-1] = ["@"]
            ckey = ".".join(parts)
            if key == "species_to_species_index":
                src = t.to(torch.int64).contiguous()
            else:
                src = t.detach().to(torch.float32).contiguous()
                if self.hypers["activation"] == "SiLU" and ".w_in." in key:
                    # activation = "SiLU" (transformer.py:32-49) on the SwiGLU kernels, exactly: with the value and
                    # gate halves both equal to W x + b, v * sigmoid(g) IS silu(W x + b), and the adjoint
                    # W^T (dv + dg) is W^T (d * silu'(a)). The packed model holds [W; W]; the tie is kept here.
                    src = torch.cat([src, src], dim=0).contiguous()
                    self._tied.add(key)
                self._ckeys[key] = (ckey, tuple(src.shape))
            check(self.lib.pet_model_set_param(self._handle, ckey.encode(), _ptr(src), src.numel(), _stream()))
            torch.cuda.current_stream().synchronize()  # src may be a temporary
        check(self.lib.pet_model_finalize(self._handle, _stream()))
        for key in self._tied:  # the fused Adam step keeps the two copies of a tied projection equal
            check(self.lib.pet_model_tie_halves(self._handle, self._ckeys[key][0].encode()))

    def load_species_table(self) -> None:
        """Upload only ``species_to_species_index`` (enough for graph building: the SOAP path shares the
        PET graph kernels but none of the PET weights)."""
        table = torch.full((max(self.atomic_types) + 1,), -1, dtype=torch.int64)
        for i, z in enumerate(self.atomic_types):
            table[z] = i
        src = table.cuda().contiguous()
        check(self.lib.pet_model_set_param(self._handle, b"species_to_species_index", _ptr(src), src.numel(), _stream()))
        torch.cuda.current_stream().synchronize()

    @property
    def num_params(self) -> int:
        return int(self.lib.pet_model_num_params(self._handle))

    # ---- training row (a16): gradient slots, one per uploaded parameter ----
    def zero_grad(self) -> None:
        check(self.lib.pet_model_zero_grad(self._handle, _stream()))

    def grad(self, key: str) -> torch.Tensor:
        """Accumulated dL/d(parameter ``key``) (state-dict key as passed to :meth:`load`)."""
        ckey, shape = self._ckeys[key]
        out = torch.empty(shape, dtype=torch.float32, device="cuda")
        check(self.lib.pet_model_get_grad(self._handle, ckey.encode(), _ptr(out), out.numel(), _stream()))
        if key in self._tied:  # d/dW of a weight used as both halves
            return out[: shape[0] // 2] + out[shape[0] // 2:]
        return out

    def grads(self) -> Dict[str, torch.Tensor]:
        return {k: self.grad(k) for k in self._ckeys}

    def param(self, key: str) -> torch.Tensor:
        ckey, shape = self._ckeys[key]
        out = torch.empty(shape, dtype=torch.float32, device="cuda")
        check(self.lib.pet_model_get_param(self._handle, ckey.encode(), _ptr(out), out.numel(), _stream()))
        if key in self._tied:
            return out[: shape[0] // 2].clone()
        return out

    def param_as_uploaded(self, key: str) -> torch.Tensor:
        """The device copy exactly as uploaded (a tied parameter: both copies, stacked)."""
        ckey, shape = self._ckeys[key]
        out = torch.empty(shape, dtype=torch.float32, device="cuda")
        check(self.lib.pet_model_get_param(self._handle, ckey.encode(), _ptr(out), out.numel(), _stream()))
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Current weights under the reference state-dict keys they were loaded with."""
        return {k: self.param(k) for k in self._ckeys}

    def flat_grad(self) -> torch.Tensor:
        out = torch.empty(self.num_params, dtype=torch.float32, device="cuda")
        check(self.lib.pet_model_flat_grad(self._handle, _ptr(out), out.numel(), 0, _stream()))
        return out

    def set_flat_grad(self, flat: torch.Tensor) -> None:
        _require_cuda(flat)
        flat = flat.to(torch.float32).contiguous()
        check(self.lib.pet_model_flat_grad(self._handle, _ptr(flat), flat.numel(), 1, _stream()))
        self._flat_keepalive = flat  # `flat` may be a temporary: it must outlive the copy queued on the stream (no host sync)

    def optimizer_state(self) -> Dict[str, torch.Tensor]:
        """Adam's moments as flat buffers in upload order (``pet_optimizer_state``): what a checkpoint keeps."""
        m = torch.empty(self.num_params, dtype=torch.float32, device="cuda")
        v = torch.empty_like(m)
        check(self.lib.pet_optimizer_state(self._handle, _ptr(m), _ptr(v), m.numel(), 0, _stream()))
        return {"exp_avg": m, "exp_avg_sq": v, "layout": self.flat_layout()}

    def flat_layout(self):
        """``[(state-dict key, numel)]`` in the order of the flat gradient / Adam buffers (= upload order). Saved with the
        optimizer state and checked on load: two state dicts with another key order or target set have the same total size."""
        import math

        return [(k, int(math.prod(shape))) for k, (_, shape) in self._ckeys.items()]

    def load_optimizer_state(self, state: Dict[str, torch.Tensor]) -> None:
        if "layout" in state and [tuple(x) for x in state["layout"]] != self.flat_layout():
            raise PetHipError("optimizer state was saved for another parameter layout (key order / targets differ): the flat "
                              "Adam moments would be misaligned")
        m = state["exp_avg"].to("cuda", torch.float32).contiguous()
        v = state["exp_avg_sq"].to("cuda", torch.float32).contiguous()
        check(self.lib.pet_optimizer_state(self._handle, _ptr(m), _ptr(v), m.numel(), 1, _stream()))
        torch.cuda.current_stream().synchronize()  # m / v may be temporaries

    def adam_step(self, lr: float, step: int, betas=(0.9, 0.999), eps: float = 1e-8,
                  weight_decay: Optional[float] = None, max_grad_norm: float = 0.0) -> torch.Tensor:
        """clip_grad_norm_ + Adam/AdamW + re-pack; returns the pre-clip gradient norm (device scalar)."""
        norm = torch.empty(1, dtype=torch.float32, device="cuda")
        wd = -1.0 if weight_decay is None else float(weight_decay)
        check(self.lib.pet_adam_step(self._handle, float(lr), float(betas[0]), float(betas[1]), float(eps), wd,
                                     float(max_grad_norm), int(step), _ptr(norm), _stream()))
        return norm


class HipGraph:
    """CSR edge graph (``pet_graph_t``) + the workspace it lives in."""

    def __init__(self, model: HipModel, positions, cells, centers, neighbors, cell_shifts, species,
                 system_indices):
        _require_cuda(positions, cells, centers, neighbors, cell_shifts, species, system_indices)
        self.lib = model.lib
        self.model = model
        dev = positions.device
        self.n_nodes = int(positions.shape[0])
        self.n_systems = int(cells.shape[0])
        self.n_edges_in = int(centers.shape[0])
        # keep the converted inputs alive: the graph build reads them asynchronously
        self._index_dtype = centers.dtype      # batch_data keeps the caller's index dtypes
        self._shift_dtype = cell_shifts.dtype  # (index_select of the inputs, structures.py:268-271)
        self._pos = positions.detach().to(torch.float32).contiguous()
        self._cells = cells.detach().to(torch.float32).contiguous()
        self._ctr = centers.to(torch.int32).contiguous()
        self._nbr = neighbors.to(torch.int32).contiguous()
        self._shift = cell_shifts.to(torch.int32).contiguous()
        self._species = species.to(torch.int32).contiguous()
        self._sys = system_indices.to(torch.int32).contiguous()
        nbytes = int(self.lib.pet_graph_workspace_bytes(self.n_nodes, self.n_edges_in))
        if nbytes < 0:
            raise PetHipError("pet_graph_workspace_bytes failed")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._handle = c_void_p()
        check(self.lib.pet_graph_build(
            model.handle, _ptr(self._pos), _ptr(self._cells), _ptr(self._ctr), _ptr(self._nbr),
            _ptr(self._shift), _ptr(self._species), _ptr(self._sys), self.n_nodes, self.n_edges_in,
            self.n_systems, _ptr(self.workspace), nbytes, byref(self._handle), _stream()))
        self.n_edges = int(self.lib.pet_graph_num_edges(self._handle))
        self.max_neighbors = int(self.lib.pet_graph_max_neighbors(self._handle))

    def __del__(self):
        h = getattr(self, "_handle", None)
        try:
            if h is not None and h.value:
                self.lib.pet_graph_destroy(h)
                self._handle = c_void_p()
        except Exception:
            pass

    @property
    def handle(self) -> c_void_p:
        return self._handle

    def export_batch(self) -> Dict[str, torch.Tensor]:
        """The 12 ``batch_data`` tensors of ``PETBackend.preprocess`` (backend.py:328-341)."""
        dev = self.workspace.device
        n, m, e = self.n_nodes, self.max_neighbors, self.n_edges
        i64, f32 = torch.int64, torch.float32
        out = {
            "element_indices_nodes": torch.empty(n, dtype=i64, device=dev),
            "element_indices_neighbors": torch.empty((n, m), dtype=i64, device=dev),
            "edge_vectors": torch.empty((n, m, 3), dtype=f32, device=dev),
            "edge_distances": torch.empty((n, m), dtype=f32, device=dev),
            "padding_mask": torch.empty((n, m), dtype=torch.uint8, device=dev),
            "reverse_neighbor_index": torch.empty((n, m), dtype=i64, device=dev),
            "cutoff_factors": torch.empty((n, m), dtype=f32, device=dev),
            "atomic_cutoffs_stats": torch.empty(n, dtype=f32, device=dev),
            "centers": torch.empty(e, dtype=i64, device=dev),
            "neighbors": torch.empty(e, dtype=i64, device=dev),
            "nef_to_edges_neighbor": torch.empty(e, dtype=i64, device=dev),
            "cell_shifts": torch.empty((e, 3), dtype=i64, device=dev),
        }
        order = ["element_indices_nodes", "element_indices_neighbors", "edge_vectors", "edge_distances",
                 "padding_mask", "reverse_neighbor_index", "cutoff_factors", "atomic_cutoffs_stats",
                 "centers", "neighbors", "nef_to_edges_neighbor", "cell_shifts"]
        check(self.lib.pet_graph_export_batch(self._handle, *[_ptr(out[k]) for k in order], _stream()))
        out["padding_mask"] = out["padding_mask"].to(torch.bool)
        out["centers"] = out["centers"].to(self._index_dtype)
        out["neighbors"] = out["neighbors"].to(self._index_dtype)
        out["cell_shifts"] = out["cell_shifts"].to(self._shift_dtype)
        return out

    def system_of_atom(self) -> torch.Tensor:
        """``[N]`` int64 structure index of every atom (the batch's ``system_indices``)."""
        return self._sys.long()

    def set_conditioning(self, charge: torch.Tensor, spin_multiplicity: torch.Tensor,
                         system_indices: Optional[torch.Tensor] = None) -> None:
        """``system_conditioning``: per-system total charge and spin multiplicity of this batch
        (``batch_data["charge"]``, ``["spin_multiplicity"]``, ``["system_indices"]``; ``pet_graph_set_conditioning``).
        Range check on the host like ``SystemConditioningEmbedding.validate`` (conditioning.py:54-80)."""
        h = self.model.hypers
        q = charge.detach().to(self.workspace.device, torch.int64).contiguous()
        sm = spin_multiplicity.detach().to(self.workspace.device, torch.int64).contiguous()
        if q.numel() and (int(q.min()) < -h["max_charge"] or int(q.max()) > h["max_charge"]):
            raise ValueError(f"charge values must be in [{-h['max_charge']}, {h['max_charge']}], got min={int(q.min())}, "
                             f"max={int(q.max())}. Increase max_charge in model hypers to support wider charge ranges.")
        if sm.numel() and (int(sm.min()) < 1 or int(sm.max()) > h["max_spin_multiplicity"]):
            raise ValueError(f"spin_multiplicity values must be in [1, {h['max_spin_multiplicity']}], got "
                             f"min={int(sm.min())}, max={int(sm.max())}. Increase max_spin_multiplicity in model hypers to "
                             f"support higher spin multiplicities.")
        si = None if system_indices is None else system_indices.detach().to(self.workspace.device, torch.int64).contiguous()
        # the training passes sum node-feature adjoints per system over contiguous runs of atoms (train.hip k_cond_accum)
        self._conditioning_sorted = si is None or si.numel() < 2 or bool((si[1:] >= si[:-1]).all())
        self._conditioning = (q, sm, si)  # the handle keeps the pointers: keep the tensors alive with it
        check(self.lib.pet_graph_set_conditioning(self.handle, _ptr(q), _ptr(sm), _ptr(si), int(q.numel())))

    @classmethod
    def from_batch(cls, model: "HipModel", batch_data: Dict[str, torch.Tensor]) -> "HipGraph":
        """CSR graph from a ``batch_data`` dictionary (the padded NEF tensors of ``PETBackend.preprocess``,
        backend.py:328-341), whoever produced it: what ``calculate_features`` / ``predict`` need of their argument."""
        self = cls.__new__(cls)
        self.lib, self.model = model.lib, model
        mask = batch_data["padding_mask"]
        _require_cuda(mask)
        dev = mask.device
        self.n_nodes, m = int(mask.shape[0]), int(mask.shape[1])
        self.n_systems, self.n_edges_in = 0, self.n_nodes * m
        i64, f32 = torch.int64, torch.float32
        self._keep = [batch_data["element_indices_nodes"].to(i64).contiguous(),
                      batch_data["element_indices_neighbors"].to(i64).contiguous(),
                      batch_data["edge_vectors"].detach().to(f32).contiguous(),
                      batch_data["edge_distances"].detach().to(f32).contiguous(),
                      mask.to(torch.uint8).contiguous(),
                      batch_data["reverse_neighbor_index"].to(i64).contiguous(),
                      batch_data["cutoff_factors"].detach().to(f32).contiguous()]
        nbytes = int(self.lib.pet_graph_from_batch_workspace_bytes(self.n_nodes, m))
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._handle = c_void_p()
        check(self.lib.pet_graph_from_batch(*[_ptr(t) for t in self._keep], self.n_nodes, m, _ptr(self.workspace), nbytes,
                                            byref(self._handle), _stream()))
        self.n_edges = int(self.lib.pet_graph_num_edges(self._handle))
        self.max_neighbors = m  # the width of the caller's NEF grid (what the NEF views of this graph must have)
        self._sys = None
        return self

    def csr(self) -> Dict[str, torch.Tensor]:
        """Copies of rowptr / ctr / nbr / rev (int32) for inspection."""
        ptrs = [c_void_p() for _ in range(4)]
        check(self.lib.pet_graph_csr(self._handle, *[byref(p) for p in ptrs]))
        torch.cuda.current_stream().synchronize()
        base = self.workspace.data_ptr()
        sizes = [self.n_nodes + 1, self.n_edges, self.n_edges, self.n_edges]
        out = {}
        for name, p, sz in zip(["rowptr", "ctr", "nbr", "rev"], ptrs, sizes):
            off = p.value - base
            out[name] = self.workspace[off : off + 4 * sz].view(torch.int32).clone()
        return out


class HipForward:
    """One forward pass' activations (kept for the backward)."""

    def __init__(self, model: HipModel, graph: HipGraph, train: bool = False):
        self.model, self.graph = model, graph
        self.lib = model.lib
        self.train = train
        if train:
            nbytes = int(self.lib.pet_train_workspace_bytes_for(model.handle, graph.handle))
        else:  # graph-aware: a graph with an atom of more than 127 neighbours runs on the size-generic path
            nbytes = int(self.lib.pet_forward_workspace_bytes_for(model.handle, graph.handle))
        if nbytes < 0:
            raise PetHipError("pet_forward_workspace_bytes failed")
        self.nbytes = nbytes
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=graph.workspace.device)

    def rebind(self, graph: "HipGraph") -> "HipForward":
        """Use this object's workspaces for another graph of the same model (micro-batches walk one allocation)."""
        if self.train:
            need = int(self.lib.pet_train_workspace_bytes_for(self.model.handle, graph.handle))
        else:
            need = int(self.lib.pet_forward_workspace_bytes_for(self.model.handle, graph.handle))
        if need > self.nbytes:
            raise PetHipError(f"workspace of {self.nbytes} bytes is too small for this graph ({need} bytes)")
        self.graph = graph
        return self

    def features(self):
        """``calculate_features`` alone (no heads): node ``[N, d_node]`` and edge ``[E, d_pet]`` (CSR rows) features,
        saved for :meth:`backward_features`."""
        g = self.graph
        dev = self.workspace.device
        nf = torch.empty((g.n_nodes, self.model.hypers["d_node"]), dtype=torch.float32, device=dev)
        ef = torch.empty((g.n_edges, self.model.hypers["d_pet"]), dtype=torch.float32, device=dev)
        check(self.lib.pet_forward(self.model.handle, g.handle, _ptr(self.workspace), self.nbytes, 2 if self.train else 1,
                                   c_void_p(0), _ptr(nf), _ptr(ef), _stream()))
        return nf, ef

    def features_layers(self):
        """``calculate_features`` for every readout layer (``pet_forward_layers``): two lists of
        ``num_readout_layers`` tensors (node ``[N, d_node]``, edge ``[E, d_pet]`` CSR rows), as ``backend.py:344-418``
        returns them (residual featuriser: one pair per GNN layer)."""
        g = self.graph
        dev = self.workspace.device
        n_l = int(self.lib.pet_model_num_readout_layers(self.model.handle))
        nfs = [torch.empty((g.n_nodes, self.model.hypers["d_node"]), dtype=torch.float32, device=dev) for _ in range(n_l)]
        efs = [torch.empty((g.n_edges, self.model.hypers["d_pet"]), dtype=torch.float32, device=dev) for _ in range(n_l)]
        pn = (c_void_p * n_l)(*[t.data_ptr() for t in nfs])
        pe = (c_void_p * n_l)(*[t.data_ptr() for t in efs])
        check(self.lib.pet_forward_layers(self.model.handle, g.handle, _ptr(self.workspace), self.nbytes, 1, pn, pe, n_l,
                                          _stream()))
        return nfs, efs

    def backward_features_layers(self, grad_node_features, grad_edge_features):
        """Adjoint of :meth:`features_layers`: lists of gradients (``None`` = zero) -> ``(d geometry [E,4], d cutoff
        factors [E])`` (``pet_backward_features_layers``)."""
        g = self.graph
        dev = self.workspace.device
        n_l = len(grad_node_features)
        gn = [None if t is None else t.detach().to(torch.float32).contiguous() for t in grad_node_features]
        ge = [None if t is None else t.detach().to(torch.float32).contiguous() for t in grad_edge_features]
        pn = (c_void_p * n_l)(*[0 if t is None else t.data_ptr() for t in gn])
        pe = (c_void_p * n_l)(*[0 if t is None else t.data_ptr() for t in ge])
        geo = torch.zeros((g.n_edges, 4), dtype=torch.float32, device=dev)
        gfc = torch.zeros(g.n_edges, dtype=torch.float32, device=dev)
        check(self.lib.pet_backward_features_layers(self.model.handle, g.handle, _ptr(self.workspace), self.nbytes, pn, pe,
                                                    n_l, _ptr(geo), _ptr(gfc), _stream()))
        return geo, gfc

    def backward_geometry(self, grad_geometry: torch.Tensor, grad_cutoff: torch.Tensor, want_cell_grad: bool = False):
        """``preprocess^T`` (``pet_backward_geometry``): ``(d geometry [E,4], d cutoff factors [E])`` -> dL/dR ``[N,3]``
        (and dL/dcell ``[S,3,3]``)."""
        g = self.graph
        dev = self.workspace.device
        geo = grad_geometry.detach().to(torch.float32).contiguous()
        gfc = grad_cutoff.detach().to(torch.float32).contiguous()
        gpos = torch.zeros((g.n_nodes, 3), dtype=torch.float32, device=dev)
        gcell = torch.zeros((g.n_systems, 3, 3), dtype=torch.float32, device=dev) if want_cell_grad else None
        check(self.lib.pet_backward_geometry(self.model.handle, g.handle, _ptr(self.workspace), self.nbytes, _ptr(geo),
                                             _ptr(gfc), _ptr(gpos), _ptr(gcell), _stream()))
        return (gpos, gcell) if want_cell_grad else gpos

    def forward(self, want_features: bool = False):
        g = self.graph
        dev = self.workspace.device
        atomic = torch.empty(g.n_nodes, dtype=torch.float32, device=dev)
        nf = torch.empty((g.n_nodes, self.model.hypers["d_node"]), dtype=torch.float32, device=dev) if want_features else None
        ef = torch.empty((g.n_edges, self.model.hypers["d_pet"]), dtype=torch.float32, device=dev) if want_features else None
        check(self.lib.pet_forward(self.model.handle, g.handle, _ptr(self.workspace), self.nbytes,
                                   2 if self.train else 1, _ptr(atomic), _ptr(nf), _ptr(ef), _stream()))
        if want_features:
            return atomic, nf, ef
        return atomic

    def aux_outputs(self, node_features: torch.Tensor, edge_features: torch.Tensor, feature: bool = True,
                    last_layer_features: bool = True):
        """The per-atom auxiliary outputs of ``pet/model.py:730-875`` from the features ``forward(want_features=True)``
        returned: ``feature`` ``[N, d_node + d_pet]`` (node features | cutoff-weighted sum of edge features) and the
        target's last-layer features ``[N, 2 d_head]`` (node-head hidden | cutoff-weighted sum of edge-head hidden)."""
        g = self.graph
        dev = self.workspace.device
        hy = self.model.hypers
        n, e = g.n_nodes, g.n_edges
        feat = torch.empty((n, hy["d_node"] + hy["d_pet"]), dtype=torch.float32, device=dev) if feature else None
        llf = torch.empty((n, 2 * hy["d_head"]), dtype=torch.float32, device=dev) if last_layer_features else None
        scratch = torch.empty(e * hy["d_head"] + max(e, n, 1), dtype=torch.float32, device=dev) if last_layer_features else None
        check(self.lib.pet_aux_outputs(self.model.handle, g.handle, _ptr(node_features.contiguous()),
                                       _ptr(edge_features.contiguous()), _ptr(feat), _ptr(llf), _ptr(scratch), _stream()))
        return feat, llf

    def backward(self, grad_atomic: torch.Tensor, want_cell_grad: bool = False):
        g = self.graph
        dev = self.workspace.device
        _require_cuda(grad_atomic)
        ga = grad_atomic.to(torch.float32).contiguous()
        gpos = torch.empty((g.n_nodes, 3), dtype=torch.float32, device=dev)
        gcell = torch.empty((g.n_systems, 3, 3), dtype=torch.float32, device=dev) if want_cell_grad else None
        check(self.lib.pet_backward(self.model.handle, g.handle, _ptr(self.workspace), self.nbytes, _ptr(ga),
                                    _ptr(gpos), _ptr(gcell), _stream()))
        if want_cell_grad:
            return gpos, gcell
        return gpos

    def _check_conditioning_for_training(self) -> None:
        if self.model.hypers.get("system_conditioning") and not getattr(self.graph, "_conditioning_sorted", True):
            raise ValueError("system conditioning: training needs non-decreasing system_indices (concatenate_structures "
                             "order); the per-system sums of the conditioning gradients run over contiguous atoms")

    def backward_train(self, grad_atomic: torch.Tensor, want_position_grad: bool = False,
                       want_cell_grad: bool = False):
        """loss.backward() for dL/d(atomic prediction) = ``grad_atomic``: accumulates dL/dtheta into
        the model's gradient slots (``HipModel.grad``); optionally also returns dL/dR."""
        if not self.train:
            raise PetHipError("backward_train needs HipForward(..., train=True)")
        self._check_conditioning_for_training()
        g = self.graph
        _require_cuda(grad_atomic)
        ga = grad_atomic.to(torch.float32).contiguous()
        dev = self.workspace.device
        gpos = torch.empty((g.n_nodes, 3), dtype=torch.float32, device=dev) if want_position_grad or want_cell_grad else None
        gcell = torch.empty((g.n_systems, 3, 3), dtype=torch.float32, device=dev) if want_cell_grad else None
        check(self.lib.pet_backward_train(self.model.handle, g.handle, _ptr(self.workspace), self.nbytes,
                                          _ptr(ga), _ptr(gpos), _ptr(gcell), _stream()))
        if want_cell_grad:
            return gpos, gcell
        return gpos

    def backward_train2(self, lambda_atomic: torch.Tensor, nu_atomic: Optional[torch.Tensor], u: torch.Tensor,
                        want_tangent: bool = False, u_cell: Optional[torch.Tensor] = None):
        """Second-order reverse pass (loss on dE/dR, and with ``u_cell`` [S,3,3] on dE/dcell: the stress term):
        accumulates d/dtheta [ sum_i nu_i E_i + <u, dE/dR> + <u_cell, dE/dcell> ] where the gradients were taken with
        seeds ``lambda_atomic``; optionally returns dE_i/d(eps) along (dR, dcell) = (u, u_cell)."""
        if not self.train:
            raise PetHipError("backward_train2 needs HipForward(..., train=True)")
        self._check_conditioning_for_training()
        g = self.graph
        dev = self.workspace.device
        _require_cuda(lambda_atomic, u)
        n2 = int(self.lib.pet_train2_workspace_bytes_for(self.model.handle, g.handle))
        if getattr(self, "workspace2", None) is None or self.workspace2.numel() < n2:
            self.workspace2 = torch.empty(n2, dtype=torch.uint8, device=dev)
        la = lambda_atomic.to(torch.float32).contiguous()
        nu = None if nu_atomic is None else nu_atomic.to(torch.float32).contiguous()
        uu = u.to(torch.float32).contiguous()
        tan = torch.empty(g.n_nodes, dtype=torch.float32, device=dev) if want_tangent else None
        uc = None if u_cell is None else u_cell.to(dev, torch.float32).reshape(g.n_systems, 3, 3).contiguous()
        check(self.lib.pet_backward_train2_cell(self.model.handle, g.handle, _ptr(self.workspace), self.nbytes,
                                                _ptr(self.workspace2), self.workspace2.numel(), _ptr(la), _ptr(nu),
                                                _ptr(uu), _ptr(uc), _ptr(tan), _stream()))
        return tan

    def sum_over_atoms(self, atomic: torch.Tensor) -> torch.Tensor:
        out = torch.zeros(self.graph.n_systems, dtype=torch.float32, device=atomic.device)
        check(self.lib.pet_sum_over_atoms(self.graph.handle, _ptr(atomic), _ptr(out), _stream()))
        return out


def _head_names(model: HipModel, target: str, block: Optional[str], readout_zero: bool = True):
    block = block or target
    if model.target is not None and target == model.target and block == model._fused_block and \
            (readout_zero or getattr(model, "_fused_all_layers", False)):
        return b"@", b"@"
    return target.encode(), block.encode()


def predict(model: HipModel, graph: HipGraph, node_features: torch.Tensor, edge_features: torch.Tensor,
            target: str, block: Optional[str] = None, readout_layer: int = 0,
            cutoff_factors: Optional[torch.Tensor] = None, want_hidden: bool = False):
    """``PETBackend.predict`` for one (target, readout layer, block) as a function of its arguments (``pet_predict``):
    per-atom predictions ``[N, P]`` from the GIVEN node ``[N, d_node]`` / edge ``[E, d_pet]`` (CSR rows) features."""
    _require_cuda(node_features)
    lib = model.lib
    tname, bname = _head_names(model, target, block, readout_layer == 0)
    p = int(lib.pet_model_block_properties(model.handle, tname, readout_layer, bname))
    if p < 1:
        raise PetHipError(f"no head / last layer uploaded for target '{target}', readout layer {readout_layer}, block "
                          f"'{block or target}'")
    dev = node_features.device
    n, e, dh = graph.n_nodes, graph.n_edges, model.hypers["d_head"]
    nf = node_features.detach().to(torch.float32).contiguous()
    ef = edge_features.detach().to(torch.float32).contiguous()
    fc = None if cutoff_factors is None else cutoff_factors.detach().to(torch.float32).contiguous()
    atomic = torch.empty((n, p), dtype=torch.float32, device=dev)
    hn = torch.empty((n, dh), dtype=torch.float32, device=dev) if want_hidden else None
    he = torch.empty((e, dh), dtype=torch.float32, device=dev) if want_hidden else None
    scratch = torch.empty(int(lib.pet_predict_scratch_floats(n, e)), dtype=torch.float32, device=dev)
    check(lib.pet_predict(model.handle, graph.handle, tname, readout_layer, bname, _ptr(nf), _ptr(ef), _ptr(fc), _ptr(atomic),
                          _ptr(hn), _ptr(he), _ptr(scratch), _stream()))
    return (atomic, hn, he) if want_hidden else atomic


def predict_backward(model: HipModel, graph: HipGraph, node_features: torch.Tensor, edge_features: torch.Tensor,
                     grad_atomic: torch.Tensor, target: str, block: Optional[str] = None, readout_layer: int = 0,
                     cutoff_factors: Optional[torch.Tensor] = None):
    """Adjoint of :func:`predict` from the same inputs: ``(d node features, d edge features (CSR), d cutoff factors)``."""
    lib = model.lib
    tname, bname = _head_names(model, target, block, readout_layer == 0)
    dev = node_features.device
    n, e = graph.n_nodes, graph.n_edges
    nf = node_features.detach().to(torch.float32).contiguous()
    ef = edge_features.detach().to(torch.float32).contiguous()
    fc = None if cutoff_factors is None else cutoff_factors.detach().to(torch.float32).contiguous()
    ga = grad_atomic.detach().to(torch.float32).reshape(n, -1).contiguous()
    g_nf = torch.empty((n, model.hypers["d_node"]), dtype=torch.float32, device=dev)
    g_ef = torch.zeros((e, model.hypers["d_pet"]), dtype=torch.float32, device=dev)
    g_fc = torch.zeros(e, dtype=torch.float32, device=dev)
    scratch = torch.empty(int(lib.pet_predict_scratch_floats(n, e)), dtype=torch.float32, device=dev)
    check(lib.pet_predict_backward(model.handle, graph.handle, tname, readout_layer, bname, _ptr(nf), _ptr(ef), _ptr(fc),
                                   _ptr(ga), _ptr(g_nf), _ptr(g_ef), _ptr(g_fc), _ptr(scratch), _stream()))
    return g_nf, g_ef, g_fc


_nl_guess: Dict[int, int] = {}  # atoms in the batch -> pairs found last time (sizes the next optimistic call)


def neighbor_list_batch(positions: torch.Tensor, cells, pbcs, first_atom: List[int], cutoff: float,
                        want_vectors: bool = True):
    """Device neighbour lists of all systems of a batch in one set of launches (``pet_nl_build_batch``):
    ``positions [N,3]`` concatenated, system ``s`` owning atoms ``first_atom[s] .. first_atom[s+1]-1``; ``cells``
    ``[S,3,3]`` (host or device tensor, or list), ``pbcs`` ``[S][3]`` booleans. Returns ``(pairs [E,5] int32, vectors
    [E,3] fp32 or None)`` with GLOBAL atom indices, rows ``(i, j, Sa, Sb, Sc)`` grouped by ``i``.
    One call when the pair buffer sized from the previous call of this size suffices, else a second one."""
    _require_cuda(positions)
    lib = _lib.load()
    pos = positions.detach().to(torch.float32).contiguous()
    n, n_sys = int(pos.shape[0]), len(first_atom) - 1
    cells_t = torch.as_tensor(cells) if not isinstance(cells, torch.Tensor) else cells
    flat = [float(x) for x in cells_t.detach().cpu().reshape(-1).tolist()]
    h_cells = (c_float * (9 * n_sys))(*flat)
    h_pbc = (c_int32 * (3 * n_sys))(*[int(bool(x)) for p in pbcs for x in p])
    h_first = (c_int64 * (n_sys + 1))(*[int(x) for x in first_atom])
    ws = torch.empty(int(lib.pet_nl_batch_workspace_bytes(n, n_sys)), dtype=torch.uint8, device=pos.device)
    count = c_int64(0)
    cap = max(1024, int(1.1 * _nl_guess.get(n, 32 * n)))
    for _ in range(2):
        pairs = torch.empty((cap, 5), dtype=torch.int32, device=pos.device)
        vectors = torch.empty((cap, 3), dtype=torch.float32, device=pos.device) if want_vectors else None
        rc = lib.pet_nl_build_batch(_ptr(pos), h_cells, h_pbc, h_first, n_sys, float(cutoff), _ptr(ws), _ptr(pairs),
                                    _ptr(vectors), cap, byref(count), _stream())
        e = int(count.value)
        if rc == 0:
            _nl_guess[n] = e
            return pairs[:e], (vectors[:e] if want_vectors else None)
        if e <= cap:  # a real error, not a short buffer
            check(rc)
        cap = e
    check(rc)


def neighbor_list(positions: torch.Tensor, cell: torch.Tensor, pbc, cutoff: float):
    """Device neighbour list of one system: ``(pairs [E,5] int32, vectors [E,3] fp32)`` with rows
    ``(i, j, Sa, Sb, Sc)`` grouped by ``i`` (replaces vesin, utils/neighbor_lists.py:131-135)."""
    return neighbor_list_batch(positions, torch.as_tensor(cell).reshape(1, 3, 3), [pbc], [0, int(positions.shape[0])],
                               cutoff)


def profile(enable: bool, stage: Optional[str] = None) -> None:
    """Bracket every stage (or only ``stage``) of forward/backward with HIP events."""
    lib = _lib.load()
    check(lib.pet_profile_reset())
    check(lib.pet_profile_select((stage or "").encode()))
    check(lib.pet_profile_enable(1 if enable else 0))


def profile_report() -> List[dict]:
    lib = _lib.load()
    n_max = 64
    names = ((ctypes.c_char * 64) * n_max)()
    ms = (c_double * n_max)()
    calls = (c_int64 * n_max)()
    flops = (c_double * n_max)()
    nbytes = (c_double * n_max)()
    n = c_int(0)
    check(lib.pet_profile_report(n_max, names, ms, calls, flops, nbytes, byref(n)))
    return [
        {"name": names[i].value.decode(), "total_ms": ms[i], "calls": calls[i], "flops": flops[i],
         "bytes": nbytes[i]}
        for i in range(n.value)
    ]


def config_set(key: str, value: int) -> None:
    """Runtime switches of the library ("side_stream", "trr")."""
    check(_lib.load().pet_config_set(key.encode(), int(value)))
