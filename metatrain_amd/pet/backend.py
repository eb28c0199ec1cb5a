"""Host-side mirror of the reference's ``PETBackend`` (src/metatrain/pet/modules/backend.py:12)
for MI355X: same constructor, same state-dict keys, same three calls

    preprocess(...)            -> Dict[str, Tensor]        (backend.py:238)
    calculate_features(batch)  -> (List[node], List[edge]) (backend.py:344)
    predict(...)               -> (Dict[name, List[Tensor]], {}, {})  (backend.py:420)

but every FLOP runs in libpet_hip (hand-written HIP for gfx950) through the C ABI of
include/pet_hip.h. torch is used for what the reference's callers expect from it: parameter
ownership (``state_dict`` interop with reference checkpoints), device memory, streams, and
the autograd *graph* -- the three calls are three ``torch.autograd.Function`` nodes whose
backward methods call ``pet_backward_{predict,features,geometry}``, so
``torch.autograd.grad(energy, [positions, strain])`` works exactly as in
pet/tests/test_backend.py and utils/output_gradient.py:34-40.

There is no CPU path: CPU tensors raise ``PetHipError``.

Training (``pet/trainer.py:417-467`` through this mirror): in ``train()`` mode with parameters that
require grad, ``predict`` routes the target through ONE fused autograd node ``_EnergyFn(positions, cells, *parameters)`` whose
backward is itself differentiable (``_EnergyGradFn``): ``autograd.grad(E, positions, create_graph=True)``
followed by ``loss.backward()`` fills ``parameter.grad`` -- first-order term from ``pet_backward_train``,
force-loss term from the forward-over-reverse pass ``pet_backward_train2`` -- so torch optimizers and DDP work
on the mirror unchanged. (``metatrain_amd.pet.trainer.TrainStep`` is the faster, fully native step.)

``activation = "SiLU"`` runs on the SwiGLU kernels with the projection uploaded as both halves (exact, see
``runtime.HipModel.load``). Not built (raise loudly): normalization != RMSNorm, transformer_type != PreLN,
featurizer_type != feedforward, the "grid" adaptive-cutoff method, system conditioning, more than one
property per block, per-edge last-layer-feature dicts (returned empty; ``auxiliary_outputs`` gives the per-atom
sums), double backward
through the three staged inference nodes, and stress (strain) terms in a training loss.
"""
from math import prod
from typing import Dict, List, Optional, Tuple

import torch

from .. import runtime as rt
from .._lib import PetHipError


class _ParamsOnly(torch.nn.Module):
    """Container whose sub-modules exist only to own parameters under the reference's names."""

    def forward(self, *args, **kwargs):  # pragma: no cover
        raise PetHipError("parameter container: compute happens in libpet_hip, not in torch modules")


def _feed_forward(d_model: int, dim_ff: int, activation: str = "SwiGLU") -> torch.nn.Module:
    m = _ParamsOnly()
    # SwiGLU: value | gate (transformer.py:28-31); SiLU: one projection (:34-36)
    m.w_in = torch.nn.Linear(d_model, (2 if activation.lower() == "swiglu" else 1) * dim_ff)
    m.w_out = torch.nn.Linear(dim_ff, d_model)
    return m


def _transformer_layer(d: int, dn: int, dff: int, activation: str = "SwiGLU") -> torch.nn.Module:
    # creation order == reference (transformer.py:169-201) so torch.manual_seed reproduces its init
    m = _ParamsOnly()
    att = _ParamsOnly()
    att.input_linear = torch.nn.Linear(d, 3 * d)
    att.output_linear = torch.nn.Linear(d, d)
    m.attention = att
    m.norm_attention = torch.nn.RMSNorm(d)
    m.norm_mlp = torch.nn.RMSNorm(d)
    m.mlp = _feed_forward(d, dff, activation)
    m.center_contraction = torch.nn.Linear(dn, d)
    m.center_expansion = torch.nn.Linear(d, dn)
    m.norm_center_features = torch.nn.RMSNorm(dn)
    m.center_mlp = _feed_forward(dn, 2 * dn, activation)
    return m


def _cartesian_transformer(d: int, dn: int, dff: int, n_layers: int, n_species: int, is_first: bool,
                           activation: str = "SwiGLU"):
    m = _ParamsOnly()
    trans = _ParamsOnly()
    trans.layers = torch.nn.ModuleList([_transformer_layer(d, dn, dff, activation) for _ in range(n_layers)])
    m.trans = trans
    m.edge_embedder = torch.nn.Linear(4, d)
    m.compress = torch.nn.Sequential(
        torch.nn.Linear((2 if is_first else 3) * d, d), torch.nn.SiLU(), torch.nn.Linear(d, d)
    )
    if not is_first:
        m.neighbor_embedder = torch.nn.Embedding(n_species, d)
    return m


# ---------------------------------------------------------------------------------------------
# autograd nodes (first order)
# ---------------------------------------------------------------------------------------------
class _Ctx:
    """What the three nodes share for one preprocess() call."""

    def __init__(self, graph: rt.HipGraph, model: rt.HipModel):
        self.graph = graph
        self.model = model
        csr = graph.csr()
        self.ctr = csr["ctr"].long()
        self.slot = torch.arange(graph.n_edges, device=self.ctr.device) - csr["rowptr"].long()[self.ctr]
        self.fwd: Optional[rt.HipForward] = None
        self.atomic: Optional[torch.Tensor] = None
        self.positions: Optional[torch.Tensor] = None  # the caller's tensors (training node inputs)
        self.cells: Optional[torch.Tensor] = None

    def to_csr(self, nef: torch.Tensor) -> torch.Tensor:
        return nef[self.ctr, self.slot].contiguous()

    def to_nef(self, csr: torch.Tensor) -> torch.Tensor:
        shape = (self.graph.n_nodes, self.graph.max_neighbors) + tuple(csr.shape[1:])
        out = torch.zeros(shape, dtype=csr.dtype, device=csr.device)
        out[self.ctr, self.slot] = csr
        return out


def _no_double_backward(*grads):
    if any(g is not None and g.requires_grad for g in grads):
        raise PetHipError("double backward through the staged inference nodes is not built: make the model "
                          "parameters require grad so that predict() uses the fused training node")


# ---------------------------------------------------------------------------------------------
# fused training nodes: E(positions, cells, theta) with a differentiable backward
# ---------------------------------------------------------------------------------------------
class _EnergyGradFn(torch.autograd.Function):
    """(g_atomic, positions, cells, *theta) -> (dL/dR, dL/dcell, *dL/dtheta) for L = <g_atomic, E_atomic>.
    Its own backward is the second-order pass: grads w.r.t. g_atomic (the JVP of the atomic energies) and
    w.r.t. theta for incoming dL/d(dL/dR); second derivatives w.r.t. positions / cells are not computed."""

    @staticmethod
    def forward(ctx, hctx, keys, g_atomic, positions, cells, *params):
        fw, model = hctx.train_fwd, hctx.model
        ga = g_atomic.detach().reshape(-1).float().contiguous()
        want_theta = any(ctx.needs_input_grad[5:])
        g = hctx.graph
        if want_theta:
            model.zero_grad()
            gpos, gcell = fw.backward_train(ga, want_cell_grad=True)
            grads = model.grads()
            gtheta = tuple(grads[k].to(p.dtype) for k, p in zip(keys, params))
        else:
            gpos, gcell = fw.backward(ga, want_cell_grad=True)
            gtheta = tuple(torch.zeros_like(p) for p in params)
        ctx.hctx, ctx.keys, ctx.ga = hctx, keys, ga
        ctx.param_dtypes = [p.dtype for p in params]
        return (gpos.to(positions.dtype), gcell.to(cells.dtype)) + gtheta

    @staticmethod
    def backward(ctx, u_pos, u_cell, *u_theta):
        if any(u is not None and bool((u != 0).any()) for u in u_theta):
            raise PetHipError("differentiating through parameter gradients is not built into libpet_hip")
        h = ctx.hctx
        model, fw = h.model, h.train_fwd
        n_in = 5 + len(ctx.keys)
        if u_cell is not None and bool((u_cell != 0).any()):
            raise PetHipError("a training loss on dE/dcell (stress) needs the strain second-order terms: not built")
        if u_pos is None:
            return (None,) * n_in
        model.zero_grad()
        tan = fw.backward_train2(ctx.ga, None, u_pos.detach().float().contiguous(), want_tangent=True)
        grads = model.grads()
        gtheta = tuple(grads[k].to(dt) for k, dt in zip(ctx.keys, ctx.param_dtypes))
        return (None, None, tan[:, None], None, None) + gtheta


class _EnergyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hctx, keys, positions, cells, *params):
        hctx.train_fwd = rt.HipForward(hctx.model, hctx.graph, train=True)
        atomic = hctx.train_fwd.forward()
        ctx.hctx, ctx.keys = hctx, keys
        ctx.save_for_backward(positions, cells, *params)
        return atomic[:, None].to(positions.dtype)

    @staticmethod
    def backward(ctx, g_atomic):
        positions, cells, *params = ctx.saved_tensors
        out = _EnergyGradFn.apply(ctx.hctx, ctx.keys, g_atomic, positions, cells, *params)
        return (None, None) + tuple(out)


class _PreprocessFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, positions, cells, hctx, ev, ed, cf):
        ctx.hctx = hctx
        return ev.clone(), ed.clone(), cf.clone()

    @staticmethod
    def backward(ctx, g_ev, g_ed, g_cf):
        _no_double_backward(g_ev, g_ed, g_cf)
        h = ctx.hctx
        g, lib = h.graph, h.graph.lib
        dev = g.workspace.device
        gpos = torch.zeros((g.n_nodes, 3), dtype=torch.float32, device=dev)
        gcell = torch.zeros((g.n_systems, 3, 3), dtype=torch.float32, device=dev)
        if g.n_edges > 0:
            geo = torch.cat([h.to_csr(g_ev), h.to_csr(g_ed)[:, None]], dim=1).float().contiguous()
            gfc = h.to_csr(g_cf).float().contiguous()
            fw = h.fwd or rt.HipForward(h.model, g)
            rt.check(lib.pet_backward_geometry(h.model.handle, g.handle, rt._ptr(fw.workspace), fw.nbytes,
                                               rt._ptr(geo), rt._ptr(gfc), rt._ptr(gpos), rt._ptr(gcell),
                                               rt._stream()))
        return gpos, gcell, None, None, None, None


class _FeaturesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ev, ed, cf, hctx):
        ctx.hctx = hctx
        h = hctx
        h.fwd = rt.HipForward(h.model, h.graph)
        atomic, nf, ef = h.fwd.forward(want_features=True)
        h.atomic = atomic
        return nf, h.to_nef(ef)

    @staticmethod
    def backward(ctx, g_nf, g_ef):
        _no_double_backward(g_nf, g_ef)
        h = ctx.hctx
        g, lib, fw = h.graph, h.graph.lib, h.fwd
        dev = g.workspace.device
        n, m = g.n_nodes, g.max_neighbors
        if g.n_edges == 0:
            z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)  # noqa: E731
            return z(n, m, 3), z(n, m), z(n, m), None
        g_nf = g_nf.float().contiguous()
        g_ef_csr = h.to_csr(g_ef).float().contiguous()
        geo = torch.empty((g.n_edges, 4), dtype=torch.float32, device=dev)
        gfc = torch.empty(g.n_edges, dtype=torch.float32, device=dev)
        rt.check(lib.pet_backward_features(h.model.handle, g.handle, rt._ptr(fw.workspace), fw.nbytes,
                                           rt._ptr(g_nf), rt._ptr(g_ef_csr), rt._ptr(geo), rt._ptr(gfc),
                                           rt._stream()))
        return h.to_nef(geo[:, :3].contiguous()), h.to_nef(geo[:, 3].contiguous()), h.to_nef(gfc), None


class _PredictFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nf, ef, cf, hctx):
        ctx.hctx = hctx
        return hctx.atomic[:, None].clone()

    @staticmethod
    def backward(ctx, g_atomic):
        _no_double_backward(g_atomic)
        h = ctx.hctx
        g, lib, fw = h.graph, h.graph.lib, h.fwd
        dev = g.workspace.device
        ga = g_atomic.reshape(-1).float().contiguous()
        g_nf = torch.empty((g.n_nodes, h.model.hypers["d_node"]), dtype=torch.float32, device=dev)
        g_ef = torch.zeros((g.n_edges, h.model.hypers["d_pet"]), dtype=torch.float32, device=dev)
        g_fc = torch.zeros(g.n_edges, dtype=torch.float32, device=dev)
        rt.check(lib.pet_backward_predict(h.model.handle, g.handle, rt._ptr(fw.workspace), fw.nbytes, rt._ptr(ga),
                                          rt._ptr(g_nf), rt._ptr(g_ef), rt._ptr(g_fc), rt._stream()))
        return g_nf, h.to_nef(g_ef), h.to_nef(g_fc), None


# ---------------------------------------------------------------------------------------------
class PETBackend(torch.nn.Module):
    """MI355X drop-in for ``metatrain.pet.modules.backend.PETBackend`` (plain tensors in / out).

    :param hypers: PET ``ModelHypers`` dict (pet/documentation.py:159-259).
    :param atomic_types: sorted list of atomic numbers the model supports.
    """

    NUM_FEATURE_TYPES: int = 2

    def __init__(self, hypers: dict, atomic_types: List[int]) -> None:
        super().__init__()
        rt.hypers_struct(hypers, atomic_types)  # validates; unsupported variants raise here
        self.hypers = dict(hypers)
        self.atomic_types = list(atomic_types)
        self.nl_is_strict = bool(hypers["long_range"]["enable"])
        self.cutoff = float(hypers["cutoff"])
        self.cutoff_function = hypers["cutoff_function"]
        self.cutoff_width = float(hypers["cutoff_width"])
        self.num_neighbors_adaptive = (float(hypers["num_neighbors_adaptive"])
                                       if hypers["num_neighbors_adaptive"] is not None else None)
        self.adaptive_cutoff_method = hypers.get("adaptive_cutoff_method", "solver")
        self.d_pet, self.d_node = hypers["d_pet"], hypers["d_node"]
        self.d_head, self.d_feedforward = hypers["d_head"], hypers["d_feedforward"]
        self.num_heads = hypers["num_heads"]
        self.num_gnn_layers = hypers["num_gnn_layers"]
        self.num_attention_layers = hypers["num_attention_layers"]
        self.featurizer_type = hypers["featurizer_type"]
        self.num_readout_layers = 1
        n_species = len(atomic_types)

        # first state-dict entry, like the reference (backend.py:63-71)
        self.register_buffer("species_to_species_index", torch.full((max(atomic_types) + 1,), -1))
        for i, species in enumerate(atomic_types):
            self.species_to_species_index[species] = i
        self.gnn_layers = torch.nn.ModuleList([
            _cartesian_transformer(self.d_pet, self.d_node, self.d_feedforward, self.num_attention_layers,
                                   n_species, g == 0, hypers["activation"])
            for g in range(self.num_gnn_layers)
        ])
        self.combination_norms = torch.nn.ModuleList(
            [torch.nn.LayerNorm(2 * self.d_pet) for _ in range(self.num_gnn_layers)])
        self.combination_mlps = torch.nn.ModuleList([
            torch.nn.Sequential(torch.nn.Linear(2 * self.d_pet, 2 * self.d_pet), torch.nn.SiLU(),
                                torch.nn.Linear(2 * self.d_pet, self.d_pet))
            for _ in range(self.num_gnn_layers)
        ])
        self.node_embedders = torch.nn.ModuleList([torch.nn.Embedding(n_species, self.d_node)])
        self.edge_embedder = torch.nn.Embedding(n_species, self.d_pet)
        self.node_heads = torch.nn.ModuleDict()
        self.edge_heads = torch.nn.ModuleDict()
        self.node_last_layers = torch.nn.ModuleDict()
        self.edge_last_layers = torch.nn.ModuleDict()

        self._hip: Dict[Tuple[str, str], rt.HipModel] = {}
        self._hip_version: Dict[Tuple[str, str], int] = {}
        self._ctx: Dict[int, _Ctx] = {}

    # ---- outputs (backend.py:157-236) ----------------------------------------------------------
    def add_output(self, target_name: str, output_shapes: Dict[str, List[int]]) -> None:
        def head(d_in):
            return torch.nn.ModuleList([torch.nn.Sequential(
                torch.nn.Linear(d_in, self.d_head), torch.nn.SiLU(),
                torch.nn.Linear(self.d_head, self.d_head), torch.nn.SiLU())])

        def last():
            return torch.nn.ModuleList([torch.nn.ModuleDict(
                {key: torch.nn.Linear(self.d_head, prod(shape), bias=True) for key, shape in output_shapes.items()})])

        self.node_heads[target_name] = head(self.d_node)
        self.edge_heads[target_name] = head(self.d_pet)
        self.node_last_layers[target_name] = last()
        self.edge_last_layers[target_name] = last()

    def remove_output(self, target_name: str) -> None:
        for d in (self.node_heads, self.edge_heads, self.node_last_layers, self.edge_last_layers):
            if target_name in d:
                del d[target_name]
        for key in [k for k in self._hip if k[0] == target_name]:
            del self._hip[key], self._hip_version[key]

    # ---- packed weights on the device, refreshed when parameters change -------------------------
    def _version(self) -> int:
        return sum(p._version for p in self.parameters()) + sum(id(p) & 0xFFFF for p in self.parameters())

    def _hip_model(self, target: str, block: str) -> rt.HipModel:
        key = (target, block)
        version = self._version()
        if key not in self._hip or self._hip_version[key] != version:
            if target not in self.node_heads:
                raise PetHipError(f"output '{target}' was never registered with add_output")
            w = self.node_last_layers[target][0][block].weight
            if w.shape[0] != 1:
                raise PetHipError("only one property per block is built into libpet_hip for now")
            model = self._hip.get(key) or rt.HipModel(self.hypers, self.atomic_types)
            model.load(dict(self.state_dict()), target, block)
            self._hip[key], self._hip_version[key] = model, version
        return self._hip[key]

    def _training_params(self, target: str, block: str):
        """State-dict keys and tensors of the parameters the packed model of (target, block) was loaded with."""
        model = self._hip[(target, block)]
        named = dict(self.named_parameters())
        keys = [k for k in model._ckeys if k in named]
        return tuple(keys), [named[k] for k in keys]

    def _any_model(self) -> rt.HipModel:
        """preprocess() needs hypers + the species table only; any target's packed model will do."""
        for target in self.node_heads:
            block = next(iter(self.node_last_layers[target][0].keys()))
            return self._hip_model(target, block)
        raise PetHipError("register an output with add_output() before calling preprocess()")

    # ---- the three calls -----------------------------------------------------------------------
    def preprocess(self, positions, centers, neighbors, species, cells, cell_shifts, system_indices,
                   cutoff_width_adaptive: float) -> Dict[str, torch.Tensor]:
        """``PETBackend.preprocess`` (backend.py:238-342): the 12 ``batch_data`` tensors."""
        model = self._any_model()
        if self.num_neighbors_adaptive is not None and abs(
                float(cutoff_width_adaptive) - float(self.hypers.get("cutoff_width_adaptive", 1.0))) > 1e-12:
            raise PetHipError("cutoff_width_adaptive differs from the value in the model hypers (it is part of "
                              "the packed model here)")
        graph = rt.HipGraph(model, positions, cells, centers, neighbors, cell_shifts, species, system_indices)
        batch = graph.export_batch()
        hctx = _Ctx(graph, model)
        hctx.positions, hctx.cells = positions, cells
        if positions.requires_grad or cells.requires_grad:
            ev, ed, cf = _PreprocessFn.apply(positions, cells, hctx, batch["edge_vectors"],
                                             batch["edge_distances"], batch["cutoff_factors"])
            batch["edge_vectors"], batch["edge_distances"], batch["cutoff_factors"] = ev, ed, cf
        dt = positions.dtype
        for k in ("edge_vectors", "edge_distances", "cutoff_factors", "atomic_cutoffs_stats"):
            if batch[k].dtype != dt:
                batch[k] = batch[k].to(dt)
        if len(self._ctx) > 8:  # graphs of stale batches
            self._ctx.pop(next(iter(self._ctx)))
        self._ctx[id(batch["reverse_neighbor_index"])] = hctx
        batch["reverse_neighbor_index"]._pet_hip_ctx = hctx  # keeps the graph alive with the batch
        return batch

    def _ctx_of(self, batch_data: Dict[str, torch.Tensor]) -> _Ctx:
        h = getattr(batch_data["reverse_neighbor_index"], "_pet_hip_ctx", None)
        if h is None:
            raise PetHipError("batch_data does not come from this backend's preprocess()")
        return h

    def calculate_features(self, batch_data: Dict[str, torch.Tensor], capture_diagnostics: bool = False
                           ) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        """``PETBackend.calculate_features`` (backend.py:344-418): node [N,d_node] and edge
        [N,M,d_pet] features of the last GNN layer (feedforward featuriser)."""
        if capture_diagnostics:
            raise PetHipError("diagnostic feature capture is not built into libpet_hip")
        h = self._ctx_of(batch_data)
        nf, ef = _FeaturesFn.apply(batch_data["edge_vectors"], batch_data["edge_distances"],
                                   batch_data["cutoff_factors"], h)
        dt = batch_data["edge_vectors"].dtype
        return [nf.to(dt)], [ef.to(dt)]

    def auxiliary_outputs(self, node_features_list, edge_features_list, batch_data, target: str = "energy",
                          feature: bool = True, last_layer_features: bool = True):
        """Per-atom ``feature`` ``[N, d_node + d_pet]`` and ``mtt::aux::<target>_last_layer_features``
        ``[N, 2 d_head]`` as ``PET._get_output_features`` / ``_get_output_last_layer_features`` assemble them
        (``pet/model.py:730-875``: cutoff-weighted edge sums next to the node parts), computed by
        ``pet_aux_outputs`` from the features of ``calculate_features`` (values only: no autograd node)."""
        h = self._ctx_of(batch_data)
        if h.fwd is None:
            raise PetHipError("auxiliary_outputs() needs calculate_features() on the same batch_data")
        blocks = list(self.node_last_layers[target][0].keys())
        model = self._hip_model(target, blocks[0])
        fw = h.fwd if model is h.model else rt.HipForward(model, h.graph)
        nf = node_features_list[-1].detach().float()
        ef = h.to_csr(edge_features_list[-1].detach()).float()
        return fw.aux_outputs(nf, ef, feature=feature, last_layer_features=last_layer_features)

    def predict(self, node_features_list, edge_features_list, batch_data, cells, system_indices,
                requested_output_names: List[str]):
        """``PETBackend.predict`` (backend.py:420-494): per-block atomic predictions ``[N, 1]``. The two last-layer
        feature dictionaries come back empty: the fused heads never store their hidden rows; the per-atom sums the
        model wrapper builds from them are served by ``auxiliary_outputs``."""
        h = self._ctx_of(batch_data)
        if h.fwd is None:
            raise PetHipError("predict() needs the features of calculate_features() on the same batch_data")
        out: Dict[str, List[torch.Tensor]] = {}
        for name in self.node_last_layers.keys():
            if name not in requested_output_names:
                continue
            if name == "non_conservative_stress":
                raise PetHipError("non_conservative_stress is not built into libpet_hip")
            blocks = list(self.node_last_layers[name][0].keys())
            if len(blocks) != 1:
                raise PetHipError("only single-block targets are built into libpet_hip for now")
            model = self._hip_model(name, blocks[0])
            if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                # training: one fused node over (positions, cells, parameters of this target)
                keys, params = self._training_params(name, blocks[0])
                h2 = _Ctx(h.graph, model)
                pred = _EnergyFn.apply(h2, keys, h.positions, h.cells, *params)
                out[name] = [pred.to(node_features_list[0].dtype)]
                continue
            if model is not h.model:
                # another target than the one the features were computed with: re-run the fused
                # forward with that target's heads (the backbone weights are identical)
                h2 = _Ctx(h.graph, model)
                nf, ef = _FeaturesFn.apply(batch_data["edge_vectors"], batch_data["edge_distances"],
                                           batch_data["cutoff_factors"], h2)
                pred = _PredictFn.apply(nf, ef, batch_data["cutoff_factors"], h2)
            else:
                pred = _PredictFn.apply(node_features_list[0].float(), edge_features_list[0].float(),
                                        batch_data["cutoff_factors"], h)
            out[name] = [pred.to(node_features_list[0].dtype)]
        return out, {}, {}
