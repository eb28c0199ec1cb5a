"""Host-side mirror of the reference's ``PETBackend`` (src/metatrain/pet/modules/backend.py:12)
for MI355X: same constructor, same state-dict keys, same three calls

    preprocess(...)            -> Dict[str, Tensor]                         (backend.py:238)
    calculate_features(batch)  -> (List[node], List[edge])                  (backend.py:344)
    predict(...)               -> (Dict[name, List[Tensor]], node_ll, edge_ll)  (backend.py:420)

Every FLOP runs in libpet_hip (hand-written HIP for gfx950) behind the C ABI of include/pet_hip.h. The three calls
are **functions of their arguments**: ``calculate_features`` and ``predict`` rebuild what they need from the
``batch_data`` / feature tensors they are handed (``pet_graph_from_batch``), so a caller that edits features or
``batch_data`` between the calls (LoRA / finetune hooks, diagnostics) gets the edited result, exactly as with the
reference. Each call is one TorchScript-visible method of ``torch.classes.pet_hip.PetHipBackend``
(csrc/torch_ops.cpp) with a C++ autograd node, so this module is ``torch.jit.script``-able
(utils/testing/torchscript.py:39-75) and ``torch.autograd.grad(energy, [positions, strain])`` works as in
pet/tests/test_backend.py and utils/output_gradient.py:34-40. torch supplies parameter ownership (state-dict interop
with reference checkpoints), device memory, streams and the autograd graph -- nothing else.

There is no CPU path: CPU tensors raise.

Training (``pet/trainer.py:417-467`` through this mirror, eager mode): in ``train()`` mode with parameters that
require grad, ``predict`` routes a single-property target through ONE fused autograd node ``_EnergyFn(positions,
cells, *parameters)`` whose backward is itself differentiable (``_EnergyGradFn``): ``autograd.grad(E, positions,
create_graph=True)`` followed by ``loss.backward()`` fills ``parameter.grad`` -- first-order term from
``pet_backward_train``, force-loss term from the forward-over-reverse pass ``pet_backward_train2`` -- so torch
optimizers and DDP work on the mirror unchanged. (``metatrain_amd.pet.trainer.TrainStep`` is the faster, fully native
step.) That node evaluates the batch ``preprocess`` saw, not edited features.

``activation = "SiLU"`` runs on the SwiGLU kernels with the projection uploaded as both halves (exact). Several
properties per block, several blocks per target and several targets are served by ``pet_predict``. Every model size the
reference accepts runs (the tuned kernels are the default size; anything else, ``d_node == d_pet`` included, and any graph
with an atom of more than 127 neighbours takes the size-generic path), and so do the architecture variants older checkpoints
use -- ``normalization = "LayerNorm"``, ``transformer_type = "PostLN"``, ``featurizer_type = "residual"``
(``pet/checkpoints.py:190-205``) --, both adaptive-cutoff methods ("solver" and the legacy "grid") and system conditioning
(charge / spin multiplicity): inference, forces AND training (energy and force loss) for all of them, edge-free batches of
isolated atoms and empty systems included. Not built (raise loudly): diagnostic capture, double backward through the three
inference nodes, long-range features.
"""
from math import prod
from typing import Dict, List, Optional, Tuple

import torch

from .. import runtime as rt
from .._lib import PetHipError

BATCH_KEYS = ["element_indices_nodes", "element_indices_neighbors", "edge_vectors", "edge_distances", "padding_mask",
              "reverse_neighbor_index", "cutoff_factors", "atomic_cutoffs_stats", "centers", "neighbors",
              "nef_to_edges_neighbor", "cell_shifts"]


# ---------------------------------------------------------------------------------------------
# parameter containers under the reference's names (creation order == reference, transformer.py:169-201,414-461,
# so that torch.manual_seed reproduces its initialisation); each lists its parameters for the kernels
# ---------------------------------------------------------------------------------------------
class _FeedForward(torch.nn.Module):
    def __init__(self, d_model: int, dim_ff: int, activation: str):
        super().__init__()
        # SwiGLU: value | gate (transformer.py:28-31); SiLU: one projection (:34-36)
        self.w_in = torch.nn.Linear(d_model, (2 if activation.lower() == "swiglu" else 1) * dim_ff)
        self.w_out = torch.nn.Linear(dim_ff, d_model)

    @torch.jit.export
    def params(self) -> List[torch.Tensor]:
        return [self.w_in.weight, self.w_in.bias, self.w_out.weight, self.w_out.bias]


class _Attention(torch.nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.input_linear = torch.nn.Linear(d, 3 * d)
        self.output_linear = torch.nn.Linear(d, d)

    @torch.jit.export
    def params(self) -> List[torch.Tensor]:
        return [self.input_linear.weight, self.input_linear.bias, self.output_linear.weight, self.output_linear.bias]


class _RMSNorm(torch.nn.RMSNorm):
    """torch.nn.RMSNorm's parameter (same state-dict key: "weight"), listable from TorchScript."""

    @torch.jit.export
    def params(self) -> List[torch.Tensor]:
        return [self.weight]


class _LayerNorm(torch.nn.LayerNorm):
    """torch.nn.LayerNorm's parameters ("weight", "bias": transformer.py:170-176 with norm = "LayerNorm")."""

    @torch.jit.export
    def params(self) -> List[torch.Tensor]:
        return [self.weight, self.bias]


def _norm(kind: str, d: int) -> torch.nn.Module:
    return _LayerNorm(d) if kind == "LayerNorm" else _RMSNorm(d)


class _TransformerLayer(torch.nn.Module):
    def __init__(self, d: int, dn: int, dff: int, activation: str, norm: str):
        super().__init__()
        self.attention = _Attention(d)
        self.norm_attention = _norm(norm, d)
        self.norm_mlp = _norm(norm, d)
        self.mlp = _FeedForward(d, dff, activation)
        self.center_contraction = torch.nn.Linear(dn, d)
        self.center_expansion = torch.nn.Linear(d, dn)
        self.norm_center_features = _norm(norm, dn)
        self.center_mlp = _FeedForward(dn, 2 * dn, activation)

    @torch.jit.export
    def params(self) -> List[torch.Tensor]:
        out = self.attention.params()
        out += self.norm_attention.params()
        out += self.norm_mlp.params()
        out += self.mlp.params()
        out += [self.center_contraction.weight, self.center_contraction.bias, self.center_expansion.weight,
                self.center_expansion.bias]
        out += self.norm_center_features.params()
        out += self.center_mlp.params()
        return out


class _TransformerLayerFlat(torch.nn.Module):
    """``d_node == d_pet`` (transformer.py:189-201): centre contraction / expansion / norm / MLP are ``Identity`` in the
    reference -- no parameters, no state-dict keys; the node features leaving the layer are the centre token."""

    def __init__(self, d: int, dff: int, activation: str, norm: str):
        super().__init__()
        self.attention = _Attention(d)
        self.norm_attention = _norm(norm, d)
        self.norm_mlp = _norm(norm, d)
        self.mlp = _FeedForward(d, dff, activation)

    @torch.jit.export
    def params(self) -> List[torch.Tensor]:
        out = self.attention.params()
        out += self.norm_attention.params()
        out += self.norm_mlp.params()
        out += self.mlp.params()
        return out


class _Transformer(torch.nn.Module):
    def __init__(self, d: int, dn: int, dff: int, n_layers: int, activation: str, norm: str):
        super().__init__()
        if dn != d:
            self.layers = torch.nn.ModuleList([_TransformerLayer(d, dn, dff, activation, norm) for _ in range(n_layers)])
        else:
            self.layers = torch.nn.ModuleList([_TransformerLayerFlat(d, dff, activation, norm) for _ in range(n_layers)])

    @torch.jit.export
    def params(self) -> List[torch.Tensor]:
        out: List[torch.Tensor] = []
        for layer in self.layers:
            out += layer.params()
        return out


class _CartesianTransformerFirst(torch.nn.Module):
    def __init__(self, d: int, dn: int, dff: int, n_layers: int, n_species: int, activation: str, norm: str):
        super().__init__()
        self.trans = _Transformer(d, dn, dff, n_layers, activation, norm)
        self.edge_embedder = torch.nn.Linear(4, d)
        # a ModuleList where the reference has a Sequential: the same "compress.0 / compress.2" keys, indexable in TorchScript
        self.compress = torch.nn.ModuleList([torch.nn.Linear(2 * d, d), torch.nn.SiLU(), torch.nn.Linear(d, d)])

    @torch.jit.export
    def params(self) -> List[torch.Tensor]:
        out = self.trans.params()
        out += [self.edge_embedder.weight, self.edge_embedder.bias]
        out += [self.compress[0].weight, self.compress[0].bias, self.compress[2].weight, self.compress[2].bias]
        return out


class _CartesianTransformerLater(torch.nn.Module):
    def __init__(self, d: int, dn: int, dff: int, n_layers: int, n_species: int, activation: str, norm: str):
        super().__init__()
        self.trans = _Transformer(d, dn, dff, n_layers, activation, norm)
        self.edge_embedder = torch.nn.Linear(4, d)
        self.compress = torch.nn.ModuleList([torch.nn.Linear(3 * d, d), torch.nn.SiLU(), torch.nn.Linear(d, d)])
        self.neighbor_embedder = torch.nn.Embedding(n_species, d)

    @torch.jit.export
    def params(self) -> List[torch.Tensor]:
        out = self.trans.params()
        out += [self.edge_embedder.weight, self.edge_embedder.bias]
        out += [self.compress[0].weight, self.compress[0].bias, self.compress[2].weight, self.compress[2].bias]
        out += [self.neighbor_embedder.weight]
        return out


class _SystemConditioning(torch.nn.Module):
    """Parameters of ``SystemConditioningEmbedding`` (conditioning.py:8-52) under the same names: charge and
    spin-multiplicity embeddings, ``project`` = Linear, SiLU, zero-initialised gate Linear."""

    required_data_keys: List[str] = ["charge", "spin_multiplicity"]

    def __init__(self, d_out: int, max_charge: int, max_spin_multiplicity: int):
        super().__init__()
        self.max_charge = max_charge
        self.max_spin_multiplicity = max_spin_multiplicity
        self.charge_embedding = torch.nn.Embedding(2 * max_charge + 1, d_out)
        self.spin_multiplicity_embedding = torch.nn.Embedding(max_spin_multiplicity, d_out)
        gate = torch.nn.Linear(d_out, d_out)
        torch.nn.init.zeros_(gate.weight)
        torch.nn.init.zeros_(gate.bias)
        self.project = torch.nn.ModuleList([torch.nn.Linear(2 * d_out, d_out), torch.nn.SiLU(), gate])

    @torch.jit.export
    def params(self) -> List[torch.Tensor]:
        return [self.charge_embedding.weight, self.spin_multiplicity_embedding.weight, self.project[0].weight,
                self.project[0].bias, self.project[2].weight, self.project[2].bias]

    @torch.jit.export
    def validate(self, charge: torch.Tensor, spin_multiplicity: torch.Tensor) -> None:
        """conditioning.py:54-80."""
        if bool((charge < -self.max_charge).any()) or bool((charge > self.max_charge).any()):
            raise ValueError("charge values must be in [-max_charge, max_charge]. Increase max_charge in model hypers to "
                             "support wider charge ranges.")
        if bool((spin_multiplicity < 1).any()) or bool((spin_multiplicity > self.max_spin_multiplicity).any()):
            raise ValueError("spin_multiplicity values must be in [1, max_spin_multiplicity]. Increase "
                             "max_spin_multiplicity in model hypers to support higher spin multiplicities.")


class _NoConditioning(torch.nn.Module):
    """``system_conditioning = False`` (the reference holds None): no parameters, nothing to validate."""

    @torch.jit.export
    def params(self) -> List[torch.Tensor]:
        out: List[torch.Tensor] = []
        return out

    @torch.jit.export
    def validate(self, charge: torch.Tensor, spin_multiplicity: torch.Tensor) -> None:
        pass


def _flat_head(d_in: int, d_head: int) -> torch.nn.ModuleList:
    """The head as a ModuleList [Linear, SiLU, Linear, SiLU]: keys "0.weight" / "2.weight" like the reference's
    Sequential, and indexable with constants in TorchScript."""
    return torch.nn.ModuleList([torch.nn.Linear(d_in, d_head), torch.nn.SiLU(), torch.nn.Linear(d_head, d_head),
                                torch.nn.SiLU()])


def process_non_conservative_stress(tensor: torch.Tensor, cells: torch.Tensor, system_indices: torch.Tensor,
                                    num_properties: int) -> torch.Tensor:
    """backend.py:780-813: reshape to 3 x 3 per atom, divide by the cell volume (zero volume = non-periodic: +inf),
    symmetrise. A handful of element-wise ops on [N, 9 P]: plain torch, as in the reference."""
    t = tensor.reshape(-1, 3, 3, num_properties)
    volumes = torch.abs(torch.det(cells))
    volumes = torch.where(volumes == 0.0, torch.full_like(volumes, float("inf")), volumes)
    t = t / volumes[system_indices].unsqueeze(1).unsqueeze(2).unsqueeze(3)
    return (t + t.transpose(1, 2)) / 2.0


# ---------------------------------------------------------------------------------------------
# fused training nodes (eager): E(positions, cells, theta) with a differentiable backward
# ---------------------------------------------------------------------------------------------
class _TrainCtx:
    """What the fused training node needs of one preprocess() call (eager mode only)."""

    def __init__(self, positions, cells, centers, neighbors, cell_shifts, species, system_indices):
        self.positions, self.cells = positions, cells
        self.args = (centers, neighbors, cell_shifts, species, system_indices)
        self.model: Optional[rt.HipModel] = None
        self.graph: Optional[rt.HipGraph] = None
        self.train_fwd: Optional[rt.HipForward] = None


class _EnergyGradFn(torch.autograd.Function):
    """(g_atomic, positions, cells, *theta) -> (dL/dR, dL/dcell, *dL/dtheta) for L = <g_atomic, E_atomic>.
    Its own backward is the second-order pass: grads w.r.t. g_atomic (the JVP of the atomic energies) and
    w.r.t. theta for incoming dL/d(dL/dR); second derivatives w.r.t. positions / cells are not computed."""

    @staticmethod
    def forward(ctx, hctx, keys, g_atomic, positions, cells, *params):
        fw, model = hctx.train_fwd, hctx.model
        ga = g_atomic.detach().reshape(-1).float().contiguous()
        want_theta = any(ctx.needs_input_grad[5:])
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
        if u_pos is None and u_cell is None:
            return (None,) * n_in
        model.zero_grad()
        up = (torch.zeros((fw.graph.n_nodes, 3), device=ctx.ga.device) if u_pos is None
              else u_pos.detach().float().contiguous())
        # a loss on dE/dcell (the stress / strain-gradient term of utils/evaluate_model.py:305-321) rides along as the
        # tangent of the cells
        uc = None if u_cell is None else u_cell.detach().float().contiguous()
        tan = fw.backward_train2(ctx.ga, None, up, want_tangent=True, u_cell=uc)
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


# ---------------------------------------------------------------------------------------------
class PETBackend(torch.nn.Module):
    """MI355X drop-in for ``metatrain.pet.modules.backend.PETBackend`` (plain tensors in / out).

    :param hypers: PET ``ModelHypers`` dict (pet/documentation.py:159-259).
    :param atomic_types: sorted list of atomic numbers the model supports.
    """

    NUM_FEATURE_TYPES: int = 2

    def __init__(self, hypers: dict, atomic_types: List[int]) -> None:
        super().__init__()
        h = rt.hypers_struct(hypers, atomic_types)  # validates; unsupported variants raise here
        fields = [float(getattr(h, name)) for name, _ in h._fields_]
        # PetHipBackend(hypers): the first 16 fields of pet_hypers_t, the SiLU flag, then the remaining fields (normalization,
        # transformer_type, featurizer_type, adaptive_cutoff_method, system_conditioning, max_charge, max_spin_multiplicity)
        self._numbers: List[float] = fields[:16] + [1.0 if hypers["activation"] == "SiLU" else 0.0] + fields[16:]
        self.hypers = dict(hypers)
        self.atomic_types: List[int] = [int(z) for z in atomic_types]
        self.nl_is_strict = bool(hypers["long_range"]["enable"])
        self.cutoff = float(hypers["cutoff"])
        self.cutoff_function: str = hypers["cutoff_function"]
        self.cutoff_width = float(hypers["cutoff_width"])
        self.num_neighbors_adaptive: Optional[float] = (float(hypers["num_neighbors_adaptive"])
                                                        if hypers["num_neighbors_adaptive"] is not None else None)
        self.adaptive_cutoff_method: str = hypers.get("adaptive_cutoff_method", "solver")
        self.cutoff_width_adaptive = float(hypers.get("cutoff_width_adaptive", 1.0))
        self.d_pet: int = hypers["d_pet"]
        self.d_node: int = hypers["d_node"]
        self.d_head: int = hypers["d_head"]
        self.d_feedforward: int = hypers["d_feedforward"]
        self.num_heads: int = hypers["num_heads"]
        self.num_gnn_layers: int = hypers["num_gnn_layers"]
        self.num_attention_layers: int = hypers["num_attention_layers"]
        self.featurizer_type: str = hypers["featurizer_type"]
        # backend.py:93-119: the residual featuriser reads out the features of every GNN layer
        self.num_readout_layers: int = self.num_gnn_layers if self.featurizer_type == "residual" else 1
        n_species = len(atomic_types)
        act = hypers["activation"]

        # first state-dict entry, like the reference (backend.py:63-71)
        self.register_buffer("species_to_species_index", torch.full((max(atomic_types) + 1,), -1))
        for i, species in enumerate(atomic_types):
            self.species_to_species_index[species] = i
        layers: List[torch.nn.Module] = []
        for g in range(self.num_gnn_layers):
            cls = _CartesianTransformerFirst if g == 0 else _CartesianTransformerLater
            layers.append(cls(self.d_pet, self.d_node, self.d_feedforward, self.num_attention_layers, n_species, act,
                              hypers["normalization"]))
        self.gnn_layers = torch.nn.ModuleList(layers)
        n_comb = self.num_gnn_layers if self.featurizer_type == "feedforward" else 0  # backend.py:93-119
        self.combination_norms = torch.nn.ModuleList([torch.nn.LayerNorm(2 * self.d_pet) for _ in range(n_comb)])
        self.combination_mlps = torch.nn.ModuleList([
            torch.nn.ModuleList([torch.nn.Linear(2 * self.d_pet, 2 * self.d_pet), torch.nn.SiLU(),
                                 torch.nn.Linear(2 * self.d_pet, self.d_pet)])
            for _ in range(n_comb)
        ])
        self.node_embedders = torch.nn.ModuleList(
            [torch.nn.Embedding(n_species, self.d_node) for _ in range(self.num_readout_layers)])
        self.edge_embedder = torch.nn.Embedding(n_species, self.d_pet)
        # backend.py:121-130 (None in the reference when off; an empty ModuleList here so that TorchScript sees one type)
        self.has_system_conditioning: bool = bool(hypers.get("system_conditioning", False))
        self.system_conditioning = (_SystemConditioning(self.d_node, int(hypers["max_charge"]),
                                                        int(hypers["max_spin_multiplicity"]))
                                    if self.has_system_conditioning else _NoConditioning())
        self.node_heads = torch.nn.ModuleDict()
        self.edge_heads = torch.nn.ModuleDict()
        self.node_last_layers = torch.nn.ModuleDict()
        self.edge_last_layers = torch.nn.ModuleDict()
        self._train_models: Dict[Tuple[str, str], rt.HipModel] = {}
        self._train_versions: Dict[Tuple[str, str], int] = {}
        self._refresh_core()

    # ---- outputs (backend.py:157-236) ----------------------------------------------------------
    def add_output(self, target_name: str, output_shapes: Dict[str, List[int]]) -> None:
        def last():
            return torch.nn.ModuleList([torch.nn.ModuleDict(
                {key: torch.nn.Linear(self.d_head, prod(shape), bias=True) for key, shape in output_shapes.items()})
                for _ in range(self.num_readout_layers)])

        self.node_heads[target_name] = torch.nn.ModuleList(
            [_flat_head(self.d_node, self.d_head) for _ in range(self.num_readout_layers)])
        self.edge_heads[target_name] = torch.nn.ModuleList(
            [_flat_head(self.d_pet, self.d_head) for _ in range(self.num_readout_layers)])
        self.node_last_layers[target_name] = last()
        self.edge_last_layers[target_name] = last()
        self._refresh_core()

    def remove_output(self, target_name: str) -> None:
        for d in (self.node_heads, self.edge_heads, self.node_last_layers, self.edge_last_layers):
            if target_name in d:
                del d[target_name]
        for key in [k for k in self._train_models if k[0] == target_name]:
            del self._train_models[key], self._train_versions[key]
        self._refresh_core()

    def _refresh_core(self) -> None:
        """(Re)create the TorchScript-visible kernel front end for the current parameter list. The keys are the
        state-dict names in the order ``_params()`` lists the tensors."""
        from .script import load_ops

        load_ops()
        named = {id(p): k for k, p in self.named_parameters()}
        keys = [named[id(p)] for p in self._params()]
        assert len(keys) == len(named) == len(set(keys)), "every parameter exactly once"
        self.core = torch.classes.pet_hip.PetHipBackend(self._numbers, self.atomic_types, keys)

    # ---- the parameters, in one fixed order ----------------------------------------------------
    @torch.jit.export
    def _params(self) -> List[torch.Tensor]:
        out: List[torch.Tensor] = []
        for layer in self.gnn_layers:
            out += layer.params()
        for norm in self.combination_norms:
            out += [norm.weight, norm.bias]
        for mlp in self.combination_mlps:
            out += [mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias]
        for emb in self.node_embedders:
            out += [emb.weight]
        out += [self.edge_embedder.weight]
        out += self.system_conditioning.params()
        for _, heads in self.node_heads.items():
            for head in heads:
                out += [head[0].weight, head[0].bias, head[2].weight, head[2].bias]
        for _, heads in self.edge_heads.items():
            for head in heads:
                out += [head[0].weight, head[0].bias, head[2].weight, head[2].bias]
        for _, layers in self.node_last_layers.items():
            for blocks in layers:
                for _, lin in blocks.items():
                    out += [lin.weight, lin.bias]
        for _, layers in self.edge_last_layers.items():
            for blocks in layers:
                for _, lin in blocks.items():
                    out += [lin.weight, lin.bias]
        return out

    # ---- the three calls -----------------------------------------------------------------------
    @torch.jit.export
    def preprocess(self, positions: torch.Tensor, centers: torch.Tensor, neighbors: torch.Tensor,
                   species: torch.Tensor, cells: torch.Tensor, cell_shifts: torch.Tensor,
                   system_indices: torch.Tensor, cutoff_width_adaptive: float) -> Dict[str, torch.Tensor]:
        """``PETBackend.preprocess`` (backend.py:238-342): the 12 ``batch_data`` tensors."""
        if self.num_neighbors_adaptive is not None and abs(cutoff_width_adaptive - self.cutoff_width_adaptive) > 1e-12:
            raise RuntimeError("cutoff_width_adaptive differs from the value in the model hypers (it is part of the "
                               "packed model here)")
        outs = self.core.preprocess(self._params(), positions, centers, neighbors, species, cells, cell_shifts,
                                    system_indices)
        batch: Dict[str, torch.Tensor] = {
            "element_indices_nodes": outs[0], "element_indices_neighbors": outs[1], "edge_vectors": outs[2],
            "edge_distances": outs[3], "padding_mask": outs[4], "reverse_neighbor_index": outs[5],
            "cutoff_factors": outs[6], "atomic_cutoffs_stats": outs[7], "centers": outs[8], "neighbors": outs[9],
            "nef_to_edges_neighbor": outs[10], "cell_shifts": outs[11],
        }
        if not torch.jit.is_scripting():
            self._stash_training_inputs(batch, positions, cells, centers, neighbors, cell_shifts, species, system_indices)
        return batch

    @torch.jit.unused
    def _stash_training_inputs(self, batch: Dict[str, torch.Tensor], positions, cells, centers, neighbors, cell_shifts,
                               species, system_indices) -> None:
        # the fused training node (eager train() mode) differentiates E w.r.t. the positions themselves
        batch["reverse_neighbor_index"]._pet_hip_train = _TrainCtx(positions, cells, centers, neighbors, cell_shifts,
                                                                    species, system_indices)

    @torch.jit.export
    def calculate_features(self, batch_data: Dict[str, torch.Tensor], capture_diagnostics: bool = False
                           ) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        """``PETBackend.calculate_features`` (backend.py:344-418): node [N, d_node] and edge [N, M, d_pet] features of
        every readout layer -- the last GNN layer (feedforward featuriser) or each of them (residual) -- computed FROM
        ``batch_data`` as given."""
        if capture_diagnostics:
            raise RuntimeError("diagnostic feature capture is not built into libpet_hip")
        conditioning: List[torch.Tensor] = []
        if self.has_system_conditioning:  # backend.py:375-378; the model wrapper puts the three keys into batch_data
            self.system_conditioning.validate(batch_data["charge"], batch_data["spin_multiplicity"])
            conditioning = [batch_data["charge"], batch_data["spin_multiplicity"], batch_data["system_indices"]]
        outs = self.core.calculate_features(
            self._params(), batch_data["element_indices_nodes"], batch_data["element_indices_neighbors"],
            batch_data["edge_vectors"], batch_data["edge_distances"], batch_data["padding_mask"],
            batch_data["reverse_neighbor_index"], batch_data["cutoff_factors"], conditioning)
        return outs[: self.num_readout_layers], outs[self.num_readout_layers:]

    @torch.jit.export
    def predict(self, node_features_list: List[torch.Tensor], edge_features_list: List[torch.Tensor],
                batch_data: Dict[str, torch.Tensor], cells: torch.Tensor, system_indices: torch.Tensor,
                requested_output_names: List[str]
                ) -> Tuple[Dict[str, List[torch.Tensor]], Dict[str, List[torch.Tensor]], Dict[str, List[torch.Tensor]]]:
        """``PETBackend.predict`` (backend.py:420-494): per-block atomic predictions ``[N, P_block]`` and the node /
        edge last-layer features of every requested output, from the features and ``batch_data`` passed in."""
        atomic: Dict[str, List[torch.Tensor]] = {}
        node_ll: Dict[str, List[torch.Tensor]] = {}
        edge_ll: Dict[str, List[torch.Tensor]] = {}
        if not torch.jit.is_scripting():
            if self.training and torch.is_grad_enabled():
                requested_output_names = self._predict_training(node_features_list, batch_data,
                                                                requested_output_names, atomic)
        params = self._params()
        mask = batch_data["padding_mask"]
        cf = batch_data["cutoff_factors"]
        for name, layers in self.node_last_layers.items():
            if name in requested_output_names:  # (no `continue`: TorchScript unrolls loops over ModuleDicts)
                node_ll[name] = []
                edge_ll[name] = []
                block_sums: List[torch.Tensor] = []
                layer_index = 0
                for blocks in layers:
                    block_index = 0
                    for block, _ in blocks.items():
                        outs = self.core.predict(params, name, layer_index, block, node_features_list[layer_index],
                                                 edge_features_list[layer_index], mask, cf)
                        if layer_index == 0:
                            block_sums.append(outs[0])
                        else:
                            block_sums[block_index] = block_sums[block_index] + outs[0]
                        if block_index == 0:
                            node_ll[name].append(outs[1])
                            edge_ll[name].append(outs[2])
                        block_index += 1
                    layer_index += 1
                if name == "non_conservative_stress":
                    num_properties = block_sums[0].shape[1] // 9
                    block_sums[0] = process_non_conservative_stress(block_sums[0], cells, system_indices.to(torch.long),
                                                                    num_properties)
                atomic[name] = block_sums
        return atomic, node_ll, edge_ll

    # ---- eager-only: training through the mirror -------------------------------------------------
    def _version(self) -> int:
        return sum(p._version for p in self.parameters()) + sum(id(p) & 0xFFFF for p in self.parameters())

    @torch.jit.unused
    def _predict_training(self, node_features_list: List[torch.Tensor], batch_data: Dict[str, torch.Tensor],
                          requested_output_names: List[str], out: Dict[str, List[torch.Tensor]]) -> List[str]:
        """train() mode with parameters that require grad: single-property targets go through the fused node
        (double-differentiable); returns the names that are still to be served."""
        if not any(p.requires_grad for p in self.parameters()):
            return requested_output_names
        tctx = getattr(batch_data["reverse_neighbor_index"], "_pet_hip_train", None)
        done: List[str] = []
        for name in self.node_last_layers.keys():
            if name not in requested_output_names:
                continue
            blocks = list(self.node_last_layers[name][0].keys())
            if len(blocks) != 1 or self.node_last_layers[name][0][blocks[0]].weight.shape[0] != 1:
                # several blocks / properties: pet_predict serves them, but its autograd node returns no PARAMETER
                # gradients (the parameters reach it as a plain list), so a loss.backward() would silently leave these
                # heads -- and the backbone's share from them -- without gradients
                heads = [self.node_heads[name], self.edge_heads[name], self.node_last_layers[name],
                         self.edge_last_layers[name]]
                if any(p.requires_grad for h in heads for p in h.parameters()):
                    raise PetHipError(
                        f"training target '{name}' has several blocks or properties: libpet_hip computes parameter "
                        "gradients only for single-property targets (the fused training node); freeze this target's "
                        "heads (requires_grad_(False)) or evaluate it in eval() mode")
                continue
            if tctx is None:
                raise PetHipError("training through the mirror needs the batch_data of this backend's preprocess()")
            key = (name, blocks[0])
            version = self._version()
            if key not in self._train_models or self._train_versions[key] != version:
                model = self._train_models.get(key) or rt.HipModel(self.hypers, self.atomic_types)
                model.load(dict(self.state_dict()), name, blocks[0])
                self._train_models[key], self._train_versions[key] = model, version
            model = self._train_models[key]
            named = dict(self.named_parameters())
            keys = tuple(k for k in model._ckeys if k in named)
            params = [named[k] for k in keys]
            h = _TrainCtx(tctx.positions, tctx.cells, *tctx.args)
            h.model = model
            c, n, s, z, sysidx = tctx.args
            h.graph = rt.HipGraph(model, tctx.positions, tctx.cells, c, n, s, z, sysidx)
            if self.has_system_conditioning:  # backend.py:375-378: the model wrapper put the three keys into batch_data
                self.system_conditioning.validate(batch_data["charge"], batch_data["spin_multiplicity"])
                h.graph.set_conditioning(batch_data["charge"], batch_data["spin_multiplicity"],
                                         batch_data["system_indices"])
            pred = _EnergyFn.apply(h, keys, tctx.positions, tctx.cells, *params)
            out[name] = [pred.to(node_features_list[0].dtype)]
            done.append(name)
        return [n for n in requested_output_names if n not in done]

    # ---- auxiliary per-atom outputs (values only) -------------------------------------------------
    @torch.jit.unused
    def auxiliary_outputs(self, node_features_list, edge_features_list, batch_data, target: str = "energy",
                          feature: bool = True, last_layer_features: bool = True):
        """Per-atom ``feature`` ``[N, L (d_node + d_pet)]`` and ``mtt::aux::<target>_last_layer_features``
        ``[N, 2 L d_head]`` (L readout layers) as ``PET._get_output_features`` / ``_get_output_last_layer_features``
        assemble them (``pet/model.py:730-875``: node parts of all layers, then the cutoff-weighted edge sums; last-layer
        features interleaved node, edge per layer), from the given features."""
        # (this is wrapper-level assembly -- pet/model.py:754-757, :813-823 -- on tensors the kernels produced)
        cf = batch_data["cutoff_factors"].detach()[..., None]
        nf = torch.cat([t.detach() for t in node_features_list], dim=1)
        ef = torch.cat([t.detach() for t in edge_features_list], dim=2)
        feat = torch.cat([nf, (ef * cf).sum(1)], dim=1) if feature else None
        llf = None
        if last_layer_features:
            _, node_ll, edge_ll = self.predict(node_features_list, edge_features_list, batch_data,
                                               torch.zeros((1, 3, 3), device=nf.device), torch.zeros(1, device=nf.device),
                                               [target])
            parts = []
            for a, b in zip(node_ll[target], edge_ll[target]):
                parts += [a, (b * cf).sum(1)]
            llf = torch.cat(parts, dim=1)
        return feat, llf
