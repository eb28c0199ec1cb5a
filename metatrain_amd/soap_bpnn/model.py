"""``SoapBpnnHip``: the plain-tensor SOAP-BPNN path on MI355X behind ``include/soap_hip.h``.

Mirrors what ``SoapBpnn.forward`` computes below its metatensor wrapping (``soap_bpnn/model.py:540-595``
and the scalar last layer ``:1204-1219``): power spectrum -> (centre encoding) -> LayerNorm -> MLP ->
bias-free linear last layer, per centre species when ``legacy``; dE/dR through the hand-written reverse pass.
**Parity unpinned** against torch-spex (not shipped with the reference); checked against ``oracle/soap.py``.
"""
from ctypes import byref, c_void_p
from typing import Dict, List, Optional

import torch

from .. import _lib
from .. import runtime as rt
from .._lib import PetHipError, check
from ..pet.hypers import default_hypers as pet_default_hypers
from . import radial


class SoapBpnnHip:
    def __init__(self, hypers: dict, atomic_types: List[int], n_grid: int = 2049):
        self.lib = _lib.load()
        self.hypers = hypers
        self.atomic_types = list(atomic_types)
        so = hypers["soap"]
        self.cutoff, self.width = float(so["cutoff"]["radius"]), float(so["cutoff"]["width"])
        self.n_per_l, zeros, norms = radial.laplacian_eigenstates(self.cutoff, so["max_radial"], so["max_angular"])
        ns = len(atomic_types)
        self.legacy = bool(hypers["legacy"])
        h = _lib.SoapHypers()
        h.cutoff, h.cutoff_width, h.max_angular = self.cutoff, self.width, so["max_angular"]
        for l, n in enumerate(self.n_per_l):
            h.n_per_l[l] = n
        h.n_species = ns
        h.n_channels = ns if self.legacy else 4
        h.legacy = int(self.legacy)
        h.layernorm = int(bool(hypers["bpnn"]["layernorm"]))
        # heads: {"energy": "mlp"} (soap_bpnn/documentation.py:117-122; model.py:117-135, 671-672, 1110-1133): one bias-free
        # Linear(H, H) + SiLU per centre species between the BPNN and the last layer -- the same shape as one more hidden layer
        # of the BPNN (model.py:50-93: [Linear(bias=False), SiLU] x num_hidden_layers), so the native model runs it as layer
        # num_hidden_layers + 1 and ``load`` / ``trainable`` map the key ``heads.energy.<s>.0.weight`` onto that layer. Omitted
        # or "linear": no extra layer (the reference's Identity).
        head = (hypers.get("heads") or {}).get("energy", "linear")
        if head not in ("linear", "mlp"):
            raise ValueError(f"Unsupported head type {head} for target energy")
        self.mlp_head = head == "mlp"
        self._nh = int(hypers["bpnn"]["num_hidden_layers"])
        if self.mlp_head and self._nh + 1 > 8:
            raise PetHipError("an mlp head on top of 8 hidden layers: the tail kernels hold at most 8 layers")
        h.num_hidden_layers = self._nh + (1 if self.mlp_head else 0)
        h.num_neurons_per_layer = hypers["bpnn"]["num_neurons_per_layer"]
        self._handle = c_void_p()
        check(self.lib.soap_model_create(byref(h), byref(self._handle)))
        self.feature_size = int(self.lib.soap_model_feature_size(self._handle))
        table = torch.from_numpy(radial.spline_table(self.cutoff, zeros, norms, n_grid)).cuda().contiguous()
        check(self.lib.soap_model_set_radial_table(self._handle, rt._ptr(table), n_grid, rt._stream()))
        torch.cuda.current_stream().synchronize()
        # the graph (CSR order, edge vectors, ij->ji map) is the PET path's, built with the SOAP cutoff
        gh = pet_default_hypers()
        gh.update(cutoff=self.cutoff, cutoff_width=self.width, cutoff_function="Cosine")
        self._graph_model = rt.HipModel(gh, self.atomic_types)
        self._graph_model.load_species_table()
        self._ws: Optional[torch.Tensor] = None

    def __del__(self):
        h = getattr(self, "_handle", None)
        try:
            if h is not None and h.value:
                self.lib.soap_model_destroy(h)
                self._handle = c_void_p()
        except Exception:
            pass

    def _native_key(self, key: str) -> str:
        """``heads.energy.<s>.0.weight`` -> the native tail's extra hidden layer ``bpnn.<s>.<2 NH>.weight``."""
        if self.mlp_head and key.startswith("heads.energy.") and key.endswith(".0.weight"):
            return f"bpnn.{key.split('.')[2]}.{2 * self._nh}.weight"
        return key

    def load(self, params: Dict[str, torch.Tensor]) -> None:
        for key, t in params.items():
            key = self._native_key(key)
            rt._require_cuda(t)
            src = t.detach().to(torch.float32).contiguous()
            check(self.lib.soap_model_set_param(self._handle, key.encode(), rt._ptr(src), src.numel(), rt._stream()))
            torch.cuda.current_stream().synchronize()
        check(self.lib.soap_model_finalize(self._handle, rt._stream()))

    def graph(self, positions, cells, centers, neighbors, cell_shifts, species, system_indices) -> rt.HipGraph:
        return rt.HipGraph(self._graph_model, positions, cells, centers, neighbors, cell_shifts, species,
                           system_indices)

    def _workspace(self, g: rt.HipGraph) -> torch.Tensor:
        n = int(self.lib.soap_workspace_bytes(self._handle, g.n_nodes, g.n_edges))
        if n < 0:
            raise PetHipError("soap_workspace_bytes failed")
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(n, dtype=torch.uint8, device=g.workspace.device)
        return self._ws

    def forward(self, g: rt.HipGraph, want_features: bool = False):
        ws = self._workspace(g)
        dev = ws.device
        atomic = torch.empty(g.n_nodes, dtype=torch.float32, device=dev)
        feats = torch.empty((g.n_nodes, self.feature_size), dtype=torch.float32, device=dev) if want_features else None
        check(self.lib.soap_forward(self._handle, g.handle, rt._ptr(ws), ws.numel(), rt._ptr(atomic), rt._ptr(feats),
                                    rt._stream()))
        return (atomic, feats) if want_features else atomic

    def backward(self, g: rt.HipGraph, grad_atomic: torch.Tensor, want_cell_grad: bool = False):
        ws = self._workspace(g)
        dev = ws.device
        ga = grad_atomic.to(torch.float32).contiguous()
        gpos = torch.empty((g.n_nodes, 3), dtype=torch.float32, device=dev)
        gcell = torch.empty((g.n_systems, 3, 3), dtype=torch.float32, device=dev) if want_cell_grad else None
        check(self.lib.soap_backward(self._handle, g.handle, rt._ptr(ws), ws.numel(), rt._ptr(ga), rt._ptr(gpos),
                                     rt._ptr(gcell), rt._stream()))
        return (gpos, gcell) if want_cell_grad else gpos

    # ---- training (soap_bpnn/trainer.py:344-391) ---------------------------------------------------------------
    def sum_over_atoms(self, g: rt.HipGraph, atomic: torch.Tensor) -> torch.Tensor:
        out = torch.zeros(g.n_systems, dtype=torch.float32, device=atomic.device)
        check(self.lib.pet_sum_over_atoms(g.handle, rt._ptr(atomic), rt._ptr(out), rt._stream()))
        return out

    def zero_grad(self) -> None:
        check(self.lib.soap_model_zero_grad(self._handle, rt._stream()))

    def train_gradients(self, g: rt.HipGraph, grad_atomic: torch.Tensor, u: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ADDS ``d/d theta [sum_i grad_atomic_i e_i + <u, dE/dR>]`` to the gradient slots (``soap_train_gradients``;
        ``forward(g)`` must have run) and returns the tangent of the atomic energies along ``u`` ``[N]``."""
        ws = self._workspace(g)
        n = int(self.lib.soap_train_workspace_bytes(self._handle, g.n_nodes, g.n_edges))
        if n < 0:
            raise PetHipError("soap_train_workspace_bytes failed")
        if getattr(self, "_tws", None) is None or self._tws.numel() < n:
            self._tws = torch.empty(n, dtype=torch.uint8, device=ws.device)
        ga = grad_atomic.detach().to(torch.float32).contiguous()
        uu = None if u is None else u.detach().to(torch.float32).contiguous()
        tangent = torch.zeros(g.n_nodes, dtype=torch.float32, device=ws.device)
        check(self.lib.soap_train_gradients(self._handle, g.handle, rt._ptr(ws), ws.numel(), rt._ptr(self._tws),
                                            self._tws.numel(), rt._ptr(ga), rt._ptr(uu), rt._ptr(tangent), rt._stream()))
        return tangent

    def _copy_out(self, fn, keys_shapes) -> Dict[str, torch.Tensor]:
        out = {}
        for key, shape in keys_shapes:
            t = torch.empty(shape, dtype=torch.float32, device="cuda")
            check(fn(self._handle, self._native_key(key).encode(), rt._ptr(t), t.numel(), rt._stream()))
            out[key] = t
        return out

    def trainable(self) -> List:
        """``(key, shape)`` of every trainable parameter: the tail's, and for ``legacy = False`` models the Alchemical species
        embedding and the centre encoding in front of it."""
        nn_, nh = self.hypers["bpnn"]["num_neurons_per_layer"], self.hypers["bpnn"]["num_hidden_layers"]
        size, out = self.feature_size, []
        if not self.legacy:
            ns = len(self.atomic_types)
            out += [("species_embedding.weight", (ns, 4)), ("center_encoding.weight", (ns, size))]
        for s in range(len(self.atomic_types) if self.legacy else 1):
            if self.hypers["bpnn"]["layernorm"]:
                out += [(f"layernorm.{s}.weight", (size,)), (f"layernorm.{s}.bias", (size,))]
            out += [(f"bpnn.{s}.{2 * k}.weight", (nn_, size if k == 0 else nn_)) for k in range(nh)]
            if self.mlp_head:
                out.append((f"heads.energy.{s}.0.weight", (nn_, nn_)))
            out.append((f"last_layers.energy.{s}.weight", (1, nn_)))
        return out

    def grads(self) -> Dict[str, torch.Tensor]:
        return self._copy_out(self.lib.soap_model_get_grad, self.trainable())

    def params(self) -> Dict[str, torch.Tensor]:
        return self._copy_out(self.lib.soap_model_get_param, self.trainable())

    def adam_step(self, lr: float, step: int, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
        check(self.lib.soap_adam_step(self._handle, lr, betas[0], betas[1], eps, step, rt._stream()))


class SoapTrainStep:
    """One optimizer step of SOAP-BPNN on one batch, the body of the reference's loop (``soap_bpnn/trainer.py:344-391``):
    ``zero_grad -> evaluate_model(is_training=True) -> per-atom average -> MSE(E/atom) + MSE(dE/dR) -> backward -> Adam ->
    lr_scheduler.step()``. Adam's learning rate follows the reference's ``LambdaLR`` (``soap_bpnn/trainer.py:54-84``: linear
    warm-up over ``warmup_fraction`` of ``num_epochs * steps_per_epoch`` steps -- so the very first step runs at lr 0 --
    then a cosine to zero), stepped once per batch; no gradient clipping (``soap_bpnn/documentation.py`` TrainerHypers:
    learning_rate 1e-3, num_epochs 100, warmup_fraction 0.01). Descriptor, tail, reverse and second-order passes, weight
    gradients and Adam run in libpet_hip; torch does the ``[S]`` / ``[N, 3]`` loss arithmetic (shared with the PET step,
    ``metatrain_amd/pet/trainer.py``). One rank: the gradients stay inside the library, there is no all-reduce hook."""

    DEFAULTS = {"learning_rate": 1e-3, "num_epochs": 100, "warmup_fraction": 0.01}

    def __init__(self, model: SoapBpnnHip, hypers: Optional[dict] = None, steps_per_epoch: int = 1,
                 loss_weights: Optional[dict] = None):
        self.model = model
        self.hypers = dict(self.DEFAULTS)
        self.hypers.update(hypers or {})
        self.total_steps = int(self.hypers["num_epochs"]) * int(steps_per_epoch)
        self.weights = {"energy": 1.0, "forces": 1.0, **(loss_weights or {})}
        self.step_index = 0  # optimizer steps taken so far (LambdaLR's last_epoch)

    def current_lr(self) -> float:
        from ..pet.trainer import lr_lambda

        return self.hypers["learning_rate"] * lr_lambda(self.step_index, self.total_steps, self.hypers["warmup_fraction"])

    def state_dict(self) -> Dict[str, object]:
        """The scheduler's side of a trainer checkpoint: the step counter drives Adam's bias correction and the schedule."""
        return {"step_index": self.step_index, "total_steps": self.total_steps, "hypers": dict(self.hypers)}

    def load_state_dict(self, state: Dict[str, object]) -> None:
        self.step_index = int(state["step_index"])
        self.total_steps = int(state["total_steps"])
        self.hypers.update(state["hypers"])

    def __call__(self, g: rt.HipGraph, target_energies: torch.Tensor, n_atoms: torch.Tensor,
                 target_gradients: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        from ..pet.trainer import energy_loss_and_seeds, force_loss_and_seeds

        m = self.model
        m.zero_grad()
        atomic = m.forward(g)
        energies = m.sum_over_atoms(g, atomic)
        loss, seeds = energy_loss_and_seeds(energies, target_energies, n_atoms, g.system_of_atom(), self.weights["energy"])
        u = None
        if target_gradients is not None:
            grad_positions = m.backward(g, torch.ones_like(atomic))   # evaluate_model: autograd.grad(E.sum(), R, create_graph)
            loss_f, u = force_loss_and_seeds(grad_positions, target_gradients, self.weights["forces"])
            loss = loss + loss_f
        tangent = m.train_gradients(g, seeds, u)
        lr = self.current_lr()  # the rate the scheduler set after the previous step
        self.step_index += 1
        m.adam_step(lr, self.step_index)
        return {"loss": loss, "energies": energies, "tangent_atomic": tangent, "lr": lr}
