"""Host side of the PET training step, mirroring the reference's loop body
``pet/trainer.py:417-472``:

    optimizer.zero_grad() -> evaluate_model(is_training=True) -> average_by_num_atoms -> loss
    -> loss.backward() -> clip_grad_norm_ -> optimizer.step() -> lr_scheduler.step()

Everything that touches activations or weights runs in libpet_hip (forward, reverse pass, weight
gradients, clip + Adam, re-pack); torch only does the per-structure loss arithmetic on ``[S]`` /
``[N,3]`` tensors and the RCCL all-reduce of the flat gradient bucket.
"""
import math
from typing import Dict, Optional

import torch

from .. import distributed as D
from ..runtime import HipForward, HipGraph, HipModel

DEFAULT_TRAINER_HYPERS = {  # pet/documentation.py TrainerHypers defaults that the step depends on
    "learning_rate": 1e-4,
    "weight_decay": None,
    "grad_clip_norm": 1.0,
    "num_epochs": 1000,
    "warmup_fraction": 0.01,
    "loss_weights": {"energy": 1.0, "forces": 1.0},
}


def lr_lambda(step: int, total_steps: int, warmup_fraction: float, min_lr_ratio: float = 0.0) -> float:
    """Linear warm-up then cosine decay (``pet/trainer.py:56-86``, ``get_scheduler``)."""
    warmup_steps = int(warmup_fraction * total_steps)
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    progress = (step - warmup_steps) / float(max(1, total_steps - warmup_steps))
    return min_lr_ratio + (1.0 - min_lr_ratio) * 0.5 * (1.0 + math.cos(math.pi * progress))


def energy_loss_and_seeds(energies: torch.Tensor, targets: torch.Tensor, n_atoms: torch.Tensor,
                          system_of_atom: torch.Tensor, weight: float = 1.0):
    """MSE (mean over structures) of per-atom-averaged energies (``utils/per_atom.py``,
    ``utils/loss.py`` "mse"/"mean") and dL/dE_i for every atom i (the seed of the reverse pass)."""
    diff = (energies - targets) / n_atoms
    loss = weight * (diff * diff).mean()
    d_energy = weight * 2.0 * diff / (n_atoms * energies.numel())  # dL/dE_s
    return loss, d_energy[system_of_atom]


def force_loss_and_seeds(grad_positions: torch.Tensor, target_gradients: torch.Tensor, weight: float = 1.0):
    """MSE (mean over the N*3 components) on the position gradient dE/dR = -forces, which the per-atom
    averaging leaves untouched (``utils/per_atom.py``: samples carry "atom"), and u = dL/d(dE/dR)."""
    diff = grad_positions - target_gradients
    loss = weight * (diff * diff).mean()
    return loss, weight * 2.0 * diff / diff.numel()


def strain_loss_and_seeds(positions: torch.Tensor, cells: torch.Tensor, system_of_atom: torch.Tensor,
                          grad_positions: torch.Tensor, grad_cells: torch.Tensor, target_strain_gradients: torch.Tensor,
                          weight: float = 1.0):
    """The strain derivative of ``utils/evaluate_model.py:305-321`` (``positions @ strain``, ``cell @ strain``, gradient at
    strain = 1) from dE/dR and dE/dcell: ``dE/deps[s] = R_s^T dE/dR_s + cell_s^T dE/dcell_s`` ``[S,3,3]``; MSE (mean over
    the 9 S components) against the targets and the seeds ``u = dL/d(dE/dR)`` ``[N,3]``, ``u_cell = dL/d(dE/dcell)``
    ``[S,3,3]`` of the second-order pass. ``[S]``-sized torch arithmetic."""
    n_sys = cells.shape[0]
    virial = torch.zeros((n_sys, 3, 3), dtype=grad_positions.dtype, device=grad_positions.device)
    virial.index_add_(0, system_of_atom, positions[:, :, None] * grad_positions[:, None, :])
    virial = virial + cells.transpose(1, 2) @ grad_cells
    diff = virial - target_strain_gradients
    loss = weight * (diff * diff).mean()
    g = weight * 2.0 * diff / diff.numel()                       # dL/d(dE/deps)
    u = (positions[:, None, :] @ g[system_of_atom]).squeeze(1)    # d(dE/deps_ab)/d(gR_ib) = R_ia
    return loss, u, cells @ g


class TrainStep:
    """One optimizer step on one batch (a ``HipGraph`` of several structures)."""

    def __init__(self, model: HipModel, hypers: Optional[dict] = None, steps_per_epoch: int = 1):
        self.model = model
        self.hypers = dict(DEFAULT_TRAINER_HYPERS)
        self.hypers.update(hypers or {})
        self.total_steps = int(self.hypers["num_epochs"]) * int(steps_per_epoch)
        self.step_index = 0  # optimizer steps taken so far (LambdaLR's last_epoch)
        self.comm_events: Optional[list] = None  # a list here collects (start, end) events around every gradient all-reduce

    def state_dict(self) -> Dict[str, object]:
        """What the reference's trainer checkpoint keeps of the optimizer and scheduler (``pet/trainer.py:697-717``:
        ``optimizer_state_dict``, ``scheduler_state_dict``, epoch): Adam's moments, the step counter that drives both
        the bias correction and the LambdaLR schedule, and the hypers the schedule was built from."""
        return {"step_index": self.step_index, "total_steps": self.total_steps, "hypers": dict(self.hypers),
                "optimizer": {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in self.model.optimizer_state().items()}}

    def load_state_dict(self, state: Dict[str, object]) -> None:
        self.step_index = int(state["step_index"])
        self.total_steps = int(state["total_steps"])
        self.hypers.update(state["hypers"])
        self.model.load_optimizer_state(state["optimizer"])

    def current_lr(self) -> float:
        h = self.hypers
        return h["learning_rate"] * lr_lambda(self.step_index, self.total_steps, h["warmup_fraction"])

    def __call__(self, graph: HipGraph, fw: HipForward, target_energies: torch.Tensor,
                 n_atoms: torch.Tensor, target_gradients: Optional[torch.Tensor] = None,
                 target_strain_gradients: Optional[torch.Tensor] = None, positions: Optional[torch.Tensor] = None,
                 cells: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """``target_gradients`` [N,3] = dE/dR targets (-forces); None trains on energies only.
        ``target_strain_gradients`` [S,3,3] = dE/dstrain targets (stress x volume; needs ``positions`` and ``cells``,
        weight ``loss_weights["strain"]``, default 1)."""
        self.model.zero_grad()
        loss, energies = self._accumulate(graph, fw, target_energies, n_atoms, target_gradients, target_strain_gradients,
                                          positions, cells, 1.0, 1.0, 1.0)
        norm = self._finish()
        return {"loss": loss, "grad_norm": norm, "energies": energies}

    def microbatched(self, batches) -> Dict[str, torch.Tensor]:
        """ONE optimizer step over several micro-batches (gradient accumulation): the training workspace holds every
        tangent of a batch (100 KB per edge), so a rank's share of a large batch -- BASELINE
        ``configs[3]``: 64 x 10 000-atom boxes per GPU -- is walked a few boxes at a time. Each element of ``batches`` is a
        dict with the arguments of :meth:`__call__` (``graph, fw, target_energies, n_atoms`` and optionally
        ``target_gradients, target_strain_gradients, positions, cells``); the losses are the full batch's means
        (``utils/loss.py`` "mean" reduction over ALL structures / components), so the result equals the one-batch step up
        to fp32 summation order."""
        batches = list(batches)
        s_tot = float(sum(int(b["target_energies"].numel()) for b in batches))
        n_tot = float(sum(int(b["graph"].n_nodes) for b in batches))
        self.model.zero_grad()
        loss, energies = None, []
        for b in batches:
            s_b, n_b = float(b["target_energies"].numel()), float(b["graph"].n_nodes)
            l_b, e_b = self._accumulate(b["graph"], b["fw"], b["target_energies"], b["n_atoms"], b.get("target_gradients"),
                                        b.get("target_strain_gradients"), b.get("positions"), b.get("cells"),
                                        s_b / s_tot, n_b / n_tot, s_b / s_tot)
            loss = l_b if loss is None else loss + l_b
            energies.append(e_b)
        norm = self._finish()
        return {"loss": loss, "grad_norm": norm, "energies": torch.cat(energies)}

    def begin(self, graph: HipGraph, fw: HipForward, target_energies: torch.Tensor, n_atoms: torch.Tensor,
              target_gradients: Optional[torch.Tensor] = None, target_strain_gradients: Optional[torch.Tensor] = None,
              positions: Optional[torch.Tensor] = None, cells: Optional[torch.Tensor] = None) -> None:
        """First half of :meth:`__call__`: the three sweeps of the batch, then the gradient all-reduce is STARTED
        (``distributed.all_reduce_gradients_async``: RCCL runs it on its own stream). Whatever the caller launches before
        :meth:`end` -- the next batch's neighbour lists and graph build, as the reference's DataLoader workers do beside
        ``loss.backward()`` (``pet/trainer.py:417-472``) -- overlaps the collective."""
        self.model.zero_grad()
        self._pending = self._accumulate(graph, fw, target_energies, n_atoms, target_gradients, target_strain_gradients,
                                         positions, cells, 1.0, 1.0, 1.0)
        self._reduce = D.all_reduce_gradients_async(self.model, self.comm_events)

    def end(self) -> Dict[str, torch.Tensor]:
        """Second half: wait for the reduced gradients (a stream dependency under RCCL), clip + AdamW + schedule."""
        loss, energies = self._pending
        norm = self._finish()
        self._pending = None
        return {"loss": loss, "grad_norm": norm, "energies": energies}

    def _finish(self) -> torch.Tensor:
        m = self.model
        reduce = getattr(self, "_reduce", None) or D.all_reduce_gradients_async(m, self.comm_events)
        self._reduce = None
        reduce.wait()
        norm = m.adam_step(self.current_lr(), self.step_index + 1, weight_decay=self.hypers["weight_decay"],
                           max_grad_norm=self.hypers["grad_clip_norm"] or 0.0)
        self.step_index += 1
        return norm

    def _accumulate(self, graph, fw, target_energies, n_atoms, target_gradients, target_strain_gradients, positions, cells,
                    share_e: float, share_f: float, share_s: float):
        """Forward + reverse passes of one (micro-)batch, parameter gradients ADDED to the model's slots; ``share_*`` =
        this batch's fraction of the structures / force components / strain components of the whole step."""
        w = self.hypers["loss_weights"]
        if fw.graph is not graph:  # micro-batches may share one workspace allocation
            fw.rebind(graph)
        atomic = fw.forward()
        energies = fw.sum_over_atoms(atomic)
        loss, seeds = energy_loss_and_seeds(energies, target_energies, n_atoms, graph.system_of_atom(), w["energy"] * share_e)
        if target_gradients is None and target_strain_gradients is None:
            fw.backward_train(seeds)
        else:
            ones = torch.ones_like(atomic)
            # evaluate_model: autograd.grad(E.sum(), [R, strain], create_graph=True)
            grad_positions, grad_cells = fw.backward(ones, want_cell_grad=True)
            u = torch.zeros_like(grad_positions)
            u_cell = None
            if target_gradients is not None:
                loss_f, u = force_loss_and_seeds(grad_positions, target_gradients, w["forces"] * share_f)
                loss = loss + loss_f
            if target_strain_gradients is not None:
                if positions is None or cells is None:
                    raise ValueError("a strain-gradient (stress) target needs `positions` and `cells`")
                loss_s, u_s, u_cell = strain_loss_and_seeds(
                    positions.to(torch.float32), cells.to(torch.float32), graph.system_of_atom().long(), grad_positions,
                    grad_cells, target_strain_gradients, w.get("strain", 1.0) * share_s)
                loss = loss + loss_s
                u = u + u_s
            fw.backward_train2(ones, seeds, u, u_cell=u_cell)
        return loss, energies
