"""The architecture-plugin skeleton (metatrain_amd/plugin/pet_hip; SURVEY section 8(b) "plugin entry": ``__model__`` /
``__trainer__`` / ``documentation.py``, pet/__init__.py:5-6, utils/architectures.py:118-152). metatrain itself cannot be
imported here (metatensor / metatomic are not installable), so the three reference modules the plugin imports are
replaced by stand-ins with the one property each that the plugin relies on."""
import importlib.util
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_plugin(monkeypatch):
    calls = {}
    ref_model = types.ModuleType("metatrain.pet.model")

    class ReferenceBackend:  # what pet/model.py:40 imports
        pass

    class PET(torch.nn.Module):  # pet/model.py:115,1035: builds the backend from the MODULE-LEVEL name, then adds outputs
        __supported_devices__ = ["cuda", "cpu"]
        __supported_dtypes__ = [torch.float32, torch.float64]

        def __init__(self, hypers, dataset_info):
            super().__init__()
            self.backend = ref_model.PETBackend(hypers, dataset_info["atomic_types"])
            for name, shapes in dataset_info["targets"].items():
                self.backend.add_output(name, shapes)
            calls["backend_class_during_init"] = ref_model.PETBackend

    ref_model.PET, ref_model.PETBackend = PET, ReferenceBackend
    ref_trainer = types.ModuleType("metatrain.pet.trainer")
    ref_trainer.Trainer = type("Trainer", (), {})
    ref_doc = types.ModuleType("metatrain.pet.documentation")
    ref_doc.ModelHypers, ref_doc.TrainerHypers = dict, dict
    for name, mod in (("metatrain", types.ModuleType("metatrain")), ("metatrain.pet", types.ModuleType("metatrain.pet")),
                      ("metatrain.experimental", types.ModuleType("metatrain.experimental")),
                      ("metatrain.pet.model", ref_model), ("metatrain.pet.trainer", ref_trainer),
                      ("metatrain.pet.documentation", ref_doc)):
        monkeypatch.setitem(sys.modules, name, mod)
    for name in [k for k in sys.modules if k.startswith("metatrain.experimental.pet_hip")]:
        monkeypatch.delitem(sys.modules, name)  # submodules of an earlier load are bound to that load's stand-ins
    for sub in ("model", "trainer", "documentation"):
        monkeypatch.setitem(sys.modules, f"metatrain.experimental.pet_hip.{sub}", None)
        monkeypatch.delitem(sys.modules, f"metatrain.experimental.pet_hip.{sub}")  # (registers them for clean-up)
    pkg_dir = os.path.join(ROOT, "metatrain_amd", "plugin", "pet_hip")
    spec = importlib.util.spec_from_file_location("metatrain.experimental.pet_hip", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    plugin = importlib.util.module_from_spec(spec)
    monkeypatch.setitem(sys.modules, "metatrain.experimental.pet_hip", plugin)
    spec.loader.exec_module(plugin)
    return plugin, ref_model, ref_trainer, ReferenceBackend, calls


def test_plugin_exposes_the_architecture_entry_points(monkeypatch):
    plugin, ref_model, ref_trainer, _, _ = _load_plugin(monkeypatch)
    assert plugin.__model__.__name__ == "PETHip" and issubclass(plugin.__model__, ref_model.PET)
    assert plugin.__trainer__ is ref_trainer.Trainer
    assert plugin.__capabilities__ == {"supported_devices": ["cuda"], "supported_dtypes": [torch.float32]}
    doc = importlib.import_module("metatrain.experimental.pet_hip.documentation")
    assert doc.ModelHypers is dict and doc.TrainerHypers is dict and "MI355X" in doc.__doc__


def test_plugin_model_builds_the_hip_backend_and_restores_the_reference_name(monkeypatch):
    from metatrain_amd.pet import PETBackend, default_hypers

    plugin, ref_model, _, ReferenceBackend, calls = _load_plugin(monkeypatch)
    model = plugin.__model__(default_hypers(), {"atomic_types": [1, 6, 7, 8], "targets": {"energy": {"energy": [1]}}})
    assert isinstance(model.backend, PETBackend) and calls["backend_class_during_init"] is PETBackend
    assert ref_model.PETBackend is ReferenceBackend  # put back, also when construction fails:
    try:
        plugin.__model__(dict(default_hypers(), normalization="BatchNorm"), {"atomic_types": [1], "targets": {}})
    except ValueError:
        pass
    assert ref_model.PETBackend is ReferenceBackend
    # reference-schema state dict (the keys a reference checkpoint carries under "backend.")
    keys = list(model.state_dict().keys())
    assert keys[0] == "backend.species_to_species_index" and "backend.node_last_layers.energy.0.energy.weight" in keys
