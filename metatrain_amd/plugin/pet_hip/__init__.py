"""``metatrain.experimental.pet_hip``: the PET architecture with its numerical core on MI355X (libpet_hip).

metatrain discovers architectures by directory under its package root (``utils/architectures.py:17-58,118-152``), so
this directory is installed by copying or symlinking it to ``<site-packages>/metatrain/experimental/pet_hip`` next to a
``default-hypers.yaml`` copied from ``metatrain/pet`` (same hypers). It needs metatrain with its PET dependencies
(metatensor, metatomic) and ``metatrain_amd`` importable; neither metatensor nor metatomic can be installed in the
environment this repository was developed in, so this package is exercised here only with stand-ins for the three
reference modules it imports (``tests/test_plugin_cpu.py``).
"""
from .model import PETHip
from .trainer import Trainer

__model__ = PETHip
__trainer__ = Trainer
__capabilities__ = {
    "supported_devices": __model__.__supported_devices__,
    "supported_dtypes": __model__.__supported_dtypes__,
}
__authors__ = [("metatrain_amd", "")]
__maintainers__ = [("metatrain_amd", "")]
