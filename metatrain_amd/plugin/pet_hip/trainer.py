"""The reference trainer, unchanged: its loop (``pet/trainer.py:417-472``: ``evaluate_model(is_training=True)`` ->
``loss.backward()`` -> clip -> optimizer -> scheduler, torch DDP) runs on the mirror because in ``train()`` mode
``PETBackend.predict`` returns energies from an autograd node whose backward is itself differentiable
(metatrain_amd/pet/backend.py). ``metatrain_amd.pet.trainer.TrainStep`` is the fully native (and faster) step for callers
that drive the loop themselves."""
from metatrain.pet.trainer import Trainer  # noqa: F401
