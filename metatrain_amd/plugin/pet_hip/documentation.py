"""
PET on MI355X (libpet_hip)
==========================

Same hyper-parameters as :mod:`metatrain.pet` (``pet/documentation.py:159-259``); what the kernels serve is listed in
``DESIGN.md`` section 0 of the metatrain_amd repository (one compiled size instantiation: ``d_pet = 128, d_node = 256,
d_feedforward = 256, d_head = 128, num_heads = 8``; training for the default architecture).
"""
from metatrain.pet.documentation import ModelHypers, TrainerHypers  # noqa: F401
