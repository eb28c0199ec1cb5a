"""
PET on MI355X (libpet_hip)
==========================

Same hyper-parameters as :mod:`metatrain.pet` (``pet/documentation.py:159-259``); what the kernels serve is listed in
``DESIGN.md`` section 0 of the metatrain_amd repository: every model size with ``d_pet % num_heads == 0`` and a head
dimension of at most 128, every architecture switch, inference and training. The matrix-core kernels are tuned for the default
size (``d_pet = 128, d_node = 256, d_feedforward = 256, d_head = 128, num_heads = 8``); other sizes, PostLN / residual
training and atoms with more than 127 neighbours run on the size-generic HIP path (``csrc/gen.hip``, ``gen_train.hip``).
"""
from metatrain.pet.documentation import ModelHypers, TrainerHypers  # noqa: F401
