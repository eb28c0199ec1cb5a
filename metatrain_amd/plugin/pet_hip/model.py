"""The reference's ``PET`` model wrapper (``pet/model.py:55``: systems -> TensorMaps, composition / scaler, export) with
``self.backend`` built from ``metatrain_amd.pet.PETBackend`` -- the one-import change of INTEGRATION.md section 1, made at
construction time. State-dict keys are identical, so reference checkpoints load (``load_checkpoint`` builds ``cls(...)``)."""
import torch

import metatrain.pet.model as _reference_model
from metatrain_amd.pet import PETBackend as _HipBackend


class PETHip(_reference_model.PET):
    __supported_devices__ = ["cuda"]       # ROCm torch reports HIP devices as "cuda"; there is no CPU path
    __supported_dtypes__ = [torch.float32]  # the kernels compute in fp32 whatever the input dtype

    def __init__(self, hypers, dataset_info) -> None:
        # PET.__init__ does `self.backend = PETBackend(self.hypers, self.atomic_types)` (pet/model.py:115) and then
        # `self.backend.add_output(...)` per target (:1035-...): give it the MI355X backend for the duration of the call
        reference_backend = _reference_model.PETBackend
        _reference_model.PETBackend = _HipBackend
        try:
            super().__init__(hypers, dataset_info)
        finally:
            _reference_model.PETBackend = reference_backend
