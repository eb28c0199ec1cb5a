"""metatrain_amd: MI355X-native (gfx950) hot path of metatrain's PET architecture.

Only the forward / force path named by BASELINE.json's north_star lives here:
``csrc/`` (hand-written HIP kernels + the C ABI of ``include/pet_hip.h``), the ctypes
binding (``_lib``), a thin torch-memory host layer (``runtime``) and the host-side
mirror of the reference's ``PETBackend`` interface (``pet``). There is no CPU path.
"""
__version__ = "0.1.0"
