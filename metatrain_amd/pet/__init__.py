from .hypers import DEFAULT_MODEL_HYPERS, default_hypers  # noqa: F401
from .backend import PETBackend  # noqa: F401
