"""MI355X host side of the SOAP-BPNN hot path."""
from .hypers import default_hypers  # noqa: F401
from .model import SoapBpnnHip, SoapTrainStep  # noqa: F401
