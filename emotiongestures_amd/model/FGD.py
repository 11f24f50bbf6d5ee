"""Mirror of model/FGD.py (MLP_Reconstruct :26-82): the FGD feature auto-encoder, on the HIP path."""
from ..harness import MLP_Reconstruct  # noqa: F401
