"""Mirror of Full_model/Modules.py (ScaledDotProductAttention :5-23) on the HIP path."""
from ..modules import ScaledDotProductAttention  # noqa: F401
