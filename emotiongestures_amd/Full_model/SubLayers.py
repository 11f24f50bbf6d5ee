"""Mirror of Full_model/SubLayers.py (MultiHeadAttention :9-59, PositionwiseFeedForward :64-84) on the HIP path."""
from ..modules import MultiHeadAttention, PositionwiseFeedForward  # noqa: F401
