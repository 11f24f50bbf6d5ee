"""Mirror of Full_model/Layers.py (EncoderLayer :10-22, DecoderLayer :41-58) on the HIP path."""
from ..modules import DecoderLayer, EncoderLayer  # noqa: F401
