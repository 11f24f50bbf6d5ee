"""Mirror of Full_model/tcn.py (TemporalBlock :16-47, TemporalConvNet :49-64) on the HIP path."""
from ..modules import TemporalBlock, TemporalConvNet  # noqa: F401
