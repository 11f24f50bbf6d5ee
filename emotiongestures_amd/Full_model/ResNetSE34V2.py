"""Mirror of Full_model/ResNetSE34V2.py (ResNetSE :13-74) on the HIP path."""
from ..modules import ResNetSE, SEBasicBlock, SELayer  # noqa: F401
