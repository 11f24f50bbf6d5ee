"""Mirror of Full_model/ResNetBlocks.py (SEBasicBlock :7-37, SELayer :81-96) on the HIP path."""
from ..modules import SEBasicBlock, SELayer  # noqa: F401
