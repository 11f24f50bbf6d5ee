"""Mirror of data_loader/motion_preprocessor_expressive.py (MotionPreprocessor :4-31)."""
from ..datapath import MotionPreprocessor  # noqa: F401
