"""Mirror of data_loader/data_preprocessor_expressive.py (DataPreprocessor :17-193)."""
from ..datapath import DataPreprocessor, MotionPreprocessor, get_words_in_time_range  # noqa: F401
