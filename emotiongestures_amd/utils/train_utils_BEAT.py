"""Mirror of the audio / pose helpers (utils/train_utils_BEAT.py:186-226, duplicated upstream in utils/data_utils_expressive.py:85-126)."""
from ..datapath import (calc_spectrogram_length_from_motion_length, extract_melspectrogram, make_audio_fixed_length,  # noqa: F401
                        resample_pose_seq)
