"""Mirror of data_loader/lmdb_loader_BEAT_full.py: SpeechMotionDataset (:120-253), one_hot_eid (:78-118),
audio_classifier_collate_fn (:63-75).  The first constructor argument is a SampleStore (..datapath), not an LMDB directory."""
from ..datapath import SpeechMotionDataset, audio_classifier_collate_fn, one_hot_eid  # noqa: F401
