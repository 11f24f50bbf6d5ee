"""emotiongestures_amd -- MI355X-native (gfx950) implementation of the EmotionGesture audio->gesture hot path.

Layout:
  csrc/ + libemogest_hip.so   hand-written HIP kernels behind the C ABI of include/emogest.h
  _lib / packing / engine / ops  ctypes binding, weight-arena packing, per-batch engines, block operators
  modules                     host mirror of the reference's nn.Module surface
  Full_model/, CAVE/, model/, skeleton_classifer/   the reference's import paths (thin re-exports of ``modules`` / ``harness``)
  harness                     the eval loop around the generator: FGD features, Frechet/diversity, emotion classifier
  synth                       platform-exact synthetic weights / inputs
  dist                        clip sharding across ranks (one process per GPU)
"""
from . import synth  # noqa: F401  (pure numpy; safe without the HIP library)

__version__ = "0.1.0"


def install_aliases() -> None:
    """Make the reference's own import lines resolve to this package, e.g.
    ``from Full_model.Models_memory import Transformer`` and ``from CAVE.BEAT_CVAE import MLP_Reconstruct_v3``
    (test_emotion_gesture_diversity_iterative.py:25-26)."""
    import importlib
    import sys

    for top in ("Full_model", "CAVE", "model", "skeleton_classifer", "data_loader", "utils"):
        pkg = importlib.import_module(f"{__name__}.{top}")
        sys.modules.setdefault(top, pkg)
    for name in ("Full_model.Models_spatial_memory", "Full_model.Models_memory", "Full_model.Layers", "Full_model.SubLayers",
                 "Full_model.Modules", "Full_model.tcn", "Full_model.ResNetSE34V2", "Full_model.ResNetBlocks", "CAVE.BEAT_CVAE",
                 "model.FGD", "model.FHD_score", "model.Beat_score_v2", "model.audio_emotion_classifer", "model.motion_ae", "model.embedding_space_evaluator",
                 "skeleton_classifer.Models", "data_loader.data_preprocessor_expressive", "data_loader.motion_preprocessor_expressive",
                 "data_loader.lmdb_loader_BEAT_full", "utils.train_utils_BEAT", "utils.data_utils_expressive"):
        sys.modules.setdefault(name, importlib.import_module(f"{__name__}.{name}"))
