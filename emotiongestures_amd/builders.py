"""Construction helpers shared by bench.py, __graft_entry__.smoke(), the tools and the tests: the argparse-style
namespaces the reference passes into its constructors, a generator mirror with synthetic weights, and the per-clip
relative-L2 measure of the north-star."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np


def make_args(chunk):
    """The argparse namespace the reference passes into its constructors
    (test_emotion_gesture_diversity_iterative.py:352,363-366,372)."""
    return SimpleNamespace(chunk=chunk, hidden_size=300, n_layers=3, freeze_wordembed=False, wordembed_dim=300,
                           dropout_prob=0.1)


def make_lang(n_words=200):
    return SimpleNamespace(n_words=n_words, word_embedding_weights=None)


def build_mirror(variant, frames, pose_dim, prior, chunk, n_words=200, seed=0, spec_len=124, precision=None):
    """Our host-mirror generator with synthetic weights (CPU tensors; no GPU needed to construct)."""
    from .synth import load_synth_weights
    if variant == "spatial":
        from .Full_model.Models_spatial_memory import Transformer
    else:
        from .Full_model.Models_memory import Transformer
    m = Transformer(make_args(chunk), make_lang(n_words), frames=frames, pose_dim=pose_dim, prior_frames=prior, d_word_vec=512,
                    d_model=512, d_inner=2048, n_layers=3, n_head=8, d_k=64, d_v=64, spec_len=spec_len, precision=precision)
    load_synth_weights(m, seed)
    return m.eval()


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def clip_rel_l2(a, b):
    """max over clips of ||a_i - b_i|| / ||b_i||: the north-star's per-clip L2 (SURVEY.md §7 hard part 1)."""
    a = np.asarray(a, np.float64).reshape(a.shape[0], -1)
    b = np.asarray(b, np.float64).reshape(b.shape[0], -1)
    return float((np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-30)).max())
