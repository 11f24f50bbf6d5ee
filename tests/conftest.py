import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


from emotiongestures_amd.builders import build_mirror, clip_rel_l2, make_args, make_lang, rel_l2  # noqa: E402,F401


def golden_meta(z):
    b, frames, pose_dim, prior, chunk, spec_len, n_words, seed, use_sampled = [int(v) for v in z["meta"]]
    return dict(batch=b, frames=frames, pose_dim=pose_dim, prior=prior, chunk=chunk, spec_len=spec_len, n_words=n_words,
                seed=seed, use_sampled=bool(use_sampled))


GENERATOR_CASES = {
    "ted_spatial_b2": "spatial", "ted_spatial_b2_sampled": "spatial", "ted_memory_b4": "memory",
    "ted_spatial_b5": "spatial", "beat_spatial_b1": "spatial", "beat_memory_b2": "memory",
}


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
