"""Import-line mirror of model/Beat_score_v2.py so that `from model.Beat_score_v2 import alignment`
(test_emotion_gesture_diversity_iterative.py:29) resolves after `install_aliases()`.

The beat-alignment metric itself is OUT of the hot-path scope: upstream computes audio onsets with librosa
(model/Beat_score_v2.py:58-77) and velocity extrema with scipy on the host; nothing of it runs on the GPU.  `alignment(sigma, order)` constructs
(the caller builds it before its loop, :185); every method that would compute something raises `BeatScoreUnavailable` with that explanation, as do
the two other metric classes of that file.  Point `sys.modules["model.Beat_score_v2"]` at the reference's own file to get the metric back
(it needs librosa and matplotlib)."""


class BeatScoreUnavailable(NotImplementedError):
    pass


def _refuse(what):
    raise BeatScoreUnavailable(f"model.Beat_score_v2.{what}: librosa-dependent host-side metric, not part of the HIP path "
                               "(import the reference's model/Beat_score_v2.py for it)")


class alignment(object):
    """model/Beat_score_v2.py:51-56: keeps (sigma, order) like upstream; see the module docstring."""

    def __init__(self, sigma, order):
        self.sigma = sigma
        self.order = order
        self.times = None
        self.oenv = None
        self.S = None
        self.rms = None
        self.pose_data = []

    def load_audio(self, *a, **k):
        _refuse("alignment.load_audio")

    def load_pose(self, *a, **k):
        _refuse("alignment.load_pose")

    def load_data(self, *a, **k):
        _refuse("alignment.load_data")

    def eval_random_pose(self, *a, **k):
        _refuse("alignment.eval_random_pose")

    def audio_beat_vis(self, *a, **k):
        _refuse("alignment.audio_beat_vis")

    def calculate_align(self, *a, **k):
        _refuse("alignment.calculate_align")


class L1div(object):
    def __init__(self, *a, **k):
        _refuse("L1div")


class SRGR(object):
    def __init__(self, *a, **k):
        _refuse("SRGR")
