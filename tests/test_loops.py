"""Host logic of the EmotionNet K-fold loop (emotiongestures_amd/train/loops.py; train_audio_classifier_K_fold.py:109-200): the fold split
against sklearn's KFold(shuffle=True) (what upstream calls, :301), class weights, the seeded / sharded batch schedule, checkpoint naming.  No GPU."""
import numpy as np
import pytest

from emotiongestures_amd.train import loops


@pytest.mark.parametrize("n,k", [(10, 10), (23, 10), (100, 7), (5, 2)])
@pytest.mark.parametrize("seed", [0, 1234])
def test_kfold_indices_match_sklearn_shuffled(n, k, seed):
    """Upstream: `KFold(n_splits=10, shuffle=True)` (train_audio_classifier_K_fold.py:301)."""
    from sklearn.model_selection import KFold
    ours = list(loops.kfold_indices(n, k, shuffle=True, seed=seed))
    ref = list(KFold(n_splits=k, shuffle=True, random_state=seed).split(np.zeros(n)))
    assert len(ours) == len(ref) == k
    for (tr, va), (rtr, rva) in zip(ours, ref):
        np.testing.assert_array_equal(tr, rtr)
        np.testing.assert_array_equal(va, rva)
    # every sample is validated exactly once, and with n > k the folds are not the contiguous blocks of an unshuffled split
    assert sorted(np.concatenate([va for _, va in ours]).tolist()) == list(range(n))
    if n > 20:
        assert any(np.any(np.diff(va) != 1) for _, va in ours)
    with pytest.raises(ValueError):
        list(loops.kfold_indices(3, 4))


def test_kfold_indices_unseeded_draws_from_numpys_global_generator_like_sklearn():
    from sklearn.model_selection import KFold
    np.random.seed(77)
    ref = list(KFold(n_splits=5, shuffle=True).split(np.zeros(31)))          # random_state=None -> np.random's global RandomState
    np.random.seed(77)
    ours = list(loops.kfold_indices(31, 5, shuffle=True, seed=None))
    for (tr, va), (rtr, rva) in zip(ours, ref):
        np.testing.assert_array_equal(tr, rtr)
        np.testing.assert_array_equal(va, rva)


def test_kfold_indices_unshuffled_option_matches_sklearn_default():
    from sklearn.model_selection import KFold
    for (tr, va), (rtr, rva) in zip(loops.kfold_indices(23, 10, shuffle=False), KFold(n_splits=10).split(np.zeros(23))):
        np.testing.assert_array_equal(tr, rtr)
        np.testing.assert_array_equal(va, rva)


def test_class_weights_follow_upstream_formula():
    labels = [0] * 10 + [1] * 5 + [3] * 5
    w = loops.class_weights(labels)
    assert w.shape == (8,)
    np.testing.assert_allclose(w[[0, 1, 3]], [20 / (8 * 10), 20 / (8 * 5), 20 / (8 * 5)])           # sum(count) / (len(count) * count) (:149)
    assert w[2] == 0 and w[7] == 0                                                                     # absent classes: no division by zero


def test_epoch_batches_are_seeded_whole_and_sharded():
    idx = np.arange(100, 143)
    a = loops.epoch_batches(idx, 8, seed=3)
    b = loops.epoch_batches(idx, 8, seed=3)
    c = loops.epoch_batches(idx, 8, seed=4)
    assert len(a) == 5 and all(len(x) == 8 for x in a)                                                # drop_last
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and not all(np.array_equal(x, y) for x, y in zip(a, c))
    flat = np.concatenate(a)
    assert len(set(flat.tolist())) == 40 and set(flat.tolist()) <= set(idx.tolist())                  # a permutation of the subset, no repeats
    r0, r1 = loops.epoch_batches(idx, 8, 3, rank=0, world=2), loops.epoch_batches(idx, 8, 3, rank=1, world=2)
    assert len(r0) == len(r1) == 2                                                                    # 5 batches -> 2 rounds of 2, the odd one dropped
    assert np.array_equal(r0[0], a[0]) and np.array_equal(r1[0], a[1]) and np.array_equal(r0[1], a[2]) and np.array_equal(r1[1], a[3])


def test_checkpoint_name_and_alpha_mode_guard(tmp_path):
    assert loops.checkpoint_name("/x", 2, 0, 100) == "/x/checkpoint_fold2_epoch0_iteraction100.pth"  # upstream's spelling (:196)
    with pytest.raises(ValueError):
        loops.train_k_fold([], device="cpu", batch_size=4, alpha_mode="positional")


def test_precision_scope_and_reset_state_are_exception_safe():
    """The training layer's configuration is per process (autograd's backward thread must see the forward's): drivers use the context manager /
    reset_state so that a failure in one model's step cannot leak its arithmetic mode into the next (emotiongestures_amd/train/functional.py)."""
    from emotiongestures_amd.train import functional as F
    assert F.get_precision() == "f32"
    with F.precision("bf16x3"):
        assert F.get_precision() == "bf16x3"
        with F.precision("f32"):
            assert F.get_precision() == "f32"
        assert F.get_precision() == "bf16x3"
    assert F.get_precision() == "f32"
    with pytest.raises(RuntimeError):
        with F.precision("bf16x3"):
            raise RuntimeError("step failed")
    assert F.get_precision() == "f32"
    with pytest.raises(ValueError):
        F.precision("fp8").__enter__()
    F.set_precision("bf16x3")
    F.manual_seed(7)
    F.reset_state()
    assert F.get_precision() == "f32" and F._DROP["seed"] == 0 and F._DROP["epoch"] is None and not F._IMAGES["reg"]
