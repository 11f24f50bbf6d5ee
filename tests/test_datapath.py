"""Data path before the generator: bit-exact against records / items produced by the reference's own
DataPreprocessor._sample_from_clip and SpeechMotionDataset.__getitem__ (tests/golden/make_golden_datapath.py)."""
import hashlib
import os

import numpy as np
import pytest

from emotiongestures_amd import datapath as D
from emotiongestures_amd.synth import hash_unit, synth_clip

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "datapath.npz"))
CASES = {"beat": dict(n_poses=60, stride=15, fps=15, eid="1_wayne_0_77_77", seed=7, duration=11.3),
         "ted": dict(n_poses=34, stride=10, fps=15, eid="2_scott_0_3_3", seed=11, duration=7.9)}


def _check(name, i, tag, arr):
    arr = np.ascontiguousarray(arr)
    key = f"{name}.{i}.{tag}"
    assert tuple(G[key + ".shape"]) == arr.shape, key
    assert str(G[key + ".dtype"]) == str(arr.dtype), key
    np.testing.assert_array_equal(np.concatenate([arr.ravel()[:8], arr.ravel()[-8:]]).astype(np.float64), G[key + ".ends"], err_msg=key)
    assert bytes(G[key + ".sha"]) == hashlib.sha256(arr.tobytes()).digest(), key


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("store_kind", ["dict", "dir"])
def test_sample_pipeline_matches_reference(name, store_kind, tmp_path):
    c = CASES[name]
    clip = synth_clip(seed=c["seed"], duration=c["duration"])
    store = D.DictStore() if store_kind == "dict" else D.DirStore(str(tmp_path / "samples"))
    pre = D.DataPreprocessor([{"eid": c["eid"], "clips": [clip]}], store, c["n_poses"], c["stride"], c["fps"])
    filtered = pre.run()
    assert pre.n_out_samples == int(G[f"{name}.n_samples"]) == len(store)
    assert sum(filtered.values()) == int(G[f"{name}.filtered"])
    assert pre.spectrogram_sample_length == int(G[f"{name}.spectrogram_sample_length"])
    assert pre.audio_sample_length == int(G[f"{name}.audio_sample_length"])
    ds = D.SpeechMotionDataset(store, c["n_poses"], c["stride"], c["fps"])
    assert ds.expected_audio_length == int(G[f"{name}.expected_audio_length"])
    assert ds.expected_spectrogram_length == int(G[f"{name}.expected_spectrogram_length"])
    assert list(store.keys())[0] == b"0000000000"
    for i in range(len(ds)):
        words, poses, audio, spec, aux = D.decode_record(store.get(D.sample_key(i)))
        _check(name, i, "rec_poses", np.asarray(poses, np.float32))
        _check(name, i, "rec_audio", audio)
        _check(name, i, "rec_spec", spec)
        assert spec.dtype == np.float16                                     # stored as fp16 (utils/train_utils_BEAT.py:189)
        assert [w[0] for w in words] == list(G[f"{name}.{i}.rec_words"])
        np.testing.assert_array_equal(
            np.array([aux["start_frame_no"], aux["end_frame_no"], aux["start_time"], aux["end_time"]], np.float64), G[f"{name}.{i}.aux"])
        a, s, p, lab, aux2 = ds[i]
        _check(name, i, "item_audio", a.numpy())
        _check(name, i, "item_spec", s.numpy())
        _check(name, i, "item_pose", p.numpy())
        np.testing.assert_array_equal(lab.numpy(), G[f"{name}.{i}.item_label"])
        assert aux2["eid"] == c["eid"]
    batch = D.audio_classifier_collate_fn([ds[0], ds[1]])
    assert batch[0].shape == (2, ds.expected_audio_length) and batch[1].shape == (2, 128, ds.expected_spectrogram_length)
    assert batch[3].shape == (2, 8) and len(batch[4]["eid"]) == 2
    with pytest.raises(IndexError):
        ds[len(ds)]


def test_scalar_helpers_match_reference():
    for nf, fps, want in G["speclen_sweep"]:
        assert D.calc_spectrogram_length_from_motion_length(int(nf), int(fps)) == int(want)
    table = np.stack([D.one_hot_eid("1_x_0_%d_%d" % (k, k)) for k in range(1, 125)])
    np.testing.assert_array_equal(table, G["eid_table"])
    a = np.arange(10, dtype=np.float32)
    np.testing.assert_array_equal(D.make_audio_fixed_length(a, 14), G["fixed_len_pad"])
    np.testing.assert_array_equal(D.make_audio_fixed_length(a, 6), G["fixed_len_cut"])
    rs = hash_unit("resample", 13 * 5, 3).astype(np.float32).reshape(13, 5)
    np.testing.assert_array_equal(rs, G["resample_in"])
    out = D.resample_pose_seq(rs, 13 / 30.0, 15)
    assert out.dtype == G["resample_out"].dtype and out.shape == G["resample_out"].shape
    np.testing.assert_allclose(out, G["resample_out"], rtol=0, atol=1e-7)


def test_edge_cases():
    # a clip shorter than one sample yields nothing; mismatched audio / skeleton lengths are refused as upstream
    clip = synth_clip(seed=3, duration=3.0)
    store = D.DictStore()
    pre = D.DataPreprocessor([{"eid": "1_a_0_1_1", "clips": [clip]}], store, 60, 15, 15)
    pre.run()
    assert len(store) == 0
    bad = synth_clip(seed=3, duration=8.0)
    bad["audio_feat"] = bad["audio_feat"][:, :100]
    with pytest.raises(AssertionError):
        D.DataPreprocessor([{"eid": "1_a_0_1_1", "clips": [bad]}], D.DictStore(), 60, 15, 15).run()
    nan = synth_clip(seed=3, duration=8.0)
    nan["skeletons"][5, 2, 1] = np.nan
    with pytest.raises(AssertionError):
        D.DataPreprocessor([{"eid": "1_a_0_1_1", "clips": [nan]}], D.DictStore(), 60, 15, 15).run()
    # samples with fewer than two words are skipped
    few = synth_clip(seed=3, duration=8.0, n_words=1)
    st = D.DictStore()
    D.DataPreprocessor([{"eid": "1_a_0_1_1", "clips": [few]}], st, 60, 15, 15).run()
    assert len(st) == 0
    with pytest.raises(ValueError):
        D.decode_record(b"\x00\x01not a record")
    with pytest.raises(RuntimeError):
        D.LmdbStore("/tmp/none")
