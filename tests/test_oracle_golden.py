"""Pin the CPU oracle (oracle/emogest_oracle.py) against golden vectors produced by the reference's own
classes (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import GENERATOR_CASES, GOLDEN, build_mirror, clip_rel_l2, golden_meta, rel_l2
from emotiongestures_amd.synth import digest, load_synth_weights, synth_inputs
from oracle import emogest_oracle as O

TOL = 2e-5      # fp32 reassociation between ATen/mkldnn module code and the functional restatement


def _run_oracle(name, variant):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = golden_meta(z)
    model = build_mirror(variant, m["frames"], m["pose_dim"], m["prior"], m["chunk"], m["n_words"], m["seed"], m["spec_len"])
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    inp = synth_inputs(m["batch"], m["frames"], m["pose_dim"], m["prior"], spec_len=m["spec_len"], n_words=m["n_words"], seed=m["seed"])
    cfg = O.GenCfg(frames=m["frames"], pose_dim=m["pose_dim"], prior_frames=m["prior"], chunk=m["chunk"], variant=variant)
    taps = {}
    with torch.no_grad():
        out = O.generator_forward(sd, cfg, torch.from_numpy(inp["spec"]), torch.from_numpy(inp["text"]),
                                  torch.from_numpy(inp["pre_pose"]),
                                  torch.from_numpy(inp["sampled"]) if m["use_sampled"] else None, taps=taps)
    return z, out, taps


@pytest.mark.parametrize("name,variant", sorted(GENERATOR_CASES.items()))
def test_generator_oracle_matches_reference_golden(name, variant):
    z, (pose, emo, sem, pred, text), taps = _run_oracle(name, variant)
    assert pose.shape == z["pose"].shape
    assert clip_rel_l2(pose.numpy(), z["pose"]) < TOL
    assert rel_l2(pred.numpy(), z["emotion_prediction"]) < TOL
    for key, t in (("emotion_feature", emo), ("semantic_feature", sem), ("text_embedding", text)):
        d = digest(t.numpy(), 8192)
        assert tuple(d["shape"]) == tuple(z[key + "/shape"])
        assert rel_l2(d["sample"], z[key + "/sample"]) < TOL, key
        assert abs(d["absmean"] - z[key + "/absmean"]) < TOL * max(1.0, z[key + "/absmean"])
    for tap in ("stem", "layer1", "layer2", "layer3", "audio_feat", "prior_enc", "fusion", "enc0", "enc1", "enc2", "dec0", "dec1", "dec2"):
        d = digest(taps[tap].numpy())
        assert rel_l2(d["sample"], z[f"tap_{tap}/sample"]) < TOL, tap


def test_tm_memory_couples_clips_across_the_batch():
    """Full_model/Models_memory.py:288-289 contracts over the batch axis: the oracle must reproduce that
    (SURVEY.md §4 probe 3), so clip 0 of a batch of 4 differs from clip 0 run alone at module level."""
    model = build_mirror("memory", 34, 126, 4, 4, seed=2)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    cfg = O.GenCfg(variant="memory")
    inp = synth_inputs(4, seed=2)
    prior = torch.from_numpy(inp["pre_pose"])
    pred = O.pred_conv(sd, "prior_seq_encoder.pred_conv", prior)
    a = O.tm_memory(sd, "prior_seq_encoder.temporal_memory", prior, pred, cfg)[0]
    b = O.tm_memory(sd, "prior_seq_encoder.temporal_memory", prior[:1], pred[:1], cfg)[0]
    assert (a - b).abs().max() > 1e-6


def test_sp_memory_v2_is_identity_on_the_output():
    """Full_model/Models_spatial_memory.py:276-295 writes into a clone and returns its input: the spatial variant's
    prior encoding must equal post_header(cat(prior, pred_conv(prior)))."""
    model = build_mirror("spatial", 34, 126, 4, 4)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    inp = synth_inputs(2)
    prior = torch.from_numpy(inp["pre_pose"])
    out = O.prior_memory_encoder(sd, "prior_seq_encoder", prior, O.GenCfg())
    cat = torch.cat((prior, O.pred_conv(sd, "prior_seq_encoder.pred_conv", prior)), 1)
    ref = torch.nn.functional.linear(torch.nn.functional.linear(cat, sd["prior_seq_encoder.post_header.0.weight"], sd["prior_seq_encoder.post_header.0.bias"]),
                                     sd["prior_seq_encoder.post_header.2.weight"], sd["prior_seq_encoder.post_header.2.bias"])
    assert torch.equal(out, ref)


def test_cvae_oracle_matches_reference_golden():
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    z = np.load(os.path.join(GOLDEN, "cvae_v3.npz"))
    n, seed = [int(v) for v in z["meta"]]
    m = load_synth_weights(MLP_Reconstruct_v3(), seed).eval()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    inp = synth_inputs(n, frames=60, seed=seed)
    with torch.no_grad():
        s = O.cvae_sample(sd, torch.from_numpy(inp["label"]), torch.from_numpy(inp["z"]))
        eps = torch.from_numpy(synth_inputs(n, seed=seed + 1)["z"])
        rec, mu, logvar = O.cvae_forward(sd, torch.from_numpy(inp["sampled"]), torch.from_numpy(inp["label"]), eps)
    assert tuple(s.shape) == (n, 60, 512)
    assert rel_l2(digest(s.numpy(), 16384)["sample"], z["sample/sample"]) < TOL
    assert rel_l2(digest(rec.numpy(), 16384)["sample"], z["recon/sample"]) < TOL
    assert rel_l2(mu.numpy(), z["mu"]) < TOL and rel_l2(logvar.numpy(), z["logvar"]) < TOL


def test_state_dict_schema_matches_reference():
    """Keys, shapes and ORDER of our mirrors' state_dicts equal the reference's (tests/golden/state_dict_schema.json,
    dumped from the reference classes), so its checkpoints load with strict=True."""
    import json
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    schema = json.load(open(os.path.join(GOLDEN, "state_dict_schema.json")))
    for name, model in (("ted_spatial", build_mirror("spatial", 34, 126, 4, 4)), ("ted_memory", build_mirror("memory", 34, 126, 4, 4)),
                        ("cvae_v3", MLP_Reconstruct_v3())):
        mine = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        assert mine == schema[name], name


def test_mel_oracle_properties():
    """Mel front-end restatement (parity unpinned: librosa absent): shape, dB range, fp16 grid, ref=max -> 0 dB peak,
    the -80 dB floor and frame count 1 + n//hop (utils/train_utils_BEAT.py:186-195)."""
    from emotiongestures_amd.synth import synth_audio
    a = synth_audio(2, 64000, seed=0)
    s = O.melspectrogram(a, out_frames=124)
    assert s.shape == (2, 128, 124) and s.dtype == np.float32
    full = O.melspectrogram(a)
    assert full.shape == (2, 128, 126)
    assert np.all(full <= 0) and np.all(full >= -80.0)
    assert np.allclose(full.reshape(2, -1).max(axis=1), 0.0)
    assert np.array_equal(s, s.astype(np.float16).astype(np.float32))
    assert O.spectrogram_length(34, 15) == 70 and O.spectrogram_length(60, 15) == 124
    fb = O.mel_filterbank()
    assert fb.shape == (128, 513) and fb.min() >= 0 and np.all(fb.sum(axis=1) > 0)
    silent = O.melspectrogram(np.zeros((1, 16000), np.float32))
    assert np.all(silent == 0.0)          # all-amin input: db == ref everywhere


def test_mel_oracle_stft_cross_checked_against_torch_stft():
    """librosa is absent (mel parity is unpinned upstream-side), but the STFT half of the restatement can be pinned by an independent
    implementation in the image: torch.stft(center=True, pad_mode="constant", periodic Hann 1024, hop 512, onesided) is librosa.stft's
    documented default configuration.  It checks framing, padding, window and frame count independently of the oracle's np.fft indexing;
    the rest of the chain (Slaney basis, power_to_db(ref=max, top_db=80), fp16) is applied to both spectra and must agree on the fp16 grid."""
    from emotiongestures_amd.synth import synth_audio
    a = synth_audio(2, 64000, seed=3)
    a[1, :5000] = 0.0                                        # a silent head: the zero padding of the first centred frames must not matter
    spec = torch.stft(torch.from_numpy(a).double(), n_fft=1024, hop_length=512, win_length=1024,
                      window=torch.hann_window(1024, periodic=True, dtype=torch.float64), center=True, pad_mode="constant",
                      normalized=False, onesided=True, return_complex=True)                  # [B, 513, frames]
    assert spec.shape == (2, 513, 1 + 64000 // 512)
    power_t = (spec.abs() ** 2).numpy().transpose(0, 2, 1)                                   # [B, frames, 513]
    # the oracle's own framing, recomputed here the way melspectrogram() does it
    x = np.pad(a.astype(np.float64), ((0, 0), (512, 512)))
    idx = np.arange(1024)[None, :] + 512 * np.arange(126)[:, None]
    power_o = np.abs(np.fft.rfft(x[:, idx] * O.hann_periodic(1024).astype(np.float64), axis=-1)) ** 2
    # the oracle keeps the window in fp32 (as scipy's get_window result is cast by librosa): 1e-7 relative
    np.testing.assert_allclose(power_o, power_t, rtol=2e-6, atol=1e-9 * power_t.max())
    fb = O.mel_filterbank().astype(np.float64)

    def to_db(power):
        mel = np.einsum("mk,bfk->bmf", fb, power)
        db = 10.0 * np.log10(np.maximum(1e-10, mel))
        db -= 10.0 * np.log10(np.maximum(1e-10, mel.reshape(2, -1).max(axis=1)))[:, None, None]
        return np.maximum(db, db.reshape(2, -1).max(axis=1)[:, None, None] - 80.0).astype(np.float16).astype(np.float32)
    ours = O.melspectrogram(a)
    via_torch = to_db(power_t)
    assert ours.shape == via_torch.shape == (2, 128, 126)
    diff = np.abs(ours - via_torch)
    assert diff.max() <= 0.0625 and (diff > 0).mean() < 1e-3          # at most one fp16 step (dB in [-80, 0]: ulp <= 1/16) on a handful of bins


def test_beat_long_oracle_matches_reference_golden():
    """BASELINE configs[3] (120 frames, 128x312 spectrogram, 120-channel CVAE): the oracle against the reference's classes with
    their hard-coded sizes replaced after construction (tests/golden/make_golden_beat_long.py)."""
    from emotiongestures_amd.builders import make_args, make_lang
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    from emotiongestures_amd.Full_model.Models_spatial_memory import Transformer
    z = np.load(os.path.join(GOLDEN, "beat_long_b2.npz"))
    B, F, D, P, chunk, T, n_words, seed = [int(v) for v in z["meta"][:8]]
    model = Transformer(make_args(chunk), make_lang(n_words), frames=F, pose_dim=D, prior_frames=P, d_word_vec=512, d_model=512, d_inner=2048,
                        n_layers=3, n_head=8, d_k=64, d_v=64, n_position=F, spec_len=T)
    load_synth_weights(model, seed)
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=F), seed)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    sdv = {k: v.detach() for k, v in vae.state_dict().items()}
    inp = synth_inputs(B, F, D, P, spec_len=T, seed=seed)
    t = {k: torch.from_numpy(v) for k, v in inp.items()}
    with torch.no_grad():
        s = O.cvae_sample(sdv, t["label"], t["z"])
        pose, emo, sem, pred, _ = O.generator_forward(sd, O.GenCfg(frames=F, pose_dim=D, prior_frames=P, chunk=chunk), t["spec"], t["text"], t["pre_pose"], s)
    assert rel_l2(digest(s.numpy(), 16384)["sample"], z["cvae_sample/sample"]) < TOL
    assert clip_rel_l2(pose.numpy(), z["pose"]) < TOL
    assert rel_l2(pred.numpy(), z["emotion_prediction"]) < TOL
    assert rel_l2(digest(emo.numpy(), 8192)["sample"], z["emotion_feature/sample"]) < TOL


def _mask_inputs():
    from emotiongestures_amd.synth import hash_unit
    B, LQ, LK, D = 3, 34, 40, 512
    q = ((hash_unit("mask.q", B * LQ * D, 31) * 2 - 1)).astype(np.float32).reshape(B, LQ, D)
    kv = ((hash_unit("mask.kv", B * LK * D, 31) * 2 - 1)).astype(np.float32).reshape(B, LK, D)
    pad = np.ones((B, 1, LK), np.int64)
    pad[0, 0, 30:] = 0
    pad[1, 0, ::3] = 0
    pad[2, 0, :] = 0
    full = np.ones((B, LQ, LK), np.int64)
    for i in range(LQ):
        full[:, i, i + 1:] = 0
    full[1, 5, :] = 0
    return q, kv, {"pad": pad, "full": full}


def _mask_mha():
    from emotiongestures_amd.modules import MultiHeadAttention
    return load_synth_weights(MultiHeadAttention(8, 512, 64, 64, dropout=0.2), 31).eval()


def test_attention_mask_oracle_matches_reference_golden():
    """The reference's optional attention mask (Modules.py:18-19, SubLayers.py:44-45; never passed on the gesture path): padding mask broadcast
    over the queries, a full causal mask, and fully masked rows (uniform over ALL keys, as -1e9 everywhere gives)."""
    z = np.load(os.path.join(GOLDEN, "attention_mask.npz"))
    q, kv, masks = _mask_inputs()
    sd = {"m." + k: v.detach() for k, v in _mask_mha().state_dict().items()}
    for name, m in masks.items():
        with torch.no_grad():
            y, attn = O.multi_head_attention(sd, "m", torch.from_numpy(q), torch.from_numpy(kv), torch.from_numpy(kv), O.GenCfg(), mask=torch.from_numpy(m))
        assert rel_l2(y.numpy()[:, :, ::4], z[f"{name}/out"]) < 2e-6
        np.testing.assert_allclose(attn.numpy()[:, ::4], z[f"{name}/attn"], atol=2e-7)
    assert abs(float(z["pad/attn"][2].max()) - 1.0 / 40) < 1e-7          # every key masked: uniform over all 40


def test_mel_oracle_filterbank_and_db_cross_checked_against_transformers_audio_utils():
    """The other half of the mel front-end's pin (the Slaney filterbank and power_to_db were single-sourced).  librosa is not in
    the image, but `transformers.audio_utils` carries an independent implementation written to reproduce librosa: `mel_filter_bank(norm="slaney",
    mel_scale="slaney")` and `spectrogram(center=True, pad_mode="constant", power=2) -> power_to_db(reference=max, db_range=80)`.  The oracle's
    filterbank equals it to 1e-8, and the whole pipeline (STFT framing, window, power, mel, dB, fp16) gives the same fp16 image on random audio,
    silence-padded audio and a loud / quiet pair."""
    A = pytest.importorskip("transformers.audio_utils")
    from oracle import emogest_oracle as O
    fb = A.mel_filter_bank(num_frequency_bins=513, num_mel_filters=128, min_frequency=0.0, max_frequency=8000.0, sampling_rate=16000,
                           norm="slaney", mel_scale="slaney")
    mine = O.mel_filterbank(16000, 1024, 128)
    assert fb.T.shape == mine.shape and float(np.abs(fb.T - mine).max()) < 1e-8 and float(np.abs(mine).max()) > 0.01
    rng = np.random.RandomState(3)
    audio = (rng.randn(3, 64000) * 0.1).astype(np.float32)
    audio[1, 30000:] = 0.0                      # trailing silence: the top_db floor decides most bins
    audio[2] *= 1e-3                            # a quiet clip: ref = its own maximum
    win = A.window_function(1024, "hann", periodic=True)
    want = []
    for a in audio:
        s = A.spectrogram(a.astype(np.float64), win, frame_length=1024, hop_length=512, fft_length=1024, power=2.0, center=True, pad_mode="constant",
                          mel_filters=fb, mel_floor=0.0, dtype=np.float64)
        want.append(A.power_to_db(s, reference=float(s.max()), min_value=1e-10, db_range=80.0))
    want = np.stack(want).astype(np.float16).astype(np.float32)
    got = O.melspectrogram(audio)
    assert got.shape == want.shape == (3, 128, 126)
    diff = got != want
    assert diff.mean() < 1e-3 and float(np.abs(got - want).max()) <= 0.0625          # at most a stray fp16 ulp near 64 dB (measured: identical)
