"""Whole-path parity on the GPU: eg_generator_forward / eg_cvae_* / eg_melspectrogram through the host mirror,
against (1) the golden vectors produced by the reference itself and (2) the CPU oracle on the same seeded inputs."""
import os

import numpy as np
import pytest
import torch

from conftest import GENERATOR_CASES, GOLDEN, build_mirror, clip_rel_l2, golden_meta, rel_l2
from emotiongestures_amd.synth import digest, load_synth_weights, synth_audio, synth_inputs

pytestmark = pytest.mark.gpu

# north-star bar: per-clip relative L2 of the pose vs the reference CPU path <= 1e-3.
POSE_TOL = {"f32": 2e-5, "bf16x3": 1e-3}


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run(name, variant, prec, keep_taps=False, fold_affine=False):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = golden_meta(z)
    model = build_mirror(variant, m["frames"], m["pose_dim"], m["prior"], m["chunk"], m["n_words"], m["seed"], m["spec_len"], precision=prec)
    model.keep_taps = keep_taps
    model.fold_affine = fold_affine
    model.to(dev())
    inp = synth_inputs(m["batch"], m["frames"], m["pose_dim"], m["prior"], spec_len=m["spec_len"], n_words=m["n_words"], seed=m["seed"])
    g = {k: torch.from_numpy(v).to(dev()) for k, v in inp.items()}
    with torch.no_grad():
        out = model(g["spec"], g["text"], g["pre_pose"], g["sampled"] if m["use_sampled"] else None)
    torch.cuda.synchronize()
    return z, m, model, out


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
@pytest.mark.parametrize("name,variant", sorted(GENERATOR_CASES.items()))
def test_generator_matches_reference_golden(name, variant, prec):
    z, m, model, (pose, emo, sem, pred, text) = _run(name, variant, prec)
    tol = POSE_TOL[prec]
    assert tuple(pose.shape) == z["pose"].shape
    assert clip_rel_l2(pose.cpu().numpy(), z["pose"]) < tol
    assert rel_l2(pred.cpu().numpy(), z["emotion_prediction"]) < tol * 5
    for key, t in (("emotion_feature", emo), ("semantic_feature", sem), ("text_embedding", text)):
        d = digest(t.cpu().numpy(), 8192)
        assert tuple(d["shape"]) == tuple(z[key + "/shape"])
        assert rel_l2(d["sample"], z[key + "/sample"]) < tol, key


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
@pytest.mark.parametrize("name,variant", [("ted_spatial_b2", "spatial"), ("ted_memory_b4", "memory"), ("beat_spatial_b1", "spatial")])
def test_fold_affine_matches_reference_golden(name, variant, prec):
    """fold_affine=True: the Dropout-only Linear chains (post_projector, emotion_proj, semantic_proj, post_header, fc1 -> fc2:
    Models_spatial_memory.py:128-130,360-364,488-496,509-517,528-536) folded into one product each at pack time.  Exact algebra
    in eval mode, different rounding: same tolerances as the unfolded path, and fewer arena entries."""
    z, m, model, (pose, emo, sem, pred, text) = _run(name, variant, prec, fold_affine=True)
    tol = POSE_TOL[prec]
    assert clip_rel_l2(pose.cpu().numpy(), z["pose"]) < tol
    assert rel_l2(pred.cpu().numpy(), z["emotion_prediction"]) < tol * 5
    for key, t in (("emotion_feature", emo), ("semantic_feature", sem)):
        assert rel_l2(digest(t.cpu().numpy(), 8192)["sample"], z[key + "/sample"]) < tol, key
    keys = [e.key.decode() for e in model.engine().entries]
    assert any("@" in k for k in keys) and not any(k.startswith("post_projector.2.") for k in keys)
    model.fold_affine = False
    assert not any("@" in e.key.decode() for e in model.engine().entries)          # the flag rebuilds the engine


def test_generator_taps_match_reference_golden():
    """Stage-by-stage: every tap the golden file holds, in fp32 mode (NHWC taps are permuted to the reference's NCHW)."""
    z, m, model, _ = _run("ted_spatial_b2", "spatial", "f32", keep_taps=True)
    eng, B = model.engine(), m["batch"]
    shapes = {"stem": (B, 128, 124, 32), "layer1": (B, 128, 124, 32), "layer2": (B, 64, 62, 64), "layer3": (B, 32, 31, 128)}
    for tap, shp in shapes.items():
        t = eng.tap(tap, B).view(shp).permute(0, 3, 1, 2).contiguous().cpu().numpy()
        assert rel_l2(digest(t)["sample"], z[f"tap_{tap}/sample"]) < 1e-5, tap
    for tap in ("audio_feat", "prior_enc", "fusion"):
        t = eng.tap(tap, B).view(B, 34, 512).cpu().numpy()
        assert rel_l2(digest(t)["sample"], z[f"tap_{tap}/sample"]) < 1e-5, tap
    t = eng.tap("enc_out", B).view(B, 34, 512).cpu().numpy()
    assert rel_l2(digest(t)["sample"], z["tap_enc2/sample"]) < 3e-5       # after 3 attention + FFN + LayerNorm layers


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_generator_matches_oracle_same_inputs(prec):
    """HIP path vs the CPU oracle on fresh seeded inputs (B=3, odd batch), full tensors."""
    from oracle import emogest_oracle as O
    model = build_mirror("spatial", 34, 126, 4, 4, seed=11, precision=prec)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    inp = synth_inputs(3, seed=11)
    t = {k: torch.from_numpy(v) for k, v in inp.items()}
    with torch.no_grad():
        ref = O.generator_forward(sd, O.GenCfg(), t["spec"], t["text"], t["pre_pose"], t["sampled"])
    model.to(dev())
    with torch.no_grad():
        got = model(t["spec"].to(dev()), t["text"].to(dev()), t["pre_pose"].to(dev()), t["sampled"].to(dev()))
    tol = POSE_TOL[prec]
    for name, a, b in zip(("pose", "emotion_feature", "semantic_feature", "emotion_prediction", "text_embedding"), got, ref):
        assert clip_rel_l2(a.cpu().numpy(), b.numpy()) < tol * (5 if name == "emotion_prediction" else 1), name


def test_batch_invariance_and_repeatability():
    """Clips are independent in the spatial variant: clip i of a batch equals clip i alone; two runs
    of the same batch are bitwise identical (no atomics anywhere on the path)."""
    model = build_mirror("spatial", 34, 126, 4, 4, seed=5).to(dev())
    inp = synth_inputs(4, seed=5)
    g = {k: torch.from_numpy(v).to(dev()) for k, v in inp.items()}
    with torch.no_grad():
        a = model(g["spec"], g["text"], g["pre_pose"])[0]
        b = model(g["spec"], g["text"], g["pre_pose"])[0]
        c = model(g["spec"][2:3], g["text"][2:3], g["pre_pose"][2:3])[0]
    assert torch.equal(a, b)
    assert clip_rel_l2(a[2:3].cpu().numpy(), c.cpu().numpy()) < 1e-5


def test_generator_draws_equal_per_draw_forward():
    """eg_generator_forward_draws (tower once, transformer per draw) == forward() called once per draw."""
    model = build_mirror("spatial", 34, 126, 4, 4, seed=9).to(dev())
    B, R = 2, 3
    inp = synth_inputs(B, seed=9)
    g = {k: torch.from_numpy(v).to(dev()) for k, v in inp.items()}
    draws = torch.stack([torch.from_numpy(synth_inputs(B, seed=100 + r)["sampled"]) for r in range(R)], 1).to(dev())
    with torch.no_grad():
        poses = model.forward_draws(g["spec"], g["pre_pose"], draws)
        for r in range(R):
            ref = model(g["spec"], g["text"], g["pre_pose"], draws[:, r].contiguous())[0]
            assert clip_rel_l2(poses[:, r].cpu().numpy(), ref.cpu().numpy()) < 1e-6


def test_diversity_sampling_64x32_matches_oracle():
    """BASELINE configs[4] at full size: 64 clips x 32 CVAE draws (label one-hot + z -> MLP_Reconstruct_v3.sample,
    CAVE/BEAT_CVAE.py:427-447) through forward_draws (audio tower once per clip, fusion -> encoder -> decoder -> post_projector per
    draw, M = 69 632 rows on the 128x128-tile GEMM), bf16x3.  A subset of (clip, draw) pairs against the CPU oracle within the
    north-star's 1e-3, the FGD auto-encoder features of all 2048 sequences finite, and the diversity statistic of the harness."""
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    from emotiongestures_amd.harness import MLP_Reconstruct, diversity_score
    from emotiongestures_amd.synth import hash_unit
    from oracle import emogest_oracle as O
    B, R, F = 64, 32, 34
    model = build_mirror("spatial", F, 126, 4, 4, seed=11, precision="bf16x3")
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=F), 11).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sdv = {k: v.detach().clone() for k, v in vae.state_dict().items()}
    inp = synth_inputs(B, F, 126, 4, seed=11)
    z = (hash_unit("draws.z", B * R * 32, 11).reshape(B, R, 32) * 2 - 1).astype(np.float32) * 1.7          # ~unit variance
    model.to(dev()); vae.to(dev())
    g = {k: torch.from_numpy(v).to(dev()) for k, v in inp.items()}
    with torch.no_grad():
        lab = g["label"][:, None, :].expand(B, R, 8).reshape(B * R, 8).contiguous()
        sampled = vae.sample(lab, z=torch.from_numpy(z).reshape(B * R, 32)).view(B, R, F, 512)
        poses = model.forward_draws(g["spec"], g["pre_pose"], sampled)
    torch.cuda.synchronize()
    assert tuple(poses.shape) == (B, R, F, 126) and bool(torch.isfinite(poses).all())
    clips, draws = [0, 17, 63], [0, 13, 31]
    t = {k: torch.from_numpy(v[clips]) for k, v in inp.items()}
    for r in draws:
        with torch.no_grad():
            s_ref = O.cvae_sample(sdv, t["label"], torch.from_numpy(z[clips, r]))
            ref = O.generator_forward(sd, O.GenCfg(), t["spec"], t["text"], t["pre_pose"], s_ref)[0]
        assert clip_rel_l2(poses[clips, r].cpu().numpy(), ref.numpy()) < 1e-3, f"draw {r}"
    # draws of one clip differ from each other (the sampling is not degenerate) ...
    assert float((poses[:, 0] - poses[:, 1]).abs().max()) > 1e-3
    # ... and the evaluator side of configs[4] runs on them (model/FGD.py features -> model/FHD_score.py:247-311 diversity)
    ae = load_synth_weights(MLP_Reconstruct(pose_dim=126), 5).eval().to(dev())
    with torch.no_grad():
        feats = ae(poses.view(B * R, F, 126).contiguous())[1].reshape(-1, 512).cpu().numpy().astype(np.float64)
    assert np.isfinite(feats).all() and feats.shape == (B * R * F, 512)
    np.random.seed(1234)
    div, _ = diversity_score(feats, frames=F)
    assert np.isfinite(div).all() and float(div[0]) > 0.0


def test_state_dict_roundtrip_and_module_prefix():
    """load_state_dict of a DataParallel-style checkpoint ('module.' prefix stripped as the reference's loaders do,
    test_emotion_gesture_diversity_iterative.py:149) gives the same poses; new weights trigger a repack."""
    a = build_mirror("spatial", 34, 126, 4, 4, seed=1)
    b = build_mirror("spatial", 34, 126, 4, 4, seed=2)
    ckpt = {"module." + k: v.clone() for k, v in a.state_dict().items()}
    inp = synth_inputs(1, seed=1)
    g = {k: torch.from_numpy(v).to(dev()) for k, v in inp.items()}
    a.to(dev()); b.to(dev())
    with torch.no_grad():
        pa = a(g["spec"], g["text"], g["pre_pose"])[0]
        pb0 = b(g["spec"], g["text"], g["pre_pose"])[0]
        b.load_state_dict({k.replace("module.", ""): v for k, v in ckpt.items()})
        pb1 = b(g["spec"], g["text"], g["pre_pose"])[0]
    assert not torch.equal(pa, pb0)
    assert torch.equal(pa, pb1)


def test_errors_are_loud():
    from emotiongestures_amd._lib import EgError
    model = build_mirror("spatial", 34, 126, 4, 4)
    inp = synth_inputs(1)
    t = {k: torch.from_numpy(v) for k, v in inp.items()}
    with pytest.raises(EgError):                       # CPU model: no fallback
        model(t["spec"], t["text"], t["pre_pose"])
    model.to(dev())
    with pytest.raises(EgError):                       # wrong spectrogram width
        model(t["spec"][:, :, :100].to(dev()), t["text"].to(dev()), t["pre_pose"].to(dev()))
    # train() mode: the spatial generator runs on the differentiable HIP operators (tests/test_gpu_training.py); modules without a
    # train-mode path still refuse instead of falling back
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=34), 0).to(dev()).train()
    with pytest.raises(NotImplementedError):           # sample() is an eval-time call (forward() has the train-mode path)
        vae.sample(t["label"].to(dev()), z=t["z"])
    model.train()
    pose = model(t["spec"].to(dev()), t["text"].to(dev()), t["pre_pose"].to(dev()))[0]
    assert pose.requires_grad and tuple(pose.shape) == (1, 34, 126)


def test_cvae_matches_reference_golden():
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    z = np.load(os.path.join(GOLDEN, "cvae_v3.npz"))
    n, seed = [int(v) for v in z["meta"]]
    m = load_synth_weights(MLP_Reconstruct_v3(), seed).eval().to(dev())
    inp = synth_inputs(n, frames=60, seed=seed)
    eps = torch.from_numpy(synth_inputs(n, seed=seed + 1)["z"]).to(dev())
    with torch.no_grad():
        s = m.sample(torch.from_numpy(inp["label"]).to(dev()), z=torch.from_numpy(inp["z"]))
        rec, mu, logvar = m(torch.from_numpy(inp["sampled"]).to(dev()), torch.from_numpy(inp["label"]).to(dev()), eps)
    assert tuple(s.shape) == (n, 60, 512)
    assert rel_l2(digest(s.cpu().numpy(), 16384)["sample"], z["sample/sample"]) < 1e-5
    assert rel_l2(digest(rec.cpu().numpy(), 16384)["sample"], z["recon/sample"]) < 1e-5
    assert rel_l2(mu.cpu().numpy(), z["mu"]) < 1e-5 and rel_l2(logvar.cpu().numpy(), z["logvar"]) < 1e-5


def test_cvae_sample_uses_cpu_generator_like_the_reference():
    """sample(y) draws torch.randn(n,32) on the CPU generator (CAVE/BEAT_CVAE.py:441): same seed -> same z."""
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    m = load_synth_weights(MLP_Reconstruct_v3(), 0).eval().to(dev())
    y = torch.from_numpy(synth_inputs(2)["label"]).to(dev())
    torch.manual_seed(123)
    zref = torch.randn(2, 32)
    torch.manual_seed(123)
    with torch.no_grad():
        a = m.sample(y)
        b = m.sample(y, z=zref)
    assert torch.equal(a, b)


def test_melspectrogram_matches_oracle():
    """HIP mel front-end vs the float64 numpy restatement of librosa's defaults (parity unpinned upstream): the output is
    fp16-rounded dB, so allow one fp16 ulp on a small fraction of bins (rounding-boundary flips)."""
    from emotiongestures_amd.engine import MelFrontEnd
    from oracle import emogest_oracle as O
    audio = synth_audio(3, 64000, seed=0)
    ref = O.melspectrogram(audio, out_frames=124)
    mel = MelFrontEnd(dev())
    got = mel(torch.from_numpy(audio).to(dev()), out_frames=124).cpu().numpy()
    assert got.shape == ref.shape
    diff = np.abs(got - ref)
    assert diff.max() <= 0.0626, diff.max()
    assert (diff > 0).mean() < 0.01
    assert np.array_equal(got, got.astype(np.float16).astype(np.float32))
    short = synth_audio(1, 16000, seed=1)                     # ragged: 1 s clip, 32 frames
    g2 = mel(torch.from_numpy(short).to(dev())).cpu().numpy()
    assert g2.shape == (1, 128, 32)
    assert np.abs(g2 - O.melspectrogram(short)).max() <= 0.0626
    silent = mel(torch.zeros(1, 16000, device=dev())).cpu().numpy()
    assert np.all(silent == 0.0)
    # the kernel transforms frames in pairs (even frame = real part, odd frame = imaginary part of one complex FFT): a quiet frame next to a
    # loud one must not pick up cross-talk above the 80 dB floor -- one second of near silence (1e-4 of full scale), then full-scale noise,
    # with the quiet / loud boundary on an odd frame index and an odd frame count (last pair half empty)
    step = synth_audio(1, 32 * 512, seed=2)
    step[:, :15 * 512 + 256] *= 1e-4
    g3 = mel(torch.from_numpy(step).to(dev())).cpu().numpy()
    r3 = O.melspectrogram(step)
    assert g3.shape == r3.shape == (1, 128, 33)
    d3 = np.abs(g3 - r3)
    assert d3.max() <= 0.0626 and (d3 > 0).mean() < 0.01, (d3.max(), (d3 > 0).mean())


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_beat_long_120_frames_matches_reference_golden_and_oracle(prec):
    """BASELINE config 4 (10 s audio -> spec 128x312, 120 frames, pose_dim 282, prior 10) against the REFERENCE's classes with
    their hard-coded sizes (32*31, 60 CVAE channels) replaced after construction
    (tests/golden/make_golden_beat_long.py -> beat_long_b2.npz), and against the oracle."""
    from conftest import make_args, make_lang
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    from emotiongestures_amd.Full_model.Models_spatial_memory import Transformer
    from oracle import emogest_oracle as O
    F, D, P, T = 120, 282, 10, 312
    model = Transformer(make_args(10), make_lang(200), frames=F, pose_dim=D, prior_frames=P, d_word_vec=512, d_model=512, d_inner=2048,
                        n_layers=3, n_head=8, d_k=64, d_v=64, n_position=F, spec_len=T, precision=prec)
    load_synth_weights(model, 21).eval()
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=F), 21).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sdv = {k: v.detach().clone() for k, v in vae.state_dict().items()}
    inp = synth_inputs(2, F, D, P, spec_len=T, seed=21)
    t = {k: torch.from_numpy(v) for k, v in inp.items()}
    with torch.no_grad():
        s_ref = O.cvae_sample(sdv, t["label"], t["z"])
        ref = O.generator_forward(sd, O.GenCfg(frames=F, pose_dim=D, prior_frames=P, chunk=10), t["spec"], t["text"], t["pre_pose"], s_ref)
    model.to(dev()); vae.to(dev())
    with torch.no_grad():
        s = vae.sample(t["label"].to(dev()), z=t["z"])
        got = model(t["spec"].to(dev()), t["text"].to(dev()), t["pre_pose"].to(dev()), s)
    assert tuple(got[0].shape) == (2, F, D)
    assert rel_l2(s.cpu().numpy(), s_ref.numpy()) < 1e-5
    assert clip_rel_l2(got[0].cpu().numpy(), ref[0].numpy()) < POSE_TOL[prec]
    z = np.load(os.path.join(GOLDEN, "beat_long_b2.npz"))                   # the reference itself
    assert [int(v) for v in z["meta"][:8]] == [2, F, D, P, 10, T, 200, 21]
    assert rel_l2(digest(s.cpu().numpy(), 16384)["sample"], z["cvae_sample/sample"]) < 1e-5
    assert clip_rel_l2(got[0].cpu().numpy(), z["pose"]) < POSE_TOL[prec]
    assert rel_l2(got[3].cpu().numpy(), z["emotion_prediction"]) < POSE_TOL[prec] * 5
    assert rel_l2(digest(got[1].cpu().numpy(), 8192)["sample"], z["emotion_feature/sample"]) < POSE_TOL[prec]
    assert rel_l2(got[3].cpu().numpy(), ref[3].numpy()) < POSE_TOL[prec] * 5
