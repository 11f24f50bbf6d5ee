"""Eval-harness pieces: FGD auto-encoder, skeleton emotion classifier, Frechet distance, diversity
score.  CPU: oracle vs goldens produced by the reference's own code.  GPU: the HIP-backed mirrors vs the same goldens."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, build_mirror, clip_rel_l2, rel_l2
from emotiongestures_amd.synth import digest, hash_uniform, load_synth_weights, synth_inputs


def _load():
    z = np.load(os.path.join(GOLDEN, "harness.npz"))
    n, frames, dim, seed = [int(v) for v in z["meta"]]
    pred = torch.from_numpy(hash_uniform("h/pred", (n, frames, dim), -1.0, 1.0, seed))
    tgt = torch.from_numpy(hash_uniform("h/tgt", (n, frames, dim), -1.0, 1.0, seed))
    return z, pred, tgt, seed


def _mirrors(seed, precision="f32"):
    from emotiongestures_amd.model.FGD import MLP_Reconstruct
    from emotiongestures_amd.skeleton_classifer.Models import Transformer as Skel
    fgd = load_synth_weights(MLP_Reconstruct(precision=precision), seed).eval()
    skel = load_synth_weights(Skel(class_dim=8, pose_dim=282, d_word_vec=512, d_model=512, d_inner=2048, n_layers=3, n_head=8, d_k=64,
                                   d_v=64, n_position=60, precision=precision), seed).eval()
    return fgd, skel


def test_harness_schema_matches_reference():
    schema = json.load(open(os.path.join(GOLDEN, "harness_schema.json")))
    fgd, skel = _mirrors(0)
    assert [[k, list(v.shape)] for k, v in fgd.state_dict().items()] == schema["fgd"]
    assert [[k, list(v.shape)] for k, v in skel.state_dict().items()] == schema["skeleton"]


def test_oracle_harness_matches_reference_golden():
    from oracle import emogest_oracle as O
    z, pred, tgt, seed = _load()
    fgd, skel = _mirrors(seed)
    sd_f = {k: v.detach() for k, v in fgd.state_dict().items()}
    sd_s = {k: v.detach() for k, v in skel.state_dict().items()}
    with torch.no_grad():
        rec, pf = O.fgd_autoencoder(sd_f, pred)
        _, tf = O.fgd_autoencoder(sd_f, tgt)
        logits, mid = O.skeleton_classifier(sd_s, pred, O.GenCfg(frames=60, pose_dim=282))
    assert rel_l2(digest(pf.numpy(), 8192)["sample"], z["fgd_latent/sample"]) < 2e-5
    assert rel_l2(digest(rec.numpy(), 8192)["sample"], z["fgd_recon/sample"]) < 2e-5
    assert rel_l2(digest(mid.numpy(), 8192)["sample"], z["skel_mid/sample"]) < 2e-5
    assert rel_l2(logits.numpy(), z["logits"]) < 1e-4
    pa, ta = pf.reshape(-1, 512).numpy().astype(np.float64), tf.reshape(-1, 512).numpy().astype(np.float64)
    fid = O.frechet_distance(np.mean(pa, 0), np.cov(pa, rowvar=False), np.mean(ta, 0), np.cov(ta, rowvar=False))
    assert abs(fid - float(z["fid"])) < 1e-3 * abs(float(z["fid"]))
    np.random.seed(1234)
    div, interval = O.diversity_score(pa, 60)
    assert abs(float(div[0]) - float(z["div"])) < 1e-3 * float(z["div"])


def test_metric_functions_host_side():
    """Frechet / diversity / acc / L2 / MPJRE of the product harness (pure host code) against goldens and identities."""
    from emotiongestures_amd import harness as H
    from oracle import emogest_oracle as O
    z, pred, tgt, seed = _load()
    rng = np.random.RandomState(0)
    a, b = rng.randn(300, 16), rng.randn(300, 16) * 1.5 + 0.3
    f1 = H.calculate_frechet_distance(a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False))
    f2 = O.frechet_distance(a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False))
    assert abs(f1 - f2) < 1e-9 and f1 > 0
    assert abs(H.calculate_frechet_distance(a.mean(0), np.cov(a, rowvar=False), a.mean(0), np.cov(a, rowvar=False))) < 1e-6
    # model/FHD_score.py:196-213: ANY ValueError inside the try returns 100 -- the imaginary-component case and scipy refusing a NaN product alike;
    # model/embedding_space_evaluator.py:156-209 has no try: the same inputs raise there
    nan_cov = np.eye(4)
    nan_cov[0, 0] = np.nan
    indefinite = np.diag([1.0, -1.0])
    assert H.calculate_frechet_distance(np.zeros(4), np.eye(4), np.zeros(4), nan_cov) == 100
    assert H.calculate_frechet_distance(np.zeros(2), indefinite, np.zeros(2), np.eye(2)) == 100
    for bad in ((np.zeros(4), np.eye(4), np.zeros(4), nan_cov), (np.zeros(2), indefinite, np.zeros(2), np.eye(2))):
        with pytest.raises(ValueError):
            H.calculate_frechet_distance(*bad, imaginary="raise")
    act = rng.randn(8 * 60, 512)
    np.random.seed(7)
    d1, i1 = H.diversity_score(act, 60)
    np.random.seed(7)
    d2, i2 = O.diversity_score(act, 60)
    assert abs(float(d1[0]) - float(d2[0])) < 1e-4 * float(d2[0])
    lab = torch.tensor([1, 2, 3, 0])
    logits = torch.eye(8)[[1, 2, 0, 0]]
    assert float(H.compute_acc(lab, logits)) == 75.0
    assert H.l2_distance_pose(pred.numpy(), pred.numpy()) == 0.0
    assert abs(H.mpjre(tgt, pred) - float((tgt - pred).abs().mean())) < 1e-7
    assert H.calc_motion(pred).shape == (4, 59, 282)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_gpu_fgd_and_classifier_match_reference_golden(prec):
    z, pred, tgt, seed = _load()
    dev = torch.device("cuda:0")
    fgd, skel = _mirrors(seed, prec)
    fgd.to(dev); skel.to(dev)
    tol = 2e-5 if prec == "f32" else 1e-3
    with torch.no_grad():
        rec, pf = fgd(pred.to(dev))
        logits, mid = skel(pred.to(dev))
    assert tuple(pf.shape) == (4, 60, 512) and tuple(rec.shape) == (4, 60, 282)
    assert rel_l2(digest(pf.cpu().numpy(), 8192)["sample"], z["fgd_latent/sample"]) < tol
    assert rel_l2(digest(rec.cpu().numpy(), 8192)["sample"], z["fgd_recon/sample"]) < tol
    assert rel_l2(digest(mid.cpu().numpy(), 8192)["sample"], z["skel_mid/sample"]) < tol
    assert rel_l2(logits.cpu().numpy(), z["logits"]) < tol * 10


@pytest.mark.gpu
def test_gpu_eval_loop_end_to_end_vs_oracle():
    """harness.evaluate (CVAE sample -> generator -> FGD / classifier / metrics) on two synthetic BEAT-shaped batches against the
    same loop assembled from the CPU oracle."""
    from emotiongestures_amd import harness as H
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    from oracle import emogest_oracle as O
    dev = torch.device("cuda:0")
    F, D, P = 60, 282, 10
    gen = build_mirror("spatial", F, D, P, 10, seed=31)
    vae = load_synth_weights(MLP_Reconstruct_v3(), 31).eval()
    fgd, skel = _mirrors(31)
    sds = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in (gen, vae, fgd, skel)]
    batches, zs = [], []
    for i in range(2):
        inp = synth_inputs(3, F, D, P, seed=40 + i)
        pose = torch.from_numpy(hash_uniform(f"h/pose{i}", (3, F, D), -0.5, 0.5, 40 + i))
        batches.append({"spec": torch.from_numpy(inp["spec"]), "text": torch.from_numpy(inp["text"]), "pose_seq": pose,
                        "label": torch.from_numpy(inp["label"])})
        zs.append(torch.from_numpy(inp["z"]))
    for m in (gen, vae, fgd, skel):
        m.to(dev)
    np.random.seed(99)
    got = H.evaluate(gen, vae, fgd, skel, batches, P, device=dev, z_list=zs)
    # oracle loop
    pf, tf, l2s, rots, accs = [], [], [], [], []
    cfg = O.GenCfg(frames=F, pose_dim=D, prior_frames=P, chunk=10)
    with torch.no_grad():
        for bt, zz in zip(batches, zs):
            s = O.cvae_sample(sds[1], bt["label"], zz)
            pose = O.generator_forward(sds[0], cfg, bt["spec"], bt["text"], bt["pose_seq"][:, :P], s)[0]
            logits, _ = O.skeleton_classifier(sds[3], pose, cfg)
            accs.append(float(H.compute_acc(torch.max(bt["label"], 1)[1], logits)))
            rots.append(H.mpjre(bt["pose_seq"], pose))
            pf.append(O.fgd_autoencoder(sds[2], pose)[1].reshape(-1, 512).numpy().astype(np.float64))
            tf.append(O.fgd_autoencoder(sds[2], bt["pose_seq"])[1].reshape(-1, 512).numpy().astype(np.float64))
            l2s.append(H.l2_distance_pose(pose.numpy(), bt["pose_seq"].numpy()))
    pa, ta = np.concatenate(pf), np.concatenate(tf)
    fid = O.frechet_distance(pa.mean(0), np.cov(pa, rowvar=False), ta.mean(0), np.cov(ta, rowvar=False))
    assert abs(got["pose_l2"] - np.mean(l2s)) < 1e-4 * np.mean(l2s)
    assert abs(got["rotation_deg"] - np.mean(rots) * 57.2958) < 1e-4 * got["rotation_deg"]
    assert abs(got["fgd"] - fid) < 2e-3 * abs(fid)
    assert got["emotion_acc"] == np.mean(accs)


@pytest.mark.gpu
def test_frechet_accumulator_matches_np_cov():
    """On-device (n, sum x, sum x x^T) accumulators against np.mean / np.cov on the same feature rows, pushed
    in ragged batches: mean / covariance within 1e-5 of the float64 estimator and the Frechet distance of two feature sets within 1e-4."""
    from emotiongestures_amd.harness import FrechetAccumulator, calculate_frechet_distance
    from emotiongestures_amd.synth import hash_unit
    dev = torch.device("cuda:0")
    rng = lambda key, n: (hash_unit(key, n * 512, 3).reshape(n, 512) * 4 - 1.5).astype(np.float32)
    a, b = rng("fa", 3000), rng("fb", 2500) * 1.1 + 0.2
    accs = []
    for arr in (a, b):
        acc = FrechetAccumulator(512, dev)
        for lo, hi in ((0, 700), (700, 701), (701, 2048), (2048, len(arr))):
            acc.push(torch.from_numpy(arr[lo:hi]).to(dev))
        mu, sig = acc.stats()
        m64, c64 = arr.astype(np.float64).mean(0), np.cov(arr.astype(np.float64), rowvar=False)
        assert np.abs(mu - m64).max() < 1e-5 and np.abs(sig - c64).max() < 1e-5 * np.abs(c64).max() + 1e-6
        accs.append((mu, sig))
    ref = calculate_frechet_distance(a.astype(np.float64).mean(0), np.cov(a.astype(np.float64), rowvar=False),
                                     b.astype(np.float64).mean(0), np.cov(b.astype(np.float64), rowvar=False))
    got = calculate_frechet_distance(accs[0][0], accs[0][1], accs[1][0], accs[1][1])
    assert abs(np.real(got) - np.real(ref)) / abs(np.real(ref)) < 1e-4
