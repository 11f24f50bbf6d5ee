"""MotionAE + EmbeddingSpaceEvaluator (the TED-style Frechet gesture distance): oracle and HIP path vs
goldens from the reference's own classes (tests/golden/make_golden_motion_ae.py)."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from emotiongestures_amd.synth import hash_unit, load_synth_weights

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, "golden", "motion_ae.npz"))


def poses(tag, n, seed):
    return ((hash_unit(tag, n * 34 * 126, seed) * 2 - 1) * 0.8).astype(np.float32).reshape(n, 34, 126)


def build():
    from emotiongestures_amd.model.motion_ae import MotionAE
    ae = MotionAE(126, 128).eval()
    load_synth_weights(ae, 41)
    return ae


def test_schema_and_oracle_match_reference():
    from oracle import emogest_oracle as O
    ae = build()
    schema = json.load(open(os.path.join(HERE, "golden", "motion_ae_schema.json")))
    assert [[k, list(v.shape)] for k, v in ae.state_dict().items()] == schema
    sd = {k: v.detach() for k, v in ae.state_dict().items()}
    with torch.no_grad():
        recon, z = O.motion_ae(sd, torch.from_numpy(poses("ae.in", 3, 1)))
    np.testing.assert_allclose(z.numpy(), G["z"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(recon.numpy(), G["recon"], rtol=1e-4, atol=1e-5)
    with pytest.raises(ValueError):
        from emotiongestures_amd.model.motion_ae import PoseEncoderConv
        PoseEncoderConv(64, 126, 128)


@pytest.mark.gpu
def test_gpu_motion_ae_and_evaluator_match_reference():
    from emotiongestures_amd.model.embedding_space_evaluator import EmbeddingSpaceEvaluator
    dev = torch.device("cuda:0")
    ae = build().to(dev)
    with torch.no_grad():
        recon, z = ae(torch.from_numpy(poses("ae.in", 3, 1)).to(dev))
    assert recon.shape == (3, 34, 126) and z.shape == (3, 128)
    assert np.linalg.norm(z.cpu().numpy() - G["z"]) / np.linalg.norm(G["z"]) < 2e-5
    assert np.linalg.norm(recon.cpu().numpy() - G["recon"]) / np.linalg.norm(G["recon"]) < 2e-5

    args = SimpleNamespace(n_pre_poses=4, n_poses=34, pose_dim=126, wordembed_dim=300)
    ckpt = {"pose_dim": 126, "latent_dim": 128, "motion_ae": build().state_dict()}
    ev = EmbeddingSpaceEvaluator(args, ckpt, None, dev)
    for i in range(3):
        real = torch.from_numpy(poses("ev.real", 48, 10 + i))
        gen = real * 0.9 + 0.1 * torch.from_numpy(poses("ev.gen", 48, 20 + i))
        ev.push_samples(None, None, gen.to(dev), real.to(dev))
    assert ev.get_no_of_samples() == int(G["n_samples"])
    fd, feat_dist = ev.get_scores()
    assert abs(fd - float(G["frechet"])) < 1e-3 * max(1.0, abs(float(G["frechet"])))
    assert abs(feat_dist - float(G["feat_dist"])) < 1e-4 * float(G["feat_dist"])
    np.testing.assert_allclose([float(v) for v in ev.recon_err_diff], G["recon_err_diff"], rtol=1e-4)
    # differences of two ~7e4-term fp32 sums of (1 - cos): noise-level quantities, compared on the scale of the sums
    np.testing.assert_allclose([float(v) for v in ev.cos_err_diff], G["cos_err_diff"], atol=0.7)
    ev.reset()
    assert ev.get_no_of_samples() == 0
    with pytest.raises(NotImplementedError):
        EmbeddingSpaceEvaluator(SimpleNamespace(n_pre_poses=4, n_poses=34, pose_dim=27), ckpt, None, dev)
