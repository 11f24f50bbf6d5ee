"""EmotionNet, the audio emotion classifier, inference forward: oracle and HIP path vs the reference's own
model.audio_emotion_classifer.EmotionNet (tests/golden/make_golden_emotion_net.py).  Exercises the 256-channel conv stage."""
import json
import os

import numpy as np
import pytest
import torch

from emotiongestures_amd.synth import hash_unit, load_synth_weights

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, "golden", "emotion_net.npz"))


def emotion_input(batch, seed):
    v = (-80.0 * hash_unit("emotion.mfcc", batch * 128 * 128, seed)).astype(np.float16).astype(np.float32)
    return v.reshape(batch, 128, 128)


def build(precision="f32"):
    from emotiongestures_amd.model.audio_emotion_classifer import EmotionNet
    net = EmotionNet(precision=precision).eval()
    load_synth_weights(net, 31)
    return net


def test_schema_and_oracle_match_reference():
    from oracle import emogest_oracle as O
    net = build()
    schema = json.load(open(os.path.join(HERE, "golden", "emotion_net_schema.json")))
    assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == schema
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    x = torch.from_numpy(emotion_input(2, 31))
    with torch.no_grad():
        feat = O.resnetse(sd, "emotion_encoder", x.unsqueeze(1), layers=(3, 4, 6, 3))
        logits = O.emotion_net(sd, x)
    np.testing.assert_allclose(feat[:, :, :4, :4].numpy(), G["feat_corner"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(feat.mean(dim=(0, 2, 3)).numpy(), G["feat_mean"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(logits.numpy(), G["logits"], rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError):
        net(torch.zeros(1, 128, 124))


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("bf16x3", 1e-3)])
def test_gpu_emotion_net_matches_reference(prec, tol):
    dev = torch.device("cuda:0")
    net = build(prec).to(dev)
    x = torch.from_numpy(emotion_input(2, 31)).to(dev)
    with torch.no_grad():
        feat = net.emotion_encoder.forward_nhwc(x)                  # [B,16,16,256]
        logits = net(x)
    assert feat.shape == (2, 16, 16, 256)
    corner = feat[:, :4, :4, :].permute(0, 3, 1, 2).cpu().numpy()
    ref = G["feat_corner"]
    assert np.linalg.norm(corner - ref) / np.linalg.norm(ref) < tol
    got = logits.cpu().numpy()
    assert np.linalg.norm(got - G["logits"]) / np.linalg.norm(G["logits"]) < tol
    assert (got.argmax(1) == G["logits"].argmax(1)).all()
