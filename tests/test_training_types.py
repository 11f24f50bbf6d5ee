"""Training-side types on the path, forward only: oracle vs goldens captured from the reference's own
SoftmaxContrastiveLoss / adjust_lr / calc_motion / Motion_Discriminator (tests/golden/make_golden_training_types.py), and the
HIP path vs both."""
import json
import os

import numpy as np
import pytest
import torch

from emotiongestures_amd.synth import hash_unit

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, "golden", "training_types.npz"))
SCL = ("small", "wide", "single", "identical")


def feats(tag, n, d, seed, corr):
    f = (hash_unit(tag + ".face", n * d, seed) * 2 - 1).astype(np.float32).reshape(n, d)
    a = (hash_unit(tag + ".audio", n * d, seed) * 2 - 1).astype(np.float32).reshape(n, d)
    return f, (corr * f + (1 - corr) * a).astype(np.float32)


def _motion():
    return (hash_unit("motion", 3 * 60 * 128, 5) * 2 - 1).astype(np.float32).reshape(3, 60, 128)


def _disc(precision="f32"):
    from emotiongestures_amd.Full_model.Models_memory import Motion_Discriminator
    from emotiongestures_amd.synth import load_synth_weights
    md = Motion_Discriminator(frames=59, pose_dim=128, d_word_vec=128, d_model=128, d_inner=1024, n_layers=2, n_head=8, d_k=64, d_v=64,
                              n_position=59, precision=precision).eval()
    load_synth_weights(md, 21)
    return md


@pytest.mark.parametrize("tag", SCL)
def test_oracle_contrastive_matches_reference(tag):
    from oracle import emogest_oracle as O
    n, d, corr = G[f"scl.{tag}.meta"]
    f, a = feats(tag, int(n), int(d), 3, float(corr))
    loss, acc, cross = O.softmax_contrastive(f, a)
    assert abs(loss - float(G[f"scl.{tag}.loss"])) <= 2e-5 * max(1.0, abs(loss))
    if tag != "identical":      # all-equal rows of 1e8: the argmax is decided by fp32 rounding noise upstream
        assert abs(acc - float(G[f"scl.{tag}.acc"])) < 1e-6
    rows = G[f"scl.{tag}.cross"].shape[0]
    np.testing.assert_allclose(cross[:rows], G[f"scl.{tag}.cross"], rtol=2e-4)


def test_oracle_schedule_motion_and_discriminator_match_reference():
    from oracle import emogest_oracle as O
    table = np.array([O.adjust_lr_value(1e-3, e) for e in range(151)])
    np.testing.assert_array_equal(table, G["adjust_lr.table"])
    with pytest.raises(ValueError):
        O.adjust_lr_value(1e-3, 151)
    m = torch.from_numpy(_motion())
    off = O.calc_motion(m)
    np.testing.assert_array_equal(off.numpy(), G["calc_motion.out"])
    md = _disc()
    schema = json.load(open(os.path.join(HERE, "golden", "motion_disc_schema.json")))
    assert [[k, list(v.shape)] for k, v in md.state_dict().items()] == schema
    sd = {k: v.detach().clone() for k, v in md.state_dict().items()}
    cfg = O.GenCfg(d_model=128, d_inner=1024, n_layers=2, n_head=8, d_k=64, d_v=64)
    with torch.no_grad():
        out = O.motion_discriminator(sd, off, cfg)
    np.testing.assert_allclose(out.numpy(), G["motion_disc.out"], rtol=1e-4, atol=1e-5)


def test_host_side_helpers():
    from emotiongestures_amd import harness as H

    class Opt:
        param_groups = [{"lr": 0.0}, {"lr": 0.0}]
    o = Opt()
    for e in (0, 15, 16, 50, 51, 80, 81, 100, 101, 150):
        H.adjust_lr(o, 1e-3, e)
        assert o.param_groups[0]["lr"] == o.param_groups[1]["lr"] == G["adjust_lr.table"][e]
    with pytest.raises(ValueError):
        H.adjust_lr(o, 1e-3, 151)
    np.testing.assert_array_equal(H.calc_motion(torch.from_numpy(_motion())).numpy(), G["calc_motion.out"])
    logits = (hash_unit("logits", 16 * 8, 1) * 2 - 1).astype(np.float32).reshape(16, 8)
    labels = (hash_unit("labels", 16, 1) * 8).astype(np.int64)
    assert float(H.compute_acc(torch.from_numpy(labels), torch.from_numpy(logits))) == float(G["compute_acc.out"])
    m = _motion()
    assert abs(H.l2_distance_pose(m[:, :, :64], m[:, :, 64:]) - float(G["l2_distance_pose.out"])) < 1e-6
    with pytest.raises(ValueError):          # the upstream default combination cannot run its own forward
        H.Motion_Discriminator()
    net = torch.nn.Linear(2, 2)
    H.set_requires_grad(net, False)
    assert not any(p.requires_grad for p in net.parameters())
    with pytest.raises(RuntimeError):        # no CPU fallback
        H.SoftmaxContrastiveLoss().evaluate(torch.zeros(2, 4), torch.zeros(2, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", SCL)
def test_gpu_contrastive_matches_reference(tag):
    from emotiongestures_amd.harness import SoftmaxContrastiveLoss
    from oracle import emogest_oracle as O
    dev = torch.device("cuda:0")
    n, d, corr = G[f"scl.{tag}.meta"]
    f, a = feats(tag, int(n), int(d), 3, float(corr))
    crit = SoftmaxContrastiveLoss()
    loss = crit(torch.from_numpy(f), torch.from_numpy(a), dev)
    acc, cross = crit.evaluate(torch.from_numpy(f).to(dev), torch.from_numpy(a).to(dev))
    ref_loss, ref_acc, ref_cross = O.softmax_contrastive(f, a)
    assert abs(float(loss) - float(G[f"scl.{tag}.loss"])) <= 2e-5 * max(1.0, abs(ref_loss))
    assert abs(float(loss) - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss))
    if tag != "identical":
        assert abs(float(acc) - float(G[f"scl.{tag}.acc"])) < 1e-6
        # 1/(dist + 1e-8) amplifies fp32 rounding of small distances; compare where the distance is well conditioned
        rows = G[f"scl.{tag}.cross"].shape[0]
        np.testing.assert_allclose(cross.cpu().numpy()[:rows], G[f"scl.{tag}.cross"], rtol=2e-4)
        np.testing.assert_allclose(cross.cpu().numpy(), ref_cross, rtol=2e-4)
    # repeatable bit for bit (fixed-order reductions)
    assert torch.equal(crit(torch.from_numpy(f), torch.from_numpy(a), dev), loss)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("bf16x3", 2e-4)])
def test_gpu_motion_discriminator_matches_reference(prec, tol):
    from emotiongestures_amd import harness as H
    dev = torch.device("cuda:0")
    md = _disc(prec).to(dev)
    off = H.calc_motion(torch.from_numpy(_motion()).to(dev))
    with torch.no_grad():
        out = md(off)
    ref = G["motion_disc.out"]
    assert out.shape == (3, 1)
    assert np.abs(out.cpu().numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max())
    with pytest.raises(ValueError):
        md(off[:, :10])


ADV = np.load(os.path.join(HERE, "golden", "adv_grads.npz"), allow_pickle=True)


def _check_fingerprint(case, named_grads, tol):
    worst = 0.0
    for k, g in named_grads.items():
        ref_norm = float(ADV[f"{case}/g/{k}/norm"])
        gd = g.detach().reshape(-1).double().cpu().numpy()
        stride = max(1, gd.size // 64)
        samp = gd[::stride][:64]
        ref = ADV[f"{case}/g/{k}/sample"].astype(np.float64)
        err = max(abs(np.linalg.norm(gd) - ref_norm) / (ref_norm + 1e-30), np.abs(samp - ref).max() / (np.abs(ref).max() + 1e-30))
        worst = max(worst, err)
        assert err < tol, (k, err)
    return worst


@pytest.mark.gpu
def test_gpu_motion_discriminator_training_gradients_match_reference():
    """Motion_Discriminator in train() mode (dropout p = 0) on the differentiable HIP operators: logit, loss, every parameter gradient and the
    input gradient against the REFERENCE's autograd (tests/golden/make_golden_adv_grad.py); the parameters the reference leaves without a
    gradient (position embedding table, the encoder's unused final LayerNorm) get none here either."""
    from emotiongestures_amd import harness as H
    from emotiongestures_amd.train import functional as F
    dev = torch.device("cuda:0")
    md = _disc("f32").to(dev).train()
    off = H.calc_motion(torch.from_numpy(_motion()).to(dev)).detach().requires_grad_(True)
    logit = md(off)
    np.testing.assert_allclose(logit.detach().cpu().numpy(), ADV["disc/logit"], rtol=2e-5, atol=2e-5)
    loss = F.smooth_l1_loss(logit, torch.ones_like(logit), 1.0, 1.0)
    loss.backward()
    assert abs(float(loss.detach()) - float(ADV["disc/loss"])) <= 1e-5 * abs(float(ADV["disc/loss"]))
    nograd = sorted(k for k, p in md.named_parameters() if p.grad is None)
    assert nograd == sorted(str(k) for k in ADV["disc/nograd"])
    _check_fingerprint("disc", {k: p.grad for k, p in md.named_parameters() if p.grad is not None}, 2e-4)
    gx = off.grad.reshape(-1).double().cpu().numpy()
    ref = ADV["disc/dx/sample"].astype(np.float64)
    assert abs(np.linalg.norm(gx) - float(ADV["disc/dx/norm"])) <= 2e-4 * float(ADV["disc/dx/norm"])
    assert np.abs(gx[:: max(1, gx.size // 64)][:64] - ref).max() <= 2e-4 * np.abs(ref).max()
    md.eval()
    with torch.no_grad():
        np.testing.assert_allclose(md(off.detach()).cpu().numpy(), G["motion_disc.out"], rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,n,d,corr", [("small", 12, 64, 0.5), ("wide", 300, 32, 0.2)])
def test_gpu_contrastive_loss_gradients_match_reference(tag, n, d, corr):
    """SoftmaxContrastiveLoss with inputs that require a gradient: the loss and both feature gradients against the REFERENCE's autograd
    (softmax over 1/(distance + 1e-8), through the pairwise L2 distances and F.normalize)."""
    from emotiongestures_amd import harness as H
    dev = torch.device("cuda:0")
    f, a = feats(tag, n, d, 3, corr)
    ft, at = torch.from_numpy(f).to(dev).requires_grad_(True), torch.from_numpy(a).to(dev).requires_grad_(True)
    loss = H.SoftmaxContrastiveLoss()(ft, at, dev)
    assert abs(float(loss.detach()) - float(ADV[f"scl/{tag}/loss"])) <= 2e-5 * abs(float(ADV[f"scl/{tag}/loss"]))
    (loss * 1.0).backward()
    for nm, t in (("dface", ft.grad), ("daudio", at.grad)):
        g = t.reshape(-1).double().cpu().numpy()
        ref_norm = float(ADV[f"scl/{tag}/{nm}/norm"])
        assert abs(np.linalg.norm(g) - ref_norm) <= 2e-4 * ref_norm, (nm, np.linalg.norm(g), ref_norm)
        samp = g if n <= 16 else g[:: max(1, g.size // 256)][:256]
        ref = ADV[f"scl/{tag}/{nm}/sample"].astype(np.float64)
        assert np.abs(samp - ref).max() <= 5e-4 * np.abs(ref).max(), nm


# ---- Pose_Discriminator (Full_model/Models_spatial_memory.py:671-704) -------------------------------------------------------------------
PD = np.load(os.path.join(HERE, "golden", "pose_disc.npz"))
PD_CFG = dict(d_word_vec=282, d_model=282, d_inner=1024, n_layers=3, n_head=8, d_k=64, d_v=64, n_position=60)


def _pose_disc(precision="f32"):
    from emotiongestures_amd.Full_model.Models_spatial_memory import Pose_Discriminator
    from emotiongestures_amd.synth import load_synth_weights
    return load_synth_weights(Pose_Discriminator(**PD_CFG, precision=precision), 23).eval()


def _pd_poses():
    return ((hash_unit("pose_disc.x", 2 * 60 * 282, 23) * 2 - 1) * 0.5).astype(np.float32).reshape(2, 60, 282)


def test_pose_discriminator_schema_and_oracle_match_reference():
    """The mirror's state_dict is the reference class's (keys, shapes, order); the oracle restatement reproduces the reference's eval-mode
    probabilities; the upstream defaults (128-wide encoder under a 282-input head) are refused up front instead of failing inside forward."""
    from emotiongestures_amd import harness as H
    from oracle import emogest_oracle as O
    pd = _pose_disc()
    schema = json.load(open(os.path.join(HERE, "golden", "pose_disc_schema.json")))
    assert [[k, list(v.shape)] for k, v in pd.state_dict().items()] == schema
    sd = {k: v.detach().clone() for k, v in pd.state_dict().items()}
    cfg = O.GenCfg(d_model=282, d_inner=1024, n_layers=3, n_head=8, d_k=64, d_v=64)
    with torch.no_grad():
        out = O.pose_discriminator(sd, torch.from_numpy(_pd_poses()), cfg)
    np.testing.assert_allclose(out.numpy(), PD["eval/out"], rtol=2e-5, atol=2e-6)
    with pytest.raises(ValueError):
        H.Pose_Discriminator()


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("f32", 2e-5), ("bf16x3", 2e-4)])
def test_gpu_pose_discriminator_eval_matches_reference(precision, tol):
    dev = torch.device("cuda:0")
    pd = _pose_disc(precision).to(dev)
    with torch.no_grad():
        out = pd(torch.from_numpy(_pd_poses()).to(dev))
    assert out.shape == (2, 60, 1)
    np.testing.assert_allclose(out.cpu().numpy(), PD["eval/out"], rtol=tol, atol=tol)


@pytest.mark.gpu
def test_gpu_pose_discriminator_training_gradients_match_reference():
    """train() mode (dropout p = 0) on the differentiable HIP operators: probabilities, loss, every parameter gradient and the input gradient
    against the REFERENCE's autograd (tests/golden/make_golden_pose_disc.py)."""
    from emotiongestures_amd.train import functional as F
    dev = torch.device("cuda:0")
    pd = _pose_disc("f32").to(dev).train()
    x = torch.from_numpy(_pd_poses()).to(dev).requires_grad_(True)
    prob = pd(x)
    np.testing.assert_allclose(prob.detach().cpu().numpy(), PD["train/out"], rtol=2e-5, atol=2e-6)
    loss = F.smooth_l1_loss(prob, torch.ones_like(prob), 1.0, 1.0)
    loss.backward()
    assert abs(float(loss.detach()) - float(PD["train/loss"])) <= 1e-5 * abs(float(PD["train/loss"]))
    assert sorted(k for k, p in pd.named_parameters() if p.grad is None) == sorted(str(k) for k in PD["train/nograd"])
    for k, p in pd.named_parameters():
        if p.grad is None:
            continue
        gd = p.grad.detach().reshape(-1).double().cpu().numpy()
        ref_norm, ref = float(PD[f"train/g/{k}/norm"]), PD[f"train/g/{k}/sample"].astype(np.float64)
        err = max(abs(np.linalg.norm(gd) - ref_norm) / (ref_norm + 1e-30), np.abs(gd[:: max(1, gd.size // 64)][:64] - ref).max() / (np.abs(ref).max() + 1e-30))
        assert err < 2e-4, (k, err)
    gx = x.grad.reshape(-1).double().cpu().numpy()
    ref = PD["train/dx/sample"].astype(np.float64)
    assert abs(np.linalg.norm(gx) - float(PD["train/dx/norm"])) <= 2e-4 * float(PD["train/dx/norm"])
    assert np.abs(gx[:: max(1, gx.size // 64)][:64] - ref).max() <= 2e-4 * np.abs(ref).max()
