#!/usr/bin/env python3
"""Generate tests/golden/training_types.npz from the reference's training-side types, forward only.

Build container only:   python tests/golden/make_golden_training_types.py

Imported in place from /root/reference: test_emotion_gesture_diversity_iterative.{SoftmaxContrastiveLoss, adjust_lr, calc_motion,
compute_acc, l2_distance_pose} and Full_model.Models_memory.Motion_Discriminator.  The eval script imports apex, librosa, lmdb,
pickle5, matplotlib, soundfile and fasttext at module scope without using them in these functions; they get empty stand-ins.
Motion_Discriminator's upstream defaults (d_model 128 on 282-d poses) do not run, so it is built with
pose_dim = d_word_vec = d_model = 128, the one consistent choice its forward admits.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from emotiongestures_amd.synth import hash_unit, load_synth_weights  # noqa: E402


def _stubs():
    names = ("lmdb", "librosa", "librosa.display", "soundfile", "pickle5", "fasttext", "matplotlib", "matplotlib.pyplot", "matplotlib.ticker",
             "matplotlib.animation", "mpl_toolkits", "mpl_toolkits.mplot3d", "torch_dct", "umap", "apex", "apex.amp", "torchvision",
             "torchvision.utils", "torchvision.transforms")
    for name in names:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["librosa"].display = sys.modules["librosa.display"]
    m = sys.modules["matplotlib"]
    m.use = lambda *a, **k: None
    m.pyplot, m.ticker, m.animation = (sys.modules["matplotlib." + n] for n in ("pyplot", "ticker", "animation"))
    sys.modules["mpl_toolkits"].mplot3d = sys.modules["mpl_toolkits.mplot3d"]
    sys.modules["apex"].amp = sys.modules["apex.amp"]
    sys.modules["matplotlib.pyplot"].figure = None        # model/Beat_score_v2.py:7 imports the name, the functions used here never call it
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision.utils"].save_image = None
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]


def feats(tag, n, d, seed, corr):
    """Two [n, d] feature sets; `corr` blends the audio side towards the face side so that the argmax accuracy is non-trivial."""
    f = (hash_unit(tag + ".face", n * d, seed) * 2 - 1).astype(np.float32).reshape(n, d)
    a = (hash_unit(tag + ".audio", n * d, seed) * 2 - 1).astype(np.float32).reshape(n, d)
    return f, (corr * f + (1 - corr) * a).astype(np.float32)


def main():
    _stubs()
    sys.path.insert(0, REF)
    os.chdir(REF)
    import test_emotion_gesture_diversity_iterative as E
    from Full_model.Models_memory import Motion_Discriminator

    out = {}
    crit = E.SoftmaxContrastiveLoss()
    for tag, n, d, corr in (("small", 12, 64, 0.5), ("wide", 300, 32, 0.2), ("single", 1, 16, 0.0), ("identical", 6, 8, 1.0)):
        f, a = feats(tag, n, d, 3, corr)
        with torch.no_grad():
            loss = crit(torch.from_numpy(f), torch.from_numpy(a), "cpu")
            acc, cross = crit.evaluate(torch.from_numpy(f), torch.from_numpy(a))
        out[f"scl.{tag}.meta"] = np.array([n, d, corr], np.float64)
        out[f"scl.{tag}.loss"] = loss.numpy()
        out[f"scl.{tag}.acc"] = acc.numpy()
        c = cross.numpy()
        out[f"scl.{tag}.cross"] = c if n <= 16 else c[:4]          # full matrix for the small cases, 4 rows + diagonal otherwise
        out[f"scl.{tag}.diag"] = np.diagonal(c).copy()
        print(tag, float(loss), float(acc))

    class Opt:
        def __init__(self):
            self.param_groups = [{"lr": -1.0}, {"lr": -1.0}]
    lrs = []
    for epoch in range(0, 151):
        o = Opt()
        E.adjust_lr(o, 1e-3, epoch)
        assert o.param_groups[0]["lr"] == o.param_groups[1]["lr"]
        lrs.append(o.param_groups[0]["lr"])
    out["adjust_lr.table"] = np.array(lrs, np.float64)

    motion = (hash_unit("motion", 3 * 60 * 128, 5) * 2 - 1).astype(np.float32).reshape(3, 60, 128)
    off = E.calc_motion(torch.from_numpy(motion))
    out["calc_motion.out"] = off.numpy()
    md = Motion_Discriminator(frames=59, pose_dim=128, d_word_vec=128, d_model=128, d_inner=1024, n_layers=2, n_head=8, d_k=64, d_v=64,
                              n_position=59).eval()
    load_synth_weights(md, 21)
    with torch.no_grad():
        out["motion_disc.out"] = md(off).numpy()
    import json
    json.dump([[k, list(v.shape)] for k, v in md.state_dict().items()],
              open(os.path.join(ROOT, "tests", "golden", "motion_disc_schema.json"), "w"))
    logits = (hash_unit("logits", 16 * 8, 1) * 2 - 1).astype(np.float32).reshape(16, 8)
    labels = (hash_unit("labels", 16, 1) * 8).astype(np.int64)
    out["compute_acc.out"] = E.compute_acc(torch.from_numpy(labels), torch.from_numpy(logits)).numpy()
    out["l2_distance_pose.out"] = np.float64(E.l2_distance_pose(motion[:, :, :64], motion[:, :, 64:]))
    path = os.path.join(ROOT, "tests", "golden", "training_types.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "motion_disc", out["motion_disc.out"].ravel())


if __name__ == "__main__":
    main()
