"""The EmotionNet K-fold loop end to end on the GPU (emotiongestures_amd/train/loops.py; train_audio_classifier_K_fold.py:109-200): samples come
from datapath.DataPreprocessor -> DictStore -> datapath.SpeechMotionDataset (the reference's record / item formats), a checkpoint written by
the loop reloads into a fresh model with bit-identical logits, and two gloo ranks sharing this box's GPU end a data-parallel run with equal
parameters."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None

# n_poses 62 at 15 fps = 4.13 s of audio -> a 128-frame spectrogram: EmotionNet's [B,128,128] input (model/audio_emotion_classifer.py:39-44)
N_POSES, STRIDE, FPS = 62, 20, 15


def build_dataset(n_clips=6, seed=0):
    from emotiongestures_amd import datapath as D
    from emotiongestures_amd.synth import synth_clip
    eids = [1, 66, 75, 82, 90, 100, 105, 115]               # one recording id per emotion class (lmdb_loader_BEAT_full.py:78-118)
    videos = []
    for i in range(n_clips):
        clip = synth_clip(seed=seed + i, duration=9.0)
        # the label is encoded in the spectrogram (a per-class band offset), so a short run has something to learn
        k = i % 8
        spec = clip["audio_feat"].astype(np.float32)
        spec[16 * k:16 * k + 16, :] = np.minimum(spec[16 * k:16 * k + 16, :] + 40.0, 0.0)
        clip["audio_feat"] = spec.astype(np.float16)
        videos.append({"eid": "1_spk_0_%d_%d" % (eids[k], eids[k]), "clips": [clip]})
    store = D.DictStore()
    D.DataPreprocessor(videos, store, N_POSES, STRIDE, FPS).run()
    ds = D.SpeechMotionDataset(store, N_POSES, STRIDE, FPS)
    assert ds.expected_spectrogram_length == 128 and len(ds) >= 12
    return ds


def test_k_fold_loop_trains_validates_saves_and_reloads(tmp_path):
    from emotiongestures_amd.model.audio_emotion_classifer import EmotionNet
    from emotiongestures_amd.train import loops
    ds = build_dataset()
    logs = []
    hist = loops.train_k_fold(ds, device=DEV, n_splits=2, total_epoch=2, batch_size=4, lr=1e-4, val_every=2, save_dir=str(tmp_path), test_dataset=ds,
                              seed=1, max_iters_per_fold=4, log=logs.append)
    assert [h["fold"] for h in hist] == [1, 2]
    for h in hist:
        assert h["iterations"] == 4 and len(h["loss"]) == 4 and all(np.isfinite(h["loss"]))
        assert [it for it, _ in h["val_acc"]] == [2, 4] and [it for it, _ in h["test_acc"]] == [2, 4]
        assert all(0.0 <= a <= 100.0 for _, a in h["val_acc"] + h["test_acc"])
        assert h["confusion"].sum() == (len(ds) // 4) * 4
        assert len(h["checkpoints"]) == 2 and all(os.path.exists(p) for p in h["checkpoints"])
        assert os.path.basename(h["checkpoints"][-1]).startswith(f"checkpoint_fold{h['fold']}_epoch")
    assert any("Val Accuracy" in l for l in logs) and any("Test Accuracy" in l for l in logs)
    # the last checkpoint of fold 2 was written right after iteration 4 = the model's final state: reload -> identical logits
    model = hist[-1]["model"].eval()
    fresh = loops.load_checkpoint(EmotionNet(precision="f32"), hist[-1]["checkpoints"][-1]).to(DEV).eval()
    spec, _ = loops._collate(ds, np.arange(4), 128)
    with torch.no_grad():
        a, b = model(spec.to(DEV)), fresh(spec.to(DEV))
    assert torch.equal(a, b)
    # and the folds differ (fresh model per fold, different training subsets)
    assert hist[0]["loss"] != hist[1]["loss"]


def test_evaluate_validation_pass_is_train_mode_batchnorm_and_confusion_is_predicted_by_true():
    """Upstream validates under no_grad with the model still in train() mode (train_audio_classifier_K_fold.py:177-190): BatchNorm uses batch
    statistics and keeps updating the running buffers that the checkpoint saved right after holds; `test_model` runs in eval() (:213) and
    fills `conf_matrix[predicted, true]` (:55-59)."""
    from emotiongestures_amd.model.audio_emotion_classifer import EmotionNet
    from emotiongestures_amd.synth import load_synth_weights
    from emotiongestures_amd.train import loops
    ds = build_dataset()
    model = load_synth_weights(EmotionNet(precision="f32"), 7).to(DEV).train()
    bufs = lambda: torch.cat([b.detach().reshape(-1).float() for n, b in model.named_buffers() if "running" in n]).cpu()
    before = bufs()
    idx = np.arange(8)
    ev = loops.evaluate(model, ds, idx, 4, DEV)                                  # eval mode: running statistics untouched, mode restored
    assert model.training and torch.equal(bufs(), before)
    labels = loops.labels_of(ds, idx)
    model.eval()
    with torch.no_grad():                                                        # the same eval-mode predictions, before validation moves the running statistics
        pred = torch.cat([model(loops._collate(ds, idx[b * 4:(b + 1) * 4], 128)[0].to(DEV)).argmax(1).cpu() for b in range(2)]).numpy()
    model.train()
    va = loops.evaluate(model, ds, idx, 4, DEV, train_mode_bn=True)             # upstream's validation pass
    assert model.training and not torch.equal(bufs(), before)
    assert all(p.grad is None for p in model.parameters())                       # no_grad: nothing recorded
    assert ev["batches"] == va["batches"] == 2
    # orientation: rows = predicted class, columns = true label
    want = np.zeros((8, 8), dtype=np.int64)
    for p_, t_ in zip(pred, labels):
        want[p_, t_] += 1
    np.testing.assert_array_equal(ev["confusion"], want)
    np.testing.assert_array_equal(ev["confusion"].sum(axis=0), np.bincount(labels, minlength=8))     # column sums = per-class sample counts


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_k_fold_loop_replayed_from_a_graph_equals_the_eager_loop(precision):
    """use_graph=True: the iteration replayed from one captured hipGraph per fold (the capture's warm-up step is the fold's first iteration, so no
    batch trains twice).  Same kernels in the same order: the losses of six iterations and the final parameters / BatchNorm statistics equal
    the eager loop's bitwise, the validation inside the loop (eval mode between replays) sees the updated weights."""
    from emotiongestures_amd.model.audio_emotion_classifer import EmotionNet
    from emotiongestures_amd.synth import load_synth_weights
    from emotiongestures_amd.train import loops
    ds = build_dataset()
    runs = []
    for use_graph in (False, True):
        factory = lambda: load_synth_weights(EmotionNet(precision="f32"), 7)
        hist = loops.train_k_fold(ds, device=DEV, n_splits=2, total_epoch=3, batch_size=4, lr=1e-4, val_every=3, seed=2, max_iters_per_fold=6, folds=[1],
                                  precision=precision, model_factory=factory, use_graph=use_graph, log=lambda s: None)
        h = hist[0]
        assert h["iterations"] == 6
        flat = torch.cat([p.detach().reshape(-1) for p in h["model"].parameters()] + [b.detach().reshape(-1).float() for b in h["model"].buffers()]).cpu()
        runs.append((h["loss"], h["val_acc"], flat))
    (l0, v0, p0), (l1, v1, p1) = runs
    assert l0 == l1, (l0, l1)
    assert v0 == v1
    assert torch.equal(p0, p1), float((p0 - p1).abs().max())


_WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, os.environ["EG_ROOT"]); sys.path.insert(0, os.path.join(os.environ["EG_ROOT"], "tests"))
    import numpy as np, torch, torch.distributed as dist
    from test_gpu_loops import build_dataset
    from emotiongestures_amd.train import loops
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from emotiongestures_amd.dist import init_process_group
    init_process_group("gloo", rank, world)          # FileStore published by the launcher (EG_DIST_STORE): no probed TCP port to lose
    print("rank", rank, "process group up", flush=True)
    dev = torch.device("cuda:0")
    ds = build_dataset()
    hist = loops.train_k_fold(ds, device=dev, n_splits=2, total_epoch=1, batch_size=2, lr=1e-4, val_every=100, seed=1, max_iters_per_fold=3, folds=[1],
                              log=lambda s: print("rank", rank, s, flush=True))
    flat = torch.cat([p.detach().reshape(-1) for p in hist[0]["model"].parameters()]).cpu()
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert torch.equal(other[0], other[1]), float((other[0] - other[1]).abs().max())       # same start (broadcast), same averaged gradients
    assert hist[0]["iterations"] == 3 and all(np.isfinite(hist[0]["loss"]))
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok", hist[0]["loss"])
''')


def _run_two_ranks(tmp_path):
    """-> (return codes or None for a rank killed at the deadline, logs)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    import time
    procs = []
    for r in range(2):
        # the ranks meet through a file (emotiongestures_amd/dist.py): round 5 probed a free port here and needed a retry when it was taken
        # before rank 0 bound it
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", EG_DIST_STORE=str(tmp_path / "store"), EG_ROOT=ROOT)
        env.pop("MASTER_PORT", None)
        log = open(tmp_path / f"rank{r}.log", "w+")      # files, not pipes: a rank that waits for a dead peer still leaves the peer's last words readable
        procs.append((subprocess.Popen([sys.executable, str(script)], env=env, stdout=log, stderr=subprocess.STDOUT, text=True), log))
    deadline = time.time() + 420
    while time.time() < deadline and any(p.poll() is None for p, _ in procs):
        if any(p.poll() not in (None, 0) for p, _ in procs):       # one rank died: do not sit out the other's collective timeout
            break
        time.sleep(0.5)
    codes = []
    for p, _ in procs:
        if p.poll() is None:
            p.kill()
            p.wait()
            codes.append(None)
        else:
            codes.append(p.returncode)
    outs = []
    for _, log in procs:
        log.seek(0)
        outs.append(log.read())
        log.close()
    return codes, outs


def test_k_fold_loop_two_gloo_ranks_share_this_gpu(tmp_path):
    codes, outs = _run_two_ranks(tmp_path)
    report = "\n".join(f"--- rank {r} (exit {c}):\n{o[-3000:]}" for r, (c, o) in enumerate(zip(codes, outs)))
    assert None not in codes, "rank(s) still running at the deadline\n" + report
    assert all(c == 0 for c in codes), report
    # the two ranks saw different batches: their per-iteration losses differ, their parameters (checked in the workers) do not
    assert outs[0].split("ok")[-1] != outs[1].split("ok")[-1]
