"""The reference's one training loop as a product piece: K-fold training of EmotionNet, the audio emotion classifier
(train_audio_classifier_K_fold.py:109-200; model/audio_emotion_classifer.py:17-49), on the HIP training path.

What upstream's `train_K_fold` does, and where it lives here:

* `KFold(n_splits=10, shuffle=True)` over the training set (:301, sklearn)                   -> `kfold_indices(n, k, shuffle=True, seed)`
* per fold a fresh EmotionNet, Adam(lr, betas, weight_decay=1e-5) (:128-132)                -> `FlatAdam` over `flatten_parameters`
* `SubsetRandomSampler` loaders, batch_size, drop_last (:136-145)                            -> `epoch_batches` (seeded permutation, whole batches)
* per epoch: class counts over the fold's training samples -> class weights
  `sum / (8 * count)` -> FocalLoss(alpha, gamma=2) (:146-153)                                 -> `class_weights`, `functional.focal_loss`
* per iteration: zero_grad -> model(in_spec) -> 100 x focal loss -> backward -> step (:155-175)
* every 100 iterations: validation accuracy over the fold's validation loader (compute_acc,
  no_grad), `torch.save(state_dict)` as checkpoint_fold{}_epoch{}_iteraction{}.pth [sic],
  then accuracy + confusion matrix on the test set in eval() mode (:177-200, :205-255)       -> `evaluate`, `save_checkpoint`

Data arrives through the reference's own item format: `datapath.SpeechMotionDataset.__getitem__` 5-tuples collated by
`datapath.audio_classifier_collate_fn` (data_loader/lmdb_loader_BEAT_full.py:63-75,171-253), on any sample store.

Two places where upstream cannot run as written, and what is done instead (both documented, neither silently):
  1. the class count loop `for _, label in train_loader: class_count[label] += 1` (:147-148) unpacks 2 values from the loader's
     5-tuples and indexes a list with a one-hot tensor; the evident intent -- count the fold's training labels -- is what runs here;
  2. `FocalLoss(alpha=[8 class weights])` multiplies the per-SAMPLE loss vector by the 8-vector (:97), which broadcasts only at
     batch size 8 and then weights by batch POSITION, not by label.  `alpha_mode="positional"` reproduces exactly that (batch size 8
     only); the default `"by_label"` weights every sample with its class's weight (the re-weighting the comment at :111 describes).

Data parallel: one process per GPU; every rank walks the same seeded permutation and takes the batches
`rank, rank + world, ...`, gradients are averaged with the bucketed all-reduce of `optim.GradBuckets` (RCCL over xGMI; gloo in the tests),
BatchNorm statistics stay per replica as with upstream's nn.DataParallel.  No CPU fallback: the model trains on the HIP operators only.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

NUM_CLASSES = 8
EMOTIONS = ("neutral", "happiness", "anger", "sadness", "contempt", "surprise", "fear", "disgust")     # train_audio_classifier_K_fold.py:65


# ---- host logic (no GPU needed: tests/test_loops.py) ------------------------------------------------------------------------
def kfold_indices(n: int, n_splits: int, shuffle: bool = True, seed: Optional[int] = 0) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
    """sklearn.model_selection.KFold(n_splits, shuffle=shuffle, random_state=seed).split (:301 calls `KFold(n_splits=10, shuffle=True)`).
    sklearn's algorithm: `indices = arange(n)`, shuffled in place by `RandomState(seed).shuffle` when `shuffle`; fold f's validation set is
    the f-th contiguous block of that (shuffled) array (the first n % k blocks one longer); both index sets are returned in ascending order
    (sklearn builds them from a boolean mask).  Upstream passes no random_state, i.e. numpy's global generator: `seed=None` does that.  The
    dataset is ordered by speaker / recording, so the shuffle decides the fold composition and the per-fold class weights."""
    if n_splits < 2 or n_splits > n:
        raise ValueError(f"kfold_indices: n_splits={n_splits} for {n} samples")
    sizes = np.full(n_splits, n // n_splits, dtype=np.int64)
    sizes[: n % n_splits] += 1
    order = np.arange(n)
    if shuffle:
        (np.random.RandomState(seed) if seed is not None else np.random.mtrand._rand).shuffle(order)
    start = 0
    for s in sizes:
        mask = np.zeros(n, dtype=bool)
        mask[order[start:start + s]] = True
        yield np.flatnonzero(~mask), np.flatnonzero(mask)
        start += s


def class_weights(labels: Sequence[int], num_classes: int = NUM_CLASSES) -> np.ndarray:
    """sum(count) / (num_classes * count) per class (:149).  A class absent from the fold would divide by zero upstream; it gets weight 0
    here (it never multiplies a loss term: no sample carries it)."""
    count = np.bincount(np.asarray(labels, dtype=np.int64), minlength=num_classes).astype(np.float64)
    w = np.zeros(num_classes)
    nz = count > 0
    w[nz] = count.sum() / (num_classes * count[nz])
    return w


def epoch_batches(indices: np.ndarray, batch_size: int, seed: int, rank: int = 0, world: int = 1) -> List[np.ndarray]:
    """SubsetRandomSampler + drop_last (:136-141): a seeded permutation of `indices` cut into whole batches; rank r of `world` takes
    batches r, r + world, ... (every rank the same number: the tail that does not fill a round is dropped, so the collectives line up)."""
    g = torch.Generator().manual_seed(int(seed))
    perm = np.asarray(indices)[torch.randperm(len(indices), generator=g).numpy()]
    nb = len(perm) // batch_size
    nb -= nb % world
    return [perm[b * batch_size:(b + 1) * batch_size] for b in range(rank, nb, world)]


def checkpoint_name(save_dir: str, fold: int, epoch: int, iteration: int) -> str:
    return os.path.join(save_dir, "checkpoint_fold{}_epoch{}_iteraction{}.pth".format(fold, epoch, iteration))      # upstream's spelling (:196)


def labels_of(dataset, indices: Sequence[int]) -> np.ndarray:
    """Integer emotion label of every listed sample (argmax of the one-hot the dataset returns, :166)."""
    return np.asarray([int(torch.argmax(dataset[int(i)][3])) for i in indices], dtype=np.int64)


def _collate(dataset, idx: np.ndarray, spec_len: Optional[int]):
    from ..datapath import audio_classifier_collate_fn
    _audio, spec, _pose, label, _aux = audio_classifier_collate_fn([dataset[int(i)] for i in idx])
    if spec_len is not None:
        spec = spec[:, :, :spec_len]
    return spec.contiguous(), torch.argmax(label, 1)


# ---- evaluation (inference kernels) -------------------------------------------------------------------------------------------
def evaluate(model, dataset, indices: Sequence[int], batch_size: int, device, spec_len: Optional[int] = 128,
             train_mode_bn: bool = False) -> Dict[str, object]:
    """Accuracy as upstream averages it -- the mean of per-batch compute_acc over whole batches (:181-190, :212-232) -- and test_model's
    confusion matrix, filled as upstream's `confusion_matrix` fills it: `conf[predicted, true] += 1` (:55-59).
    train_mode_bn=False: eval() mode, running-statistics BatchNorm on the inference engine -- upstream's `test_model` (:213).
    train_mode_bn=True: upstream's *validation* pass (:177-190) runs under `torch.no_grad()` with the model still in train() mode, so BatchNorm
    normalises with batch statistics and keeps updating its running buffers, and the checkpoint saved right after (:197) holds those buffers."""
    from ..harness import compute_acc
    was_training = model.training
    model.train() if train_mode_bn else model.eval()
    conf = np.zeros((NUM_CLASSES, NUM_CLASSES), dtype=np.int64)
    accs = []
    idx = np.asarray(indices)
    with torch.no_grad():
        for b in range(len(idx) // batch_size):
            spec, label = _collate(dataset, idx[b * batch_size:(b + 1) * batch_size], spec_len)
            out = model(spec.to(device))
            accs.append(float(compute_acc(label.to(device), out)))
            for t, p in zip(label.tolist(), out.argmax(1).cpu().tolist()):
                conf[p, t] += 1
    model.train() if was_training else model.eval()
    return {"accuracy": float(np.mean(accs)) if accs else float("nan"), "batches": len(accs), "confusion": conf}


def save_checkpoint(model, path: str) -> None:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)        # torch.save(model.state_dict(), ...) (:197)


def load_checkpoint(model, path: str):
    sd = torch.load(path, map_location="cpu")
    model.load_state_dict({k.replace("module.", ""): v for k, v in sd.items()})           # the loaders strip DataParallel's prefix (:121)
    return model


# ---- the loop ------------------------------------------------------------------------------------------------------------------
def train_k_fold(train_dataset, *, device, n_splits: int = 10, total_epoch: int = 100, batch_size: int = 8, lr: float = 3e-4,
                 betas=(0.5, 0.999), weight_decay: float = 1e-5, gamma: float = 2.0, val_every: int = 100, save_dir: Optional[str] = None,
                 test_dataset=None, seed: int = 0, precision: str = "f32", alpha_mode: str = "by_label", spec_len: Optional[int] = 128,
                 max_iters_per_fold: Optional[int] = None, folds: Optional[Sequence[int]] = None, model_factory: Optional[Callable] = None,
                 use_graph: bool = False, shuffle_folds: bool = True, keep_models: str = "last", log: Callable[[str], None] = print) -> List[Dict[str, object]]:
    """train_K_fold (train_audio_classifier_K_fold.py:109-200).  Returns one record per fold: losses, validation / test accuracies,
    checkpoint paths.  `precision`: "f32" (gradient-parity arithmetic) or "bf16x3" (split-bf16 MFMA) for the convolutions / Linear products.
    Under torch.distributed (initialised by the caller) the loop is data parallel as described in the module docstring.
    use_graph: replay the iteration (zero_grad, forward, loss, backward, gradient collection and -- on one rank -- Adam) from one captured
    hipGraph per fold instead of issuing its ~600 launches through autograd (a batch of 8 is host-bound otherwise); with several ranks the
    bucket all-reduces and Adam follow each replay.  Same kernels, same order: the losses equal the eager loop's.
    shuffle_folds / seed: the fold split is sklearn's KFold(n_splits, shuffle=True, random_state=seed) (:301).
    keep_models: "last" keeps only the final fold's model on the device in its record (`rec["model"]`; None in the other records), "none" none,
    "all" every fold's;
    every record carries the fold's final `state_dict` on the CPU (`rec["state_dict"]`).  Validation (:177-190) runs as upstream does, with
    train()-mode BatchNorm under no_grad; the test pass (:205-255) in eval() mode."""
    import torch.distributed as dist
    from ..model.audio_emotion_classifer import EmotionNet
    from . import functional as F
    from .optim import FlatAdam, GradBuckets, flatten_parameters
    if alpha_mode not in ("by_label", "positional"):
        raise ValueError("alpha_mode: 'by_label' or 'positional'")
    if keep_models not in ("last", "none", "all"):
        raise ValueError(f"keep_models: 'last', 'none' or 'all' (got {keep_models!r})")
    if alpha_mode == "positional" and batch_size != NUM_CLASSES:
        raise ValueError("alpha_mode='positional' is upstream's literal broadcast of 8 class weights over the batch axis: batch_size must be 8")
    device = torch.device(device)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = len(train_dataset)
    all_labels = labels_of(train_dataset, range(n))
    history = []
    F.set_precision(precision)
    try:
        for fold, (train_index, val_index) in enumerate(kfold_indices(n, n_splits, shuffle=shuffle_folds, seed=seed), start=1):
            if folds is not None and fold not in folds:
                continue
            model = (model_factory() if model_factory is not None else EmotionNet(precision=precision)).to(device)
            if world > 1:                       # every replica starts from rank 0's initialisation, as DataParallel's replicas do
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t.data, src=0)
            model.train()
            fp = flatten_parameters(model)
            if precision != "f32":
                from . import nets
                fp.enable_weight_images(*nets.weight_image_plan(model))
            opt = FlatAdam(fp, lr=lr, betas=betas, weight_decay=weight_decay)
            gb = GradBuckets(fp).attach() if world > 1 else None     # plain buckets: this loop reduces behind the whole backward (reduce_deferred / hooks), no SegmentedStep
            try:
                rec = {"fold": fold, "loss": [], "val_acc": [], "test_acc": [], "checkpoints": [], "iterations": 0}
                global_iter = 0
                done = False
                graphed, static = None, None            # use_graph: built at the first full batch of the fold

                def eager_iteration(spec_d, label_d, alpha_d):
                    opt.zero_grad()
                    if gb is not None:
                        gb.begin()
                    loss = F.focal_loss(model(spec_d), label_d, alpha_d, gamma, 100.0)          # criterion(output, label) * 100 (:168)
                    loss.backward()
                    if gb is not None:
                        gb.finish()
                    opt.step(collected=gb is not None)
                    return loss

                for epoch in range(total_epoch):
                    w = class_weights(all_labels[train_index])                       # :146-150 (recomputed every epoch, as upstream)
                    for idx in epoch_batches(train_index, batch_size, seed * 100003 + fold * 1009 + epoch, rank, world):
                        spec, label = _collate(train_dataset, idx, spec_len)
                        alpha = torch.tensor(w if alpha_mode == "positional" else w[label.numpy()], dtype=torch.float32)
                        model.train()
                        global_iter += 1
                        batch = {"spec": spec.to(device), "label": label.to(device), "alpha": alpha.to(device)}
                        if use_graph and len(idx) == batch_size:
                            if graphed is None:
                                from .graph import GraphedStep
                                static = {k: v.clone() for k, v in batch.items()}
                                if gb is None:
                                    graphed = GraphedStep(lambda _i: eager_iteration(static["spec"], static["label"], static["alpha"]), static, opt, warmup=1)
                                else:       # data parallel: the graph holds forward + backward + collection; collectives and Adam follow the replay
                                    gb.deferred = True

                                    def fwd_bwd(_i):
                                        opt.zero_grad()
                                        gb.begin()
                                        ls = F.focal_loss(model(static["spec"]), static["label"], static["alpha"], gamma, 100.0)
                                        ls.backward()
                                        gb.finish()
                                        return ls
                                    graphed = GraphedStep(fwd_bwd, static, None, warmup=1, device=device)
                                loss = graphed.warmup_loss          # the capture's one warm-up step WAS this batch's iteration (executed eagerly)
                            else:
                                loss = graphed.run(batch)
                            if gb is not None:
                                gb.reduce_deferred()
                                opt.step(collected=True)
                        else:
                            if gb is not None:
                                gb.deferred = False
                            loss = eager_iteration(batch["spec"], batch["label"], batch["alpha"])
                        rec["loss"].append(float(loss.detach()))
                        if global_iter % val_every == 0:                             # :177
                            va = evaluate(model, train_dataset, val_index, batch_size, device, spec_len, train_mode_bn=True)
                            rec["val_acc"].append((global_iter, va["accuracy"]))
                            log("Fold {}, Epoch {}, Val Accuracy: {:.2f}%".format(fold, epoch, va["accuracy"]))
                            if save_dir is not None and rank == 0:
                                path = checkpoint_name(save_dir, fold, epoch, global_iter)
                                save_checkpoint(model, path)
                                rec["checkpoints"].append(path)
                            if test_dataset is not None:
                                ta = evaluate(model, test_dataset, range(len(test_dataset)), batch_size, device, spec_len)
                                rec["test_acc"].append((global_iter, ta["accuracy"]))
                                rec["confusion"] = ta["confusion"]
                                log("Fold {}, Epoch {}/{}, Iteraction {}, Test Accuracy: {:.2f}%".format(fold, epoch, total_epoch, global_iter, ta["accuracy"]))
                        if max_iters_per_fold is not None and global_iter >= max_iters_per_fold:
                            done = True
                            break
                    if done:
                        break
                rec["iterations"] = global_iter
                rec["state_dict"] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
                keep = keep_models == "all" or (keep_models == "last" and (folds is None and fold == n_splits or folds is not None and fold == max(folds)))
                rec["model"] = model if keep else None        # the key always exists; None = not kept on the device (rec["state_dict"] holds the weights)
                history.append(rec)
            finally:                            # also on an exception mid-fold: no stale weight images in the registry, no fold's graph / scratch kept alive
                if fp.images is not None:
                    F.unregister_weight_images(fp.images)
                if graphed is not None:
                    graphed.close()             # the fold's graph and the scratch buffers registered under its stream
                graphed = static = opt = gb = None
    finally:
        F.set_precision("f32")
    return history
