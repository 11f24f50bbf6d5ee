"""The step before the generator: whole-clip mel -> per-sample slices -> stored records -> dataset items.

Host-side mirror of the reference's sample pipeline.  The arithmetic that matters for parity is integer index arithmetic
(slice starts, symmetric padding, clipping) plus the fp16 storage of the dB spectrogram, so everything here is bit-exact
against the reference except the mel front-end itself (librosa in the reference: parity unpinned, see oracle header).

  reference                                                          here
  -----------------------------------------------------------------  ---------------------------------------------
  utils/train_utils_BEAT.py:186-190  extract_melspectrogram           extract_melspectrogram (HIP mel kernels, device)
  utils/train_utils_BEAT.py:193-195  calc_spectrogram_length_...      calc_spectrogram_length_from_motion_length
  utils/train_utils_BEAT.py:198-208  resample_pose_seq                resample_pose_seq
  utils/train_utils_BEAT.py:220-226  make_audio_fixed_length          make_audio_fixed_length
  data_loader/data_preprocessor_expressive.py:17-68   DataPreprocessor          DataPreprocessor (any SampleStore, not only LMDB)
  data_loader/data_preprocessor_expressive.py:70-171  _sample_from_clip         DataPreprocessor._sample_from_clip
  data_loader/data_preprocessor_expressive.py:178-193 get_words_in_time_range   get_words_in_time_range
  data_loader/motion_preprocessor_expressive.py:4-31  MotionPreprocessor        MotionPreprocessor
  data_loader/lmdb_loader_BEAT_full.py:78-118         one_hot_eid               one_hot_eid
  data_loader/lmdb_loader_BEAT_full.py:120-253        SpeechMotionDataset       SpeechMotionDataset
  data_loader/lmdb_loader_BEAT_full.py:63-75          audio_classifier_collate_fn  audio_classifier_collate_fn

On-disk format.  The reference stores `pyarrow.serialize([words, poses, audio, spectrogram, aux])` under the ASCII key
'{:010}' in LMDB.  Neither `lmdb` nor `pyarrow.serialize` (removed upstream in pyarrow 2.0+) exists in this image, so the
store is an interface: `DictStore` (in memory), `DirStore` (one file per key) and, when `lmdb` is importable, `LmdbStore`.
Records are encoded by `encode_record` (npz container: arrays verbatim incl. the fp16 spectrogram, words/aux as JSON);
`decode_record` also accepts legacy pyarrow payloads when `pyarrow.deserialize` exists.
"""
from __future__ import annotations

import io
import json
import math
import os
from collections import defaultdict
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

SAMPLE_RATE = 16000
N_FFT = 1024
HOP = 512


# ---------------------------------------------------------------------------------------------------------------- utils

def calc_spectrogram_length_from_motion_length(n_frames: int, fps: float) -> int:
    """utils/train_utils_BEAT.py:193-195 (python round: banker's rounding, as upstream)."""
    ret = (n_frames / fps * SAMPLE_RATE - N_FFT) / HOP + 1
    return int(round(ret))


def make_audio_fixed_length(audio: np.ndarray, expected_audio_length: int) -> np.ndarray:
    """utils/train_utils_BEAT.py:220-226: symmetric right padding, or truncation."""
    n_padding = expected_audio_length - len(audio)
    if n_padding > 0:
        return np.pad(audio, (0, n_padding), mode="symmetric")
    return audio[0:expected_audio_length]


def resample_pose_seq(poses: np.ndarray, duration_in_sec: float, fps: float) -> np.ndarray:
    """utils/train_utils_BEAT.py:198-208: linear interpolation (with linear extrapolation past the last frame) of a
    [n, ...] pose sequence onto `np.arange(0, n, n / (duration * fps))`."""
    poses = np.asarray(poses)
    n = len(poses)
    expected_n = duration_in_sec * fps
    x_new = np.arange(0, n, n / expected_n)
    # scipy interp1d(kind='linear', fill_value='extrapolate'): segment index clipped to [0, n-2], same line extended
    lo = np.clip(np.floor(x_new).astype(np.int64), 0, max(n - 2, 0))
    frac = (x_new - lo).reshape((-1,) + (1,) * (poses.ndim - 1))
    y = poses.astype(np.float64, copy=False)
    out = y[lo] + (y[np.minimum(lo + 1, n - 1)] - y[lo]) * frac
    return out.astype(poses.dtype) if hasattr(poses, "dtype") else out


def get_words_in_time_range(word_list: Sequence[Sequence], start_time: float, end_time: float) -> List[Sequence]:
    """data_loader/data_preprocessor_expressive.py:178-193 (word = [text, start, end], list sorted by start)."""
    words = []
    for word in word_list:
        word_s, word_e = word[1], word[2]
        if word_s >= end_time:
            break
        if word_e <= start_time:
            continue
        words.append(word)
    return words


_EID_BOUNDS = (64, 72, 80, 86, 94, 102, 110, 118)        # data_loader/lmdb_loader_BEAT_full.py:78-118


def one_hot_eid(eid: str) -> np.ndarray:
    """BEAT recording id -> 8-way emotion one-hot (float64 as upstream); ids above 118 give the all-zero vector upstream
    (its `assert 'label one_hot error!'` never fires), reproduced here."""
    index = int(eid.split("_", 4)[-1])
    label = np.zeros(8, dtype=float)
    for k, hi in enumerate(_EID_BOUNDS):
        if index <= hi:
            label[k] = 1
            break
    return label


class MotionPreprocessor:
    """data_loader/motion_preprocessor_expressive.py:4-31: all upstream filters are commented out; what remains is the
    NaN assertion and the conversion to nested lists."""

    def __init__(self, skeletons):
        self.skeletons = np.array(skeletons)
        self.filtering_message = "PASS"

    def get(self):
        if self.skeletons.size:
            if np.isnan(self.skeletons).any():
                raise AssertionError("missing joints (NaN) in a sample")
            return self.skeletons.tolist(), self.filtering_message
        return [], self.filtering_message


# ------------------------------------------------------------------------------------------------------------ mel (device)

_MEL = {}


def extract_melspectrogram(y, sr: int = SAMPLE_RATE, device="cuda:0") -> np.ndarray:
    """utils/train_utils_BEAT.py:186-190 on the GPU: whole-clip log-mel [128, 1 + n // 512], dB relative to the clip maximum,
    clamped at -80 dB, stored as float16.  `y` is a 1-D float array (host) or tensor; there is no CPU fallback."""
    import torch

    from .engine import MelFrontEnd
    if sr != SAMPLE_RATE:
        raise ValueError(f"extract_melspectrogram: the path is defined at 16 kHz (got sr={sr})")
    dev = torch.device(device)
    fe = _MEL.get(str(dev))
    if fe is None:
        fe = _MEL[str(dev)] = MelFrontEnd(dev)
    t = torch.as_tensor(np.asarray(y, dtype=np.float32) if not torch.is_tensor(y) else y, dtype=torch.float32).reshape(1, -1).to(dev)
    spec = fe(t)[0]
    return spec.cpu().numpy().astype(np.float16)        # values are already fp16-representable (rounded in the kernel)


# ---------------------------------------------------------------------------------------------------------------- stores

def sample_key(index: int) -> bytes:
    """data_loader/data_preprocessor_expressive.py:160, lmdb_loader_BEAT_full.py:173."""
    return "{:010}".format(index).encode("ascii")


def encode_record(words, poses, audio, spectrogram, aux) -> bytes:
    buf = io.BytesIO()
    meta = json.dumps({"words": [[w[0], float(w[1]), float(w[2])] for w in words], "aux": aux})
    np.savez(buf, poses=np.asarray(poses), audio=np.asarray(audio), spectrogram=np.asarray(spectrogram),
             meta=np.frombuffer(meta.encode("utf-8"), dtype=np.uint8))
    return buf.getvalue()


def decode_record(blob: bytes):
    if blob[:2] == b"PK":                                # npz container written by encode_record
        z = np.load(io.BytesIO(blob), allow_pickle=False)
        meta = json.loads(bytes(z["meta"]).decode("utf-8"))
        return [meta["words"], z["poses"], z["audio"], z["spectrogram"], meta["aux"]]
    try:                                                 # legacy upstream payload
        import pyarrow
        return list(pyarrow.deserialize(blob))
    except (ImportError, AttributeError) as e:
        raise ValueError("record is not in the npz format and pyarrow.deserialize is unavailable") from e


class DictStore:
    """In-memory SampleStore: `put(key, blob)`, `get(key)`, `__len__`, `keys()`."""

    def __init__(self):
        self._d: Dict[bytes, bytes] = {}

    def put(self, key: bytes, blob: bytes) -> None:
        self._d[bytes(key)] = bytes(blob)

    def get(self, key: bytes) -> Optional[bytes]:
        return self._d.get(bytes(key))

    def keys(self) -> Iterator[bytes]:
        return iter(sorted(self._d))

    def __len__(self) -> int:
        return len(self._d)


class DirStore(DictStore):
    """One file per key under `path` (the on-disk stand-in where LMDB is unavailable)."""

    def __init__(self, path: str):
        self.path = path
        os.makedirs(path, exist_ok=True)

    def put(self, key, blob):
        tmp = os.path.join(self.path, key.decode("ascii") + ".tmp")
        with open(tmp, "wb") as f:
            f.write(blob)
        os.replace(tmp, os.path.join(self.path, key.decode("ascii") + ".rec"))

    def get(self, key):
        p = os.path.join(self.path, key.decode("ascii") + ".rec")
        if not os.path.exists(p):
            return None
        with open(p, "rb") as f:
            return f.read()

    def keys(self):
        return iter(sorted(n[:-4].encode("ascii") for n in os.listdir(self.path) if n.endswith(".rec")))

    def __len__(self):
        return sum(1 for n in os.listdir(self.path) if n.endswith(".rec"))


class LmdbStore(DictStore):
    """The reference's container (lmdb.open(dir) / txn.put / txn.get); only usable where `lmdb` is installed."""

    def __init__(self, path: str, readonly: bool = False, map_size: int = (1024 * 100) << 20):
        try:
            import lmdb
        except ImportError as e:
            raise RuntimeError("LmdbStore needs the `lmdb` module, which this environment does not have; use DirStore") from e
        self._env = lmdb.open(path, readonly=readonly, lock=not readonly, map_size=map_size)

    def put(self, key, blob):
        with self._env.begin(write=True) as txn:
            txn.put(key, blob)

    def get(self, key):
        with self._env.begin(write=False) as txn:
            return txn.get(key)

    def keys(self):
        with self._env.begin(write=False) as txn:
            return iter([k for k, _ in txn.cursor()])

    def __len__(self):
        with self._env.begin() as txn:
            return txn.stat()["entries"]


# ---------------------------------------------------------------------------------------------------------- preprocessor

class DataPreprocessor:
    """data_loader/data_preprocessor_expressive.py:17-171.  `clips` is an iterable of video dicts
    {'eid': str, 'clips': [ {'skeletons','audio_feat','audio_raw','words','start_frame_no','end_frame_no','start_time','end_time'} ]}
    (what the upstream source LMDB holds); samples are written to `out_store` under sample_key(n)."""

    def __init__(self, videos: Iterable[dict], out_store, n_poses: int, subdivision_stride: int, pose_resampling_fps: float,
                 disable_filtering: bool = False):
        self.videos = videos
        self.out_store = out_store
        self.n_poses = n_poses
        self.subdivision_stride = subdivision_stride
        self.skeleton_resampling_fps = pose_resampling_fps
        self.disable_filtering = disable_filtering
        self.spectrogram_sample_length = calc_spectrogram_length_from_motion_length(n_poses, pose_resampling_fps)
        self.audio_sample_length = int(n_poses / pose_resampling_fps * SAMPLE_RATE)
        self.n_out_samples = 0

    def run(self) -> Dict[str, int]:
        n_filtered_out: Dict[str, int] = defaultdict(int)
        for video in self.videos:
            for clip in video["clips"]:
                for kind, n in self._sample_from_clip(video["eid"], clip).items():
                    n_filtered_out[kind] += n
        return dict(n_filtered_out)

    def _sample_from_clip(self, eid: str, clip: dict) -> Dict[str, int]:
        clip_audio = np.asarray(clip["audio_feat"])
        clip_audio_raw = np.asarray(clip["audio_raw"])
        clip_s_f = clip["start_frame_no"]
        clip_s_t, clip_e_t = clip["start_time"], clip["end_time"]
        n_filtered_out: Dict[str, int] = defaultdict(int)

        clip_skeleton = resample_pose_seq(clip["skeletons"], clip_e_t - clip_s_t, self.skeleton_resampling_fps)
        n_skel = len(clip_skeleton)
        num_subdivision = math.floor((n_skel - self.n_poses) / self.subdivision_stride) + 1
        expected_audio_length = calc_spectrogram_length_from_motion_length(n_skel, self.skeleton_resampling_fps)
        if abs(expected_audio_length - clip_audio.shape[1]) > 5:
            raise AssertionError("audio and skeleton lengths are different")

        out = []
        for i in range(num_subdivision):
            start_idx = i * self.subdivision_stride
            fin_idx = start_idx + self.n_poses
            sample_skeletons = clip_skeleton[start_idx:fin_idx]
            t0 = clip_s_t + start_idx / self.skeleton_resampling_fps
            t1 = clip_s_t + fin_idx / self.skeleton_resampling_fps
            sample_words = get_words_in_time_range(clip["words"], t0, t1)

            a0 = math.floor(start_idx / n_skel * clip_audio.shape[1])           # spectrogram columns
            a1 = a0 + self.spectrogram_sample_length
            if a1 > clip_audio.shape[1]:
                padded = np.pad(clip_audio, ((0, 0), (0, a1 - clip_audio.shape[1])), mode="symmetric")
                sample_spectrogram = padded[:, a0:a1]
            else:
                sample_spectrogram = clip_audio[:, a0:a1]

            r0 = math.floor(start_idx / n_skel * len(clip_audio_raw))           # raw samples
            r1 = r0 + self.audio_sample_length
            if r1 > len(clip_audio_raw):
                sample_audio = np.pad(clip_audio_raw, (0, r1 - len(clip_audio_raw)), mode="symmetric")[r0:r1]
            else:
                sample_audio = clip_audio_raw[r0:r1]

            if len(sample_words) >= 2:
                skel, message = MotionPreprocessor(sample_skeletons).get()
                ok = skel != []
                aux = {"eid": eid, "start_frame_no": clip_s_f + start_idx, "end_frame_no": clip_s_f + fin_idx,
                       "start_time": t0, "end_time": t1, "is_correct_motion": ok, "filtering_message": message}
                if ok or self.disable_filtering:
                    out.append((sample_words, np.asarray(skel), sample_audio, sample_spectrogram, aux))
                else:
                    n_filtered_out[message] += 1

        for words, poses, audio, spectrogram, aux in out:
            self.out_store.put(sample_key(self.n_out_samples), encode_record(words, poses, audio, spectrogram, aux))
            self.n_out_samples += 1
        return n_filtered_out


# --------------------------------------------------------------------------------------------------------------- dataset

class SpeechMotionDataset:
    """data_loader/lmdb_loader_BEAT_full.py:120-253 over any SampleStore.  `__getitem__` returns
    (audio f32 [expected_audio_length], spectrogram f32 [128, expected_spectrogram_length], pose_seq f32 [n, D], eid_label f32 [8], aux)."""

    def __init__(self, store, n_poses: int, subdivision_stride: int, pose_resampling_fps: float, speaker_model=None,
                 remove_word_timing: bool = False):
        self.store = store
        self.n_poses = n_poses
        self.subdivision_stride = subdivision_stride
        self.skeleton_resampling_fps = pose_resampling_fps
        self.remove_word_timing = remove_word_timing
        self.expected_audio_length = int(round(n_poses / pose_resampling_fps * SAMPLE_RATE))
        self.expected_spectrogram_length = calc_spectrogram_length_from_motion_length(n_poses, pose_resampling_fps)
        self.lang_model = None
        self.speaker_model = speaker_model
        self.n_samples = len(store)

    def __len__(self) -> int:
        return self.n_samples

    def set_lang_model(self, lang_model) -> None:
        self.lang_model = lang_model

    def __getitem__(self, idx: int):
        import torch
        blob = self.store.get(sample_key(idx))
        if blob is None:
            raise IndexError(idx)
        _words, pose_seq, audio, spectrogram, aux_info = decode_record(blob)
        audio = make_audio_fixed_length(np.asarray(audio), self.expected_audio_length)
        spectrogram = np.asarray(spectrogram)[:, 0:self.expected_spectrogram_length]
        pose_seq = np.asarray(pose_seq)
        pose_t = torch.from_numpy(np.array(pose_seq)).reshape((pose_seq.shape[0], -1)).float()
        audio_t = torch.from_numpy(np.array(audio)).float()
        spec_t = torch.from_numpy(np.array(spectrogram)).float()        # fp16 storage -> fp32 (`.float()`, :242)
        label = torch.from_numpy(one_hot_eid(aux_info["eid"])).float()
        return audio_t, spec_t, pose_t, label, aux_info


def audio_classifier_collate_fn(data):
    """data_loader/lmdb_loader_BEAT_full.py:63-75."""
    from torch.utils.data.dataloader import default_collate
    audio, spectrogram, poses_seq, eid_label, aux_info = zip(*data)
    aux = {key: default_collate([d[key] for d in aux_info]) for key in aux_info[0]}
    return default_collate(audio), default_collate(spectrogram), default_collate(poses_seq), default_collate(eid_label), aux


def clips_from_raw_audio(eid: str, audio_raw: np.ndarray, skeletons: np.ndarray, words, fps_in: float, device="cuda:0") -> dict:
    """Build one upstream-shaped video dict from raw 16 kHz audio: the whole-clip mel is computed on the GPU, so that
    raw audio -> DataPreprocessor -> SpeechMotionDataset -> generator starts from samples, as the north star asks."""
    duration = len(audio_raw) / SAMPLE_RATE
    return {"eid": eid, "clips": [{
        "skeletons": np.asarray(skeletons), "audio_feat": extract_melspectrogram(audio_raw, device=device),
        "audio_raw": np.asarray(audio_raw), "words": [list(w) for w in words],
        "start_frame_no": 0, "end_frame_no": int(round(duration * fps_in)), "start_time": 0.0, "end_time": duration}]}
