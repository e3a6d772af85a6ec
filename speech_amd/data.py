"""Feature frontend and batch contract -- same module surface as the reference's data.py:
``N_MELS``, ``spectrogram_transform``, ``train_audio_transforms``, ``collate_fn``,
``SortedDataset``.  collate_fn returns (spectrograms f32 [B,1,80,Tmax], labels i64 [B,Lmax],
label_lengths i32 [B]) exactly as data.py:160-178 does, but the whole batch is featurised by ONE
launch of the fused HIP frontend on the GPU (the reference runs torchaudio per utterance inside
CPU DataLoader workers), so the spectrograms are device tensors.

SpecAugment (data.py:20-24): one frequency band (< 15 mels) and one time band (< 35 frames) per
utterance are zeroed.  The (start, width) draws are made on the host from torch's global RNG in
the order torchaudio makes them -- per utterance: rand()*15, rand()*(80 - value), rand()*35,
rand()*(T_b - value) -- and handed to the kernel.
"""
import gc
import json
import os
import pickle
import wave

import numpy as np
import torch

from . import ops, text

N_MELS = 80
SAMPLE_RATE = 16000
FREQ_MASK_PARAM = 15
TIME_MASK_PARAM = 35


def _device():
    if not torch.cuda.is_available():
        raise ops._lib.SpeechHipError("speech_amd.data needs a GPU: the frontend has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _mask_draw(mask_param, size):
    """torchaudio.functional.mask_along_axis: two uniforms -> (start, width)."""
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (size - value)
    return int(min_value.long()), int(value.long())


def draw_masks(frame_counts):
    """[(f0, fw, t0, tw)] per utterance, consuming torch's global CPU RNG like the reference."""
    masks = []
    for tb in frame_counts:
        f0, fw = _mask_draw(FREQ_MASK_PARAM, N_MELS)
        t0, tw = _mask_draw(TIME_MASK_PARAM, tb)
        masks.append((f0, fw, t0, tw))
    return torch.tensor(masks, dtype=torch.int32).reshape(-1, 4)


def featurize(waveforms, augment=False, t_max=None, device=None):
    """list of [1,N] / [N] waveforms (any device) -> [B,1,80,Tmax] on the GPU, one launch.  Waveforms that
    are ALL int16 PCM (what load_audio returns for 16-bit files) go to the GPU as int16 -- half the bytes
    over PCIe and out of HBM -- and the kernel reads sample / 32768, the value torchaudio.load(normalize=True)
    gives the reference; anything else is converted to float32 first."""
    device = device or _device()
    flats = [w.reshape(-1) for w in waveforms]
    lens = [int(w.numel()) for w in flats]
    offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64)
    if all(w.dtype == torch.int16 for w in flats):
        flat = torch.cat([w.to(device=device, non_blocking=True) for w in flats])
    else:
        flat = torch.cat([_to_float(w).to(device=device, dtype=torch.float32, non_blocking=True) for w in flats])
    masks = draw_masks([ops.num_frames(n) for n in lens]) if augment else None
    return ops.melspec(flat, offs, t_max=t_max, masks=masks)


def _to_float(w):
    """int16 PCM -> float as torchaudio.load(normalize=True); other dtypes unchanged."""
    return w.to(torch.float32) / 32768.0 if w.dtype == torch.int16 else w


class _SpectrogramTransform:
    """Callable stand-in for torchaudio.transforms.MelSpectrogram(16000, n_fft=400, hop=160,
    n_mels=80, power=1.0) (data.py:16-18): waveform [..., N] -> [..., 80, 1 + N//160]."""

    augment = False

    def __call__(self, waveform):
        w = torch.as_tensor(waveform)
        lead = w.shape[:-1]
        rows = w.reshape(-1, w.shape[-1])
        out = featurize(list(rows), augment=self.augment)
        return out.reshape(*lead, N_MELS, out.shape[-1])


class _TrainTransform(_SpectrogramTransform):
    augment = True


spectrogram_transform = _SpectrogramTransform()
train_audio_transforms = _TrainTransform()


def collate_fn(data, data_type="train", vocab=text):
    """data.py:160-178: list of (waveform [1,N], utterance str) -> the batch contract."""
    waveforms = [w for (w, _) in data]
    spectrograms = featurize(waveforms, augment=(data_type == "train"))
    label_list = [torch.tensor(vocab.text_to_int(u), dtype=torch.int64) for (_, u) in data]
    label_lengths = [len(x) for x in label_list]
    labels = torch.nn.utils.rnn.pad_sequence(label_list, batch_first=True)
    return spectrograms, labels, torch.tensor(label_lengths, dtype=torch.int32)


class SortedDataset(torch.utils.data.Dataset):
    """data.py:63-79: one whole pre-sorted batch per __getitem__."""

    def __init__(self, batch_size):
        super().__init__()
        self.batch_size = batch_size
        self.paths = []

    def __len__(self):
        return len(self.paths) // self.batch_size

    def __getitem__(self, i):
        return [self.get_clip(j) for j in range(i * self.batch_size, (i + 1) * self.batch_size)]

    def get_clip(self, i):
        raise NotImplementedError


# ------------------------------------------------------------------------------------------
# Dataset / IO path (data.py:30-157): manifests of (path, n_samples) sorted by duration, LibriSpeech
# transcript lookup, TV-style wav + txt pairs.  Audio decoding: the reference calls torchaudio.load;
# here 16-bit PCM WAV is read with the standard library and handed on as RAW int16 (the GPU featuriser
# scales it), other containers need `soundfile` if it is installed.
# ------------------------------------------------------------------------------------------
class _ManifestUnpickler(pickle.Unpickler):
    """Manifests are pickled lists of (path, n_samples) tuples (data.py:31-32): plain containers only."""

    def find_class(self, module, name):
        raise pickle.UnpicklingError(f"manifest may hold only lists/tuples/str/numbers, found {module}.{name}")


def load_manifest(dataset_path):
    """[(path, n_samples), ...] from a .pkl (the reference's format) or .json manifest."""
    if dataset_path.endswith(".json"):
        with open(dataset_path) as f:
            return [(str(p), n) for p, n in json.load(f)]
    with open(dataset_path, "rb") as f:
        return [(str(t[0]), t[1]) for t in _ManifestUnpickler(f).load()]


def get_librispeech_paths(dataset_path):
    """data.py:30-33: (path, seconds) pairs."""
    return [(p, n / SAMPLE_RATE) for p, n in load_manifest(dataset_path)]


def load_audio(audio_path):
    """-> (waveform [1, N], sample_rate).  16-bit PCM WAV: int16 samples, undivided (featurize / the
    kernel apply the 1/32768 of torchaudio's normalize=True)."""
    if audio_path.lower().endswith(".wav"):
        with wave.open(audio_path, "rb") as f:
            if f.getsampwidth() != 2:
                raise ValueError(f"{audio_path}: only 16-bit PCM WAV is read natively")
            nch, sr = f.getnchannels(), f.getframerate()
            pcm = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2").reshape(-1, nch).T.copy()
        return torch.from_numpy(pcm), sr
    try:
        import soundfile
    except ImportError as e:
        raise RuntimeError(f"{audio_path}: decoding this container needs the soundfile package") from e
    pcm, sr = soundfile.read(audio_path, dtype="int16", always_2d=True)
    return torch.from_numpy(pcm.T.copy()), sr


def get_librispeech_clip(audio_path):
    """data.py:36-52: waveform + lower-cased transcript line of <speaker>-<chapter>.trans.txt."""
    path, filename = os.path.split(audio_path)
    fileid = filename.split(".")[0]
    speaker_id, chapter_id, _ = fileid.split("-")
    waveform, _ = load_audio(audio_path)
    utterance = None
    with open(os.path.join(path, speaker_id + "-" + chapter_id + ".trans.txt")) as f:
        for line in f:
            fileid_text, utt = line.strip().split(" ", 1)
            if fileid == fileid_text:
                utterance = utt
                break
    if utterance is None:
        raise KeyError(f"{fileid} not in its chapter's .trans.txt")
    return waveform, utterance.lower()


def get_tv_clip(audio_path):
    """data.py:55-60: waveform + the .txt next to the .wav."""
    text_path = audio_path[:-4] + ".txt" if audio_path.endswith(".wav") else audio_path + ".txt"
    waveform, _ = load_audio(audio_path)
    with open(text_path) as f:
        return waveform, f.read().strip()


def _remap(path, root_map):
    for old, new in root_map:
        path = path.replace(old, new)
    return path


class IBMDataset(torch.utils.data.Dataset):
    """data.py:82-102: every *.wav under one directory with its transcript in the .txt next to it, one clip
    per item (the loader batches it).  root replaces the reference's hard-coded "/hd1/ibm"."""

    def __init__(self, root="/hd1/ibm"):
        super().__init__()
        self.paths = sorted(os.path.join(root, f) for f in os.listdir(root) if f.endswith("wav"))

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        audio_path = self.paths[i]
        waveform, _ = load_audio(audio_path)
        with open(audio_path.replace(".wav", ".txt")) as f:
            utterance = f.read().strip()
        return waveform, utterance


class SortedLibriSpeech(SortedDataset):
    """data.py:105-115.  root_map replaces the reference's hard-coded "/data" -> "/hd1"."""

    def __init__(self, dataset_path, batch_size, root_map=(("/data", "/hd1"),)):
        super().__init__(batch_size)
        assert dataset_path.endswith((".pkl", ".json"))
        self.paths = [t[0] for t in get_librispeech_paths(dataset_path)]
        if "train" in dataset_path:
            self.paths = self.paths[:-1000]      # remove the longest clips
        self.root_map = tuple(root_map)

    def get_clip(self, i):
        return get_librispeech_clip(_remap(self.paths[i], self.root_map))


class SortedTV(SortedDataset):
    """data.py:118-130: wav + txt pairs from several manifests; `device` picks the disk /hd<device+1>."""

    def __init__(self, dataset_paths, batch_size, device, root_map=None):
        super().__init__(batch_size)
        hd = "/hd" + str(device + 1)
        self.root_map = tuple(root_map) if root_map is not None else (("/data", hd), ("/hd1", hd))
        for dataset_path in dataset_paths:
            self.paths.extend(_remap(t[0], self.root_map) for t in load_manifest(dataset_path))

    def get_clip(self, i):
        return get_tv_clip(self.paths[i])


class CombinedTVLibriSpeech(SortedDataset):
    """data.py:133-157: LibriSpeech (minus its 2000 longest) and TV manifests merged and re-sorted by
    duration; paths kept as one numpy bytes array (the reference does this to keep fork()ed DataLoader
    workers from copying a Python list).  As in the reference, only the LibriSpeech manifest's second
    field is divided by 16000: TV manifests merged here carry seconds."""

    def __init__(self, librispeech_dataset_path, tv_dataset_paths, batch_size, device, root_map=None):
        super().__init__(batch_size)
        tuples = get_librispeech_paths(librispeech_dataset_path)[:-2000]
        for dataset_path in tv_dataset_paths:
            tuples.extend(load_manifest(dataset_path))
        tuples = sorted(tuples, key=lambda t: t[1])
        self.paths = np.array([t[0] for t in tuples]).astype(np.bytes_)
        del tuples
        gc.collect()
        self.hd = "/hd" + str(device + 1)
        self.root_map = root_map

    def get_clip(self, i):
        audio_path = str(self.paths[i], encoding="utf-8")
        if self.root_map is not None:
            audio_path = _remap(audio_path, self.root_map)
        if "LibriSpeech" in audio_path:
            return get_librispeech_clip(audio_path if self.root_map is not None else audio_path.replace("/data", self.hd))
        if "clean" in audio_path:
            return get_tv_clip(audio_path if self.root_map is not None else audio_path.replace("/hd1", self.hd))
        if "gigaspeech" in audio_path:
            return get_tv_clip(audio_path)
        raise ValueError(f"{audio_path} not in datasets")


class SyntheticSorted(SortedDataset):
    """Synthetic 16 kHz corpus with the SortedDataset protocol: duration-sorted clips of noise plus
    one sine, random transcripts at ~15 characters per second.  Stands in
    for the LibriSpeech/TV manifests (data.py:105-157), which need corpora that are not here."""

    def __init__(self, n_clips, batch_size, min_s=1.0, max_s=15.0, seed=7, chars_per_s=15.0):
        super().__init__(batch_size)
        g = torch.Generator().manual_seed(seed)
        durs = torch.sort(torch.rand(n_clips, generator=g) * (max_s - min_s) + min_s).values
        self.paths = [f"synthetic/{i:07d}" for i in range(n_clips)]
        self.samples = (durs * SAMPLE_RATE).long().tolist()
        self.seed = seed
        self.chars_per_s = chars_per_s

    def get_clip(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        n = self.samples[i]
        t = torch.arange(n) / SAMPLE_RATE
        f = float(torch.randint(80, 4000, (1,), generator=g))
        wav = (0.1 * torch.randn(n, generator=g) + 0.2 * torch.sin(2 * np.pi * f * t)).clamp(-1, 1)
        n_chars = max(1, int(n / SAMPLE_RATE * self.chars_per_s * (0.5 + 0.5 * float(torch.rand(1, generator=g)))))
        ids = torch.randint(1, len(text.CHARS), (n_chars,), generator=g).tolist()
        return wav.unsqueeze(0), text.int_to_text(ids)
