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
    """list of [1,N] / [N] waveforms (any device) -> [B,1,80,Tmax] on the GPU, one launch."""
    device = device or _device()
    flats = [w.reshape(-1) for w in waveforms]
    lens = [int(w.numel()) for w in flats]
    offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64)
    flat = torch.cat([w.to(device=device, dtype=torch.float32, non_blocking=True) for w in flats])
    masks = draw_masks([ops.num_frames(n) for n in lens]) if augment else None
    return ops.melspec(flat, offs, t_max=t_max, masks=masks)


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
