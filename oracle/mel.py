"""Oracle restatement of the feature frontend (reference data.py:15-24, :160-178).

Test infrastructure only.  PARITY UNPINNED at the reference level: torchaudio is not
installed here, so ``torchaudio.transforms.MelSpectrogram`` itself cannot be run.  This file
restates its documented defaults for the reference's constructor call

    MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80, power=1.0)

(win_length=n_fft, periodic Hann, center=True, pad_mode="reflect", normalized=False,
onesided=True, f_min=0, f_max=sample_rate/2, mel_scale="htk", norm=None) on top of
``torch.stft`` -- the routine torchaudio's Spectrogram calls -- and cross-checks it with an
independent float64 DFT (``mel_spectrogram_f64``).
"""
import math

import numpy as np
import torch

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160
N_MELS = 80
N_FREQS = N_FFT // 2 + 1  # 201


def melscale_fbanks_htk(n_freqs=N_FREQS, f_min=0.0, f_max=SAMPLE_RATE / 2.0, n_mels=N_MELS):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") in float32 torch ops.

    fb[f, m] = max(0, min(down, up)); shape [n_freqs, n_mels].  SURVEY.md section 8 F1.
    """
    all_freqs = torch.linspace(0, SAMPLE_RATE // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def num_frames(n_samples):
    """center=True: 1 + N // hop."""
    return 1 + n_samples // HOP


def mel_spectrogram(waveform, fb=None):
    """data.py:16-18 -- waveform [..., N] f32 -> magnitude mel [..., 80, 1 + N//160] f32."""
    waveform = torch.as_tensor(waveform, dtype=torch.float32)
    if fb is None:
        fb = melscale_fbanks_htk()
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1])
    spec = torch.stft(
        x, N_FFT, HOP, N_FFT, torch.hann_window(N_FFT, periodic=True), center=True,
        pad_mode="reflect", normalized=False, onesided=True, return_complex=True,
    ).abs()  # power=1.0
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)
    return mel.reshape(shape[:-1] + mel.shape[-2:])


def mel_spectrogram_f64(waveform, fb=None):
    """Independent float64 check: explicit reflect pad, framing, numpy rfft."""
    x = np.asarray(waveform, dtype=np.float64)
    assert x.ndim == 1
    if fb is None:
        fb = melscale_fbanks_htk().double().numpy()
    n = np.arange(N_FFT)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)
    xp = np.pad(x, (N_FFT // 2, N_FFT // 2), mode="reflect")
    T = num_frames(len(x))
    frames = np.stack([xp[t * HOP: t * HOP + N_FFT] * win for t in range(T)])
    mag = np.abs(np.fft.rfft(frames, axis=1))  # [T, 201]
    return (mag @ fb).T  # [80, T]


def mask_params(u1, u2, mask_param, size):
    """torchaudio.functional.mask_along_axis: (start, width) from two uniforms in [0, 1).

    value = u1 * mask_param; min_value = u2 * (size - value);
    start = floor(min_value); end = start + floor(value).  SURVEY.md section 8 F2.
    """
    value = np.float32(u1) * np.float32(mask_param)
    min_value = np.float32(u2) * (np.float32(size) - value)
    return int(min_value), int(value)


def apply_masks(spec, f0, fw, t0, tw):
    """Zero rows [f0, f0+fw) and columns [t0, t0+tw) of spec [80, T] (mask_value=0)."""
    spec = spec.clone()
    spec[..., f0: f0 + fw, :] = 0.0
    spec[..., :, t0: t0 + tw] = 0.0
    return spec


def train_audio_transforms(waveform, masks):
    """data.py:20-24 with the (f0, fw, t0, tw) draw made explicit."""
    return apply_masks(mel_spectrogram(waveform), *masks)


def collate(waveforms, masks=None):
    """data.py:160-178, spectrogram half: per-utterance features, zero pad at the tail,
    -> [B, 1, 80, Tmax] (returned contiguous)."""
    specs = []
    for i, w in enumerate(waveforms):
        s = mel_spectrogram(w)
        if s.dim() == 3:
            s = s.squeeze(0)
        if masks is not None:
            s = apply_masks(s, *[int(v) for v in masks[i]])
        specs.append(s.transpose(0, 1))
    out = torch.nn.utils.rnn.pad_sequence(specs, batch_first=True)
    return out.unsqueeze(1).transpose(2, 3).contiguous()
