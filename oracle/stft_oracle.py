"""STFT / power-spectrogram front end oracle.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

PARITY UNPINNED: the reference builds its spectrograms with
``torchaudio.transforms.Spectrogram(n_fft=400, hop_length=160, power=None,
center=True, pad=0, normalized=True)`` followed by ``.abs().pow(2)``
(/root/reference/scripts/genereate_dataset.py:90-91, 37, 39, 47-49).  torchaudio
is not installed here (nor pinned by the reference's pyproject.toml) and the
reference holds no spectrogram fixture, so this restates torchaudio's published
semantics on ``torch.stft``: reflect-pad n_fft/2, periodic Hann window, one-sided
DFT, divide by sqrt(sum(window^2)) ("window" normalisation is what a boolean
``normalized=True`` selects), then |.|^2.
"""
import numpy as np
import torch


def stft_complex(wave, n_fft=400, hop=160):
    window = torch.hann_window(n_fft, periodic=True, dtype=wave.dtype)
    spec = torch.stft(wave, n_fft, hop_length=hop, win_length=n_fft, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    return spec / window.pow(2.0).sum().sqrt()


def stft_power(wave, n_fft=400, hop=160):
    """wave (..., S) -> power spectrogram (..., n_fft//2+1, 1+S//hop)."""
    return stft_complex(wave, n_fft, hop).abs().pow(2)


def stft_power_direct(wave, n_fft=400, hop=160):
    """Independent float64 restatement (explicit frames + DFT matrix) used to
    cross-check ``stft_power`` and, at small sizes, the HIP kernel."""
    x = np.asarray(wave, dtype=np.float64)
    pad = n_fft // 2
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode="reflect")
    n_frames = 1 + x.shape[-1] // hop
    n = np.arange(n_fft)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)
    k = np.arange(n_fft // 2 + 1)
    ang = 2 * np.pi * np.outer(k, n) / n_fft
    cos_m, sin_m = np.cos(ang), np.sin(ang)
    frames = np.stack([xp[..., t * hop:t * hop + n_fft] for t in range(n_frames)], axis=-2) * win
    re = frames @ cos_m.T
    im = -(frames @ sin_m.T)
    power = (re ** 2 + im ** 2) / np.sum(win ** 2)
    return np.swapaxes(power, -1, -2)
