"""CPU restatement of the dataset generator's per-sample arithmetic -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/scripts/genereate_dataset.py:35-49 statement by statement, with scipy.signal.convolve (the
reference's own library, present here) and the torch.stft restatement of torchaudio's Spectrogram from
oracle/stft_oracle.py (torchaudio absent: PARITY UNPINNED for the STFT itself, see that file).
"""
import numpy as np
import scipy.signal as ss
import torch

from . import stft_oracle


def convert_speech_to_specs(waveform, h_RIR, n_fft=400, hop=160, waveform_h=None):
    """waveform (1,S) float32 tensor, h_RIR (Nh,) float64 array -> (speech_spec, rir_spec, echoed_spec, wiener_est).
    ``waveform_h``: use this echoed waveform instead of convolving (isolates the spectrogram arithmetic in tests)."""
    audio_transformer = lambda w: stft_oracle.stft_complex(w, n_fft, hop)          # noqa: E731
    speech_spec = audio_transformer(waveform).squeeze(0)                           # :37  complex64 (F,T)
    if waveform_h is None:
        waveform_h = ss.convolve(waveform.squeeze().numpy(), np.squeeze(h_RIR), mode="same")   # :38  float64
    echoed_spec = audio_transformer(torch.from_numpy(waveform_h))                  # :39  complex128 (F,T)
    rir_spec = speech_spec.to(torch.complex128) / (echoed_spec + 1e-8)             # :41
    rir_spec = rir_spec / rir_spec.abs().max()                                     # :42
    wiener_est = (torch.sum(echoed_spec * torch.conj(speech_spec), dim=1) /
                  (torch.sum(speech_spec * torch.conj(speech_spec), dim=1) + 1e-8))       # :44-45
    return (speech_spec.abs().pow(2), rir_spec.abs().pow(2), echoed_spec.abs().pow(2), wiener_est.abs().pow(2))   # :46-49
