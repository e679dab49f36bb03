"""Waveform front end on the device.

The reference prepares its training samples offline (scripts/genereate_dataset.py:35-49): a torchaudio complex
spectrogram of the clean speech, the same of the speech convolved with a room impulse response, and from the two the
"RIR spectrogram", the Wiener estimate and three power spectrograms.  These functions do that arithmetic with the
HIP kernels of csrc/stft.hip, so the train loops can start from waveforms instead of from a pre-generated dataset:

    speech_spec, rir_spec, echoed_spec, wiener_est = specs_from_waveform(wave, h_rir)      # the sample 6-tuple's tensors
    x = speech_input_from_waveform(wave)                                                   # straight into Trainer.step

The room impulse response itself comes from the ``rir_generator`` C++ package in the reference
(genereate_dataset.py:21-29), which is out of scope; any (Nh,) or (B, Nh) float64 response can be passed in.
"""
import torch

from . import _native as N
from .data_preprocessing import SPEC_FRAMES

N_FFT, HOP = 400, 160        # genereate_dataset.py:73-74 at fs = 16 kHz


def specs_from_waveform(wave, h_rir, n_fft=N_FFT, hop=HOP):
    """wave (B,S) float32 on the GPU, h_rir (Nh,) or (B,Nh) float64 ->
    (speech_spec (B,F,T) fp32, rir_spec fp64, echoed_spec fp64, wiener_est (B,F) fp64), all powers, with the
    reference's precisions: the clean branch is complex64, everything touched by the convolution is float64."""
    wave = wave.contiguous()
    echoed_wave = N.fir_same(wave, h_rir.contiguous())                      # :38  ss.convolve(..., mode='same')
    speech = N.stft_complex(wave, n_fft, hop)                               # :37
    echoed = N.stft_complex(echoed_wave, n_fft, hop)                        # :39
    speech_pow, echoed_pow, rir_pow, wiener = N.spec_rir_wiener(speech, echoed)   # :41-49
    return speech_pow, rir_pow, echoed_pow, wiener


def speech_input_from_waveform(wave, n_fft=N_FFT, hop=HOP, frames=SPEC_FRAMES):
    """wave (B,S) float32 -> the raw batch a speech train step takes: power spectrogram cropped to ``frames`` frames
    (the collate's crop, data_preprocessing.py:67-69); ``Trainer.step`` standardises it on the device."""
    power = N.stft_power(wave.contiguous(), n_fft, hop)
    if power.shape[2] < frames:
        raise ValueError("waveform gives %d frames, the collate needs %d" % (power.shape[2], frames))
    return power[:, :, :frames].contiguous()
