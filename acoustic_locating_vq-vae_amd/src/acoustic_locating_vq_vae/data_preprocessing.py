"""Collate functions of the train loops (host side).

Reference: data_preprocessing.py -- ``batchify_spectrograms`` (:7-16), ``combine_tensors_with_min_dim`` (:18-52),
``spec_dataset_preprocessing`` (:55-89).  Same names, arguments, return values and error behaviour.  The reference
module also imports ``rir_generator`` and ``scipy.signal`` at the top without using them in these functions, which
makes every training script depend on a C++ package; this one needs torch only.
"""
import torch

SPEC_FRAMES = 500      # every sample is cropped to its first 500 frames; shorter ones are dropped (:64-69)


def combine_tensors_with_min_dim(tensor_list):
    """[(1, H, x_i)] -> complex64 (N, H, min_i x_i): every tensor cut to the shortest last dimension."""
    if not tensor_list:
        raise ValueError("Input tensor list cannot be empty")
    height = tensor_list[0].shape[1]
    if any(t.shape[1] != height for t in tensor_list):
        raise ValueError("All tensors in the list must have the same height (H)")
    width = min(t.shape[2] for t in tensor_list)
    return torch.cat([t[:, :, :width] for t in tensor_list], dim=0).to(torch.complex64)


def batchify_spectrograms(data, NFFT, noverlap):
    """LibriSpeech-style 6-tuples -> (combined first fields, the LAST item's sixth field) -- as the reference does."""
    sample_rate = None
    firsts = []
    for item in data:
        firsts.append(item[0].unsqueeze(0))
        sample_rate = item[5]
    return combine_tensors_with_min_dim(firsts), sample_rate


def spec_dataset_preprocessing(data):
    """[6-tuple] -> (speech_specs, rir_specs, echoed_specs, fs, theta, wiener_est), each stacked over the samples that
    have at least 500 frames (spectrograms cropped to 500).  If none qualifies all six are the same empty list."""
    kept = [item for item in data if item[0].shape[1] >= SPEC_FRAMES]
    if not kept:
        empty = []
        return empty, empty, empty, empty, empty, empty
    speech, rir, echoed, fs, theta, wiener = zip(*kept)
    crop = lambda specs: torch.stack([s[:, :SPEC_FRAMES] for s in specs])   # noqa: E731
    return (crop(speech), crop(rir), crop(echoed), torch.stack([torch.as_tensor(f) for f in fs]), torch.stack(theta),
            torch.stack(wiener))
