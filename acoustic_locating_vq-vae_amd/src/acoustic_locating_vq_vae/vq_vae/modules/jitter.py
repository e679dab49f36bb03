"""``Jitter`` -- training-time latent jitter.

Reference: vq_vae/modules/jitter.py:42-70.  The per-column decisions are drawn on the host with the
reference's exact ``np.random`` call order (so a seeded run picks the same columns), including its inverted
probability (a column is replaced with probability 1 - p, :55).  The copy itself is one HIP gather; replaced
columns receive no gradient because the reference copies from a detached clone (:48,68).
"""
import torch
import torch.nn as nn

from ... import _ops


class Jitter(nn.Module):
    def __init__(self, probability=0.12):
        super().__init__()
        self._probability = probability

    def draw(self, length, device):
        src = _ops.jitter_source_index(length, self._probability)
        return torch.from_numpy(src).to(device, non_blocking=True)

    def forward(self, quantized):
        _ops._need_gpu(quantized, "Jitter")
        return _ops.JitterFn.apply(quantized, self.draw(quantized.size(2), quantized.device))
