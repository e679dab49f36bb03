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
        """Device int32[length] source-index vector for this call.  If ``pin_buffer`` was used, the persistent
        buffer is returned unchanged (the owner refreshes it between hipGraph replays)."""
        static = self.__dict__.get("_static_src")
        if static is not None and static.numel() == length and static.device == device:
            return static
        src = _ops.jitter_source_index(length, self._probability)
        return torch.from_numpy(src).to(device, non_blocking=True)

    def pin_buffer(self, length, device):
        """Switch to a persistent device buffer (fixed address, as a captured graph needs).  It starts as the
        identity (no column replaced) and draws nothing, so the np.random stream stays aligned with an eager run;
        the owner calls ``refresh()`` before every step."""
        self.__dict__["_static_src"] = torch.arange(length, dtype=torch.int32, device=device)
        return self.__dict__["_static_src"]

    def refresh(self):
        """Draw a new index vector (same np.random call order) into the persistent buffer."""
        static = self.__dict__["_static_src"]
        src = _ops.jitter_source_index(static.numel(), self._probability)
        static.copy_(torch.from_numpy(src), non_blocking=True)

    def unpin_buffer(self):
        self.__dict__.pop("_static_src", None)

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_static_src", None)     # never pickled: reference checkpoints carry no such attribute
        return state

    def forward(self, quantized):
        _ops._need_gpu(quantized, "Jitter")
        return _ops.JitterFn.apply(quantized, self.draw(quantized.size(2), quantized.device))
