"""``Residual`` -- parameter holder + standalone forward for one residual layer.

Reference: vq_vae/modules/residual.py:33-66.  ``_block`` keeps the reference's four-slot layout
(ReLU, Conv1d k3 no-bias, ReLU, Conv1d k1 no-bias) so state_dict keys ``_block.1.weight`` / ``_block.3.weight``
and whole-module pickles stay interchangeable.  Only ``_block.1`` gets the Kaiming re-initialisation: the
reference re-initialises conv_1 twice and never conv_2 (residual.py:45,55 -- SURVEY App. B.5).
"""
import torch.nn as nn

from ... import _ops


class Residual(nn.Module):
    def __init__(self, in_channels, num_hiddens, num_residual_hiddens):
        super().__init__()
        k3 = nn.Conv1d(in_channels, num_residual_hiddens, kernel_size=3, stride=1, padding=1, bias=False)
        k1 = nn.Conv1d(num_residual_hiddens, num_hiddens, kernel_size=1, stride=1, bias=False)
        nn.init.kaiming_uniform_(k3.weight, a=0, mode="fan_in", nonlinearity="relu")
        self._block = nn.Sequential(nn.ReLU(True), k3, nn.ReLU(True), k1)

    @property
    def weights(self):
        return self._block[1].weight, self._block[3].weight

    def forward(self, x):
        """relu(x) + W2 *1 relu(W1 *3 relu(x)) -- the value ``x + block(x)`` takes after the in-place ReLU
        (residual.py:36,66).  Unlike the reference this does not overwrite the caller's ``x``."""
        w1, w2 = self.weights
        return _ops.ResidualLayerFn.apply(x, w1, w2)
