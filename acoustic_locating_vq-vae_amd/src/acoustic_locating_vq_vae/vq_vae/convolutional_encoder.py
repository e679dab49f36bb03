"""``ConvolutionalEncoder`` -- Conv1d(k3) -> shared residual stack -> outer skip.

Reference: vq_vae/convolutional_encoder.py:9-44.  Output is ``relu(h_R) + relu(h_0)``: the stack's in-place
ReLU has already rewritten ``x_conv_1`` when the outer add reads it (:42).
Runs as one autograd node: 1 + 2R fused HIP convs forward, hand-scheduled backward.
"""
import torch.nn as nn

from . import _init
from .modules.residual_stack import ResidualStack
from .. import _ops


class ConvolutionalEncoder(nn.Module):
    def __init__(self, in_channels: int, num_hiddens: int, num_residual_layers: int, num_residual_hiddens: int):
        super().__init__()
        self._conv_1 = _init.kaiming_conv(nn.Conv1d(in_channels, num_hiddens, kernel_size=3, stride=1, padding=1))
        self._relu = nn.ReLU()   # unused, as in the reference (:26); kept so pickled attribute sets match
        self._residual_stack = ResidualStack(num_hiddens, num_hiddens, num_residual_layers, num_residual_hiddens)

    def forward(self, inputs):
        _ops._need_gpu(inputs, "ConvolutionalEncoder")
        w1, w2 = self._residual_stack.weights
        return _ops.EncoderFn.apply(inputs, self._conv_1.weight, self._conv_1.bias, w1, w2,
                                    self._residual_stack._num_residual_layers)
