"""``DeconvolutionalDecoder`` -- [Jitter] -> Conv1d(k3) -> residual stack -> 3x ConvTranspose1d(k3,s1,p1).

Reference: vq_vae/deconvolutional_decoder.py:9-79.  A stride-1 ConvTranspose1d is a Conv1d with flipped,
transposed weights, so all three run on the same HIP conv kernel with the ALVQ_W_IOK
weight read pattern; no weight re-layout is materialised.
"""
import torch.nn as nn

from . import _init
from .modules.jitter import Jitter
from .modules.residual_stack import ResidualStack
from .. import _ops


class DeconvolutionalDecoder(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, num_hiddens: int, num_residual_layers: int,
                 num_residual_hiddens: int, use_jitter: bool, jitter_probability: float):
        super().__init__()
        self._use_jitter = use_jitter
        if self._use_jitter:
            self._jitter = Jitter(jitter_probability)
        self._conv_1 = _init.kaiming_conv(nn.Conv1d(in_channels, num_hiddens, kernel_size=3, stride=1, padding=1))
        self._residual_stack = ResidualStack(num_hiddens, num_hiddens, num_residual_layers, num_residual_hiddens)

        def up(cin, cout):
            return _init.kaiming_conv(nn.ConvTranspose1d(cin, cout, kernel_size=3, stride=1, padding=1))

        self._conv_trans_1 = up(num_hiddens, num_hiddens)
        self._conv_trans_2 = up(num_hiddens, num_hiddens)
        self._conv_trans_3 = up(num_hiddens, out_channels)

    def forward(self, inputs):
        _ops._need_gpu(inputs, "DeconvolutionalDecoder")
        src = None
        if self._use_jitter and self.training:
            src = self._jitter.draw(inputs.size(2), inputs.device)
        w1, w2 = self._residual_stack.weights
        c1, t1, t2, t3 = self._conv_1, self._conv_trans_1, self._conv_trans_2, self._conv_trans_3
        return _ops.DecoderFn.apply(inputs, src, c1.weight, c1.bias, w1, w2, t1.weight, t1.bias, t2.weight, t2.bias,
                                    t3.weight, t3.bias, self._residual_stack._num_residual_layers)
