"""``ResidualStack`` -- R applications of ONE shared ``Residual`` followed by ReLU.

Reference: vq_vae/modules/residual_stack.py:36-46.  ``[Residual(...)] * n`` (:40-41) repeats one module
object, so there are 2 unique weight tensors, the state_dict shows R aliased copies, and the weight grads
are sums over the R uses.  The forward is one autograd node chaining 2R fused HIP convs.
"""
import torch.nn as nn

# same import root as the reference (residual_stack.py:28) -- resolved by the alias finder in src/__init__.py
from src.acoustic_locating_vq_vae.vq_vae.modules.residual import Residual

from ... import _ops


class ResidualStack(nn.Module):
    def __init__(self, in_channels, num_hiddens, num_residual_layers, num_residual_hiddens):
        super().__init__()
        self._num_residual_layers = num_residual_layers
        shared = Residual(in_channels, num_hiddens, num_residual_hiddens)
        self._layers = nn.ModuleList([shared] * num_residual_layers)

    @property
    def weights(self):
        return self._layers[0].weights

    def forward(self, x):
        _ops._need_gpu(x, "ResidualStack")
        w1, w2 = self.weights
        return _ops.StackFn.apply(x, w1, w2, self._num_residual_layers)
