"""Initialisers shared by the module mirrors (distributional parity only: SURVEY App. A.5)."""
import torch.nn as nn


def kaiming_conv(conv):
    """kaiming_uniform_(a=0, fan_in, relu) as the reference applies to every conv weight except the residual
    k1 conv (e.g. convolutional_encoder.py:24).  For ConvTranspose1d PyTorch's fan_in is weight.size(1)*k,
    i.e. Cout*k -- same call, same quirk."""
    nn.init.kaiming_uniform_(conv.weight, a=0, mode="fan_in", nonlinearity="relu")
    return conv
