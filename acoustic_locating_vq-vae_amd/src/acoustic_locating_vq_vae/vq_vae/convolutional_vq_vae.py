"""``ConvolutionalVQVAE`` -- encoder -> pre-VQ conv -> vector quantiser -> decoder.

Reference: vq_vae/convolutional_vq_vae.py:18-105.  Same constructor, attributes, state_dict keys and
return values; every tensor op runs as a hand-written gfx950 kernel through libalvq.so.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import optim
from torch.utils.data import DataLoader

from acoustic_locating_vq_vae.vq_vae.convolutional_encoder import ConvolutionalEncoder
from acoustic_locating_vq_vae.vq_vae.deconvolutional_decoder import DeconvolutionalDecoder
from acoustic_locating_vq_vae.vq_vae.vector_quantizer import VectorQuantizer

from . import _init
from .. import _native, _ops

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class ConvolutionalVQVAE(nn.Module):
    def __init__(self, in_channels: int, num_hiddens: int, embedding_dim: int, num_residual_layers: int,
                 num_residual_hiddens: int, commitment_cost: float, num_embeddings: int, use_jitter: bool = True,
                 encoder_average_pooling: bool = False, out_channels: int = None):
        out_channels = in_channels if out_channels is None else out_channels
        super().__init__()
        self.encoder_average_pooling = encoder_average_pooling
        self._encoder = ConvolutionalEncoder(in_channels, num_hiddens, num_residual_layers, num_residual_hiddens)
        self._pre_vq_conv = _init.kaiming_conv(nn.Conv1d(num_hiddens, embedding_dim, kernel_size=3, padding=1))
        self._vq = VectorQuantizer(num_embeddings, embedding_dim, commitment_cost)
        self._decoder = DeconvolutionalDecoder(embedding_dim, out_channels, num_hiddens, num_residual_layers,
                                               num_residual_hiddens, use_jitter, 0.25)

    def get_embedding_dim(self):
        return self._vq.get_embedding_dim()

    def _latent(self, x):
        """``_pre_vq_conv(_encoder(x))`` (:94-95) as one autograd node (the encoder output never leaves the
        compute layout between the two)."""
        _ops._need_gpu(x, "ConvolutionalVQVAE")
        enc, pre = self._encoder, self._pre_vq_conv
        w1, w2 = enc._residual_stack.weights
        return _ops.LatentFn.apply(x, enc._conv_1.weight, enc._conv_1.bias, w1, w2, pre.weight, pre.bias,
                                   enc._residual_stack._num_residual_layers)

    def forward(self, x):
        if self.training:
            _ops.note_training_forward(x.device)
        z = self._latent(x)
        if _ops._LATENT_TAP is not None and z.requires_grad:
            # train_step.Trainer runs the backward in two parts (decoder + quantiser, then encoder) so that the
            # first gradient bucket's all-reduce overlaps the second part: cut the autograd graph here.
            z = _ops.tap_latent(z)
        if self.encoder_average_pooling:
            z = _ops.MeanPoolFn.apply(z)
        loss, quantized, perplexity, _ = self._vq.quantize(z)
        x_recon = self._decoder(quantized)
        return loss, x_recon, perplexity

    def get_latent_representation(self, x):
        return self._vq(self._latent(x))

    def get_latent_indices(self, x):
        """(loss, quantized, perplexity, indices[N] int64): the sparse form of get_latent_representation."""
        return self._vq.quantize(self._latent(x))

    def train_on_data(self, optimizer: optim, dataloader: DataLoader, num_training_updates, data_variance):
        """The alternative loop the reference keeps on the class (convolutional_vq_vae.py:58-91): a fresh loader
        iterator per update, MSE scaled by 1/data_variance, a crop of one trailing frame when shapes differ, progress
        printed every 100 updates, and the two history lists left on the instance."""
        self.train()
        history = {"recon": [], "perp": []}
        for update in range(1, num_training_updates + 1):
            batch = next(iter(dataloader))[0].to(device)
            optimizer.zero_grad()
            vq_loss, recon, perplexity = self(batch)
            target = batch if batch.shape == recon.shape else batch[:, :, :-1]
            recon_error = F.mse_loss(recon, target) / data_variance
            (recon_error + vq_loss).backward()
            optimizer.step()
            history["recon"].append(recon_error.item())
            history["perp"].append(perplexity.item())
            if update % 100 == 0:
                print("%d iterations\nrecon_error: %.3f\nperplexity: %.3f\n"
                      % (update, np.mean(history["recon"][-100:]), np.mean(history["perp"][-100:])))
        self.train_res_recon_error, self.train_res_perplexity = history["recon"], history["perp"]
