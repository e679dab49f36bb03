"""``EchoedSpeechReconModel`` -- two frozen VQ-VAE encoders (speech + RIR) feeding a fresh decoder.

Reference: vq_vae/echoed_speech_model.py:9-56.
"""
import torch
from torch import nn

from acoustic_locating_vq_vae.vq_vae.deconvolutional_decoder import DeconvolutionalDecoder

from .. import _ops

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class EchoedSpeechReconModel(nn.Module):
    def __init__(self, rir_model, speech_model, out_channels, num_hiddens, num_residual_layers,
                 num_residual_hiddens, use_jitter):
        super().__init__()
        self.rir_model = rir_model.to(device)
        self.speech_model = speech_model.to(device)
        self.rir_model._vq.set_train_vq(False)
        self.speech_model._vq.set_train_vq(False)
        self.flag_train_encoder = False
        self.embedding_dim = self.rir_model.get_embedding_dim() + self.speech_model.get_embedding_dim()
        self._decoder = DeconvolutionalDecoder(self.embedding_dim, out_channels, num_hiddens, num_residual_layers,
                                               num_residual_hiddens, use_jitter, 0.25)

    def set_train_encoder(self, flag):
        self.flag_train_encoder = flag

    def forward(self, spec_in, spec_in_rir):
        if self.training:
            _ops.note_training_forward(spec_in.device)
        grad = self.flag_train_encoder
        with torch.set_grad_enabled(grad and torch.is_grad_enabled()):
            # frozen encoders build no graph and keep no activations (the reference detaches afterwards, :53-54)
            _, rir_q, rir_perplexity, _ = self.rir_model.get_latent_indices(spec_in_rir)
            _, speech_q, speech_perplexity, _ = self.speech_model.get_latent_indices(spec_in)
        size_diff = speech_q.size(2) - rir_q.size(2)
        if size_diff > 0:
            rir_q = nn.functional.pad(rir_q, (0, size_diff))      # right zero-pad to the speech length (:41-49)
        quantized = torch.cat((speech_q, rir_q), dim=1)
        return self._decoder(quantized), speech_perplexity, rir_perplexity
