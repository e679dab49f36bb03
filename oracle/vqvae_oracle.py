"""Functional PyTorch-CPU restatement of the reference VQ-VAE hot path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``) -- not product code.

Every function takes plain tensors / a ``{state_dict key: tensor}`` mapping with
the reference's own key names, so weights captured from the reference (or from
the product modules) can be fed in unchanged.  Gradients come from autograd on
this restatement.  Citations are into ``/root/reference/src/acoustic_locating_vq_vae``
(``SRC/``) and ``/root/reference/scripts`` (``SCR/``).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- residual stack
def residual_stack(h, w1, w2, num_layers):
    """SRC/vq_vae/modules/residual_stack.py:36-46 + residual.py:33-66.

    One shared (w1, w2) pair is applied ``num_layers`` times
    (``[Residual(...)] * n``, residual_stack.py:40-41).  ``nn.ReLU(True)`` is the
    first op of the block (residual.py:36) so the skip operand of ``x + block(x)``
    (residual.py:66) is already ReLU'd.
    """
    for _ in range(num_layers):
        t = F.relu(h)
        u = F.relu(F.conv1d(t, w1, None, padding=1))
        h = t + F.conv1d(u, w2, None)
    return F.relu(h)


def encoder(x, p, prefix, num_layers):
    """SRC/vq_vae/convolutional_encoder.py:39-44.

    ``x_conv_1`` is mutated by the stack's in-place ReLU before the outer add
    (convolutional_encoder.py:42), hence ``+ relu(h0)``.
    """
    h0 = F.conv1d(x, p[prefix + "_conv_1.weight"], p[prefix + "_conv_1.bias"], padding=1)
    w1 = p[prefix + "_residual_stack._layers.0._block.1.weight"]
    w2 = p[prefix + "_residual_stack._layers.0._block.3.weight"]
    return residual_stack(h0, w1, w2, num_layers) + F.relu(h0)


# --------------------------------------------------------------------------- vector quantiser
def vq_distances(flat, codebook):
    """SRC/vq_vae/vector_quantizer.py:34-36, same op order (fl(fl(a+b) - 2c))."""
    return (torch.sum(flat ** 2, dim=1, keepdim=True)
            + torch.sum(codebook ** 2, dim=1)
            - 2 * torch.matmul(flat, codebook.t()))


def vector_quantizer(z, codebook, commitment_cost, train_vq=True):
    """SRC/vq_vae/vector_quantizer.py:29-58.

    Returns (loss, quantized_st, perplexity, indices).  Rows are D consecutive
    floats of the contiguous (B, D, L) buffer -- no permute (vector_quantizer.py:32).
    The reference's ``onehot @ E`` (:43) is a gather (bit-identical).
    """
    shape = z.shape
    d = codebook.shape[1]
    flat = z.reshape(-1, d)
    idx = torch.argmin(vq_distances(flat, codebook), dim=1)
    cb = codebook if train_vq else codebook.detach()
    q = cb[idx].view(shape)
    e_latent = F.mse_loss(q.detach(), z)
    q_latent = F.mse_loss(q, z.detach())
    loss = q_latent + commitment_cost * e_latent
    q_st = z + (q - z).detach()
    k = codebook.shape[0]
    probs = torch.bincount(idx, minlength=k).to(z.dtype) / idx.numel()
    perplexity = torch.exp(-torch.sum(probs * torch.log(probs + 1e-10)))
    return loss, q_st.contiguous(), perplexity, idx


def onehot(idx, k, dtype=torch.float32):
    """Dense encodings as returned by the reference (vector_quantizer.py:39-40)."""
    enc = torch.zeros(idx.numel(), k, dtype=dtype)
    enc.scatter_(1, idx.view(-1, 1), 1)
    return enc


# --------------------------------------------------------------------------- jitter
def jitter_source_index(length, probability, rng=np.random):
    """SRC/vq_vae/modules/jitter.py:50-68, same ``np.random`` call order.

    Returns int64[length]: the column each output column copies from (``i`` itself
    if kept).  A column is replaced with probability ``1 - probability`` (the
    index inversion at jitter.py:55).
    """
    src = np.arange(length, dtype=np.int64)
    for i in range(length):
        replace = [True, False][rng.choice([1, 0], p=[probability, 1 - probability])]
        if replace:
            if i == 0:
                src[i] = 1
            elif i == length - 1:
                src[i] = i - 1
            else:
                src[i] = i + rng.choice([-1, 1], p=[0.5, 0.5])
    return src


def jitter(q, src):
    """Apply a jitter source-index vector; replaced columns carry no gradient
    (they are copied from ``quantized.detach()``, jitter.py:48,68)."""
    src_t = torch.as_tensor(src, dtype=torch.long)
    keep = (src_t == torch.arange(q.shape[2])).view(1, 1, -1)
    return torch.where(keep, q, q.detach()[:, :, src_t])


# --------------------------------------------------------------------------- decoder
def decoder(q, p, prefix, num_layers, jitter_src=None):
    """SRC/vq_vae/deconvolutional_decoder.py:62-79."""
    x = q if jitter_src is None else jitter(q, jitter_src)
    x = F.conv1d(x, p[prefix + "_conv_1.weight"], p[prefix + "_conv_1.bias"], padding=1)
    w1 = p[prefix + "_residual_stack._layers.0._block.1.weight"]
    w2 = p[prefix + "_residual_stack._layers.0._block.3.weight"]
    x = residual_stack(x, w1, w2, num_layers)
    x = F.relu(F.conv_transpose1d(x, p[prefix + "_conv_trans_1.weight"], p[prefix + "_conv_trans_1.bias"], padding=1))
    x = F.relu(F.conv_transpose1d(x, p[prefix + "_conv_trans_2.weight"], p[prefix + "_conv_trans_2.bias"], padding=1))
    return F.conv_transpose1d(x, p[prefix + "_conv_trans_3.weight"], p[prefix + "_conv_trans_3.bias"], padding=1)


# --------------------------------------------------------------------------- whole model
def latent(x, p, num_layers, commitment_cost, train_vq=True, average_pooling=False, prefix=""):
    """ConvolutionalVQVAE.get_latent_representation (convolutional_vq_vae.py:102-105)
    plus the optional pooling of forward (:96-97)."""
    z = encoder(x, p, prefix + "_encoder.", num_layers)
    z = F.conv1d(z, p[prefix + "_pre_vq_conv.weight"], p[prefix + "_pre_vq_conv.bias"], padding=1)
    if average_pooling:
        z = torch.mean(z, dim=2, keepdim=True)
    loss, q_st, perplexity, idx = vector_quantizer(z, p[prefix + "_vq._embedding.weight"], commitment_cost, train_vq)
    return z, loss, q_st, perplexity, idx


def vqvae_forward(x, p, num_layers, commitment_cost, jitter_src=None, train_vq=True, average_pooling=False):
    """ConvolutionalVQVAE.forward (convolutional_vq_vae.py:93-100).

    Returns dict(z, vq_loss, q_st, perplexity, idx, recon).
    """
    z, loss, q_st, perplexity, idx = latent(x, p, num_layers, commitment_cost, train_vq, average_pooling)
    recon = decoder(q_st, p, "_decoder.", num_layers, jitter_src)
    return dict(z=z, vq_loss=loss, q_st=q_st, perplexity=perplexity, idx=idx, recon=recon)


def echoed_forward(x, x_rir, p, speech_layers, rir_layers, dec_layers, commitment_cost=0.25,
                   jitter_src=None, train_encoder=False):
    """EchoedSpeechReconModel.forward (echoed_speech_model.py:36-56).

    ``p`` uses the echoed model's own state_dict keys (``rir_model.*``,
    ``speech_model.*``, ``_decoder.*``).  Both codebooks are frozen (:17-18).
    """
    _, _, rir_q, rir_perp, rir_idx = latent(x_rir, p, rir_layers, commitment_cost, False, prefix="rir_model.")
    _, _, sp_q, sp_perp, sp_idx = latent(x, p, speech_layers, commitment_cost, False, prefix="speech_model.")
    diff = sp_q.size(2) - rir_q.size(2)
    if diff > 0:
        rir_q = F.pad(rir_q, (0, diff))
    if not train_encoder:
        sp_q, rir_q = sp_q.detach(), rir_q.detach()
    quantized = torch.cat((sp_q, rir_q), dim=1)
    recon = decoder(quantized, p, "_decoder.", dec_layers, jitter_src)
    return dict(recon=recon, speech_perplexity=sp_perp, rir_perplexity=rir_perp,
                speech_idx=sp_idx, rir_idx=rir_idx, quantized=quantized)


# --------------------------------------------------------------------------- callers' step arithmetic
def standardise(x):
    """SCR/train_speech.py:64 / train_rir.py:44 -- per-(b, l) over dim=1, unbiased std."""
    return (x - torch.mean(x, dim=1, keepdim=True)) / (torch.std(x, dim=1, keepdim=True) + 1e-8)


def speech_preprocess(x_raw):
    """SCR/train_speech.py:63-66."""
    return standardise(torch.abs(x_raw))


def rir_preprocess(rir_spec, wiener_est):
    """SCR/train_rir.py:42-49 -> (x (B, T, F) non-contiguous, target (B, 1, F))."""
    x = standardise(rir_spec.float()).permute(0, 2, 1)
    w = standardise(wiener_est.float()).unsqueeze(1)
    return x, w


def adam_step(params, grads, state, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam(lr, amsgrad=False) single-tensor arithmetic (SCR/train_speech.py:154).

    ``state`` is {'step': int, 'm': [...], 'v': [...]}; updates ``params`` in place.
    """
    state["step"] += 1
    t = state["step"]
    b1, b2 = betas
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    with torch.no_grad():
        for p_, g, m, v in zip(params, grads, state["m"], state["v"]):
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
            p_.addcdiv_(m, denom, value=-(lr / bc1))


class OracleTrainer:
    """A whole speech/RIR train step on CPU: the ``cpu_baseline`` "port" of bench.py.

    Same ATen op sequence as the reference step (SCR/train_speech.py:62-74,88-91).
    """

    def __init__(self, params, num_layers, commitment_cost=0.25, use_jitter=True,
                 jitter_probability=0.25, lr=1e-3):
        self.p = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        self.num_layers = num_layers
        self.beta = commitment_cost
        self.use_jitter = use_jitter
        self.jp = jitter_probability
        self.opt = torch.optim.Adam(list(self.p.values()), lr=lr, amsgrad=False)

    def step(self, x, target=None):
        self.opt.zero_grad()
        src = jitter_source_index(x.shape[2], self.jp) if self.use_jitter else None
        out = vqvae_forward(x, self.p, self.num_layers, self.beta, src)
        recon_error = F.mse_loss(out["recon"], x if target is None else target)
        loss = recon_error + out["vq_loss"]
        loss.backward()
        self.opt.step()
        return float(loss.detach()), float(recon_error.detach()), float(out["perplexity"].detach())


# --------------------------------------------------------------------------- exactly reproducible fills
def hashed_uniform(n, seed, scale):
    """float32[n] in [-scale, scale): integer-hash based, bit-reproducible on any
    machine (used instead of RNG so goldens do not depend on the torch version)."""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = i * np.uint64(6364136223846793005) + np.uint64(seed) * np.uint64(1442695040888963407) + np.uint64(1)
        h ^= h >> np.uint64(33)
        h *= np.uint64(0xFF51AFD7ED558CCD)
        h ^= h >> np.uint64(33)
        h *= np.uint64(0xC4CEB9FE1A85EC53)
        h ^= h >> np.uint64(33)
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return ((2.0 * u - 1.0) * scale).astype(np.float32)


def closed_form_params(shapes, codebook_scale=None, gain=1.0):
    """{key: tensor} for a {key: shape} mapping: Kaiming-like bound per tensor
    (gain*sqrt(6/fan_in) for weights, 0.05 for biases), codebook U(+-codebook_scale)."""
    out = {}
    for n, (key, shape) in enumerate(shapes.items()):
        numel = int(np.prod(shape))
        if key.endswith("_embedding.weight"):
            scale = codebook_scale if codebook_scale is not None else 1.0 / shape[0]
        elif key.endswith("bias"):
            scale = 0.05
        elif "_conv_trans_" in key:
            # keep the 1-channel RIR head (fan_in = Cout*k = 3 in the reference init) tame
            scale = gain * (6.0 / (max(shape[0], shape[1]) * shape[2])) ** 0.5
        else:
            scale = gain * (6.0 / (shape[1] * shape[2])) ** 0.5
        out[key] = torch.from_numpy(hashed_uniform(numel, 1000 + n, scale).reshape(shape))
    return out


def vqvae_param_shapes(in_channels, num_hiddens, embedding_dim, num_residual_hiddens,
                       num_embeddings, out_channels=None, prefix=""):
    """Unique parameter tensors of ConvolutionalVQVAE (aliases only at layer 0)."""
    oc = in_channels if out_channels is None else out_channels
    h, rh, d, k = num_hiddens, num_residual_hiddens, embedding_dim, num_embeddings
    s = {
        "_encoder._conv_1.weight": (h, in_channels, 3), "_encoder._conv_1.bias": (h,),
        "_encoder._residual_stack._layers.0._block.1.weight": (rh, h, 3),
        "_encoder._residual_stack._layers.0._block.3.weight": (h, rh, 1),
        "_pre_vq_conv.weight": (d, h, 3), "_pre_vq_conv.bias": (d,),
        "_vq._embedding.weight": (k, d),
    }
    s.update(decoder_param_shapes(d, oc, h, rh))
    return {prefix + key: v for key, v in s.items()}


def decoder_param_shapes(in_channels, out_channels, num_hiddens, num_residual_hiddens):
    h, rh = num_hiddens, num_residual_hiddens
    return {
        "_decoder._conv_1.weight": (h, in_channels, 3), "_decoder._conv_1.bias": (h,),
        "_decoder._residual_stack._layers.0._block.1.weight": (rh, h, 3),
        "_decoder._residual_stack._layers.0._block.3.weight": (h, rh, 1),
        "_decoder._conv_trans_1.weight": (h, h, 3), "_decoder._conv_trans_1.bias": (h,),
        "_decoder._conv_trans_2.weight": (h, h, 3), "_decoder._conv_trans_2.bias": (h,),
        "_decoder._conv_trans_3.weight": (h, out_channels, 3), "_decoder._conv_trans_3.bias": (out_channels,),
    }
