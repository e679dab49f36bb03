"""CPU oracle for the location head -- TEST INFRASTRUCTURE, never imported by the product path.

Restates ``LocationModule`` (/root/reference/src/acoustic_locating_vq_vae/vq_vae/location_model/location_model.py:5-29)
as a pure function of a parameter dict, plus the caller's loss (scripts/train_location.py:74-78).  Pinned: the real
module imports in the build container (torch only), ``oracle/check_against_reference.py`` compares this restatement
with it (outputs and all ten gradients), and ``tests/golden/g7_location.npz`` (made by the real module) is re-checked
on any machine by ``tests/test_oracle_goldens.py``.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .vqvae_oracle import hashed_uniform

LAYERS = (1024, 512, 512, 64)          # hidden widths fixed by location_model.py:10-17


def location_param_shapes(encoder_output_dim, num_hiddens, output_dim):
    """state_dict keys of LocationModule (location_model.py:10-18)."""
    dims = (encoder_output_dim * num_hiddens,) + LAYERS + (output_dim,)
    s = {}
    for i in range(5):
        s["fc_%d.weight" % (i + 1)] = (dims[i + 1], dims[i])
        s["fc_%d.bias" % (i + 1)] = (dims[i + 1],)
    return s


def closed_form_location_params(shapes, gain=1.0):
    """Kaiming-like closed-form fill (no RNG): weights U(+-gain*sqrt(6/fan_in)), biases U(+-0.05).  fc_1 sees one-hot
    rows (201 ones among 205 824 inputs), so its bound uses the number of ACTIVE inputs to keep activations O(1)."""
    out = {}
    for n, (key, shape) in enumerate(shapes.items()):
        numel = int(np.prod(shape))
        if key.endswith("bias"):
            scale = 0.05
        else:
            scale = gain * (6.0 / shape[1]) ** 0.5
        out[key] = torch.from_numpy(hashed_uniform(numel, 7000 + n, scale).reshape(shape))
    return out


def location_forward(x, p):
    """location_model.py:20-29: flatten -> fc_1 -> relu -> ... -> fc_5 (x: dense (B, L, K) one-hot or anything)."""
    z = F.linear(torch.flatten(x, start_dim=1), p["fc_1.weight"], p["fc_1.bias"])
    for i in (2, 3, 4, 5):
        z = F.linear(F.relu(z), p["fc_%d.weight" % i], p["fc_%d.bias" % i])
    return z


def location_loss(location, theta):
    """train_location.py:77-78: mse(location, theta / pi), mean reduction (theta (B,) broadcasts against (B, 1) exactly
    as in the script)."""
    return F.mse_loss(location, torch.as_tensor(theta).float() / torch.pi, reduction="mean")


def onehot_codes(idx, K):
    """(B, L) int -> dense (B, L, K) fp32 one-hot, the shape train_location.py:74 builds."""
    return F.one_hot(torch.as_tensor(idx).long(), K).float()


def hashed_indices(B, L, K, seed):
    u = hashed_uniform(B * L, seed, 0.5).astype(np.float64) + 0.5
    return np.minimum((u * K).astype(np.int64), K - 1).reshape(B, L)
