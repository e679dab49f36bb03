"""G8: six steps of the reference's own training loop body on the DEFAULT speech model at B = 16, recorded from the real
reference (tests/golden/make_goldens.py::speech_steps: |x|, per-frame standardise, model(x) with the jitter stream running on
from np.random.seed(9), mse + vq loss, backward, torch.optim.Adam(lr=1e-3); fresh batch per step).  With the closed-form
weights the trajectory is violent -- the loss goes 1.84 -> 19.3 -> 1.66 and the perplexity 317 -> 6 -> 157 -- which makes it a
sharp test of the whole step (preprocessing, jitter stream, forward, backward, the flat Adam): a wrong bias correction or a
dropped gradient shows within one step.

Measured (loss / perplexity, worst of the six steps, relative): f32 4.8e-5 / 2.1e-3; f16mx 8.9e-5 / 1.6e-3; f16mx_hb 3.4e-4 /
3.0e-3; bf16x3 4.1e-4 / 2.1e-3; f16mx_hd 7.2e-4 / 4.8e-3; bf16 2.8e-3 / 7.0e-3."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from acoustic_locating_vq_vae import _ops  # noqa: E402
from g3_cases import O, _build  # noqa: E402

BARS = {"f32": (5e-4, 1e-2), "bf16x3": (3e-3, 2e-2), "f16mx": (3e-3, 2e-2), "f16mx_hb": (3e-3, 2e-2), "f16mx_hd": (5e-3, 3e-2),
        "bf16": (2e-2, 5e-2)}


@pytest.mark.parametrize("mode", list(BARS))
def test_six_reference_train_steps_at_the_default_config(mode, golden_dir):
    from acoustic_locating_vq_vae.train_step import Trainer
    g = np.load(os.path.join(golden_dir, "g8_speech_steps.npz"))
    B = int(g["batch"])
    _ops.set_compute_dtype(mode)
    try:
        p = O.closed_form_params(O.vqvae_param_shapes(201, 1024, 128, 1024, 1024), float(g["cb_scale"]), float(g["gain"]))
        m = _build((201, 1024, 128, 3, 1024, 0.25, 1024), p).train()
        tr = Trainer(m, "speech")
        np.random.seed(9)
        rows = []
        for s in range(g["curve"].shape[0]):
            raw = torch.from_numpy(O.hashed_uniform(B * 201 * 500, 40 + s, 2.0).reshape(B, 201, 500)).cuda()
            loss, rec, perp = tr.step(raw)
            rows.append([float(loss), float(rec), float(perp)])
    finally:
        _ops.set_compute_dtype("f32")
    rows, ref = np.array(rows), g["curve"][:, [0, 1, 3]]
    dl = np.abs(rows[:, 0] - ref[:, 0]) / ref[:, 0]
    dr = np.abs(rows[:, 1] - ref[:, 1]) / ref[:, 1]
    dp = np.abs(rows[:, 2] - ref[:, 2]) / ref[:, 2]
    print("%s: loss %s recon %s perplexity %s" % (mode, np.round(dl, 6).tolist(), np.round(dr, 6).tolist(), np.round(dp, 5).tolist()))
    assert dl[0] < (1e-3 if mode == "bf16" else 1e-5)                  # step 0 is a pure forward
    assert dl.max() < BARS[mode][0] and dr.max() < BARS[mode][0] and dp.max() < BARS[mode][1]
