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

# the user-selectable modes (round 4: f16mx / bf16x3 / f16mx_hd are internal engines; their forwards live on inside the _hb modes)
BARS = {"f32": (5e-4, 1e-2), "x3mx_hb": (3e-3, 2e-2), "bf16x3_hb": (3e-3, 2e-2), "f16mx_hb": (3e-3, 2e-2), "bf16": (2e-2, 5e-2)}


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


# (loss after the first update, loss over all four steps, perplexity over all four steps)
RIR_BARS = {"f32": (1.5e-3, 6e-3, 0.1), "x3mx_hb": (3e-3, 1e-2, 0.1), "bf16x3_hb": (3e-3, 1e-2, 0.1), "f16mx_hb": (3e-3, 1e-2, 0.1),
            "bf16": (0.15, 0.3, 0.3)}


@pytest.mark.parametrize("mode", list(RIR_BARS))
def test_four_reference_train_steps_of_the_rir_loop(mode, golden_dir):
    """train_rir.py:42-58, 72-75 on the default RIR model (B = 8), recorded from the real reference: permuted input, Wiener
    target, mse + vq loss, Adam.  The trajectory is violent here too (loss 1.70 -> 114.9 -> 3.6 -> 5.7), and past the second
    step every arithmetic is equally far from the reference -- measured, loss / perplexity over the four steps: f32 1.8e-3 /
    3.2e-2, bf16x3 2.7e-3 / 2.0e-2, f16mx 1.3e-3 / 2.2e-2, f16mx_hb 6.3e-4 / 8.4e-3, bf16 0.11 / 0.11 -- so the sharp bars are
    on the forward (step 0) and on the loss right after the first update (f32 3.9e-4, split modes 1e-3, bf16 4.3e-2)."""
    from acoustic_locating_vq_vae.train_step import Trainer
    g = np.load(os.path.join(golden_dir, "g8_rir_steps.npz"))
    B = int(g["batch"])
    _ops.set_compute_dtype(mode)
    try:
        p = O.closed_form_params(O.vqvae_param_shapes(500, 1024, 64, 64, 1024, 1), float(g["cb_scale"]), float(g["gain"]))
        m = _build((500, 1024, 64, 2, 64, 0.25, 1024), p, use_jitter=False, out_channels=1).train()
        tr = Trainer(m, "rir")
        rows = []
        for s in range(g["curve"].shape[0]):
            raw = torch.from_numpy(O.hashed_uniform(B * 201 * 500, 50 + s, 2.0).reshape(B, 201, 500)).abs().cuda()
            wien = torch.from_numpy(O.hashed_uniform(B * 201, 60 + s, 2.0).reshape(B, 201)).cuda()
            loss, rec, perp = tr.step(raw, wien)
            rows.append([float(loss), float(rec), float(perp)])
    finally:
        _ops.set_compute_dtype("f32")
    rows, ref = np.array(rows), g["curve"][:, [0, 1, 3]]
    dl = np.abs(rows[:, 0] - ref[:, 0]) / ref[:, 0]
    dp = np.abs(rows[:, 2] - ref[:, 2]) / ref[:, 2]
    print("rir %s: loss %s perplexity %s" % (mode, np.round(dl, 6).tolist(), np.round(dp, 5).tolist()))
    assert dl[0] < (1e-3 if mode == "bf16" else 1e-5)
    assert dl[1] < RIR_BARS[mode][0] and dl.max() < RIR_BARS[mode][1] and dp.max() < RIR_BARS[mode][2]


ECHOED_BARS = {"f32": (1e-4, 1e-5), "x3mx_hb": (1e-3, 1e-5), "bf16x3_hb": (1e-3, 1e-5), "f16mx_hb": (1e-3, 1e-5), "bf16": (2e-2, 2e-2)}


@pytest.mark.parametrize("mode", list(ECHOED_BARS))
def test_four_reference_train_steps_of_the_echoed_loop(mode, golden_dir):
    """train_echoed_speech.py:62-75, 89-92 (B = 4), recorded from the real reference: both encoders frozen, the decoder trained
    on the reconstruction error alone, Adam handed every parameter.  The perplexities are pure forward quantities of the
    frozen encoders on each step's batch: they must match to rounding in every parity mode."""
    from acoustic_locating_vq_vae.train_step import Trainer
    from acoustic_locating_vq_vae.vq_vae.echoed_speech_model import EchoedSpeechReconModel
    from g3_cases import _expand
    g = np.load(os.path.join(golden_dir, "g8_echoed_steps.npz"))
    B, gain = int(g["batch"]), float(g["gain"])
    _ops.set_compute_dtype(mode)
    try:
        sp_p = O.closed_form_params(O.vqvae_param_shapes(201, 1024, 128, 1024, 1024), float(g["speech_cb_scale"]), gain)
        rir_p = O.closed_form_params(O.vqvae_param_shapes(500, 1024, 64, 64, 1024, 1), float(g["rir_cb_scale"]), gain)
        sp = _build((201, 1024, 128, 3, 1024, 0.25, 1024), sp_p)
        rir = _build((500, 1024, 64, 2, 64, 0.25, 1024), rir_p, use_jitter=False, out_channels=1)
        model = EchoedSpeechReconModel(rir, sp, 201, 1024, 2, 1024, True)
        dec_p = O.closed_form_params(O.decoder_param_shapes(192, 201, 1024, 1024), gain=gain)
        model._decoder.load_state_dict({k[len("_decoder."):]: v for k, v in _expand(dec_p, 2).items()})
        tr = Trainer(model.cuda().train(), "echoed")
        np.random.seed(9)
        rows = []
        for s in range(g["curve"].shape[0]):
            raw = torch.from_numpy(O.hashed_uniform(B * 201 * 500, 70 + s, 2.0).reshape(B, 201, 500)).abs().cuda()
            loss, rec, sperp = tr.step(raw)
            rows.append([float(rec), float(sperp)])
    finally:
        _ops.set_compute_dtype("f32")
    rows, ref = np.array(rows), g["curve"][:, [0, 1]]
    dl = np.abs(rows[:, 0] - ref[:, 0]) / ref[:, 0]
    dp = np.abs(rows[:, 1] - ref[:, 1]) / ref[:, 1]
    print("echoed %s: recon error %s speech perplexity %s" % (mode, np.round(dl, 7).tolist(), np.round(dp, 7).tolist()))
    assert dl.max() < ECHOED_BARS[mode][0] and dp.max() < ECHOED_BARS[mode][1]
