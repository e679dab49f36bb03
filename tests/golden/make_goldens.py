"""Generate tests/golden/*.npz by running the REAL reference (build container only).

Run:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo:/root/reference:/root/reference/src \
      python3 /root/repo/tests/golden/make_goldens.py

The fixtures are data only (inputs + expected outputs); weights and inputs come
from ``oracle.vqvae_oracle.hashed_uniform`` (integer hash, bit-reproducible) so the
big configs need not store them.  G6 (STFT) comes from the torch.stft restatement
because torchaudio is absent -- it is marked parity-unpinned.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import vqvae_oracle as O
from oracle import stft_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def expand_aliases(p, num_layers):
    out = {}
    for k, v in p.items():
        if "_layers.0." in k:
            for r in range(num_layers):
                out[k.replace("_layers.0.", "_layers.%d." % r)] = v
        else:
            out[k] = v
    return out


def top2(flat, codebook):
    """(two smallest fp32 distances per row, argmin) -- argmin is torch.argmin
    (lowest index on ties, what the reference calls); topk gives values only."""
    d = O.vq_distances(flat, codebook)
    v, _ = torch.topk(d, 2, dim=1, largest=False)
    return v.numpy(), torch.argmin(d, dim=1).numpy()


def sl(t, n=64):
    f = t.detach().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def checksum(t):
    f = t.detach().double().flatten()
    return np.array([float(f.sum()), float(f.abs().sum()), float((f * f).sum())])


def g1_tiny():
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    cfg = (7, 16, 4, 2, 8, 0.25, 16)
    shapes = O.vqvae_param_shapes(7, 16, 4, 8, 16)
    p = O.closed_form_params(shapes, codebook_scale=0.8)
    model = ConvolutionalVQVAE(*cfg)
    model.load_state_dict(expand_aliases(p, 2))
    model.train()
    x_raw = torch.from_numpy(O.hashed_uniform(2 * 7 * 13, 77, 2.0).reshape(2, 7, 13))
    x = O.speech_preprocess(x_raw)
    np.random.seed(5)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, amsgrad=False)
    opt.zero_grad()
    z = model._pre_vq_conv(model._encoder(x))
    with torch.no_grad():
        _, q_st, _, enc = model.get_latent_representation(x)      # before the optimiser step
    np.random.seed(5)
    vq_loss, recon, perp = model(x)
    recon_error = F.mse_loss(recon, x)
    (recon_error + vq_loss).backward()
    grads = {k: pp.grad.clone() for k, pp in model.named_parameters()}
    opt.step()
    after = {k: pp.detach().clone() for k, pp in model.named_parameters()}
    np.random.seed(5)
    src = O.jitter_source_index(13, 0.25)
    out = {"x_raw": x_raw.numpy(), "x": x.numpy(), "z": z.detach().numpy(), "idx": enc.argmax(1).numpy().astype(np.int64),
           "q_st": q_st.numpy(), "vq_loss": vq_loss.detach().numpy(), "recon_error": recon_error.detach().numpy(),
           "perplexity": perp.detach().numpy(), "recon": recon.detach().numpy(), "jitter_src": src,
           "encodings": enc.numpy()}
    for k, v in p.items():
        out["param:" + k] = v.numpy()
        out["grad:" + k] = grads[k].numpy()
        out["after:" + k] = after[k].numpy()
    np.savez_compressed(os.path.join(HERE, "g1_tiny_vqvae.npz"), **out)
    print("g1", float(perp), float(vq_loss), float(recon_error))


def g2_vq():
    from acoustic_locating_vq_vae.vq_vae.vector_quantizer import VectorQuantizer
    out = {}
    n, k, d = 2000, 1024, 128
    x = torch.from_numpy(O.hashed_uniform(n * d, 11, 1.7).reshape(4, d, n // 4))  # (4,128,500): N = 2000 rows
    regimes = {
        "data": torch.from_numpy(O.hashed_uniform(k * d, 12, 1.7).reshape(k, d)),
        "init": torch.from_numpy(O.hashed_uniform(k * d, 13, 1.0 / k).reshape(k, d)),
    }
    ties = torch.from_numpy(O.hashed_uniform(k * d, 14, 1.7).reshape(k, d)).clone()
    ties[5] = ties[2]
    ties[7] = ties[2]
    ties[900] = ties[33]
    regimes["ties"] = ties
    out["x"] = x.numpy()
    for name, cb in regimes.items():
        vq = VectorQuantizer(k, d, 0.25)
        with torch.no_grad():
            vq._embedding.weight.copy_(cb)
        xin = x.clone().requires_grad_(True)
        loss, q_st, perp, enc = vq(xin)
        g = torch.from_numpy(O.hashed_uniform(x.numel(), 15, 1.0).reshape(x.shape))
        (loss + (q_st * g).sum()).backward()
        idx = enc.argmax(1)
        v, i = top2(x.reshape(-1, d), cb)
        assert (i == idx.numpy()).all()
        out[name + ":idx"] = idx.numpy().astype(np.int16)
        out[name + ":top2_val"] = v
        out[name + ":loss"] = loss.detach().numpy()
        out[name + ":perplexity"] = perp.detach().numpy()
        out[name + ":q_st_slice"] = sl(q_st, 256)
        out[name + ":q_st_sum"] = checksum(q_st)
        out[name + ":dx_slice"] = sl(xin.grad, 256)
        out[name + ":dx_sum"] = checksum(xin.grad)
        out[name + ":dE_slice"] = sl(vq._embedding.weight.grad, 256)
        out[name + ":dE_sum"] = checksum(vq._embedding.weight.grad)
        print("g2", name, float(perp), float(loss))
    # rows mapped to duplicate codes must pick the lowest index
    np.savez_compressed(os.path.join(HERE, "g2_vq.npz"), **out)


def big(tag, cfg, shape, permuted, out_channels, use_jitter, cb_scale):
    """cb_scale: codebook U(+-cb_scale), chosen ~1.5x std(z) so the argmin has data-scale margins."""
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    in_c, h, d, r, rh, beta, k = cfg
    shapes = O.vqvae_param_shapes(in_c, h, d, rh, k, out_channels)
    p = O.closed_form_params(shapes, codebook_scale=cb_scale, gain=GAIN)
    model = ConvolutionalVQVAE(*cfg, use_jitter=use_jitter, out_channels=out_channels)
    model.load_state_dict(expand_aliases(p, r))
    model.train()
    x_raw = torch.from_numpy(O.hashed_uniform(int(np.prod(shape)), 21, 2.0).reshape(shape))
    x = O.speech_preprocess(x_raw)
    if permuted:
        x = x.permute(0, 2, 1)
    if out_channels is None:
        target = x
    else:
        tr = torch.from_numpy(O.hashed_uniform(shape[0] * x.shape[2], 22, 2.0).reshape(shape[0], x.shape[2]))
        target = O.standardise(tr).unsqueeze(1)
    z = model._pre_vq_conv(model._encoder(x)).detach()
    np.random.seed(9)
    vq_loss, recon, perp = model(x)
    recon_error = F.mse_loss(recon, target)
    (recon_error + vq_loss).backward()
    cb = p["_vq._embedding.weight"]
    v, i = top2(z.reshape(-1, d), cb)
    out = {"idx": i.astype(np.int16), "top2_val": v,
           "vq_loss": vq_loss.detach().numpy(), "recon_error": recon_error.detach().numpy(),
           "perplexity": perp.detach().numpy(), "z_slice": sl(z), "z_sum": checksum(z),
           "recon_slice": sl(recon), "recon_sum": checksum(recon), "z_std": np.array(float(z.std())),
           "cb_scale": np.array(cb_scale), "gain": np.array(GAIN),
           # round 3: wide strided slices (thousands of elements) beside the 64-element ones; the *_sum checksums cover
           # the whole tensors
           "z_wide": sl(z, WIDE), "recon_wide": sl(recon, WIDE)}
    for key, pp in model.named_parameters():
        out["grad_slice:" + key] = sl(pp.grad)
        out["grad_wide:" + key] = sl(pp.grad, WIDE_GRAD)
        out["grad_sum:" + key] = checksum(pp.grad)
    np.savez_compressed(os.path.join(HERE, "g3_%s.npz" % tag), **out)
    gap = (v[:, 1] - v[:, 0]) / np.abs(v[:, 0])
    print("g3", tag, "perp", float(perp), "vq", float(vq_loss), "rec", float(recon_error), "zstd", float(z.std()),
          "min rel gap", gap.min(), "n<1e-4", int((gap < 1e-4).sum()))


def speech_b16():
    """Round 3: the speech config at a training-like batch (8000 codebook rows).  At B = 2 a flipped ReLU gate is one of 1000
    terms of a weight-gradient element; this golden shows the split modes' gradient agreement where the batch averages the
    flips down (tests/analysis/gate_flips.py predicts ~1/sqrt(B)) -- and eight times as many indices that must all be bit-exact."""
    big("speech_b16", (201, 1024, 128, 3, 1024, 0.25, 1024), (16, 201, 500), False, None, True, SPEECH_CB)


def speech_b64():
    """Round 3: the bench workload itself -- BASELINE configs[1], B = 64, 32 000 codebook rows."""
    big("speech_b64", (201, 1024, 128, 3, 1024, 0.25, 1024), (64, 201, 500), False, None, True, SPEECH_CB)


def speech_steps():
    """Round 3: the reference's own loop body (train_speech.py:62-74, 88-91) on the default speech model at B = 16 for six
    steps -- |x|, per-frame standardise, model(x) with the jitter stream running on from np.random.seed(9), mse + vq loss,
    backward, torch.optim.Adam(lr=1e-3) -- fresh hashed batches per step.  Stored: per-step losses / perplexities and the
    parameter checksums afterwards."""
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    cfg = (201, 1024, 128, 3, 1024, 0.25, 1024)
    p = O.closed_form_params(O.vqvae_param_shapes(201, 1024, 128, 1024, 1024), codebook_scale=SPEECH_CB, gain=GAIN)
    model = ConvolutionalVQVAE(*cfg)
    model.load_state_dict(expand_aliases(p, 3))
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, amsgrad=False)
    np.random.seed(9)
    shape, steps = (16, 201, 500), 6
    rows = []
    for s in range(steps):
        x = O.speech_preprocess(torch.from_numpy(O.hashed_uniform(int(np.prod(shape)), 40 + s, 2.0).reshape(shape)))
        opt.zero_grad()
        vq_loss, recon, perp = model(x)
        recon_error = F.mse_loss(recon, x)
        (recon_error + vq_loss).backward()
        opt.step()
        rows.append([float(recon_error + vq_loss), float(recon_error), float(vq_loss), float(perp)])
        print("steps", s, rows[-1])
    out = {"curve": np.array(rows), "cb_scale": np.array(SPEECH_CB), "gain": np.array(GAIN), "batch": np.array(16)}
    for key, pp in model.named_parameters():
        out["after_sum:" + key] = checksum(pp)
        out["after_wide:" + key] = sl(pp, WIDE_GRAD)
    np.savez_compressed(os.path.join(HERE, "g8_speech_steps.npz"), **out)


def rir_steps():
    """Round 3: the loop body of train_rir.py:42-58, 72-75 on the default RIR model (B = 8, four steps): frames standardised
    over the frequency axis, permuted input, Wiener target standardised and unsqueezed, mse + vq loss, Adam."""
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    cfg = (500, 1024, 64, 2, 64, 0.25, 1024)
    p = O.closed_form_params(O.vqvae_param_shapes(500, 1024, 64, 64, 1024, 1), codebook_scale=RIR_CB, gain=GAIN)
    model = ConvolutionalVQVAE(*cfg, use_jitter=False, out_channels=1)
    model.load_state_dict(expand_aliases(p, 2))
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, amsgrad=False)
    B, rows = 8, []
    for s in range(4):
        raw = torch.from_numpy(O.hashed_uniform(B * 201 * 500, 50 + s, 2.0).reshape(B, 201, 500)).abs()
        wien = torch.from_numpy(O.hashed_uniform(B * 201, 60 + s, 2.0).reshape(B, 201))
        x, target = O.rir_preprocess(raw, wien)
        opt.zero_grad()
        vq_loss, recon, perp = model(x)
        recon_error = F.mse_loss(recon, target)
        (recon_error + vq_loss).backward()
        opt.step()
        rows.append([float(recon_error + vq_loss), float(recon_error), float(vq_loss), float(perp)])
        print("rir steps", s, rows[-1])
    np.savez_compressed(os.path.join(HERE, "g8_rir_steps.npz"), curve=np.array(rows), cb_scale=np.array(RIR_CB),
                        gain=np.array(GAIN), batch=np.array(B))


def echoed_steps():
    """Round 3: the loop body of train_echoed_speech.py:62-75, 89-92 (B = 4, four steps): two frozen encoders built from the
    closed-form fills, the decoder trained on the reconstruction error alone, Adam handed every parameter."""
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    from acoustic_locating_vq_vae.vq_vae.echoed_speech_model import EchoedSpeechReconModel
    sp_p = O.closed_form_params(O.vqvae_param_shapes(201, 1024, 128, 1024, 1024), codebook_scale=SPEECH_CB, gain=GAIN)
    rir_p = O.closed_form_params(O.vqvae_param_shapes(500, 1024, 64, 64, 1024, 1), codebook_scale=RIR_CB, gain=GAIN)
    sp = ConvolutionalVQVAE(201, 1024, 128, 3, 1024, 0.25, 1024)
    sp.load_state_dict(expand_aliases(sp_p, 3))
    rir = ConvolutionalVQVAE(500, 1024, 64, 2, 64, 0.25, 1024, use_jitter=False, out_channels=1)
    rir.load_state_dict(expand_aliases(rir_p, 2))
    model = EchoedSpeechReconModel(rir, sp, 201, 1024, 2, 1024, True)
    dec_p = O.closed_form_params(O.decoder_param_shapes(192, 201, 1024, 1024), gain=GAIN)
    model._decoder.load_state_dict({k[len("_decoder."):]: v for k, v in expand_aliases(dec_p, 2).items()})
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, amsgrad=False)
    np.random.seed(9)
    B, rows = 4, []
    for s in range(4):
        raw = torch.from_numpy(O.hashed_uniform(B * 201 * 500, 70 + s, 2.0).reshape(B, 201, 500)).abs()
        x = O.standardise(raw)
        opt.zero_grad()
        recon, sperp, rperp = model(x, x.permute(0, 2, 1))
        err = F.mse_loss(recon, x)
        err.backward()
        opt.step()
        rows.append([float(err), float(sperp), float(rperp)])
        print("echoed steps", s, rows[-1])
    np.savez_compressed(os.path.join(HERE, "g8_echoed_steps.npz"), curve=np.array(rows), speech_cb_scale=np.array(SPEECH_CB),
                        rir_cb_scale=np.array(RIR_CB), gain=np.array(GAIN), batch=np.array(B))


def rir_b32():
    """Round 3: BASELINE configs[2]'s per-GPU share (B = 256 over 8 GPUs), 6 432 codebook rows."""
    big("rir_b32", (500, 1024, 64, 2, 64, 0.25, 1024), (32, 201, 500), True, 1, False, RIR_CB)


def echoed_b32():
    """Round 3: BASELINE configs[4]'s per-GPU share (B = 128 over 4 GPUs)."""
    g3_echoed(32, "echoed_b32")


def g3_echoed(batch=2, tag="echoed"):
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    from acoustic_locating_vq_vae.vq_vae.echoed_speech_model import EchoedSpeechReconModel
    sp_cfg = (201, 1024, 128, 3, 1024, 0.25, 1024)
    rir_cfg = (500, 1024, 64, 2, 64, 0.25, 1024)
    sp_p = O.closed_form_params(O.vqvae_param_shapes(201, 1024, 128, 1024, 1024), codebook_scale=SPEECH_CB, gain=GAIN)
    rir_p = O.closed_form_params(O.vqvae_param_shapes(500, 1024, 64, 64, 1024, 1), codebook_scale=RIR_CB, gain=GAIN)
    sp = ConvolutionalVQVAE(*sp_cfg)
    sp.load_state_dict(expand_aliases(sp_p, 3))
    rir = ConvolutionalVQVAE(*rir_cfg, use_jitter=False, out_channels=1)
    rir.load_state_dict(expand_aliases(rir_p, 2))
    model = EchoedSpeechReconModel(rir, sp, 201, 1024, 2, 1024, True)
    dec_p = O.closed_form_params(O.decoder_param_shapes(192, 201, 1024, 1024), gain=GAIN)
    model._decoder.load_state_dict({k[len("_decoder."):]: v for k, v in expand_aliases(dec_p, 2).items()})
    model.train()
    shape = (batch, 201, 500)
    x = O.standardise(torch.from_numpy(O.hashed_uniform(int(np.prod(shape)), 21, 2.0).reshape(shape)).abs())
    np.random.seed(9)
    recon, sperp, rperp = model(x, x.permute(0, 2, 1))
    err = F.mse_loss(recon, x)
    err.backward()
    out = {"recon_error": err.detach().numpy(), "speech_perplexity": sperp.detach().numpy(),
           "rir_perplexity": rperp.detach().numpy(), "recon_slice": sl(recon), "recon_sum": checksum(recon),
           "speech_cb_scale": np.array(SPEECH_CB), "rir_cb_scale": np.array(RIR_CB), "gain": np.array(GAIN),
           "recon_wide": sl(recon, WIDE)}
    for key, pp in model._decoder.named_parameters():
        out["grad_slice:_decoder." + key] = sl(pp.grad)
        out["grad_wide:_decoder." + key] = sl(pp.grad, WIDE_GRAD)
        out["grad_sum:_decoder." + key] = checksum(pp.grad)
    np.savez_compressed(os.path.join(HERE, "g3_%s.npz" % tag), **out)
    print("g3", tag, float(err), float(sperp), float(rperp))


def g5_jitter():
    from acoustic_locating_vq_vae.vq_vae.modules.jitter import Jitter
    out = {}
    for length in (13, 201, 500):
        for seed in (0, 1):
            q = torch.arange(length, dtype=torch.float32).view(1, 1, length).clone()
            np.random.seed(seed)
            res = Jitter(0.25)(q)
            out["L%d_s%d" % (length, seed)] = res.view(-1).numpy().astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "g5_jitter.npz"), **out)
    print("g5 ok")


def g6_stft():
    t = np.arange(4000, dtype=np.float64) / 16000.0
    chirp = np.sin(2 * np.pi * (200.0 * t + 0.5 * 12000.0 * t * t)) * (0.6 + 0.4 * np.cos(2 * np.pi * 3 * t))
    wave32 = torch.from_numpy(chirp.astype(np.float32)).view(1, -1)
    p32 = stft_oracle.stft_power(wave32)
    p64 = stft_oracle.stft_power(torch.from_numpy(chirp).view(1, -1))
    direct = stft_oracle.stft_power_direct(chirp.reshape(1, -1))
    assert np.allclose(direct, p64.numpy(), rtol=1e-9, atol=1e-12)
    np.savez_compressed(os.path.join(HERE, "g6_stft_unpinned.npz"), wave=chirp, power_f32=p32.numpy(), power_f64=p64.numpy())
    print("g6", p32.shape)


GAIN = 0.5
WIDE = 4096          # elements of the wide z / recon slices of the default-config goldens
WIDE_GRAD = 2048     # ... and of each parameter gradient
SPEECH_CB = 1.0
RIR_CB = 1.0

def g7_location():
    """LocationModule (vq_vae/location_model/location_model.py) at a small size (all tensors) and at the script's size
    (scripts/train_location.py:23-24,41: 201 x 1024 -> 1; slices + checksums, fc_1 alone is 843 MB)."""
    from acoustic_locating_vq_vae.vq_vae.location_model.location_model import LocationModule
    from oracle import location_oracle as LO
    out = {}
    for tag, (L, K, od, B) in (("small", (5, 8, 3, 4)), ("full", (201, 1024, 1, 16))):
        gain = 1.0 if tag == "small" else 12.0
        p = LO.closed_form_location_params(LO.location_param_shapes(L, K, od), gain)
        m = LocationModule(L, K, od)
        m.load_state_dict(p)
        idx = LO.hashed_indices(B, L, K, 31)
        if tag == "small":
            idx[1] = idx[0]                                  # two samples hitting the same columns: the scatter-add case
        theta = torch.from_numpy(O.hashed_uniform(B, 32, 3.0))
        loc = m(LO.onehot_codes(idx, K))
        loss = LO.location_loss(loc, theta if od == 1 else theta.view(B, 1).expand(B, od))
        loss.backward()
        out[tag + ":cfg"] = np.array([L, K, od, B])
        out[tag + ":gain"] = np.float64(gain)
        out[tag + ":location"] = loc.detach().numpy()
        out[tag + ":loss"] = np.float64(loss.item())
        for k, v in m.named_parameters():
            if tag == "small" and v.grad.numel() <= 50000:
                out[tag + ":grad:" + k] = v.grad.numpy()
            else:
                out[tag + ":grad_slice:" + k] = sl(v.grad, 256 if k == "fc_1.weight" else 64)
                out[tag + ":grad_sum:" + k] = checksum(v.grad)
        if tag == "full":
            # fc_1.weight.grad is zero except in the B*L touched columns: store those exactly (one row of the weight)
            g1 = m.fc_1.weight.grad
            cols = (np.arange(L)[None, :] * K + idx).reshape(-1)
            out["full:fc1_grad_row7_touched"] = g1[7, torch.from_numpy(cols)].numpy()
            out["full:fc1_grad_nonzero_cols"] = np.int64(int((g1.abs().sum(dim=0) != 0).sum()))
    np.savez_compressed(os.path.join(HERE, "g7_location.npz"), **out)
    print("g7_location", {k: getattr(v, "shape", None) for k, v in out.items() if "grad" not in k})


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:                      # e.g. `make_goldens.py g7_location`: regenerate a single fixture
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g1_tiny()
    g2_vq()
    big("speech", (201, 1024, 128, 3, 1024, 0.25, 1024), (2, 201, 500), False, None, True, SPEECH_CB)
    big("rir", (500, 1024, 64, 2, 64, 0.25, 1024), (2, 201, 500), True, 1, False, RIR_CB)
    speech_b16()
    speech_b64()
    rir_b32()
    echoed_b32()
    speech_steps()
    rir_steps()
    echoed_steps()
    g3_echoed()
    g5_jitter()
    g6_stft()
    g7_location()
