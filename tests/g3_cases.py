"""Shared driver for the default-config goldens (G3-speech / G3-rir / G3-echoed, made by tests/golden/make_goldens.py
running the real reference): builds the model with the closed-form weights the golden was made with, runs one
forward + backward in the CURRENT compute dtype and returns the measured parity numbers.  Used by the per-mode
tests (tests/test_default_configs_modes_gpu.py) and, as the checker of its `north_star` block, by bench.py.

Every number is always computed -- nothing here is conditional on another comparison succeeding."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PKG = os.path.join(_ROOT, "acoustic_locating_vq-vae_amd")
for _q in (_ROOT, _PKG, os.path.join(_PKG, "src")):
    if _q not in sys.path:
        sys.path.insert(0, _q)

from oracle import vqvae_oracle as O  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(a):
    return torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()


def rel_max(a, b):
    a, b = _t(a), _t(b)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_l2(a, b):
    a, b = _t(a).flatten(), _t(b).flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def sl(t, n=64):
    f = t.detach().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].cpu().numpy()


def wide(t, ref):
    """The strided slice of ``t`` that matches the golden array ``ref`` (64 elements in round 1-2, thousands since round 3)."""
    return sl(t, int(np.asarray(ref).size))


def sum_rel(t, ref_sum):
    """Full-tensor checksums against the golden's fp64 (sum, sum|.|, sum .^2): the three relative errors' maximum, the
    plain sum being judged against sum|.| (it cancels)."""
    f = t.detach().double().flatten()
    got = (float(f.sum()), float(f.abs().sum()), float((f * f).sum()))
    s, a, q = (float(v) for v in ref_sum)
    return max(abs(got[0] - s) / (a + 1e-300), abs(got[1] - a) / (a + 1e-300), abs(got[2] - q) / (q + 1e-300))


def _expand(p, R):
    out = {}
    for k, v in p.items():
        if "_layers.0." in k:
            for r in range(R):
                out[k.replace("_layers.0.", "_layers.%d." % r)] = v
        else:
            out[k] = v
    return out


def _build(cfg, p, **kw):
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    m = ConvolutionalVQVAE(*cfg, **kw)
    m.load_state_dict(_expand(p, cfg[3]))
    return m.cuda()


def _grad_report(named_params, g, prefix=""):
    worst, key_w, l2s, sums, enc_worst = 0.0, None, [], 0.0, 0.0
    for key, pp in named_params:
        want = g["grad_wide:" + prefix + key]
        r = rel_max(wide(pp.grad, want), want)
        l2s.append(rel_l2(wide(pp.grad, want), want))
        sums = max(sums, sum_rel(pp.grad, g["grad_sum:" + prefix + key]))
        if r > worst:
            worst, key_w = r, prefix + key
        if key.startswith(("_encoder", "_pre_vq_conv")):
            enc_worst = max(enc_worst, r)
    return {"grad_rel_max": worst, "grad_rel_max_key": key_w, "grad_rel_l2_median": float(np.median(l2s)),
            "grad_rel_l2_max": float(np.max(l2s)), "grad_sum_rel_max": sums, "encoder_grad_rel_max": enc_worst}


def run_vqvae(tag, golden_dir=GOLDEN):
    """tag in {"speech", "speech_b16", "speech_b64", "rir", "rir_b32"}.  Returns the parity numbers of the current compute dtype against the golden."""
    g = np.load(os.path.join(golden_dir, "g3_%s.npz" % tag))
    if tag == "speech":
        cfg, shape, permuted, oc, jit = (201, 1024, 128, 3, 1024, 0.25, 1024), (2, 201, 500), False, None, True
    elif tag in ("speech_b16", "speech_b64"):
        cfg, shape, permuted, oc, jit = (201, 1024, 128, 3, 1024, 0.25, 1024), (int(tag[-2:]), 201, 500), False, None, True
    else:
        cfg, shape, permuted, oc, jit = (500, 1024, 64, 2, 64, 0.25, 1024), (32 if tag == "rir_b32" else 2, 201, 500), True, 1, False
    in_c, h, d, r, rh, beta, k = cfg
    p = O.closed_form_params(O.vqvae_param_shapes(in_c, h, d, rh, k, oc), float(g["cb_scale"]), float(g["gain"]))
    m = _build(cfg, p, use_jitter=jit, out_channels=oc).train()
    x = O.speech_preprocess(torch.from_numpy(O.hashed_uniform(int(np.prod(shape)), 21, 2.0).reshape(shape)))
    if permuted:
        x = x.permute(0, 2, 1)
    if oc is None:
        target = x
    else:
        tr = torch.from_numpy(O.hashed_uniform(shape[0] * x.shape[2], 22, 2.0).reshape(shape[0], x.shape[2]))
        target = O.standardise(tr).unsqueeze(1)
    xg = x.cuda()
    z = m._latent(xg)
    out = {"tag": tag, "z_rel_max": rel_max(wide(z, g["z_wide"]), g["z_wide"]), "z_rel_l2": rel_l2(wide(z, g["z_wide"]), g["z_wide"]),
           "z_sum_rel": sum_rel(z, g["z_sum"]), "slice_elems": int(g["z_wide"].size)}
    _, _, _, idx = m.get_latent_indices(xg)
    idx = idx.cpu().numpy().astype(np.int16)
    bad = np.nonzero(idx != g["idx"])[0]
    gap = (g["top2_val"][:, 1] - g["top2_val"][:, 0]) / np.abs(g["top2_val"][:, 0])
    out.update(idx_total=int(idx.size), idx_mismatches=int(bad.size), idx_agree=float(1.0 - bad.size / idx.size),
               mismatch_gap_max=float(gap[bad].max()) if bad.size else 0.0)
    np.random.seed(9)
    vq_loss, recon, perp = m(xg)
    err = F.mse_loss(recon, target.cuda())
    (err + vq_loss).backward()
    out.update(vq_loss_rel=rel_max(vq_loss, g["vq_loss"]), recon_error_rel=rel_max(err, g["recon_error"]),
               perplexity_rel=rel_max(perp, g["perplexity"]),
               recon_rel_max=rel_max(wide(recon, g["recon_wide"]), g["recon_wide"]),
               recon_rel_l2=rel_l2(wide(recon, g["recon_wide"]), g["recon_wide"]), recon_sum_rel=sum_rel(recon, g["recon_sum"]))
    out.update(_grad_report(list(m.named_parameters()), g))
    return out


def run_echoed(golden_dir=GOLDEN, tag="echoed"):
    from acoustic_locating_vq_vae.vq_vae.echoed_speech_model import EchoedSpeechReconModel
    g = np.load(os.path.join(golden_dir, "g3_%s.npz" % tag))
    gain = float(g["gain"])
    sp_p = O.closed_form_params(O.vqvae_param_shapes(201, 1024, 128, 1024, 1024), float(g["speech_cb_scale"]), gain)
    rir_p = O.closed_form_params(O.vqvae_param_shapes(500, 1024, 64, 64, 1024, 1), float(g["rir_cb_scale"]), gain)
    sp = _build((201, 1024, 128, 3, 1024, 0.25, 1024), sp_p)
    rir = _build((500, 1024, 64, 2, 64, 0.25, 1024), rir_p, use_jitter=False, out_channels=1)
    model = EchoedSpeechReconModel(rir, sp, 201, 1024, 2, 1024, True)
    dec_p = O.closed_form_params(O.decoder_param_shapes(192, 201, 1024, 1024), gain=gain)
    model._decoder.load_state_dict({k[len("_decoder."):]: v for k, v in _expand(dec_p, 2).items()})
    model = model.cuda().train()
    shape = (32 if tag == "echoed_b32" else 2, 201, 500)
    raw = torch.from_numpy(O.hashed_uniform(int(np.prod(shape)), 21, 2.0).reshape(shape)).abs()
    x = O.standardise(raw).cuda()
    np.random.seed(9)
    recon, sperp, rperp = model(x, x.permute(0, 2, 1))
    err = F.mse_loss(recon, x)
    err.backward()
    out = {"tag": tag, "recon_error_rel": rel_max(err, g["recon_error"]),
           "speech_perplexity_rel": rel_max(sperp, g["speech_perplexity"]),
           "rir_perplexity_rel": rel_max(rperp, g["rir_perplexity"]),
           "recon_rel_max": rel_max(wide(recon, g["recon_wide"]), g["recon_wide"]),
           "recon_rel_l2": rel_l2(wide(recon, g["recon_wide"]), g["recon_wide"]), "recon_sum_rel": sum_rel(recon, g["recon_sum"])}
    out.update(_grad_report(list(model._decoder.named_parameters()), g, "_decoder."))
    out["encoders_grad_free"] = all(p.grad is None for p in model.speech_model.parameters())
    return out


def run(tag, golden_dir=GOLDEN):
    return run_echoed(golden_dir, tag) if tag.startswith("echoed") else run_vqvae(tag, golden_dir)


if __name__ == "__main__":      # python tests/g3_cases.py [modes...]  -> one JSON line per (mode, config)
    import json
    from acoustic_locating_vq_vae import _ops
    for mode in (sys.argv[1:] or ["f32", "bf16x3", "f16mx", "f16mx_hb", "bf16"]):
        _ops.set_compute_dtype(mode)
        for tag in ("speech", "rir", "echoed"):
            print(json.dumps({"mode": mode, **run(tag)}), flush=True)
