"""T0: pin the oracle restatement against the real reference (build container only).

TEST INFRASTRUCTURE.  Imports the reference read-only from /root/reference
(PYTHONPATH must hold both /root/reference and /root/reference/src -- SURVEY 8c)
and checks that ``oracle.vqvae_oracle`` reproduces it: outputs, indices, every
parameter gradient, one Adam step, jitter index stream, echoed model, and ``oracle.location_oracle``
against the real LocationModule.

Run:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo:/root/reference:/root/reference/src \
      python3 /root/repo/oracle/check_against_reference.py
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

from oracle import vqvae_oracle as O


def _ref():
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    from acoustic_locating_vq_vae.vq_vae.echoed_speech_model import EchoedSpeechReconModel
    from acoustic_locating_vq_vae.vq_vae.modules.jitter import Jitter
    return ConvolutionalVQVAE, EchoedSpeechReconModel, Jitter


def unique_params(model):
    """state_dict restricted to layer-0 aliases (the 17 unique tensors)."""
    return {k: v for k, v in model.state_dict().items() if "_layers." not in k or "_layers.0." in k}


def maxdiff(a, b):
    return float((a.detach() - b.detach()).abs().max())


def check_vqvae(cfg, shape, use_jitter, seed, out_channels=None, permuted=False, tol=0.0):
    ConvolutionalVQVAE, _, _ = _ref()
    torch.manual_seed(seed)
    model = ConvolutionalVQVAE(*cfg, use_jitter=use_jitter, out_channels=out_channels)
    # data-scale codebook so the argmin has wide margins
    with torch.no_grad():
        model._vq._embedding.weight.normal_(0, 0.5)
    model.train()
    x = torch.randn(*shape)
    if permuted:
        x = x.permute(0, 2, 1)
    x = O.standardise(x.abs())
    target = x if out_channels is None else torch.randn(shape[0], out_channels, x.shape[2])

    np.random.seed(seed)
    vq_loss, recon, perp = model(x)
    loss = F.mse_loss(recon, target) + vq_loss
    loss.backward()
    ref_grads = {k: p.grad.clone() for k, p in model.named_parameters()}

    p = {k: v.clone().requires_grad_(True) for k, v in unique_params(model).items()}
    np.random.seed(seed)
    src = O.jitter_source_index(x.shape[2], 0.25) if use_jitter else None
    out = O.vqvae_forward(x, p, cfg[3], cfg[5], src)
    oloss = F.mse_loss(out["recon"], target) + out["vq_loss"]
    oloss.backward()

    with torch.no_grad():
        _, _, _, enc = model.eval().get_latent_representation(x)
    ref_idx = enc.argmax(1)
    res = {
        "recon": maxdiff(recon, out["recon"]), "vq_loss": maxdiff(vq_loss, out["vq_loss"]),
        "perplexity": maxdiff(perp, out["perplexity"]),
        "idx_mismatch": int((ref_idx != out["idx"]).sum()),
    }
    gmax = 0.0
    for k, g in ref_grads.items():
        if k in p:
            denom = float(g.abs().max()) + 1e-30
            gmax = max(gmax, maxdiff(g, p[k].grad) / denom)
    res["grad_rel"] = gmax
    ok = res["idx_mismatch"] == 0 and res["recon"] <= tol and res["grad_rel"] <= max(tol, 1e-5)
    print(("OK  " if ok else "FAIL"), cfg, shape, res)
    return ok


def check_jitter():
    _, _, Jitter = _ref()
    ok = True
    for length in (13, 201, 500):
        q = torch.arange(2 * 3 * length, dtype=torch.float32).view(2, 3, length)
        np.random.seed(7)
        ref = Jitter(0.25)(q.clone())
        np.random.seed(7)
        src = O.jitter_source_index(length, 0.25)
        ok &= bool(torch.equal(ref, O.jitter(q, src)))
    print("OK  " if ok else "FAIL", "jitter stream")
    return ok


def check_echoed():
    ConvolutionalVQVAE, Echoed, _ = _ref()
    torch.manual_seed(3)
    rir = ConvolutionalVQVAE(20, 16, 4, 2, 8, 0.25, 16, use_jitter=False, out_channels=1)
    sp = ConvolutionalVQVAE(9, 16, 6, 3, 16, 0.25, 32)
    for m in (rir, sp):
        with torch.no_grad():
            m._vq._embedding.weight.normal_(0, 0.5)
    model = Echoed(rir, sp, 9, 16, 2, 16, True)
    model.train()
    x = O.standardise(torch.randn(2, 9, 20).abs())
    np.random.seed(11)
    recon, sperp, rperp = model(x, x.permute(0, 2, 1))
    loss = F.mse_loss(recon, x)
    loss.backward()
    p = {k: v.clone().requires_grad_(True) for k, v in unique_params(model).items()}
    np.random.seed(11)
    src = O.jitter_source_index(20, 0.25)
    out = O.echoed_forward(x, x.permute(0, 2, 1), p, 3, 2, 2, 0.25, src)
    F.mse_loss(out["recon"], x).backward()
    d = maxdiff(recon, out["recon"])
    g = max(maxdiff(pp.grad, p[k].grad) for k, pp in model.named_parameters() if pp.grad is not None and k in p)
    none_ok = all((p[k].grad is None or float(p[k].grad.abs().max()) == 0) for k in p if not k.startswith("_decoder"))
    ok = d == 0.0 and g < 1e-6 and none_ok and maxdiff(sperp, out["speech_perplexity"]) == 0
    print("OK  " if ok else "FAIL", "echoed", d, g, none_ok)
    return ok


def check_adam():
    torch.manual_seed(0)
    ps = [torch.randn(5, 3), torch.randn(7)]
    gs = [[torch.randn(5, 3), torch.randn(7)] for _ in range(3)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.Adam(ref, lr=1e-3, amsgrad=False)
    mine = [p.clone() for p in ps]
    st = {"step": 0, "m": [torch.zeros_like(p) for p in ps], "v": [torch.zeros_like(p) for p in ps]}
    for g in gs:
        for r, gg in zip(ref, g):
            r.grad = gg.clone()
        opt.step()
        O.adam_step(mine, g, st)
    d = max(maxdiff(a, b) for a, b in zip(ref, mine))
    print("OK  " if d < 1e-7 else "FAIL", "adam", d)
    return d < 1e-7


def check_location():
    """oracle.location_oracle vs the real LocationModule (location_model.py:5-29) and the script's loss
    (train_location.py:77-78, including its (B,) vs (B,1) broadcast)."""
    from acoustic_locating_vq_vae.vq_vae.location_model.location_model import LocationModule
    from oracle import location_oracle as LO
    ok = True
    for L, K, od, B in ((5, 8, 3, 4), (13, 32, 1, 6)):
        torch.manual_seed(11)
        m = LocationModule(L, K, od)
        assert list(m.state_dict()) == list(LO.location_param_shapes(L, K, od)), "state_dict keys / order"
        assert all(tuple(v.shape) == LO.location_param_shapes(L, K, od)[k] for k, v in m.state_dict().items())
        p = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        idx = LO.hashed_indices(B, L, K, 5)
        x = LO.onehot_codes(idx, K)
        theta = torch.from_numpy(O.hashed_uniform(B, 6, 3.0))
        tgt = theta if od == 1 else theta.view(B, 1).expand(B, od)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = m(x)
            F.mse_loss(ref, tgt / torch.pi, reduction="mean").backward()
            got = LO.location_forward(x, p)
            LO.location_loss(got, tgt).backward()
        d = maxdiff(ref, got)
        gd = max(maxdiff(q.grad, p[k].grad) for k, q in m.named_parameters())
        print("location L=%d K=%d out=%d: forward diff %.1e, grad diff %.1e" % (L, K, od, d, gd))
        ok &= d == 0.0 and gd < 1e-7
    return ok


def main():
    ok = True
    ok &= check_vqvae((7, 16, 4, 2, 8, 0.25, 16), (2, 7, 13), True, 0)
    ok &= check_vqvae((7, 16, 4, 2, 8, 0.25, 16), (2, 7, 13), False, 1)
    ok &= check_vqvae((20, 32, 8, 2, 8, 0.25, 16), (3, 13, 20), False, 2, out_channels=1, permuted=True)
    ok &= check_vqvae((201, 1024, 128, 3, 1024, 0.25, 1024), (1, 201, 500), True, 3, tol=2e-5)
    ok &= check_vqvae((500, 1024, 64, 2, 64, 0.25, 1024), (2, 201, 500), False, 4, out_channels=1, permuted=True, tol=2e-5)
    ok &= check_jitter()
    ok &= check_echoed()
    ok &= check_adam()
    ok &= check_location()
    print("ALL OK" if ok else "SOME FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
