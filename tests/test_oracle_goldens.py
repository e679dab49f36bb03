"""T1: the CPU oracle restatement reproduces the goldens captured from the real
reference (tests/golden/make_goldens.py).  Runs anywhere, no GPU, no reference."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vqvae_oracle as O
from oracle import stft_oracle


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _np(a):
    return a.detach().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)


def rel(a, b):
    a, b = _np(a), _np(b)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def sl(t, n=64):
    f = t.detach().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy()


def test_g1_tiny_full_step(golden_dir):
    g = load(golden_dir, "g1_tiny_vqvae.npz")
    p = {k[len("param:"):]: torch.from_numpy(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith("param:")}
    x = O.speech_preprocess(torch.from_numpy(g["x_raw"]))
    assert rel(x, g["x"]) < 1e-6
    out = O.vqvae_forward(x, p, 2, 0.25, g["jitter_src"])
    assert np.array_equal(out["idx"].numpy(), g["idx"])
    assert rel(out["z"], g["z"]) < 1e-6
    assert rel(out["recon"], g["recon"]) < 1e-5
    assert rel(out["vq_loss"], g["vq_loss"]) < 1e-6
    assert rel(out["perplexity"], g["perplexity"]) < 1e-6
    recon_error = F.mse_loss(out["recon"], x)
    (recon_error + out["vq_loss"]).backward()
    keys = list(p)
    for k in keys:
        assert rel(p[k].grad, g["grad:" + k]) < 1e-5, k
    state = {"step": 0, "m": [torch.zeros_like(p[k]) for k in keys], "v": [torch.zeros_like(p[k]) for k in keys]}
    params = [p[k].detach().clone() for k in keys]
    O.adam_step(params, [p[k].grad for k in keys], state)
    for k, v in zip(keys, params):
        assert np.abs(v.numpy() - g["after:" + k]).max() < 2e-6, k
    enc = O.onehot(out["idx"], 16)
    assert np.array_equal(enc.numpy(), g["encodings"])


@pytest.mark.parametrize("regime,scale,seed", [("data", 1.7, 12), ("init", 1.0 / 1024, 13), ("ties", 1.7, 14)])
def test_g2_vq_regimes(golden_dir, regime, scale, seed):
    g = load(golden_dir, "g2_vq.npz")
    cb = torch.from_numpy(O.hashed_uniform(1024 * 128, seed, scale).reshape(1024, 128)).clone()
    if regime == "ties":
        cb[5] = cb[2]
        cb[7] = cb[2]
        cb[900] = cb[33]
    cb.requires_grad_(True)
    x = torch.from_numpy(g["x"]).clone().requires_grad_(True)
    assert np.array_equal(g["x"].ravel(), O.hashed_uniform(2000 * 128, 11, 1.7))
    loss, q_st, perp, idx = O.vector_quantizer(x, cb, 0.25)
    assert np.array_equal(idx.numpy().astype(np.int16), g[regime + ":idx"])
    if regime == "ties":
        assert not np.isin(idx.numpy(), [5, 7, 900]).any()      # lowest index wins
    gr = torch.from_numpy(O.hashed_uniform(x.numel(), 15, 1.0).reshape(x.shape))
    (loss + (q_st * gr).sum()).backward()
    assert rel(loss, g[regime + ":loss"]) < 1e-6
    assert rel(perp, g[regime + ":perplexity"]) < 1e-6
    assert rel(sl(q_st, 256), g[regime + ":q_st_slice"]) < 1e-6
    assert rel(sl(x.grad, 256), g[regime + ":dx_slice"]) < 1e-5
    assert rel(sl(cb.grad, 256), g[regime + ":dE_slice"]) < 1e-5


def test_g4_layout_quirks():
    # VQ rows are memory-order chunks (vector_quantizer.py:32): row 0 == z[0, 0, :D]
    z = torch.arange(2 * 4 * 6, dtype=torch.float32).view(2, 4, 6)
    assert torch.equal(z.reshape(-1, 4)[0], z[0, 0, :4])
    # ConvTranspose1d(k3,s1,p1) == conv1d with flipped+transposed weights
    torch.manual_seed(0)
    x = torch.randn(2, 5, 9)
    w = torch.randn(5, 3, 3)
    a = F.conv_transpose1d(x, w, None, padding=1)
    b = F.conv1d(x, w.flip(2).transpose(0, 1), None, padding=1)
    assert float((a - b).abs().max()) < 1e-5
    # shared-weight grads are the sum over the R uses; frozen codebook -> no dE
    w1 = torch.randn(4, 6, 3, requires_grad=True)
    w2 = torch.randn(6, 4, 1, requires_grad=True)
    h = torch.randn(2, 6, 7)
    O.residual_stack(h, w1, w2, 3).sum().backward()
    assert w1.grad.abs().sum() > 0
    cb = torch.randn(8, 4, requires_grad=True)
    loss, q, _, _ = O.vector_quantizer(torch.randn(2, 4, 6, requires_grad=True), cb, 0.25, train_vq=False)
    (loss + q.sum()).backward()
    assert cb.grad is None or float(cb.grad.abs().max()) == 0.0


def test_g5_jitter_stream(golden_dir):
    g = load(golden_dir, "g5_jitter.npz")
    for length in (13, 201, 500):
        for seed in (0, 1):
            np.random.seed(seed)
            src = O.jitter_source_index(length, 0.25)
            assert np.array_equal(src, g["L%d_s%d" % (length, seed)])
    # inverted probability (jitter.py:55): ~75 % of columns replaced at p=0.25
    np.random.seed(3)
    src = O.jitter_source_index(500, 0.25)
    frac = float((src != np.arange(500)).mean())
    assert 0.68 < frac < 0.82


def test_g6_stft_unpinned(golden_dir):
    g = load(golden_dir, "g6_stft_unpinned.npz")
    p64 = stft_oracle.stft_power(torch.from_numpy(g["wave"]).view(1, -1))
    assert rel(p64, g["power_f64"]) < 1e-10
    direct = stft_oracle.stft_power_direct(g["wave"].reshape(1, -1))
    assert rel(direct, g["power_f64"]) < 1e-9
    assert g["power_f32"].shape == (1, 201, 26)


@pytest.mark.parametrize("tag", ["speech", "rir", "speech_b16", "speech_b64", "rir_b32"])
def test_g3_default_configs(golden_dir, tag):
    """Default speech / RIR configs at B=2 (about 10 s of CPU); the speech config at B=16 and at the
    bench batch B=64 (8 000 / 32 000 codebook rows; a few seconds each on 8 cores)."""
    g = load(golden_dir, "g3_%s.npz" % tag)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    if tag.startswith("speech"):
        cfg, shape, permuted, oc, jit = (201, 1024, 128, 3, 1024, 0.25, 1024), (int(tag[-2:]) if tag[-2:].isdigit() else 2, 201, 500), False, None, True
    else:
        cfg, shape, permuted, oc, jit = (500, 1024, 64, 2, 64, 0.25, 1024), (32 if tag == "rir_b32" else 2, 201, 500), True, 1, False
    in_c, h, d, r, rh, beta, k = cfg
    p = O.closed_form_params(O.vqvae_param_shapes(in_c, h, d, rh, k, oc), float(g["cb_scale"]), float(g["gain"]))
    p = {key: v.requires_grad_(True) for key, v in p.items()}
    x = O.speech_preprocess(torch.from_numpy(O.hashed_uniform(int(np.prod(shape)), 21, 2.0).reshape(shape)))
    if permuted:
        x = x.permute(0, 2, 1)
    if oc is None:
        target = x
    else:
        tr = torch.from_numpy(O.hashed_uniform(shape[0] * x.shape[2], 22, 2.0).reshape(shape[0], x.shape[2]))
        target = O.standardise(tr).unsqueeze(1)
    np.random.seed(9)
    src = O.jitter_source_index(x.shape[2], 0.25) if jit else None
    out = O.vqvae_forward(x, p, r, beta, src)
    assert np.array_equal(out["idx"].numpy().astype(np.int16), g["idx"])
    err = F.mse_loss(out["recon"], target)
    (err + out["vq_loss"]).backward()
    assert rel(out["vq_loss"], g["vq_loss"]) < 1e-5
    assert rel(err, g["recon_error"]) < 1e-5
    assert rel(sl(out["recon"]), g["recon_slice"]) < 1e-4
    # round 3: 4096-element slices and the whole-tensor checksums the GPU tests are held to
    from g3_cases import sum_rel, wide
    assert rel(wide(out["z"], g["z_wide"]), g["z_wide"]) < 1e-5 and sum_rel(out["z"], g["z_sum"]) < 1e-6
    assert rel(wide(out["recon"], g["recon_wide"]), g["recon_wide"]) < 1e-4 and sum_rel(out["recon"], g["recon_sum"]) < 1e-6
    for key in p:
        assert rel(sl(p[key].grad), g["grad_slice:" + key]) < 1e-4, key
        assert rel(wide(p[key].grad, g["grad_wide:" + key]), g["grad_wide:" + key]) < 1e-4, key
        assert sum_rel(p[key].grad, g["grad_sum:" + key]) < 1e-5, key


def test_g7_location_oracle_reproduces_reference(golden_dir):
    """oracle.location_oracle vs the golden made by the real LocationModule: small config in full, the script's
    201 x 1024 config through outputs, loss, gradient slices / checksums and the touched columns of fc_1's gradient."""
    import warnings
    from oracle import location_oracle as LO
    g = load(golden_dir, "g7_location.npz")
    for tag in ("small", "full"):
        L, K, od, B = (int(v) for v in g[tag + ":cfg"])
        p = {k: v.requires_grad_(True) for k, v in
             LO.closed_form_location_params(LO.location_param_shapes(L, K, od), float(g[tag + ":gain"])).items()}
        idx = LO.hashed_indices(B, L, K, 31)
        if tag == "small":
            idx[1] = idx[0]
        theta = torch.from_numpy(O.hashed_uniform(B, 32, 3.0))
        loc = LO.location_forward(LO.onehot_codes(idx, K), p)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                 # the script's (B,) vs (B,1) target broadcast, kept on purpose
            loss = LO.location_loss(loc, theta if od == 1 else theta.view(B, 1).expand(B, od))
        loss.backward()
        assert rel(loc, g[tag + ":location"]) < 1e-6 and abs(float(loss) - float(g[tag + ":loss"])) < 1e-6 * abs(float(g[tag + ":loss"]))
        for k, v in p.items():
            if tag + ":grad:" + k in g.files:
                assert rel(v.grad, g[tag + ":grad:" + k]) < 1e-5, k
            else:
                n = 256 if k == "fc_1.weight" else 64
                assert rel(sl(v.grad, n), g[tag + ":grad_slice:" + k]) < 1e-5 or np.abs(g[tag + ":grad_slice:" + k]).max() == 0, k
        if tag == "full":
            cols = torch.from_numpy((np.arange(L)[None, :] * K + idx).reshape(-1))
            assert rel(p["fc_1.weight"].grad[7, cols], g["full:fc1_grad_row7_touched"]) < 1e-5
            assert int((p["fc_1.weight"].grad.abs().sum(dim=0) != 0).sum()) == int(g["full:fc1_grad_nonzero_cols"])


def test_g8_speech_train_steps(golden_dir):
    """The oracle's trainer against the six reference train steps of the default speech model (B = 16; ~15 s of CPU)."""
    g = load(golden_dir, "g8_speech_steps.npz")
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    B = int(g["batch"])
    p = O.closed_form_params(O.vqvae_param_shapes(201, 1024, 128, 1024, 1024), float(g["cb_scale"]), float(g["gain"]))
    ot = O.OracleTrainer(p, 3, 0.25, use_jitter=True)
    np.random.seed(9)
    for s in range(g["curve"].shape[0]):
        x = O.speech_preprocess(torch.from_numpy(O.hashed_uniform(B * 201 * 500, 40 + s, 2.0).reshape(B, 201, 500)))
        loss, rec, perp = ot.step(x)
        assert abs(loss - g["curve"][s, 0]) < 1e-5 * abs(g["curve"][s, 0]), (s, loss, g["curve"][s, 0])
        assert abs(perp - g["curve"][s, 3]) < 1e-4 * abs(g["curve"][s, 3]), (s, perp)


def test_g8_rir_train_steps(golden_dir):
    """The oracle's trainer against the four reference steps of the RIR loop (B = 8)."""
    g = load(golden_dir, "g8_rir_steps.npz")
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    B = int(g["batch"])
    p = O.closed_form_params(O.vqvae_param_shapes(500, 1024, 64, 64, 1024, 1), float(g["cb_scale"]), float(g["gain"]))
    ot = O.OracleTrainer(p, 2, 0.25, use_jitter=False)
    for s in range(g["curve"].shape[0]):
        raw = torch.from_numpy(O.hashed_uniform(B * 201 * 500, 50 + s, 2.0).reshape(B, 201, 500)).abs()
        wien = torch.from_numpy(O.hashed_uniform(B * 201, 60 + s, 2.0).reshape(B, 201))
        x, target = O.rir_preprocess(raw, wien)
        loss, rec, perp = ot.step(x, target)
        assert abs(loss - g["curve"][s, 0]) < 1e-5 * abs(g["curve"][s, 0]), (s, loss, g["curve"][s, 0])
        assert abs(perp - g["curve"][s, 3]) < 1e-4 * abs(g["curve"][s, 3]), (s, perp)
