"""bf16 compute mode at the module level: same API, bf16 storage/MFMA between convs.  Judged against the fp32
CPU oracle at bf16 tolerances (relative L2), and against the HIP fp32 mode on the same weights."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import vqvae_oracle as O  # noqa: E402


@pytest.fixture(autouse=True)
def _bf16_mode():
    from acoustic_locating_vq_vae import _ops
    _ops.set_compute_dtype("bf16")
    yield
    _ops.set_compute_dtype("f32")


def l2(a, b):
    a = a.detach().double().cpu().flatten()
    b = b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def build(cfg, **kw):
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    return ConvolutionalVQVAE(*cfg, **kw).cuda()


def oracle_params(m):
    return {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()
            if "_layers." not in k or "_layers.0." in k}


CASES = [
    ((7, 16, 4, 2, 8, 0.25, 16), (2, 7, 13), dict(), False),
    ((20, 48, 8, 3, 24, 0.25, 64), (3, 20, 40), dict(use_jitter=False), False),
    ((50, 64, 8, 2, 16, 0.25, 64), (4, 24, 50), dict(use_jitter=False, out_channels=1), True),
    ((201, 128, 32, 2, 128, 0.25, 128), (2, 201, 96), dict(), False),
]


@pytest.mark.parametrize("cfg,shape,kw,permuted", CASES)
def test_bf16_forward_backward_close_to_fp32_oracle(cfg, shape, kw, permuted):
    torch.manual_seed(21)
    m = build(cfg, **kw)
    with torch.no_grad():
        m._vq._embedding.weight.normal_(0, 0.7)
    m.train()
    p = oracle_params(m)
    x = O.standardise(torch.randn(*shape).abs())
    if permuted:
        x = x.permute(0, 2, 1)
    oc = kw.get("out_channels")
    target = x if oc is None else torch.randn(shape[0], oc, x.shape[2])
    np.random.seed(3)
    src = O.jitter_source_index(x.shape[2], 0.25) if kw.get("use_jitter", True) else None
    out = O.vqvae_forward(x, p, cfg[3], cfg[5], src)
    (F.mse_loss(out["recon"], target) + out["vq_loss"]).backward()

    np.random.seed(3)
    vq_loss, recon, perp = m(x.cuda())
    (F.mse_loss(recon, target.cuda()) + vq_loss).backward()
    z = m._latent(x.cuda())
    assert l2(z, out["z"]) < 2e-2
    _, _, _, idx = m.eval().get_latent_indices(x.cuda())
    agree = float((idx.cpu() == out["idx"]).float().mean())
    assert agree > 0.9, agree
    # Downstream of the quantiser the comparison must not depend on which codes flipped: feed the ORACLE's quantised
    # latent to the decoder (same jitter columns) and compare its output and its parameter gradients -- always runs.
    m.train()
    m.zero_grad()
    np.random.seed(3)
    recon2 = m._decoder(out["q_st"].detach().cuda())
    F.mse_loss(recon2, target.cuda()).backward()
    assert l2(recon2, out["recon"]) < 3e-2
    named = dict(m.named_parameters())
    for k, v in p.items():
        if k.startswith("_decoder") and v.numel() >= 1024:
            # small widths: bf16 rounding flips individual ReLU gates, so only the larger tensors average out
            assert l2(named[k].grad, v.grad) < 1e-1, k


def test_bf16_matches_f32_mode_on_same_weights_and_codes():
    """Decoder alone (no argmin in the way): bf16 vs the HIP fp32 mode, outputs and all gradients."""
    from acoustic_locating_vq_vae import _ops
    from acoustic_locating_vq_vae.vq_vae.deconvolutional_decoder import DeconvolutionalDecoder
    torch.manual_seed(22)
    dec = DeconvolutionalDecoder(32, 40, 128, 2, 64, False, 0.25).cuda()
    q = torch.randn(3, 32, 77).cuda()
    g = torch.randn(3, 40, 77).cuda()
    res = {}
    for mode in ("f32", "bf16"):
        _ops.set_compute_dtype(mode)
        dec.zero_grad()
        qq = q.clone().requires_grad_(True)
        y = dec(qq)
        y.backward(g)
        res[mode] = (y.detach(), qq.grad.clone(), {k: v.grad.clone() for k, v in dec.named_parameters()})
    assert l2(res["bf16"][0], res["f32"][0]) < 2e-2
    # Backward through a ReLU gates on (activation > 0); bf16 rounding of the forward flips ~0.3 % of the gates
    # (those with |h| below the rounding error) and each flipped gate is a full-size gradient error, i.e. ~5 % in
    # relative L2 per gated layer, adding in quadrature (measured: 0.5 % at the last layer -> 9 % at the input).
    # That is a property of bf16 storage, not of the kernels (which match bf16-rounded references to 2^-8).
    assert l2(res["bf16"][1], res["f32"][1]) < 0.15
    for k in res["f32"][2]:
        assert l2(res["bf16"][2][k], res["f32"][2][k]) < 0.15, k
    assert l2(res["bf16"][2]["_conv_trans_3.weight"], res["f32"][2]["_conv_trans_3.weight"]) < 2e-2


def test_bf16_standalone_modules_roundtrip_layout():
    from acoustic_locating_vq_vae.vq_vae.convolutional_encoder import ConvolutionalEncoder
    from acoustic_locating_vq_vae.vq_vae.modules.residual_stack import ResidualStack
    torch.manual_seed(23)
    enc = ConvolutionalEncoder(20, 64, 2, 32).cuda()
    x = torch.randn(2, 20, 33)
    p = {("_encoder." + k): v.detach().cpu() for k, v in enc.state_dict().items()}
    ref = O.encoder(x, p, "_encoder.", 2)
    xg = x.cuda().requires_grad_(True)
    y = enc(xg)
    assert y.shape == ref.shape and l2(y, ref) < 2e-2
    y.sum().backward()
    assert xg.grad.shape == x.shape and enc._conv_1.weight.grad is not None
    st = ResidualStack(16, 16, 3, 8).cuda()
    h = torch.randn(2, 16, 9)
    w1, w2 = (w.detach().cpu() for w in st.weights)
    assert l2(st(h.cuda()), O.residual_stack(h, w1, w2, 3)) < 2e-2


def test_bf16_training_reduces_loss_like_fp32():
    """A few steps on a fixed batch: bf16 loss curve stays within a few percent of the CPU oracle's."""
    from acoustic_locating_vq_vae.train_step import Trainer
    torch.manual_seed(5)
    cfg = (20, 48, 8, 2, 24, 0.25, 64)
    m = build(cfg, use_jitter=False)
    with torch.no_grad():
        m._vq._embedding.weight.normal_(0, 0.7)
    m.train()
    ot = O.OracleTrainer(oracle_params(m), 2, 0.25, use_jitter=False)
    tr = Trainer(m, "speech")
    raw = torch.randn(4, 20, 40)
    x = O.speech_preprocess(raw)
    for step in range(5):
        want = ot.step(x)
        got = tr.step(raw.cuda())
        assert abs(float(got[0]) - want[0]) < 5e-2 * abs(want[0]), (step, float(got[0]), want[0])
