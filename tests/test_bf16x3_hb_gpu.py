"""bf16x3_hb mode: bf16x3 forward + bf16 backward on the hi planes.  The forward must be bf16x3's BIT FOR BIT (same kernels,
same operands) -- the most accurate split format: no flipped index in the goldens' 47 834 rows -- and the gradients carry the
rounding of bf16 operands (2^-9 per operand) under an exact forward: rel-L2 ~2e-3 per tensor, between f16mx_hb's 5e-4 and the
bf16 mode's 0.1 (whose error is its FORWARD's, flipped gates and codes)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from acoustic_locating_vq_vae import _ops  # noqa: E402
from oracle import vqvae_oracle as O  # noqa: E402


@pytest.fixture(autouse=True)
def _restore():
    yield
    _ops.set_compute_dtype("f32")


def _model(cfg, seed, **kw):
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    torch.manual_seed(seed)
    m = ConvolutionalVQVAE(*cfg, **kw)
    with torch.no_grad():
        m._vq._embedding.weight.normal_(0, 0.7)
    return m.cuda().train()


@pytest.mark.parametrize("cfg,shape,kw", [((20, 48, 8, 3, 24, 0.25, 64), (3, 20, 40), dict(use_jitter=False)),
                                          ((201, 128, 32, 2, 128, 0.25, 128), (2, 201, 96), dict()),
                                          ((50, 1024, 8, 2, 1024, 0.25, 64), (3, 50, 130), dict(use_jitter=False))])
def test_forward_is_bf16x3_bit_for_bit_and_gradients_are_bf16_operand_grade(cfg, shape, kw):
    x = O.standardise(torch.randn(*shape, generator=torch.Generator().manual_seed(5)).abs()).cuda()
    outs = {}
    for mode in ("bf16x3", "bf16x3_hb", "f32"):
        _ops.set_compute_dtype(mode)
        m = _model(cfg, 11, **kw)
        np.random.seed(3)
        vq_loss, recon, perp = m(x)
        (F.mse_loss(recon, x) + vq_loss).backward()
        _, _, _, idx = m.get_latent_indices(x)
        outs[mode] = (vq_loss.detach(), recon.detach(), perp.detach(), idx, {k: p.grad.detach().clone() for k, p in m.named_parameters()})
    a, b, ref = outs["bf16x3"], outs["bf16x3_hb"], outs["f32"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    worst = 0.0
    for k, g in b[4].items():
        assert torch.isfinite(g).all()
        l2 = float((g - ref[4][k]).norm() / ref[4][k].norm())
        worst = max(worst, l2)
        assert l2 < 2e-2, (k, l2)
    print("bf16x3_hb gradient rel-L2 vs f32, worst tensor: %.2e" % worst)


def test_trainer_steps_eager_and_replayed_track_bf16x3():
    from acoustic_locating_vq_vae.train_step import Trainer
    cfg = (40, 128, 16, 2, 64, 0.25, 64)
    raws = [torch.randn(4, 40, 60, generator=torch.Generator().manual_seed(10 + i)).cuda() for i in range(5)]
    losses = {}
    for mode, graph in (("bf16x3", True), ("bf16x3_hb", True), ("bf16x3_hb", False)):
        _ops.set_compute_dtype(mode)
        m = _model(cfg, 7)
        tr = Trainer(m, "speech")
        np.random.seed(5)
        if graph:
            tr.capture(raws[0], warmup=1)
        else:
            tr.step(raws[0])
        losses[mode, graph] = [float(tr.step(r)[0]) for r in raws]
    assert np.isfinite(losses["bf16x3_hb", True]).all()
    assert losses["bf16x3_hb", True] == losses["bf16x3_hb", False]
    np.testing.assert_allclose(losses["bf16x3_hb", True], losses["bf16x3", True], rtol=5e-3)


@pytest.fixture
def wide_min_tiles(request):
    from acoustic_locating_vq_vae import _native as N
    prev = N.set_option("wide_min_tiles", request.param)
    yield request.param
    N.set_option("wide_min_tiles", prev)


@pytest.mark.parametrize("tag", ["speech", "rir", "echoed"])
@pytest.mark.parametrize("wide_min_tiles", [192, 1], ids=["default_dispatch", "wide_forced"], indirect=True)
def test_default_configs_against_reference_golden_under_both_dispatches(tag, wide_min_tiles, golden_dir):
    """The B = 2 goldens under the dispatch a user gets (the bf16 backward then runs the 128 x 128 kernels, reading the hi
    planes of bf16x3 activations as operands and masks) and under the forced 256 x 256 kernels: same bars."""
    import json
    from g3_cases import run
    _ops.set_compute_dtype("bf16x3_hb")
    r = run(tag, golden_dir)
    print("g3-%s bf16x3_hb wide_min_tiles=%d: %s" % (tag, wide_min_tiles, json.dumps(r)))
    if tag != "echoed":
        assert r["idx_mismatches"] == 0 and r["z_rel_max"] < 1e-4 and r["vq_loss_rel"] < 1e-5, r
    assert r["recon_error_rel"] < 1e-5 and r["recon_rel_max"] < 1e-3 and r["recon_sum_rel"] < 1e-5, r
    assert r["grad_rel_max"] < 0.1 and r["grad_rel_l2_max"] < 1.5e-2 and r["grad_rel_l2_median"] < 5e-3 and r["grad_sum_rel_max"] < 3e-3, r
