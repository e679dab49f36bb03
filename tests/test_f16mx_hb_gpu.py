"""f16mx_hb mode: f16mx forward + fp16 backward.  The forward pass must be f16mx's BIT FOR BIT (it is the same code on the
same operands); the gradients are mixed-precision-training grade: fp16 products with fp32 accumulation under a loss scale."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from acoustic_locating_vq_vae import _ops  # noqa: E402
from g3_cases import run  # noqa: E402
from oracle import vqvae_oracle as O  # noqa: E402


def rel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


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
def test_forward_is_f16mx_bit_for_bit_and_gradients_are_fp16_grade(cfg, shape, kw):
    x = O.standardise(torch.randn(*shape, generator=torch.Generator().manual_seed(5)).abs()).cuda()
    outs = {}
    for mode in ("f16mx", "f16mx_hb", "f32"):
        _ops.set_compute_dtype(mode)
        m = _model(cfg, 11, **kw)
        np.random.seed(3)
        vq_loss, recon, perp = m(x)
        (F.mse_loss(recon, x) + vq_loss).backward()
        _, _, _, idx = m.get_latent_indices(x)
        outs[mode] = (vq_loss.detach(), recon.detach(), perp.detach(), idx, {k: p.grad.detach().clone() for k, p in m.named_parameters()})
    a, b, ref = outs["f16mx"], outs["f16mx_hb"], outs["f32"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    worst = 0.0
    for k, g in b[4].items():
        l2 = float((g - ref[4][k]).norm() / ref[4][k].norm())
        worst = max(worst, l2)
        assert l2 < 1e-2, (k, l2)                       # fp16-grade: measured ~1e-3 (f16mx: ~4e-4); never bf16's 0.1
        assert torch.isfinite(g).all()
    print("f16mx_hb gradient rel-L2 vs f32, worst tensor: %.2e" % worst)


@pytest.fixture
def wide_min_tiles(request):
    from acoustic_locating_vq_vae import _native as N
    prev = N.set_option("wide_min_tiles", request.param)
    yield request.param
    N.set_option("wide_min_tiles", prev)


@pytest.mark.parametrize("tag", ["speech", "rir", "echoed"])
@pytest.mark.parametrize("wide_min_tiles", [192, 1], ids=["default_dispatch", "wide_forced"], indirect=True)
def test_default_configs_against_reference_golden(tag, wide_min_tiles, golden_dir):
    """The north star's forward bar exactly as for f16mx (tests/test_default_configs_modes_gpu.py): indices bit-exact,
    outputs 1e-3 (measured 2e-5), losses 1e-5; gradients at the bars of the split modes -- under the dispatch a user gets
    (at the goldens' B = 2 the fp16 backward then runs the 128 x 128 kernel) and under the forced 256 x 256 kernels."""
    _ops.set_compute_dtype("f16mx_hb")
    r = run(tag, golden_dir)
    print("g3-%s f16mx_hb: %s" % (tag, json.dumps(r)))
    if tag != "echoed":
        assert r["idx_mismatches"] == 0, r
        assert r["z_rel_max"] < 1e-4 and r["z_sum_rel"] < 1e-5, r
        assert r["vq_loss_rel"] < 1e-5 and r["perplexity_rel"] < 1e-5, r
    assert r["recon_error_rel"] < 1e-5, r
    assert r["recon_rel_max"] < 1e-3 and r["recon_sum_rel"] < 1e-5, r
    assert r["grad_rel_max"] < 0.1 and r["grad_rel_l2_max"] < 1.5e-2 and r["grad_rel_l2_median"] < 5e-3 and r["grad_sum_rel_max"] < 3e-3, r


def test_trainer_steps_track_the_f16mx_mode():
    """A few Trainer steps (graph replay included) in f16mx_hb against f16mx: same first loss to the bit (same forward),
    later losses within the gradient noise."""
    from acoustic_locating_vq_vae.train_step import Trainer
    cfg = (40, 128, 16, 2, 64, 0.25, 64)
    raws = [torch.randn(4, 40, 60, generator=torch.Generator().manual_seed(10 + i)).cuda() for i in range(5)]
    losses = {}
    for mode in ("f16mx", "f16mx_hb"):
        _ops.set_compute_dtype(mode)
        m = _model(cfg, 7)
        tr = Trainer(m, "speech")
        np.random.seed(5)
        tr.capture(raws[0], warmup=1)
        losses[mode] = [float(tr.step(r)[0]) for r in raws]
    assert np.isfinite(losses["f16mx_hb"]).all()
    np.testing.assert_allclose(losses["f16mx_hb"], losses["f16mx"], rtol=5e-3)


@pytest.mark.parametrize("mode", ["f16mx_hb", "bf16"])
@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
@pytest.mark.parametrize("buckets", [1, 2])
def test_deferred_batched_split_reduction_is_bitwise_the_immediate_one(mode, graph, buckets, monkeypatch):
    """Inside a Trainer step the weight-gradient launches of the bf16 / fp16 family leave their split partials in arena
    scratch and ONE launch sums them all at the end of the backward (alvq_wgrad_reduce_batch).  Same sums, same order:
    the parameters after three steps are bit-identical to the per-launch reductions (ALVQ_DEFER_REDUCE=0)."""
    from acoustic_locating_vq_vae.train_step import Trainer
    cfg = (40, 1024, 16, 3, 256, 0.25, 64)            # wide enough for several splits per weight, both reduce layouts
    raws = [torch.randn(4, 40, 60, generator=torch.Generator().manual_seed(20 + i)).cuda() for i in range(3)]
    flats = {}
    for defer in ("1", "0"):
        monkeypatch.setenv("ALVQ_DEFER_REDUCE", defer)
        _ops.set_compute_dtype(mode)
        m = _model(cfg, 9)
        tr = Trainer(m, "speech", grad_buckets=buckets)
        np.random.seed(5)
        if graph:
            tr.capture(raws[0], warmup=1)
        for r in raws:
            tr.step(r)
        torch.cuda.synchronize()
        flats[defer] = tr.buffers.flat.detach().clone()
    assert torch.isfinite(flats["1"]).all() and torch.equal(flats["1"], flats["0"])


@pytest.mark.parametrize("mode", ["x3mx_hb", "f16mx_hb", "bf16", "bf16x3", "f16mx"])
@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
def test_fused_adam_pack_is_bitwise_adam_then_pack(mode, graph, monkeypatch):
    """The optimiser launch that also emits the packed images of the conv weights (alvq_adam_pack_batch + one segmented
    launch for biases / codebook) against the separate Adam and re-pack launches (ALVQ_ADAM_PACK=0): the same parameters,
    bit for bit, after three steps -- and a load_state_dict between steps (parameters modified behind the fused images) is
    picked up, in eager steps and under graph replay."""
    from acoustic_locating_vq_vae.train_step import Trainer
    cfg = (40, 200, 16, 2, 72, 0.25, 64)              # ragged channel counts: partially filled 32 x 64 tiles in both layouts
    raws = [torch.randn(4, 40, 60, generator=torch.Generator().manual_seed(30 + i)).cuda() for i in range(4)]
    flats, after_load = {}, {}
    for fused in ("1", "0"):
        monkeypatch.setenv("ALVQ_ADAM_PACK", fused)
        _ops.set_compute_dtype(mode)
        m = _model(cfg, 13)
        sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
        tr = Trainer(m, "speech")
        np.random.seed(5)
        if graph:
            tr.capture(raws[0], warmup=1)
        for r in raws[:3]:
            tr.step(r)
        torch.cuda.synchronize()
        flats[fused] = tr.buffers.flat.detach().clone()
        m.load_state_dict(sd0)                        # back to the initial weights, behind the optimiser's back
        loss = tr.step(raws[3])[0]
        after_load[fused] = (float(loss), tr.buffers.flat.detach().clone())
    assert torch.isfinite(flats["1"]).all() and torch.equal(flats["1"], flats["0"])
    assert after_load["1"][0] == after_load["0"][0] and torch.equal(after_load["1"][1], after_load["0"][1])


def test_trainer_reports_fp16_saturation():
    """Inputs far outside fp16's range: the Trainer's periodic range check (every step here) warns; sane inputs do not."""
    import warnings
    from acoustic_locating_vq_vae import _native as N
    from acoustic_locating_vq_vae.train_step import Trainer
    _ops.set_compute_dtype("f16mx_hb")
    m = _model((20, 48, 8, 2, 24, 0.25, 64), 3, use_jitter=False)
    with torch.no_grad():
        m._encoder._conv_1.weight.mul_(3e4)          # activations of the first layer leave fp16's range
    tr = Trainer(m, "speech", range_check_every=1)
    N.f16mx_range_flag(reset=True)
    raw = torch.randn(3, 20, 40, generator=torch.Generator().manual_seed(1)).cuda()
    tr.step(raw)                                     # step 1 runs; the check at the START of step 2 sees its flag
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tr.step(raw)
    assert any("fp16 range flag" in str(x.message) for x in w), [str(x.message) for x in w]
    m2 = _model((20, 48, 8, 2, 24, 0.25, 64), 3, use_jitter=False)
    tr2 = Trainer(m2, "speech", range_check_every=1)
    N.f16mx_range_flag(reset=True)
    with warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter("always")
        tr2.step(raw)
        tr2.step(raw)
    assert not any("fp16 range flag" in str(x.message) for x in w2)


def test_loss_curves_separate_no_faster_than_an_fp32_run_does_from_itself():
    """How far, how fast (round-2 verdict, weak item 4): 80 Trainer steps of a mid-size model in f32, in f32 with the initial
    weights nudged by one ulp (the control: an equally exact fp32 run), and in the default mode.  Training is sensitive to
    its own rounding -- codebook assignments are discrete -- so the default mode is held to the CONTROL's distance from
    f32, not to zero: it must not leave the f32 curve sooner or further than a second fp32 run does (within a factor, the
    curves being single samples), and it must reach the same loss level.  Measured: control 3.0e-2 (outside 1e-3 from step
    17), f16mx_hb 4.4e-2 (from step 5: its ~1e-3 gradient noise is a larger seed than one ulp); speech config, 200 steps:
    tools/long_run_modes.py, profiles/r03_long_run_modes.txt."""
    from acoustic_locating_vq_vae.train_step import Trainer
    cfg = (40, 128, 16, 2, 64, 0.25, 64)
    pool = [torch.randn(8, 40, 60, generator=torch.Generator().manual_seed(30 + i)).abs().cuda() * (1 + i % 3) for i in range(4)]
    curves = {}
    for leg in ("f32", "f32+ulp", "f16mx_hb"):
        _ops.set_compute_dtype(leg.split("+")[0])
        torch.manual_seed(21)
        from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
        m = ConvolutionalVQVAE(*cfg, use_jitter=False).cuda().train()
        if leg.endswith("+ulp"):
            with torch.no_grad():
                for q in m.parameters():
                    q.mul_(1.0 + 2.0 ** -23)
        tr = Trainer(m, "speech")
        curves[leg] = np.array([float(tr.step(pool[s % 4])[0]) for s in range(80)])
    ref = curves["f32"]
    dev = {k: np.abs(v - ref) / np.abs(ref) for k, v in curves.items() if k != "f32"}
    first = {k: int(np.argmax(d > 1e-3)) if (d > 1e-3).any() else 80 for k, d in dev.items()}
    print("max relative deviation from f32 over 80 steps: control %.2e, f16mx_hb %.2e; first step outside 1e-3: control %d, f16mx_hb %d; "
          "final losses %.4f / %.4f / %.4f" % (dev["f32+ulp"].max(), dev["f16mx_hb"].max(), first["f32+ulp"], first["f16mx_hb"],
                                              ref[-1], curves["f32+ulp"][-1], curves["f16mx_hb"][-1]))
    assert np.isfinite(curves["f16mx_hb"]).all()
    assert dev["f16mx_hb"][:3].max() < 2e-3        # the first steps: forward parity + fp16-grade gradients (measured: inside 1e-3 for 5 steps)
    assert dev["f16mx_hb"].max() < max(10 * dev["f32+ulp"].max(), 5e-2)
    assert abs(curves["f16mx_hb"][-10:].mean() - ref[-10:].mean()) < 0.1 * abs(ref[-10:].mean())
