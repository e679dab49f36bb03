"""T3/T5: the module API on the HIP path against the CPU oracle and the goldens captured from the real
reference.  Tolerance 1e-3 relative (north_star) on outputs/grads -- observed ~1e-5; indices bit-exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import vqvae_oracle as O  # noqa: E402

TOL = 1e-3


def rel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def sl(t, n=64):
    f = t.detach().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].cpu().numpy()


def expand(p, R):
    out = {}
    for k, v in p.items():
        if "_layers.0." in k:
            for r in range(R):
                out[k.replace("_layers.0.", "_layers.%d." % r)] = v
        else:
            out[k] = v
    return out


def build(cfg, p=None, **kw):
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    m = ConvolutionalVQVAE(*cfg, **kw)
    if p is not None:
        m.load_state_dict(expand(p, cfg[3]))
    return m.cuda()


def oracle_params(m):
    return {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()
            if "_layers." not in k or "_layers.0." in k}


def test_g1_tiny_against_reference_golden(golden_dir):
    from acoustic_locating_vq_vae.train_step import Trainer
    g = np.load(os.path.join(golden_dir, "g1_tiny_vqvae.npz"))
    p = {k[len("param:"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param:")}
    m = build((7, 16, 4, 2, 8, 0.25, 16), p)
    m.train()
    tr = Trainer(m, "speech")
    x_raw = torch.from_numpy(g["x_raw"]).cuda()
    x, _ = tr.preprocess(x_raw)
    assert rel(x, g["x"]) < 1e-5
    z = m._latent(x)
    assert rel(z, g["z"]) < 1e-5
    _, q_st, _, enc = m.get_latent_representation(x)
    assert np.array_equal(enc.argmax(1).cpu().numpy(), g["idx"])
    assert np.array_equal(enc.cpu().numpy(), g["encodings"])
    assert rel(q_st, g["q_st"]) < 1e-5
    np.random.seed(5)
    loss, recon_error, perp = tr.step(x_raw)
    assert rel(recon_error, g["recon_error"]) < 1e-4
    assert rel(loss, g["recon_error"] + g["vq_loss"]) < 1e-4
    assert rel(perp, g["perplexity"]) < 1e-5
    named = dict(m.named_parameters())
    for k in p:
        assert rel(named[k].grad, g["grad:" + k]) < TOL, k
        assert rel(named[k].grad, g["grad:" + k]) < 1e-4, k
        assert float((named[k].detach().cpu() - torch.from_numpy(g["after:" + k])).abs().max()) < 5e-6, k
    # same seed -> same jitter columns -> same reconstruction
    m2 = build((7, 16, 4, 2, 8, 0.25, 16), p).train()
    np.random.seed(5)
    _, recon, _ = m2(x)
    assert rel(recon, g["recon"]) < 1e-4


CASES = [
    # cfg (in,H,D,R,RH,beta,K), input shape (B,C,L), kwargs, permuted
    ((7, 16, 4, 2, 8, 0.25, 16), (2, 7, 13), dict(), False),
    ((20, 48, 8, 3, 24, 0.25, 64), (3, 20, 40), dict(use_jitter=False), False),
    ((33, 64, 16, 1, 64, 0.5, 32), (2, 33, 64), dict(), False),
    ((50, 64, 8, 2, 16, 0.25, 64), (4, 24, 50), dict(use_jitter=False, out_channels=1), True),
    ((201, 128, 32, 2, 128, 0.25, 128), (2, 201, 96), dict(), False),
]


@pytest.mark.parametrize("cfg,shape,kw,permuted", CASES)
def test_forward_backward_matches_oracle(cfg, shape, kw, permuted):
    torch.manual_seed(11)
    m = build(cfg, **kw)
    with torch.no_grad():
        m._vq._embedding.weight.normal_(0, 0.7)      # data-scale codebook: wide argmin margins
    m.train()
    p = oracle_params(m)
    x = O.standardise(torch.randn(*shape).abs())
    if permuted:
        x = x.permute(0, 2, 1)
    oc = kw.get("out_channels")
    target = x if oc is None else torch.randn(shape[0], oc, x.shape[2])
    use_jitter = kw.get("use_jitter", True)
    np.random.seed(3)
    src = O.jitter_source_index(x.shape[2], 0.25) if use_jitter else None
    out = O.vqvae_forward(x, p, cfg[3], cfg[5], src)
    (F.mse_loss(out["recon"], target) + out["vq_loss"]).backward()

    xg = x.cuda().requires_grad_(True)
    np.random.seed(3)
    vq_loss, recon, perp = m(xg)
    (F.mse_loss(recon, target.cuda()) + vq_loss).backward()
    _, _, _, idx = m.eval().get_latent_indices(x.cuda())
    assert torch.equal(idx.cpu(), out["idx"])
    assert rel(recon, out["recon"]) < 1e-4
    assert rel(vq_loss, out["vq_loss"]) < 1e-5 and rel(perp, out["perplexity"]) < 1e-5
    named = dict(m.named_parameters())
    for k, v in p.items():
        assert rel(named[k].grad, v.grad) < 1e-4, k
    assert xg.grad is not None and xg.grad.shape == x.shape


def test_eval_mode_has_no_jitter_and_is_deterministic():
    torch.manual_seed(1)
    m = build((7, 16, 4, 2, 8, 0.25, 16)).eval()
    x = torch.randn(2, 7, 13).cuda()
    a = m(x)[1]
    b = m(x)[1]
    assert torch.equal(a, b)
    p = oracle_params(m)
    out = O.vqvae_forward(x.cpu(), p, 2, 0.25, None)
    assert rel(a, out["recon"]) < 1e-4


@pytest.mark.parametrize("mode,tol,gtol", [("f32", 1e-4, 1e-4), ("x3mx_hb", 1e-3, 5e-2)])
def test_average_pooling_path_matches_oracle(mode, tol, gtol):
    """encoder_average_pooling=True (convolutional_vq_vae.py:96-97; no script enables it): the latent is averaged over time
    by alvq_row_mean_f32 and its adjoint -- values and every gradient against the oracle, use_jitter=False (with jitter the
    one-column latent raises in the reference and here alike, _ops.jitter_source_index).  In the default mode too: the decoder
    then runs on ONE position per sample (L = 1)."""
    from acoustic_locating_vq_vae import _ops
    _ops.set_compute_dtype(mode)
    try:
        _average_pooling_case(tol, gtol)
    finally:
        _ops.set_compute_dtype("f32")


def _average_pooling_case(tol, gtol):
    torch.manual_seed(4)
    m = build((7, 16, 4, 2, 8, 0.25, 16), encoder_average_pooling=True, out_channels=3, use_jitter=False).train()
    with torch.no_grad():
        m._vq._embedding.weight.normal_(0, 0.7)
    p = oracle_params(m)
    x = torch.randn(5, 7, 13)
    target = torch.randn(5, 3, 1)
    out = O.vqvae_forward(x, p, 2, 0.25, None, average_pooling=True)
    assert out["recon"].shape == (5, 3, 1)
    (F.mse_loss(out["recon"], target) + out["vq_loss"]).backward()
    vq_loss, recon, perp = m(x.cuda())
    assert recon.shape == (5, 3, 1)
    (F.mse_loss(recon, target.cuda()) + vq_loss).backward()
    assert rel(recon, out["recon"]) < tol and rel(vq_loss, out["vq_loss"]) < tol / 10 and rel(perp, out["perplexity"]) < 1e-5
    named = dict(m.named_parameters())
    for k, v in p.items():
        assert rel(named[k].grad, v.grad) < gtol, k


def test_submodules_standalone():
    from acoustic_locating_vq_vae.vq_vae.modules.residual_stack import ResidualStack
    from acoustic_locating_vq_vae.vq_vae.modules.residual import Residual
    from acoustic_locating_vq_vae.vq_vae.modules.jitter import Jitter
    from acoustic_locating_vq_vae.vq_vae.vector_quantizer import VectorQuantizer
    torch.manual_seed(2)
    st = ResidualStack(12, 12, 3, 6).cuda()
    h = torch.randn(2, 12, 9)
    w1, w2 = (w.detach().cpu().requires_grad_(True) for w in st.weights)
    hc = h.clone().requires_grad_(True)
    ref = O.residual_stack(hc, w1, w2, 3)
    ref.sum().backward()
    hg = h.cuda().requires_grad_(True)
    got = st(hg)
    got.sum().backward()
    assert rel(got, ref) < 1e-5 and rel(hg.grad, hc.grad) < 1e-5
    assert rel(st.weights[0].grad, w1.grad) < 1e-5 and rel(st.weights[1].grad, w2.grad) < 1e-5
    res = Residual(12, 12, 6).cuda()
    r1, r2 = (w.detach().cpu() for w in res.weights)
    t = F.relu(h)
    want = t + F.conv1d(F.relu(F.conv1d(t, r1, padding=1)), r2)
    assert rel(res(h.cuda()), want) < 1e-5
    np.random.seed(4)
    q = torch.randn(2, 3, 17).cuda().requires_grad_(True)
    y = Jitter(0.25)(q)
    np.random.seed(4)
    src = O.jitter_source_index(17, 0.25)
    assert torch.equal(y.detach().cpu(), q.detach().cpu()[:, :, torch.from_numpy(src)])
    y.sum().backward()
    assert torch.equal(q.grad.cpu()[0, 0], torch.from_numpy((src == np.arange(17)).astype(np.float32)))
    vq = VectorQuantizer(16, 4, 0.25).cuda()
    vq.set_train_vq(False)
    z = torch.randn(2, 4, 6).cuda().requires_grad_(True)
    loss, q2, perp, enc = vq(z)
    (loss + q2.sum()).backward()
    assert vq._embedding.weight.grad is None or float(vq._embedding.weight.grad.abs().max()) == 0.0
    assert enc.shape == (12, 16) and float(enc.sum()) == 12.0


def _default(tag):
    if tag == "speech":
        return (201, 1024, 128, 3, 1024, 0.25, 1024), (2, 201, 500), False, None, True
    return (500, 1024, 64, 2, 64, 0.25, 1024), (2, 201, 500), True, 1, False


@pytest.mark.parametrize("tag", ["speech", "rir"])
def test_g3_default_configs_against_reference_golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "g3_%s.npz" % tag))
    cfg, shape, permuted, oc, jit = _default(tag)
    in_c, h, d, r, rh, beta, k = cfg
    p = O.closed_form_params(O.vqvae_param_shapes(in_c, h, d, rh, k, oc), float(g["cb_scale"]), float(g["gain"]))
    m = build(cfg, p, use_jitter=jit, out_channels=oc).train()
    x = O.speech_preprocess(torch.from_numpy(O.hashed_uniform(int(np.prod(shape)), 21, 2.0).reshape(shape)))
    if permuted:
        x = x.permute(0, 2, 1)
    if oc is None:
        target = x
    else:
        tr = torch.from_numpy(O.hashed_uniform(shape[0] * x.shape[2], 22, 2.0).reshape(shape[0], x.shape[2]))
        target = O.standardise(tr).unsqueeze(1)
    xg = x.cuda()                                   # RIR: a permuted (non-contiguous) view, as train_rir.py:45 hands over
    from g3_cases import sum_rel, wide
    z = m._latent(xg)
    assert rel(sl(z), g["z_slice"]) < 1e-4
    assert rel(wide(z, g["z_wide"]), g["z_wide"]) < 1e-4 and sum_rel(z, g["z_sum"]) < 1e-5      # 4096 elements + the whole tensor
    _, _, _, idx = m.get_latent_indices(xg)
    idx = idx.cpu().numpy().astype(np.int16)
    bad = np.nonzero(idx != g["idx"])[0]
    gap = (g["top2_val"][:, 1] - g["top2_val"][:, 0]) / np.abs(g["top2_val"][:, 0])
    assert all(gap[i] < 1e-5 for i in bad), (bad, gap[bad])      # bit-exact except on reference near-ties
    assert len(bad) == 0
    np.random.seed(9)
    vq_loss, recon, perp = m(xg)
    err = F.mse_loss(recon, target.cuda())
    (err + vq_loss).backward()
    assert rel(vq_loss, g["vq_loss"]) < 1e-4 and rel(err, g["recon_error"]) < 1e-4
    assert rel(perp, g["perplexity"]) < 1e-5
    assert rel(sl(recon), g["recon_slice"]) < TOL
    assert rel(wide(recon, g["recon_wide"]), g["recon_wide"]) < TOL and sum_rel(recon, g["recon_sum"]) < 1e-5
    for key, pp in m.named_parameters():
        assert rel(sl(pp.grad), g["grad_slice:" + key]) < TOL, key
        assert rel(wide(pp.grad, g["grad_wide:" + key]), g["grad_wide:" + key]) < TOL, key
        assert sum_rel(pp.grad, g["grad_sum:" + key]) < 1e-3, key


def test_echoed_model_against_oracle_and_golden(golden_dir):
    from acoustic_locating_vq_vae.vq_vae.echoed_speech_model import EchoedSpeechReconModel
    from acoustic_locating_vq_vae.train_step import Trainer
    g = np.load(os.path.join(golden_dir, "g3_echoed.npz"))
    gain = float(g["gain"])
    sp_p = O.closed_form_params(O.vqvae_param_shapes(201, 1024, 128, 1024, 1024), float(g["speech_cb_scale"]), gain)
    rir_p = O.closed_form_params(O.vqvae_param_shapes(500, 1024, 64, 64, 1024, 1), float(g["rir_cb_scale"]), gain)
    sp = build((201, 1024, 128, 3, 1024, 0.25, 1024), sp_p)
    rir = build((500, 1024, 64, 2, 64, 0.25, 1024), rir_p, use_jitter=False, out_channels=1)
    model = EchoedSpeechReconModel(rir, sp, 201, 1024, 2, 1024, True)
    dec_p = O.closed_form_params(O.decoder_param_shapes(192, 201, 1024, 1024), gain=gain)
    model._decoder.load_state_dict({k[len("_decoder."):]: v for k, v in expand(dec_p, 2).items()})
    model = model.cuda().train()
    shape = (2, 201, 500)
    raw = torch.from_numpy(O.hashed_uniform(int(np.prod(shape)), 21, 2.0).reshape(shape)).abs()
    tr = Trainer(model, "echoed")
    np.random.seed(9)
    loss, err, sperp = tr.step(raw.cuda())
    assert rel(err, g["recon_error"]) < 1e-4
    assert rel(sperp, g["speech_perplexity"]) < 1e-5
    for key, pp in model._decoder.named_parameters():
        assert rel(sl(pp.grad), g["grad_slice:_decoder." + key]) < TOL, key
    assert all(p.grad is None for p in model.speech_model.parameters())
    # fine-tuning mode lets gradients into both encoders (encoder_training_echoed_model.py:43-46)
    for p in model.parameters():
        p.requires_grad_(True)
    model.set_train_encoder(True)
    x = O.standardise(raw).cuda()
    recon, _, _ = model(x, x.permute(0, 2, 1))
    recon.square().mean().backward()
    assert model.speech_model._encoder._conv_1.weight.grad is not None
    assert model.rir_model._pre_vq_conv.weight.grad.abs().sum() > 0
    assert model.speech_model._vq._embedding.weight.grad is None or float(model.speech_model._vq._embedding.weight.grad.abs().max()) == 0


def test_train_steps_track_oracle_trainer():
    """T5: K steps from identical init/inputs: loss curve vs the CPU oracle trainer."""
    from acoustic_locating_vq_vae.train_step import Trainer
    torch.manual_seed(5)
    cfg = (20, 48, 8, 2, 24, 0.25, 64)
    m = build(cfg)
    with torch.no_grad():
        m._vq._embedding.weight.normal_(0, 0.7)
    m.train()
    ot = O.OracleTrainer(oracle_params(m), 2, 0.25, use_jitter=True)
    tr = Trainer(m, "speech")
    raw = torch.randn(4, 20, 40)
    x = O.speech_preprocess(raw)
    for step in range(4):
        np.random.seed(100 + step)
        want = ot.step(x)
        np.random.seed(100 + step)
        got = tr.step(raw.cuda())
        assert abs(float(got[0]) - want[0]) < 2e-3 * abs(want[0]), (step, float(got[0]), want[0])
        assert abs(float(got[2]) - want[2]) < 1e-2 * abs(want[2]) + 1e-3


def test_graph_replay_equals_eager_steps():
    """A captured hipGraph of the step (fwd+bwd+Adam) replays to the same parameters as eager launches."""
    from acoustic_locating_vq_vae.train_step import Trainer
    cfg = (20, 48, 8, 2, 24, 0.25, 64)
    raws = [torch.randn(4, 20, 40, generator=torch.Generator().manual_seed(s)).cuda() for s in range(6)]
    finals = []
    for use_graph in (False, True):
        torch.manual_seed(7)
        m = build(cfg)
        with torch.no_grad():
            m._vq._embedding.weight.normal_(0, 0.7)
        m.train()
        tr = Trainer(m, "speech")
        np.random.seed(42)
        if use_graph:
            tr.capture(raws[0], warmup=3)           # 3 real steps on raws[0]
        else:
            for _ in range(3):
                tr.step(raws[0])
        losses = [float(tr.step(r)[0]) for r in raws[1:]]
        finals.append((losses, tr.buffers.flat.clone()))
    assert np.allclose(finals[0][0], finals[1][0], rtol=1e-5), (finals[0][0], finals[1][0])
    assert float((finals[0][1] - finals[1][1]).abs().max()) < 1e-5


@pytest.mark.parametrize("mode", ["f32", "x3mx_hb"])
def test_ragged_batch_after_capture_runs_eagerly(mode):
    """The last batch of an epoch is smaller than the captured one: it must run (as eager launches) with exactly the result
    a Trainer without graphs gives, and the replays must carry on afterwards.  A one-sample batch is the nasty case: copy_
    into the graph's static buffer would broadcast it over the whole batch without an error."""
    from acoustic_locating_vq_vae import _ops
    from acoustic_locating_vq_vae.train_step import Trainer
    cfg = (20, 48, 8, 2, 24, 0.25, 64)
    gen = lambda b, s: torch.randn(b, 20, 40, generator=torch.Generator().manual_seed(s)).cuda()
    seq = [gen(4, 1), gen(3, 2), gen(4, 3), gen(1, 4), gen(4, 5)]
    finals = []
    _ops.set_compute_dtype(mode)
    try:
        for use_graph in (False, True):
            torch.manual_seed(7)
            m = build(cfg, use_jitter=False)
            with torch.no_grad():
                m._vq._embedding.weight.normal_(0, 0.7)
            m.train()
            tr = Trainer(m, "speech")
            if use_graph:
                tr.capture(gen(4, 0), warmup=2)
            else:
                for _ in range(2):
                    tr.step(gen(4, 0))
            losses = [float(tr.step(r)[0]) for r in seq]
            finals.append((losses, tr.buffers.flat.clone()))
    finally:
        _ops.set_compute_dtype("f32")
    assert np.allclose(finals[0][0], finals[1][0], rtol=1e-5), (finals[0][0], finals[1][0])
    assert float((finals[0][1] - finals[1][1]).abs().max()) < 1e-5


def test_graph_replay_survives_workspace_growth():
    """A captured graph records the raw pointer of the scratch its weight-gradient / quantiser launches used.  A later
    eager call that needs MORE scratch (a bigger batch, another model, another dtype -- bench.py does all three) must
    not hand that memory back to the allocator: replay afterwards has to give the same steps as an undisturbed run."""
    from acoustic_locating_vq_vae import _native as N
    from acoustic_locating_vq_vae.train_step import Trainer
    cfg = (20, 48, 8, 2, 24, 0.25, 64)
    raws = [torch.randn(4, 20, 40, generator=torch.Generator().manual_seed(s)).cuda() for s in range(5)]
    finals = []
    for disturb in (False, True):
        torch.manual_seed(7)
        m = build(cfg)
        with torch.no_grad():
            m._vq._embedding.weight.normal_(0, 0.7)
        m.train()
        tr = Trainer(m, "speech")
        np.random.seed(42)
        tr.capture(raws[0], warmup=2)
        if disturb:
            # far larger split-K partials than the captured net; no jitter, so the np.random stream the captured
            # trainer's jitter draws from is left alone
            big = build((64, 256, 16, 2, 128, 0.25, 128), use_jitter=False).train()
            xb = torch.randn(8, 64, 300).cuda()
            loss, recon, _ = big(xb)
            (loss + recon.square().mean()).backward()
            junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]   # whatever was freed gets reused
            torch.cuda.synchronize()
            del junk
        losses = [float(tr.step(r)[0]) for r in raws[1:]]
        finals.append((losses, tr.buffers.flat.clone()))
    assert np.all(np.isfinite(finals[1][0]))
    assert finals[0][0] == finals[1][0], (finals[0][0], finals[1][0])
    assert torch.equal(finals[0][1], finals[1][1])
    # the mechanism itself: a workspace that is outgrown is retired, never released, and scratch is per stream
    dev = torch.device("cuda", torch.cuda.current_device())
    small = N._workspace(1 << 20, dev)
    ptr, n_retired = small.data_ptr(), len(N._WS_RETIRED)
    large = N._workspace(small.numel() + (64 << 20), dev)
    assert large.data_ptr() != ptr and len(N._WS_RETIRED) == n_retired + 1 and N._WS_RETIRED[-1].data_ptr() == ptr
    assert N._workspace(1 << 20, dev).data_ptr() == large.data_ptr()          # grow-only
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert N._workspace(1 << 20, dev).data_ptr() != large.data_ptr()      # another stream, another buffer


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_two_part_backward_equals_single_backward(dtype):
    """The trainers' default (backward cut at the encoder output so the first gradient bucket's all-reduce can
    overlap the second part) produces bit-identical parameters to one plain backward, eagerly and from the two
    captured hipGraphs."""
    from acoustic_locating_vq_vae import _ops
    from acoustic_locating_vq_vae.train_step import Trainer
    cfg = (20, 48, 8, 2, 24, 0.25, 64)
    raws = [torch.randn(4, 20, 40, generator=torch.Generator().manual_seed(s)).cuda() for s in range(5)]
    finals = {}
    prev = _ops.get_compute_dtype()
    _ops.set_compute_dtype(dtype)
    try:
        for tag, buckets, use_graph in (("single", 1, False), ("split", 2, False), ("split+graph", 2, True)):
            torch.manual_seed(7)
            m = build(cfg)
            with torch.no_grad():
                m._vq._embedding.weight.normal_(0, 0.7)
            m.train()
            tr = Trainer(m, "speech", grad_buckets=buckets)
            assert (tr._buckets is not None) == (buckets == 2)
            np.random.seed(42)
            if use_graph:
                tr.capture(raws[0], warmup=2)
                assert tr._graph_late is not None
            else:
                for _ in range(2):
                    tr.step(raws[0])
            for r in raws[1:]:
                tr.step(r)
            finals[tag] = tr.buffers.flat.clone()
    finally:
        _ops.set_compute_dtype(prev)
    assert torch.equal(finals["single"], finals["split"])
    assert torch.equal(finals["single"], finals["split+graph"])


def test_trainer_checkpoint_resume_is_bitwise(tmp_path):
    """Trainer.state_dict() / load_state_dict(): model + flat Adam moments + step count; a run resumed from the file
    continues with exactly the parameters an uninterrupted run reaches."""
    from acoustic_locating_vq_vae.train_step import Trainer
    cfg = (20, 48, 8, 2, 24, 0.25, 64)
    raws = [torch.randn(4, 20, 40, generator=torch.Generator().manual_seed(40 + s)).cuda() for s in range(5)]

    def fresh():
        torch.manual_seed(7)
        m = build(cfg)
        with torch.no_grad():
            m._vq._embedding.weight.normal_(0, 0.7)
        return Trainer(m.train(), "speech")

    np.random.seed(5)
    full = fresh()
    for r in raws:
        full.step(r)
    np.random.seed(5)
    first = fresh()
    for r in raws[:3]:
        first.step(r)
    path = str(tmp_path / "trainer.pt")
    torch.save(first.state_dict(), path)
    rng = np.random.get_state()                      # the jitter stream is the caller's to carry over
    resumed = fresh()
    resumed.load_state_dict(torch.load(path))
    assert resumed.opt.step_count == 3 and float(resumed.opt.scalars[3]) == 3.0
    np.random.set_state(rng)
    for r in raws[3:]:
        resumed.step(r)
    assert torch.equal(resumed.buffers.flat, full.buffers.flat)
    assert torch.equal(resumed.opt.exp_avg_sq, full.opt.exp_avg_sq)
    with pytest.raises(ValueError, match="different trainer"):
        Trainer(build((20, 32, 8, 2, 24, 0.25, 64)).train(), "speech").load_state_dict(torch.load(path))


def test_rir_trainer_tracks_oracle_and_graph():
    """Trainer(kind="rir"): device-side standardise + transpose of the (B,F,T) spectrogram, the Wiener target, jitter off,
    1-channel output -- loss curve vs the CPU oracle over several steps, eagerly and from the two captured graphs."""
    from acoustic_locating_vq_vae.train_step import Trainer
    cfg = (33, 32, 6, 2, 8, 0.25, 16)
    shapes = O.vqvae_param_shapes(33, 32, 6, 8, 16, out_channels=1)
    p0 = O.closed_form_params(shapes, 0.8, gain=0.5)
    rir = [torch.from_numpy(O.hashed_uniform(3 * 21 * 33, 170 + i, 2.0).reshape(3, 21, 33)) for i in range(5)]
    wien = [torch.from_numpy(O.hashed_uniform(3 * 21, 190 + i, 1.0).reshape(3, 21)) for i in range(5)]
    ot = O.OracleTrainer(p0, 2, 0.25, use_jitter=False)
    want = []
    for r, w in zip(rir, wien):
        x, tgt = O.rir_preprocess(r, w)
        want.append(ot.step(x, tgt))
    for use_graph in (False, True):
        m = build(cfg, p0, use_jitter=False, out_channels=1).train()
        tr = Trainer(m, "rir")
        got = []
        if use_graph:
            # capture() runs its warm-up steps on the first batch: give it a throw-away trainer state by capturing on
            # batch 0 with zero warm-up steps, then replay every batch
            tr.capture(rir[0].cuda(), wien[0].cuda(), warmup=0)
        for r, w in zip(rir, wien):
            got.append(tuple(float(v) for v in tr.step(r.cuda(), w.cuda())))
        for g, w_ in zip(got, want):
            assert abs(g[0] - w_[0]) < 2e-3 * abs(w_[0]) and abs(g[2] - w_[2]) < 1e-2 * abs(w_[2]) + 1e-3, (use_graph, got, want)
