"""f16mx mode: every tensor as an fp16 plane + an fp8 (hi, lo) plane; products as fp16*fp16 + hi8*lo8 + lo8*hi8 -- one
fp16 MFMA and one block-scaled fp8 MFMA per product instead of the three bf16 MFMAs of bf16x3.  It must deliver the same
class of parity (the 1e-3 bar with more than an order of margin; indices bit-exact on data-scale codebooks), judged
against the fp32 CPU oracle and the reference goldens like the other modes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from acoustic_locating_vq_vae import _native as N  # noqa: E402
from oracle import vqvae_oracle as O  # noqa: E402

KERNEL = 2e-4       # per-launch bound (measured ~3e-5: ~1.5e-5 rms per product, max-norm over the tensor)


def rel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def fx(t, gscale=None):
    return N.ncl_to_nlc(t.cuda(), 2, "f16mx", gscale)


@pytest.fixture(autouse=True)
def _mode():
    from acoustic_locating_vq_vae import _ops
    _ops.set_compute_dtype("f16mx")
    yield
    _ops.set_compute_dtype("f32")


@pytest.fixture(params=["128", "256"])
def rows_tile(request):
    """Both row tiles of conv1d_f16mx_kernel (256 rows per workgroup; 128 for launches that would leave CUs idle)."""
    prev = N.set_option("fx_rows", int(request.param))
    yield request.param
    N.set_option("fx_rows", prev)


SHAPES = [(2, 7, 16, 13, 3), (2, 16, 7, 13, 3), (2, 8, 16, 13, 1), (3, 5, 1, 201, 3), (2, 201, 1024, 500, 3),
          (2, 1024, 128, 500, 3), (2, 1024, 1024, 201, 1), (2, 500, 1024, 201, 3), (2, 1024, 201, 500, 3),
          (5, 130, 130, 129, 3)]


def test_split_roundtrip_and_planes():
    torch.manual_seed(0)
    x = torch.randn(3, 201, 37) * 10
    n = fx(x)
    assert n.planes == 2 and n.fmt == "f16mx"
    assert rel(N.nlc_to_ncl(n), x) < 1e-4                   # fp16 + 4 bits of the remainder
    h = n.matrix(0).view(torch.float16).float().cpu()       # the H plane alone is the fp16 rounding of x
    want = torch.zeros(n.rows, n.Cp)
    want[1:1 + 3 * 38].view(3, 38, n.Cp)[:, :37, :201] = x.permute(0, 2, 1).half().float()
    assert torch.equal(h, want)
    # saturation instead of infinities / NaNs outside fp16's and e4m3's range
    big = torch.tensor([[[1e6, -1e9, 70000.0, 3e-9]]]).expand(1, 1, 4).contiguous()
    back = N.nlc_to_ncl(fx(big))
    assert torch.isfinite(back).all() and abs(float(back[0, 0, 0]) - 65504.0) < 1 and abs(float(back[0, 0, 1]) + 65504.0) < 1


@pytest.mark.parametrize("B,C,M,L,KW", SHAPES)
def test_conv_f16mx_matches_fp32(B, C, M, L, KW, rows_tile):
    torch.manual_seed(1)
    x, b = torch.randn(B, C, L), torch.randn(M)
    w = torch.randn(M, C, KW) / (C * KW) ** 0.5
    ref = F.conv1d(x, w, b, padding=KW // 2)
    xn = fx(x)
    pk = N.pack_weight(w.cuda(), N.W_OIK, planes=3)
    assert rel(N.conv1d_bf16(xn, pk, b.cuda(), out_ncl=True), ref) < KERNEL
    y = N.conv1d_bf16(xn, pk, b.cuda())
    assert rel(y.to_ncl(), ref) < KERNEL
    wt = torch.randn(C, M, KW) / (C * KW) ** 0.5
    reft = F.conv_transpose1d(x, wt, b, padding=KW // 2)
    assert rel(N.conv1d_bf16(xn, N.pack_weight(wt.cuda(), N.W_IOK, planes=3), b.cuda(), out_ncl=True), reft) < KERNEL


def test_conv_output_saturates_instead_of_overflowing(rows_tile):
    """The conversions INTO the format saturate through the wave's MODE register (FP16_OVFL), with no clamp
    instructions: a convolution output beyond fp16's range is stored as +-65504 (+ the largest lo8), never inf / NaN,
    in the epilogue of both tiles as in the boundary conversion."""
    B, C, M, L = 1, 32, 64, 40
    x = torch.full((B, C, L), 300.0)
    w = torch.full((M, C, 1), 50.0)
    w[1::2] = -50.0
    y = N.conv1d_bf16(fx(x), N.pack_weight(w.cuda(), N.W_OIK, planes=3)).to_ncl().cpu()     # +-480 000
    assert torch.isfinite(y).all()
    assert float((y[:, 0::2] - 65504.0).abs().max()) < 1 and float((y[:, 1::2] + 65504.0).abs().max()) < 1
    h = N.conv1d_bf16(fx(x), N.pack_weight(w.cuda(), N.W_OIK, planes=3)).matrix(0).view(torch.float16)
    assert torch.isfinite(h.float()).all()


def test_conv_f16mx_epilogue_fusions(rows_tile):
    torch.manual_seed(2)
    B, C, M, L = 2, 24, 40, 50
    x, w, b = torch.randn(B, C, L), torch.randn(M, C, 3) / 8, torch.randn(M)
    s1, s2, mk, post = (torch.randn(B, M, L) for _ in range(4))
    v = F.relu(F.conv1d(x, w, b, padding=1) + s1 + s2)
    v = torch.where(mk > 0, v, torch.zeros_like(v))
    y, y2 = N.conv1d_bf16(fx(x), N.pack_weight(w.cuda(), N.W_OIK, planes=3), b.cuda(), fx(s1), fx(s2), fx(mk), fx(post), relu=True)
    assert rel(y.to_ncl(), v) < KERNEL and rel(y2.to_ncl(), v + post) < KERNEL
    # gap rows, tail rows and padded channels of an output stay exactly zero in both planes
    hm = y.matrix(0).view(torch.float16).float().cpu()
    qm = y.matrix(1).view(torch.int16).cpu()
    assert float(hm[0].abs().sum()) == 0 and float(hm[:, M:].abs().sum()) == 0 and float(hm[1 + B * (L + 1):].abs().sum()) == 0
    assert int(qm[0].abs().sum()) == 0 and int(qm[1 + B * (L + 1):].abs().sum()) == 0


@pytest.mark.parametrize("M,KW", [(1024, 3), (1000, 1), (72, 3)])
def test_relu_sign_bits_replace_the_mask_tensor_f16mx(M, KW, rows_tile):
    """A ReLU'd f16mx output leaves the sign bits of its H plane behind (one byte per 8 channels of a row); the masked
    launch that later needs "* (y > 0)" gives bit-identical results whether it reads those bits or the H plane."""
    torch.manual_seed(31)
    B, C, L = 3, 72, 140
    x = torch.randn(B, C, L)
    w = torch.randn(M, C, KW) / (C * KW) ** 0.5
    t = N.conv1d_bf16(fx(x), N.pack_weight(w.cuda(), N.W_OIK, planes=3), relu=True)
    assert t.has_bits
    bits = t.storage.view(torch.uint8)[2 * (t.rows + 2 * t.guard) * t.Cp * 2:][:t.rows * t.Cp // 8].view(t.rows, t.Cp // 8)
    want = (t.matrix(0).view(torch.float16).float() > 0).view(t.rows, t.Cp // 8, 8).to(torch.int32)
    want = (want << torch.arange(8, device="cuda", dtype=torch.int32)).sum(dim=2).to(torch.uint8)
    assert torch.equal(bits, want)
    dy = fx(torch.randn(B, M, L))
    pk = N.pack_weight(torch.randn(M, M, 1).cuda() / M ** 0.5, N.W_OIK, planes=3)
    got_bits = N.conv1d_bf16(dy, pk, mask=t)                       # t.has_bits -> passed as bits
    t_plain = N.NLC.wrap(t.storage, t.B, t.L, t.C, 2, has_bits=False, fmt="f16mx")
    got_tensor = N.conv1d_bf16(dy, pk, mask=t_plain)
    assert torch.equal(got_bits.matrix(0), got_tensor.matrix(0)) and torch.equal(got_bits.matrix(1), got_tensor.matrix(1))


@pytest.mark.parametrize("B,C,M,L,KW", SHAPES)
def test_wgrad_f16mx_matches_fp32(B, C, M, L, KW):
    torch.manual_seed(3)
    x = torch.randn(B, C, L, requires_grad=True)
    w = (torch.randn(M, C, KW) / (C * KW) ** 0.5).requires_grad_(True)
    b = torch.randn(M, requires_grad=True)
    dy = torch.randn(B, M, L)
    F.conv1d(x, w, b, padding=KW // 2).backward(dy)
    dyn, xn = fx(dy), fx(x.detach())
    dw, db = N.conv1d_wgrad_bf16(dyn, xn, KW, N.W_OIK, want_bias=True)
    assert rel(dw, w.grad) < KERNEL
    assert float((db.cpu() - b.grad).abs().max()) < 1e-4 * float(dy.abs().sum(dim=(0, 2)).max())
    wt = (torch.randn(C, M, KW) / (C * KW) ** 0.5).requires_grad_(True)
    F.conv_transpose1d(x.detach(), wt, None, padding=KW // 2).backward(dy)
    assert rel(N.conv1d_wgrad_bf16(dyn, xn, KW, N.W_IOK), wt.grad) < KERNEL


@pytest.mark.parametrize("KW", [1, 3])
@pytest.mark.parametrize("nseg", [1, 2, 3, 4])
def test_wgrad_f16mx_multi_sums_the_uses_of_a_shared_weight(nseg, KW):
    """One launch over nseg (dy, x) pairs == the sum of nseg single launches (residual_stack.py:40-41 shares one weight
    between the R layers); a loss scale common to the chain is divided out once; accumulate adds to what dw held."""
    torch.manual_seed(6)
    B, C, M, L = 3, 72, 136, 95
    mag = 1e-6
    xs = [torch.randn(B, C, L) for _ in range(nseg)]
    dys = [torch.randn(B, M, L) * mag for _ in range(nseg)]
    w = (torch.randn(M, C, KW) / (C * KW) ** 0.5).requires_grad_(True)
    for x, dy in zip(xs, dys):
        F.conv1d(x, w, None, padding=KW // 2).backward(dy)
    state = N.grad_scale(torch.stack(dys).cuda())
    pairs = [(fx(dy, state), fx(x)) for dy, x in zip(dys, xs)]
    dw = N.conv1d_wgrad_bf16_multi(pairs, KW, N.W_OIK)
    assert rel(dw, w.grad) < KERNEL
    singles = sum(N.conv1d_wgrad_bf16(dy, x, KW, N.W_OIK) for dy, x in pairs)
    assert rel(dw, singles) < 1e-6
    base = torch.randn(M, C, KW, device="cuda") * mag
    acc = N.conv1d_wgrad_bf16_multi(pairs, KW, N.W_OIK, dw_out=base.clone(), accumulate=True)
    assert rel(acc, base.cpu() + w.grad) < KERNEL
    other = N.grad_scale(dys[0].cuda())
    if nseg > 1:
        with pytest.raises(RuntimeError, match="loss-scale"):
            N.conv1d_wgrad_bf16_multi([pairs[0], (fx(dys[1], other), pairs[1][1])], KW)


@pytest.mark.parametrize("mag", [1e-9, 3e-5, 1.0, 4e4])
def test_loss_scale_carries_gradients_of_any_magnitude(mag):
    """A gradient chain enters the format multiplied by a power of two chosen on the device from its amax and leaves it
    divided by the same factor: data-gradient, weight-gradient and bias-gradient of a conv are as accurate for 1e-9
    gradients (far below fp16's range) as for O(1) ones."""
    torch.manual_seed(4)
    B, C, M, L = 2, 40, 72, 60
    x = torch.randn(B, C, L, requires_grad=True)
    w = (torch.randn(M, C, 3) / (C * 3) ** 0.5).requires_grad_(True)
    b = torch.randn(M, requires_grad=True)
    dy = torch.randn(B, M, L) * mag
    F.conv1d(x, w, b, padding=1).backward(dy)
    state = N.grad_scale(dy.cuda())
    S, invS = float(state[0]), float(state[1])
    assert S * invS == 1.0 and np.log2(S) == int(np.log2(S))              # an exact power of two
    assert 128.0 <= S * float(dy.abs().max()) < 256.0
    dyn = fx(dy, state)
    dx = N.conv1d_bf16(dyn, N.pack_weight(w.detach().cuda(), N.W_IOK, planes=3), out_ncl=True)   # leaves the format: / S
    assert rel(dx, x.grad) < KERNEL
    dw, db = N.conv1d_wgrad_bf16(dyn, fx(x.detach()), 3, N.W_OIK, want_bias=True)
    assert rel(dw, w.grad) < KERNEL and rel(db, b.grad) < KERNEL
    # the scale rides along through NLC -> NLC launches and the way out
    mid = N.conv1d_bf16(dyn, N.pack_weight(w.detach().cuda(), N.W_IOK, planes=3))
    assert mid.gscale is state and rel(N.nlc_to_ncl(mid), x.grad) < KERNEL
    assert rel(N.nlc_to_ncl(N.relu_mask_bf16(dyn, fx(x.new_ones(B, M, L)))), dy) < 1e-4


def test_range_flag_reports_inputs_outside_fp16s_range():
    """f16mx carries fp16's range (round-2 advisor finding: no overflow signal).  A value of magnitude >= 65504, or a NaN,
    entering the format raises a sticky device flag (and is stored saturated, not as inf); in-range data leaves it clear."""
    N.f16mx_range_flag(reset=True)
    x = torch.randn(2, 40, 33).cuda() * 100
    fx(x)
    assert N.f16mx_range_flag(reset=False) == 0
    x[1, 3, 5] = 7e4
    back = N.nlc_to_ncl(fx(x))
    assert N.f16mx_range_flag(reset=False) == 1 and 65504.0 <= float(back[1, 3, 5]) < 65505.0     # saturated, not inf
    assert N.f16mx_range_flag(reset=True) == 1 and N.f16mx_range_flag() == 0        # sticky until cleared
    x[1, 3, 5] = float("nan")
    fx(x)
    assert N.f16mx_range_flag() == 2
    # a gradient entering a backward chain is brought into range by its loss scale whatever its magnitude: no flag
    g = torch.randn(2, 40, 33).cuda() * 1e7
    fx(g, N.grad_scale(g))
    assert N.f16mx_range_flag() == 0
    # values PRODUCED inside a chain: a convolution whose output leaves fp16's range raises bit 2 (f16mx and fp16 kernels)
    x = torch.full((2, 64, 40), 200.0).cuda()
    w = torch.full((256, 64, 3), 1.0).cuda()                      # sums of 192 x 200 = 38 400 ... 76 800 at the edges / interior
    N.conv1d_bf16(fx(x * 0.1), N.pack_weight(w, N.W_OIK, 3))     # 3 840 .. 7 680: in range
    assert N.f16mx_range_flag() == 0
    y = N.conv1d_bf16(fx(x * 2), N.pack_weight(w, N.W_OIK, 3))    # 76 800 .. : saturates
    assert N.f16mx_range_flag() == 4 and float(N.nlc_to_ncl(y).max()) < 65505.0
    N.conv1d_bf16(N.ncl_to_nlc(x * 2, 1, "f16"), N.pack_weight(w, N.W_OIK, 3))
    assert N.f16mx_range_flag() == 4


def test_relu_mask_f16mx():
    torch.manual_seed(4)
    d, t = torch.randn(2, 70, 33), torch.randn(2, 70, 33)
    out = N.relu_mask_bf16(fx(d), fx(t))
    assert rel(out.to_ncl(), torch.where(t > 0, d, torch.zeros_like(d))) < 1e-4


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


CASES = [
    ((7, 16, 4, 2, 8, 0.25, 16), (2, 7, 13), dict(), False),
    ((20, 48, 8, 3, 24, 0.25, 64), (3, 20, 40), dict(use_jitter=False), False),
    ((50, 64, 8, 2, 16, 0.25, 64), (4, 24, 50), dict(use_jitter=False, out_channels=1), True),
    ((201, 128, 32, 2, 128, 0.25, 128), (2, 201, 96), dict(), False),
]


@pytest.mark.parametrize("cfg,shape,kw,permuted", CASES)
def test_module_forward_backward_matches_oracle(cfg, shape, kw, permuted, rows_tile):
    torch.manual_seed(11)
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
    xg = x.cuda().requires_grad_(True)
    np.random.seed(3)
    vq_loss, recon, perp = m(xg)
    (F.mse_loss(recon, target.cuda()) + vq_loss).backward()
    assert rel(m._latent(x.cuda()), out["z"]) < 3e-4
    full = {k: q.grad.detach().clone() for k, q in m.named_parameters()}      # gradients of the whole chain, encoder included
    # downstream of the quantiser: also on the ORACLE's codes (decoder fed the oracle's quantised latent), so these
    # comparisons do not depend on the codes chosen above
    m.zero_grad()
    np.random.seed(3)
    recon2 = m._decoder(out["q_st"].detach().cuda())
    F.mse_loss(recon2, target.cuda()).backward()
    assert rel(recon2, out["recon"]) < 1e-3
    _, _, _, idx = m.eval().get_latent_indices(x.cuda())
    bad = np.nonzero((idx.cpu() != out["idx"]).numpy())[0]
    d2 = torch.topk(O.vq_distances(out["z"].detach().reshape(-1, cfg[2]), p["_vq._embedding.weight"].detach()), 2, dim=1,
                    largest=False).values
    gap = ((d2[:, 1] - d2[:, 0]) / d2[:, 0].abs()).numpy()
    assert all(gap[i] < 1e-5 for i in bad), (bad, gap[bad])     # bit-exact except where the oracle's own top-2 nearly tie
    assert len(bad) == 0, bad
    named = dict(m.named_parameters())
    for k, v in p.items():
        # a ReLU gate whose pre-activation is within the forward noise of zero can flip; in nets this small one
        # flipped gate moves a gradient by ~1e-2 (the default-size goldens keep the tighter bar)
        if k.startswith("_decoder"):
            assert rel(named[k].grad, v.grad) < 3e-2, k
        assert rel(full[k], v.grad) < 3e-2, k                   # every parameter of the full chain: encoder, pre-VQ conv, codebook
    assert rel(recon, out["recon"]) < 1e-3 and rel(vq_loss, out["vq_loss"]) < 1e-4
    assert xg.grad.shape == x.shape
