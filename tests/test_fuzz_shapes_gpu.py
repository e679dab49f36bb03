"""Seeded shape fuzz: 40 random (B, C, M, L, KW) problems per precision through forward, data-gradient and
weight-gradient, against torch's CPU convolution on operands rounded to what the kernels store.  Shapes are drawn to
straddle every dispatch boundary (narrow / 256-tile kernels, ragged channel counts, single-row problems, several row
tiles, M within and beyond the last m-tile)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from acoustic_locating_vq_vae import _native as N  # noqa: E402


def draw(rng):
    C = int(rng.choice([1, 3, 7, 20, 64, 65, 130, 201, 256, 500]))
    M = int(rng.choice([1, 5, 64, 100, 128, 201, 230, 256, 300, 480, 512, 1000]))
    return int(rng.integers(1, 5)), C, M, int(rng.choice([1, 2, 13, 77, 201, 340])), int(rng.choice([1, 3]))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


CASES = [draw(np.random.default_rng(1000 + i)) for i in range(40)]


@pytest.mark.parametrize("B,C,M,L,KW", CASES)
def test_fuzz_f32(B, C, M, L, KW):
    g = torch.Generator().manual_seed(B * 7919 + C * 31 + M * 17 + L * 3 + KW)
    x = torch.randn(B, C, L, generator=g).requires_grad_(True)
    w = (torch.randn(M, C, KW, generator=g) / (C * KW) ** 0.5).requires_grad_(True)
    b = torch.randn(M, generator=g, requires_grad=True)
    dy = torch.randn(B, M, L, generator=g)
    y = F.conv1d(x, w, b, padding=KW // 2)
    y.backward(dy)
    xd, wd, bd, dyd = x.detach().cuda(), w.detach().cuda(), b.detach().cuda(), dy.cuda()
    assert rel(N.conv1d(xd, wd, bd), y) < 2e-5
    assert rel(N.conv1d(dyd, wd, w_layout=N.W_IOK), x.grad) < 2e-5
    dw, db = N.conv1d_wgrad(dyd, xd, KW, want_bias=True)
    assert rel(dw, w.grad) < 5e-5 and rel(db, b.grad) < 5e-5 + 1e-6 * float(dy.abs().sum()) / (float(b.grad.abs().max()) + 1e-30)


@pytest.mark.parametrize("planes,tol", [(1, 3e-5), (2, 6e-5)])
@pytest.mark.parametrize("B,C,M,L,KW", CASES)
def test_fuzz_bf16(B, C, M, L, KW, planes, tol):
    g = torch.Generator().manual_seed(B * 7919 + C * 31 + M * 17 + L * 3 + KW + planes)
    rnd = (lambda t: t.bfloat16().float()) if planes == 1 else (lambda t: t)
    x = rnd(torch.randn(B, C, L, generator=g)).requires_grad_(True)
    w = rnd(torch.randn(M, C, KW, generator=g) / (C * KW) ** 0.5).requires_grad_(True)
    b = torch.randn(M, generator=g, requires_grad=True)
    dy = rnd(torch.randn(B, M, L, generator=g))
    y = F.conv1d(x, w, b, padding=KW // 2)
    y.backward(dy)
    xn, dyn = N.ncl_to_nlc(x.detach().cuda(), planes), N.ncl_to_nlc(dy.cuda(), planes)
    wd = w.detach().cuda()
    assert rel(N.conv1d_bf16(xn, N.pack_weight(wd, N.W_OIK, planes), b.detach().cuda(), out_ncl=True), y) < tol
    assert rel(N.conv1d_bf16(dyn, N.pack_weight(wd, N.W_IOK, planes), out_ncl=True), x.grad) < tol
    dw, db = N.conv1d_wgrad_bf16(dyn, xn, KW, want_bias=True)
    assert rel(dw, w.grad) < 2 * tol
    # without a bias gradient the bf16 launch takes the 32x32-MFMA kernels (256 x 256 tile for width 1)
    assert rel(N.conv1d_wgrad_bf16(dyn, xn, KW), w.grad) < 2 * tol
    # column sums of dy as stored: exact bf16 values (planes = 1) or hi + lo pairs carrying 2^-17 of each element
    assert float((db.cpu() - b.grad).abs().max()) <= (2e-6 if planes == 1 else 2e-5) * float(dy.abs().sum(dim=(0, 2)).max()) + 1e-30
    # the bf16 NLC output keeps its gap / tail rows and padded channels at zero for any shape
    out = N.conv1d_bf16(xn, N.pack_weight(wd, N.W_OIK, planes), b.detach().cuda(), relu=True)
    mat = out.storage.view(out.planes, -1, out.Cp)[:, out.guard:out.guard + out.rows].float()
    gaps = torch.arange(0, B * (L + 1) + 1, L + 1, device="cuda")
    assert float(mat[:, gaps].abs().max()) == 0.0
    if B * (L + 1) + 1 < out.rows:
        assert float(mat[:, B * (L + 1) + 1:].abs().max()) == 0.0
    if M < out.Cp:
        assert float(mat[:, :, M:].abs().max()) == 0.0


@pytest.fixture
def fx_rows(request):
    prev = N.set_option("fx_rows", int(request.param))
    yield request.param
    N.set_option("fx_rows", prev)


@pytest.mark.parametrize("fx_rows", ["128", "256"], indirect=True)
@pytest.mark.parametrize("B,C,M,L,KW", CASES)
def test_fuzz_f16mx(B, C, M, L, KW, fx_rows):
    """The same 40 shapes through the f16mx kernels (fp16 + block-scaled fp8 MFMA per product), both row tiles of the
    convolution: ~1.5e-5 per product against the fp32 result; gap / tail rows and padded channels of both planes zero;
    the sign bits a ReLU'd output leaves behind are the signs of its H plane."""
    tol = 1.5e-4
    g = torch.Generator().manual_seed(B * 7919 + C * 31 + M * 17 + L * 3 + KW + 5)
    x = torch.randn(B, C, L, generator=g).requires_grad_(True)
    w = (torch.randn(M, C, KW, generator=g) / (C * KW) ** 0.5).requires_grad_(True)
    b = torch.randn(M, generator=g, requires_grad=True)
    dy = torch.randn(B, M, L, generator=g)
    y = F.conv1d(x, w, b, padding=KW // 2)
    y.backward(dy)
    xn, dyn = N.ncl_to_nlc(x.detach().cuda(), 2, "f16mx"), N.ncl_to_nlc(dy.cuda(), 2, "f16mx")
    wd = w.detach().cuda()
    assert rel(N.conv1d_bf16(xn, N.pack_weight(wd, N.W_OIK, 3), b.detach().cuda(), out_ncl=True), y) < tol
    assert rel(N.conv1d_bf16(dyn, N.pack_weight(wd, N.W_IOK, 3), out_ncl=True), x.grad) < tol
    dw, db = N.conv1d_wgrad_bf16(dyn, xn, KW, want_bias=True)
    assert rel(dw, w.grad) < 2 * tol
    assert float((db.cpu() - b.grad).abs().max()) <= 1e-4 * float(dy.abs().sum(dim=(0, 2)).max()) + 1e-30
    out = N.conv1d_bf16(xn, N.pack_weight(wd, N.W_OIK, 3), b.detach().cuda(), relu=True)
    assert rel(out.to_ncl(), F.relu(y)) < tol
    hm = out.matrix(0).view(torch.float16).float()
    qm = out.matrix(1).view(torch.int16)
    gaps = torch.arange(0, B * (L + 1) + 1, L + 1, device="cuda")
    assert float(hm[gaps].abs().max()) == 0.0 and int(qm[gaps].abs().max()) == 0
    if B * (L + 1) + 1 < out.rows:
        assert float(hm[B * (L + 1) + 1:].abs().max()) == 0.0 and int(qm[B * (L + 1) + 1:].abs().max()) == 0
    if M < out.Cp:
        assert float(hm[:, M:].abs().max()) == 0.0
    assert out.has_bits
    bits = out.storage.view(torch.uint8)[2 * (out.rows + 2 * out.guard) * out.Cp * 2:][:out.rows * out.Cp // 8].view(out.rows, out.Cp // 8)
    want = (hm > 0).view(out.rows, out.Cp // 8, 8).to(torch.int32)
    want = (want << torch.arange(8, device="cuda", dtype=torch.int32)).sum(dim=2).to(torch.uint8)
    assert torch.equal(bits, want)


VQ_CASES = [(int(r.choice([1, 63, 64, 65, 777, 4097])), int(r.choice([1, 2, 16, 100, 513, 1024])),
             int(r.choice([1, 4, 31, 64, 100, 128, 200, 256, 300, 512])))
            for r in (np.random.default_rng(2000 + i) for i in range(30))]


@pytest.mark.parametrize("n,K,D", VQ_CASES)
def test_fuzz_vq(n, K, D):
    """Nearest-code search, gather/loss and backward on random (rows, codes, dims): the chosen code's distance (the
    reference's fl(fl(|x|^2 + |e|^2) - 2 x.e) arithmetic, CPU oracle) is the row minimum or within 2 ulp of it, ties go
    to the lowest index, and the quantiser's loss / gradients match the oracle."""
    from oracle import vqvae_oracle as O
    g = torch.Generator().manual_seed(n * 31 + K * 7 + D)
    x, e = torch.randn(n, D, generator=g), torch.randn(K, D, generator=g) * 0.8
    d = O.vq_distances(x, e)
    ref = torch.argmin(d, dim=1)
    got = N.vq_argmin(x.cuda(), e.cuda()).cpu()
    for r in (got != ref).nonzero().view(-1).tolist():
        dm, dg = d[r, ref[r]], d[r, got[r]]
        ulp = float(torch.nextafter(dm.abs(), torch.tensor(float("inf"))) - dm.abs())
        assert float(dg - dm) <= 2 * ulp, (r, float(dm), float(dg))
    assert int((got != ref).sum()) <= max(1, n // 500)
    q_st, out = N.vq_gather_loss(x.cuda(), e.cuda(), got.cuda(), 0.25)
    m = float(((e[got] - x) ** 2).double().mean())
    assert abs(float(out[0]) - 1.25 * m) <= 1e-5 * 1.25 * m + 1e-12
    gq, gl = torch.randn(n, D, generator=g), torch.tensor([0.7])
    dx, dE = N.vq_backward(gq.cuda(), gl.cuda(), x.cuda(), e.cuda(), got.cuda(), 0.25)
    diff = (e[got] - x).double()
    want_dx = gq.double() - 0.7 * (2 * 0.25 / (n * D)) * diff
    want_dE = torch.zeros(K, D, dtype=torch.float64).index_add_(0, got, diff) * (0.7 * 2.0 / (n * D))
    assert rel(dx, want_dx) < 1e-6 and rel(dE, want_dE) < 1e-5
