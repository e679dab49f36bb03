"""fp16 element type of the 16-bit kernels (the backward pass of the f16mx_hb mode) at kernel level: the HIP kernels
against a PyTorch-CPU statement that uses the SAME fp16-rounded operands with fp32 accumulation -- fp32 outputs agree to
accumulation order, fp16 outputs to one fp16 rounding (2^-11) -- through every kernel of the family (128 x 128, the two
256 x 256 forms, both weight-gradient generations), with plain fp16 operands and with the H planes of f16mx tensors / f16mx
packed weights (what the mode actually feeds them), under a loss scale."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from acoustic_locating_vq_vae import _native as N  # noqa: E402


def hf(t):
    return t.to(torch.float16).float()


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def close_f16(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    tol = ref.abs() * 2.0 ** -10 + 1e-6 * float(ref.abs().max())
    return bool(((got - ref).abs() <= tol).all())


def h(t, gs=None):
    return N.ncl_to_nlc(t.cuda(), 1, "f16", gs)


def fx(t, gs=None):
    return N.ncl_to_nlc(t.cuda(), 2, "f16mx", gs)


SHAPES = [(2, 7, 16, 13, 3), (2, 8, 16, 13, 1), (3, 5, 1, 201, 3), (2, 201, 1024, 500, 3), (2, 1024, 128, 500, 3),
          (2, 1024, 1024, 201, 1), (2, 1024, 201, 500, 3), (5, 130, 130, 129, 3), (2, 1024, 1024, 300, 3), (2, 96, 1000, 150, 3),
          (2, 64, 1000, 150, 1)]


def test_layout_roundtrip_scale_and_padding():
    torch.manual_seed(0)
    x = torch.randn(3, 201, 37) * 1e-6
    gs = N.grad_scale(x.cuda())
    S = float(gs[0])
    n = h(x, gs)
    assert (n.planes, n.fmt, n.Cp, n.rows) == (1, "f16", 256, 256)
    assert rel(N.nlc_to_ncl(n), x) < 2.0 ** -10                  # scaled into fp16's range, rounded once, scaled back
    m = n.matrix(0).view(torch.float16).float().cpu()
    assert torch.equal(m[1:38, :201], (x[0].t() * S).half().float())
    assert float(m[0].abs().sum()) == 0 and float(m[38].abs().sum()) == 0 and float(m[:, 201:].abs().sum()) == 0
    big = torch.tensor([[[1e9, -1e9, 3.0]]])
    assert N.nlc_to_ncl(h(big)).flatten().tolist() == [65504.0, -65504.0, 3.0]     # saturating, never inf


@pytest.mark.parametrize("operands", ["f16", "f16mx_h_planes"])
@pytest.mark.parametrize("B,C,M,L,KW", SHAPES)
def test_conv_f16_both_layouts_and_outputs(B, C, M, L, KW, operands):
    torch.manual_seed(1)
    x, b = torch.randn(B, C, L), torch.randn(M)
    w = torch.randn(M, C, KW) / (C * KW) ** 0.5
    ref = F.conv1d(hf(x), hf(w), b, padding=KW // 2)
    xn = h(x) if operands == "f16" else None
    pk = N.pack_weight(w.cuda(), N.W_OIK, 3)                      # f16mx packed weight: its H image is the fp16 weight
    if operands == "f16":
        y32 = N.conv1d_bf16(xn, pk, b.cuda(), out_ncl=True)
        assert rel(y32, ref) < 2e-5
        y = N.conv1d_bf16(xn, pk, b.cuda())
        assert y.fmt == "f16" and close_f16(y.to_ncl(), ref)
        mat = y.matrix(0).view(torch.float16).float().cpu()
        assert float(mat[0].abs().sum()) == 0 and float(mat[:, M:].abs().sum()) == 0 and float(mat[1 + B * (L + 1):].abs().sum()) == 0
    wt = torch.randn(C, M, KW) / (C * KW) ** 0.5
    reft = F.conv_transpose1d(hf(x), hf(wt), None, padding=KW // 2)
    # skip operand given as fp16 or as an f16mx tensor (read through its H plane)
    s1 = torch.randn(B, M, L)
    skip = h(s1) if operands == "f16" else fx(s1)
    yt = N.conv1d_bf16(h(x), N.pack_weight(wt.cuda(), N.W_IOK, 3), skip1=skip)
    assert close_f16(yt.to_ncl(), reft + hf(s1))


@pytest.mark.parametrize("M,KW", [(1024, 3), (1024, 1), (1000, 3), (40, 3)])
def test_conv_f16_epilogue_fusions_and_loss_scale(M, KW):
    torch.manual_seed(21)
    B, C, L = 3, 72, 140
    mag = 3e-7                                                    # gradients far below fp16's range: the loss scale carries them
    x, w = torch.randn(B, C, L) * mag, torch.randn(M, C, KW) / (C * KW) ** 0.5
    s1, s2, post = (torch.randn(B, M, L) * mag for _ in range(3))
    mk = torch.randn(B, M, L)
    gs = N.grad_scale(x.cuda())
    S = float(gs[0])
    hs = lambda t: (t * S).half().float() / S                     # what the scaled fp16 plane holds
    acc = F.conv1d(hs(x), hf(w), None, padding=KW // 2) + hs(s1) + hs(s2)
    v = torch.where(mk > 0, acc, torch.zeros_like(acc))
    pk = N.pack_weight(w.cuda(), N.W_OIK, 3)
    t = N.conv1d_bf16(fx(mk), N.pack_weight(torch.eye(M).view(M, M, 1).contiguous().cuda(), N.W_OIK, 3), relu=True)   # carries sign bits
    assert t.has_bits
    for mask in (fx(mk), t):                                      # mask as a tensor (H plane) and as sign bits
        y, y2 = N.conv1d_bf16(h(x, gs), pk, None, h(s1, gs), h(s2, gs), mask, h(post, gs))
        assert y.gscale is gs and close_f16(N.nlc_to_ncl(y) * S, v * S) and close_f16(N.nlc_to_ncl(y2) * S, (v + hs(post)) * S)
    out = N.conv1d_bf16(h(x, gs), pk, None, out_ncl=True)         # leaves the format: divided by S in the epilogue
    assert rel(out, F.conv1d(hs(x), hf(w), None, padding=KW // 2)) < 2e-5
    ry = N.conv1d_bf16(h(x * 0 + 1.0), pk, None, relu=True)       # relu'd fp16 output leaves its sign bits behind
    assert ry.has_bits


@pytest.mark.parametrize("operands", ["f16", "f16mx_h_planes"])
@pytest.mark.parametrize("B,C,M,L,KW", SHAPES)
def test_wgrad_f16(B, C, M, L, KW, operands):
    torch.manual_seed(3)
    mag = 1e-5
    x = hf(torch.randn(B, C, L)).requires_grad_(True)
    w = (torch.randn(M, C, KW) / (C * KW) ** 0.5).requires_grad_(True)
    b = torch.randn(M, requires_grad=True)
    dy = torch.randn(B, M, L) * mag
    gs = N.grad_scale(dy.cuda())
    S = float(gs[0])
    dyh = (dy * S).half().float() / S
    F.conv1d(x, w, b, padding=KW // 2).backward(dyh)
    xn = h(x.detach()) if operands == "f16" else fx(x.detach())
    dw, db = N.conv1d_wgrad_bf16(h(dy, gs), xn, KW, N.W_OIK, want_bias=True)
    assert rel(dw, w.grad) < 3e-5 and rel(db, b.grad) < 3e-5
    dw2 = N.conv1d_wgrad_bf16(h(dy, gs), xn, KW, N.W_OIK, dw_out=dw.clone(), accumulate=True)
    assert rel(dw2, 2 * w.grad) < 3e-5
    wt = (torch.randn(C, M, KW) / (C * KW) ** 0.5).requires_grad_(True)
    F.conv_transpose1d(x.detach(), wt, None, padding=KW // 2).backward(dyh)
    assert rel(N.conv1d_wgrad_bf16(h(dy, gs), xn, KW, N.W_IOK), wt.grad) < 3e-5
    # the R uses of a shared residual weight in one launch (no bias: the v3 kernels)
    multi = N.conv1d_wgrad_bf16_multi([(h(dy, gs), xn)] * 3, KW, N.W_OIK)
    w.grad = None
    for _ in range(3):
        F.conv1d(x, w, None, padding=KW // 2).backward(dyh)
    assert rel(multi, w.grad) < 3e-5


def test_relu_mask_f16_with_an_f16mx_mask():
    torch.manual_seed(4)
    d, t = torch.randn(2, 70, 33), torch.randn(2, 70, 33)
    out = N.relu_mask_bf16(h(d), fx(t))
    assert out.fmt == "f16" and torch.equal(out.to_ncl().cpu(), torch.where(t > 0, hf(d), torch.zeros_like(d)))
