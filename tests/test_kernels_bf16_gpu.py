"""bf16 throughput path, kernel level: the HIP kernels against a PyTorch-CPU statement that uses the SAME
bf16-rounded operands with fp32 accumulation.  fp32 outputs agree to accumulation order (1e-5); bf16 outputs to
one bf16 rounding (2^-8)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from acoustic_locating_vq_vae import _native as N  # noqa: E402


def bf(t):
    return t.to(torch.bfloat16).float()


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def close_bf16(got, ref):
    """|got - ref| <= one bf16 ulp of ref (plus accumulation-order slack)."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    tol = ref.abs() * 2.0 ** -7 + 1e-6 * float(ref.abs().max())
    return bool(((got - ref).abs() <= tol).all())


SHAPES = [(2, 7, 16, 13, 3), (2, 16, 7, 13, 3), (2, 8, 16, 13, 1), (3, 5, 1, 201, 3), (2, 201, 1024, 500, 3),
          (2, 1024, 128, 500, 3), (2, 1024, 1024, 201, 1), (2, 500, 1024, 201, 3), (2, 192, 1024, 77, 3),
          (2, 1024, 201, 500, 3), (5, 130, 130, 129, 3), (1, 64, 64, 1, 3),
          # wide layers (256 x 256-tile kernels): the main 1024 -> 1024 width-3 shape, a ragged M inside the last m-tile
          # (1000 of 1024, 480 of 512), ragged C, more rows than one tile and fewer
          (2, 1024, 1024, 300, 3), (2, 96, 1000, 150, 3), (2, 64, 1000, 150, 1), (2, 70, 480, 90, 3), (3, 201, 2048, 40, 3)]


def test_layout_roundtrip_and_zero_padding():
    torch.manual_seed(0)
    x = torch.randn(3, 201, 37)
    n = N.ncl_to_nlc(x.cuda())
    assert n.Cp == 256 and n.rows == 256 and n.guard == 8
    assert torch.equal(n.to_ncl().cpu(), bf(x))
    m = n.matrix().float().cpu()
    assert float(m[0].abs().sum()) == 0 and float(m[38].abs().sum()) == 0           # row 0 and the gap after sample 0
    assert float(m[1 + 3 * 38:].abs().sum()) == 0 and float(m[:, 201:].abs().sum()) == 0


@pytest.mark.parametrize("B,C,M,L,KW", SHAPES)
def test_conv_bf16_forward_both_layouts_and_outputs(B, C, M, L, KW):
    torch.manual_seed(1)
    x, b = torch.randn(B, C, L), torch.randn(M)
    w = torch.randn(M, C, KW) / (C * KW) ** 0.5
    ref = F.conv1d(bf(x), bf(w), b, padding=KW // 2)
    xn = N.ncl_to_nlc(x.cuda())
    pk = N.pack_weight(w.cuda(), N.W_OIK)
    y32 = N.conv1d_bf16(xn, pk, b.cuda(), out_ncl=True)
    assert rel(y32, ref) < 2e-5
    y = N.conv1d_bf16(xn, pk, b.cuda())
    assert close_bf16(y.to_ncl(), ref)
    mat = y.matrix().float().cpu()
    assert float(mat[0].abs().sum()) == 0 and float(mat[:, M:].abs().sum()) == 0 and float(mat[1 + B * (L + 1):].abs().sum()) == 0
    assert float(mat[L + 1].abs().sum()) == 0                                        # gap rows stay zero
    wt = torch.randn(C, M, KW) / (C * KW) ** 0.5
    reft = F.conv_transpose1d(bf(x), bf(wt), b, padding=KW // 2)
    yt = N.conv1d_bf16(xn, N.pack_weight(wt.cuda(), N.W_IOK), b.cuda(), out_ncl=True)
    assert rel(yt, reft) < 2e-5


def test_conv_bf16_epilogue_fusions():
    torch.manual_seed(2)
    B, C, M, L = 2, 24, 40, 50
    x, w, b = torch.randn(B, C, L), torch.randn(M, C, 3) / 8, torch.randn(M)
    s1, s2, mk, post = (torch.randn(B, M, L) for _ in range(4))
    acc = F.conv1d(bf(x), bf(w), b, padding=1) + bf(s1) + bf(s2)
    v = F.relu(acc)
    v = torch.where(bf(mk) > 0, v, torch.zeros_like(v))
    cu = lambda t: N.ncl_to_nlc(t.cuda())
    y, y2 = N.conv1d_bf16(cu(x), N.pack_weight(w.cuda(), N.W_OIK), b.cuda(), cu(s1), cu(s2), cu(mk), cu(post), relu=True)
    assert close_bf16(y.to_ncl(), v) and close_bf16(y2.to_ncl(), v + bf(post))


@pytest.mark.parametrize("M,KW", [(1024, 3), (1024, 1), (1000, 3), (480, 1)])
def test_conv_bf16_epilogue_fusions_wide_layers(M, KW):
    """The register-direct epilogue of the 256 x 256-tile kernels (permlane swap, 16-byte I/O): bias, two skips, ReLU,
    ReLU-mask and the second output, on full and ragged last m-tiles."""
    torch.manual_seed(21)
    B, C, L = 3, 72, 140
    x, w, b = torch.randn(B, C, L), torch.randn(M, C, KW) / (C * KW) ** 0.5, torch.randn(M)
    s1, s2, mk, post = (torch.randn(B, M, L) for _ in range(4))
    acc = F.conv1d(bf(x), bf(w), b, padding=KW // 2) + bf(s1) + bf(s2)
    v = F.relu(acc)
    v = torch.where(bf(mk) > 0, v, torch.zeros_like(v))
    cu = lambda t: N.ncl_to_nlc(t.cuda())
    y, y2 = N.conv1d_bf16(cu(x), N.pack_weight(w.cuda(), N.W_OIK), b.cuda(), cu(s1), cu(s2), cu(mk), cu(post), relu=True)
    assert close_bf16(y.to_ncl(), v) and close_bf16(y2.to_ncl(), v + bf(post))
    for out in (y, y2):                                                     # zero gap / tail rows and padded channels
        mat = out.matrix().float().cpu()
        assert float(mat[0].abs().sum()) == 0 and float(mat[L + 1].abs().sum()) == 0
        assert float(mat[1 + B * (L + 1):].abs().sum()) == 0 and float(mat[:, M:].abs().sum()) == 0


@pytest.mark.parametrize("B,C,M,L,KW", SHAPES)
def test_wgrad_bf16(B, C, M, L, KW):
    torch.manual_seed(3)
    x = bf(torch.randn(B, C, L)).requires_grad_(True)
    w = (torch.randn(M, C, KW) / (C * KW) ** 0.5).requires_grad_(True)
    b = torch.randn(M, requires_grad=True)
    dy = bf(torch.randn(B, M, L))
    F.conv1d(x, w, b, padding=KW // 2).backward(dy)
    dw, db = N.conv1d_wgrad_bf16(N.ncl_to_nlc(dy.cuda()), N.ncl_to_nlc(x.detach().cuda()), KW, N.W_OIK, want_bias=True)
    assert rel(dw, w.grad) < 3e-5
    assert float((db.cpu() - b.grad).abs().max()) < 2e-6 * float(dy.abs().sum(dim=(0, 2)).max())
    dw2 = N.conv1d_wgrad_bf16(N.ncl_to_nlc(dy.cuda()), N.ncl_to_nlc(x.detach().cuda()), KW, N.W_OIK, dw_out=dw.clone(), accumulate=True)
    assert rel(dw2, 2 * w.grad) < 3e-5
    # ConvTranspose weight layout
    wt = (torch.randn(C, M, KW) / (C * KW) ** 0.5).requires_grad_(True)
    F.conv_transpose1d(x.detach(), wt, None, padding=KW // 2).backward(dy)
    dwt = N.conv1d_wgrad_bf16(N.ncl_to_nlc(dy.cuda()), N.ncl_to_nlc(x.detach().cuda()), KW, N.W_IOK)
    assert rel(dwt, wt.grad) < 3e-5


def test_relu_mask_bf16():
    torch.manual_seed(4)
    d, t = torch.randn(2, 70, 33), torch.randn(2, 70, 33)
    out = N.relu_mask_bf16(N.ncl_to_nlc(d.cuda()), N.ncl_to_nlc(t.cuda()))
    assert torch.equal(out.to_ncl().cpu(), torch.where(bf(t) > 0, bf(d), torch.zeros_like(d)))


def test_wgrad_bf16_multi_segment_sums_uses():
    """Shared residual weights: sum_i wgrad(dy_i, x_i) in one launch == the sum of the single launches."""
    torch.manual_seed(6)
    B, C, M, L = 2, 96, 160, 77
    for KW in (1, 3):
        pairs, want = [], 0
        for _ in range(3):
            x, dy = bf(torch.randn(B, C, L)), bf(torch.randn(B, M, L))
            w = torch.zeros(M, C, KW, requires_grad=True)
            F.conv1d(x, w, None, padding=KW // 2).backward(dy)
            want = want + w.grad
            pairs.append((N.ncl_to_nlc(dy.cuda()), N.ncl_to_nlc(x.cuda())))
        got = N.conv1d_wgrad_bf16_multi(pairs, KW)
        assert rel(got, want) < 3e-5
        acc = N.conv1d_wgrad_bf16_multi(pairs[:2], KW, dw_out=got.clone(), accumulate=True)
        single = N.conv1d_wgrad_bf16(pairs[0][0], pairs[0][1], KW) + N.conv1d_wgrad_bf16(pairs[1][0], pairs[1][1], KW)
        assert rel(acc, got + single) < 3e-5


@pytest.mark.parametrize("planes", [1, 2])
def test_pack_weights_batch_matches_single_packs(planes):
    """One batched launch (alvq_pack_weights_bf16_batch) == the per-weight packs, bit for bit: both layouts, widths 1
    and 3, ragged M / C (zero padding included), more descriptors than one launch holds (32)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    shapes = [(1024, 201, 3), (128, 1024, 1), (201, 1024, 3), (64, 64, 1), (33, 70, 3), (256, 128, 3), (1, 1024, 1)]
    entries, singles = [], []
    for rep in range(3):                                  # 7 shapes x 2 layouts x 3 = 42 descriptors
        for (M, C, KW) in shapes:
            for layout in (N.W_OIK, N.W_IOK):
                w = torch.randn((M, C, KW) if layout == N.W_OIK else (C, M, KW), device="cuda", generator=g)
                img, tag = N.packed_weight_alloc(w, layout, planes)
                img.fill_(float("nan"))                   # every element must be written, padding included
                entries.append((w, img, layout))
                singles.append(N.pack_weight(w, layout, planes))
    N.pack_weights_batch(entries, planes)
    for (w, img, layout), (ref, tag) in zip(entries, singles):
        assert tag[3] == planes
        assert torch.equal(img.view(torch.int16), ref.view(torch.int16)), (tuple(w.shape), layout)


@pytest.mark.parametrize("M,KW", [(1024, 3), (1000, 1), (72, 3)])
def test_relu_sign_bits_replace_the_mask_tensor(M, KW):
    """A ReLU'd output leaves its sign bits behind (one byte per 8 channels of a row); the masked launch that later needs
    "* (y > 0)" gives bit-identical results whether it reads those bits or the tensor itself -- wide and narrow kernels."""
    torch.manual_seed(31)
    B, C, L = 3, 72, 140
    x = torch.randn(B, C, L)
    w = torch.randn(M, C, KW) / (C * KW) ** 0.5
    t = N.conv1d_bf16(N.ncl_to_nlc(x.cuda()), N.pack_weight(w.cuda(), N.W_OIK), relu=True)
    assert t.has_bits
    # the bits are the signs of the stored tensor, padding and gap rows included
    bits = t.storage.view(torch.uint8)[(t.rows + 2 * t.guard) * t.Cp * 2:][:t.rows * t.Cp // 8].view(t.rows, t.Cp // 8)
    want = (t.matrix().float() > 0).view(t.rows, t.Cp // 8, 8).to(torch.int32)
    want = (want << torch.arange(8, device="cuda", dtype=torch.int32)).sum(dim=2).to(torch.uint8)
    assert torch.equal(bits, want)
    dy = N.ncl_to_nlc(torch.randn(B, M, L).cuda())
    wd = torch.randn(M, M, 1).cuda() / M ** 0.5
    pk = N.pack_weight(wd, N.W_OIK)
    got_bits = N.conv1d_bf16(dy, pk, mask=t)                       # t.has_bits -> passed as bits
    t_plain = N.NLC.wrap(t.storage, t.B, t.L, t.C, 1, has_bits=False)
    got_tensor = N.conv1d_bf16(dy, pk, mask=t_plain)
    assert torch.equal(got_bits.matrix(), got_tensor.matrix())
