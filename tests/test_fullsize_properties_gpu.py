"""Size-independent properties at BASELINE.json's FULL sizes (speech config: B=64, 201/1024 channels, L=500).

The oracle cannot run these sizes in test time, so the kernels are pinned here through identities that hold for
any correct implementation, every one of them evaluated on the full-size launch geometry (hundreds of tiles, gap rows
between all 64 samples, ragged channel counts, split reductions):

* adjointness      <conv(x; W), dy> == <x, dgrad(dy; W)>             (forward vs data-gradient / ConvTranspose)
* weight-gradient  <wgrad(dy, x), V> == <conv(x; V), dy>             (weight-gradient vs forward)
* linearity        conv(a*x1 + x2) == a*conv(x1) + conv(x2)
* sample independence: permuting the batch permutes the output, bit for bit (a halo never crosses a sample)
* quantiser idempotence and optimality (sampled rows against the CPU oracle), histogram checksum
* layout round trips, jitter = column gather, standardise moments, Adam fixed point
* one full train step: finite, sample-permutation invariant loss, identical when repeated from identical state
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from acoustic_locating_vq_vae import _native as N  # noqa: E402
from oracle import vqvae_oracle as O  # noqa: E402

B, L = 64, 500
SHAPES = [(201, 1024, 3), (1024, 1024, 3), (1024, 1024, 1), (1024, 128, 3), (1024, 201, 3)]   # (C, M, KW) of the speech net


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, device="cuda", generator=g) * scale


def dot(a, b):
    return float((a.double() * b.double()).sum())


def close(lhs, rhs, tol):
    assert abs(lhs - rhs) <= tol * max(abs(lhs), abs(rhs), 1e-30), (lhs, rhs, abs(lhs - rhs) / max(abs(lhs), abs(rhs)))


# ------------------------------------------------------------------------------------------------- fp32 convolutions
@pytest.mark.parametrize("C,M,KW", SHAPES)
def test_f32_conv_adjoint_wgrad_linearity_fullsize(C, M, KW):
    x, dy = rnd(B, C, L, seed=1), rnd(B, M, L, seed=2)
    W = rnd(M, C, KW, seed=3, scale=(C * KW) ** -0.5)
    V = rnd(M, C, KW, seed=4, scale=(C * KW) ** -0.5)
    y = N.conv1d(x, W)
    dx = N.conv1d(dy, W, w_layout=N.W_IOK)                   # the data-gradient launch (= ConvTranspose1d forward)
    assert dx.shape == x.shape
    close(dot(y, dy), dot(x, dx), 2e-5)
    dW = N.conv1d_wgrad(dy, x, KW)
    close(dot(dW, V), dot(N.conv1d(x, V), dy), 2e-5)
    x2 = rnd(B, C, L, seed=5)
    lin = N.conv1d(x * 0.5 + x2, W)
    ref = y * 0.5 + N.conv1d(x2, W)
    assert float((lin - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_f32_conv_sample_independence_fullsize():
    C, M, KW = 1024, 1024, 3
    x = rnd(B, C, L, seed=6)
    W, b = rnd(M, C, KW, seed=7, scale=(C * KW) ** -0.5), rnd(M, seed=8)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(9)).cuda()
    y = N.conv1d(x, W, b, relu=True)
    yp = N.conv1d(x[perm].contiguous(), W, b, relu=True)
    assert torch.equal(yp, y[perm])


# ------------------------------------------------------------------------------------------------- bf16 convolutions
def _bf(x):
    return x.bfloat16().float()


@pytest.mark.parametrize("planes,tol", [(1, 1e-4), (2, 5e-4)])
@pytest.mark.parametrize("C,M,KW", SHAPES)
def test_bf16_conv_adjoint_wgrad_fullsize(C, M, KW, planes, tol):
    """planes=1: bf16 kernels (k3 / v2 / narrow, by shape); planes=2: split-bf16.  Operands are pre-rounded to what the
    kernels store, outputs leave through the fp32 epilogue, so only accumulation order (planes=1) or the dropped
    lo*lo products (planes=2, ~1e-5 of each term) separate the two sides; the inner products of independent random
    operands are ~sqrt(N) smaller than the norms, which is what the tolerances (relative to the VALUE) absorb."""
    x, dy = rnd(B, C, L, seed=11), rnd(B, M, L, seed=12)
    W = rnd(M, C, KW, seed=13, scale=(C * KW) ** -0.5)
    V = rnd(M, C, KW, seed=14, scale=(C * KW) ** -0.5)
    if planes == 1:
        x, dy, W, V = _bf(x), _bf(dy), _bf(W), _bf(V)
    xn, dyn = N.ncl_to_nlc(x, planes), N.ncl_to_nlc(dy, planes)
    y = N.conv1d_bf16(xn, N.pack_weight(W, N.W_OIK, planes), out_ncl=True)
    dx = N.conv1d_bf16(dyn, N.pack_weight(W, N.W_IOK, planes), out_ncl=True)
    assert tuple(y.shape) == (B, M, L) and tuple(dx.shape) == (B, C, L)
    close(dot(y, dy), dot(x, dx), tol)
    dW = N.conv1d_wgrad_bf16(dyn, xn, KW)
    yv = N.conv1d_bf16(xn, N.pack_weight(V, N.W_OIK, planes), out_ncl=True)
    close(dot(dW, V), dot(yv, dy), tol)


@pytest.mark.parametrize("C,M,KW", SHAPES)
def test_f16mx_conv_adjoint_wgrad_fullsize(C, M, KW):
    """The same size-independent identities through the f16mx kernels at BASELINE's full sizes (B=64 x 500 rows): the
    forward and the data-gradient launch are adjoint, the weight gradient is the forward's derivative in W; ~1.5e-5 per
    product, and the inner products of independent random operands are ~sqrt(N) smaller than the norms."""
    x, dy = rnd(B, C, L, seed=31), rnd(B, M, L, seed=32)
    W = rnd(M, C, KW, seed=33, scale=(C * KW) ** -0.5)
    V = rnd(M, C, KW, seed=34, scale=(C * KW) ** -0.5)
    xn, dyn = N.ncl_to_nlc(x, 2, "f16mx"), N.ncl_to_nlc(dy, 2, "f16mx")
    y = N.conv1d_bf16(xn, N.pack_weight(W, N.W_OIK, 3), out_ncl=True)
    dx = N.conv1d_bf16(dyn, N.pack_weight(W, N.W_IOK, 3), out_ncl=True)
    assert tuple(y.shape) == (B, M, L) and tuple(dx.shape) == (B, C, L)
    close(dot(y, dy), dot(x, dx), 1e-3)
    dW = N.conv1d_wgrad_bf16(dyn, xn, KW)
    yv = N.conv1d_bf16(xn, N.pack_weight(V, N.W_OIK, 3), out_ncl=True)
    close(dot(dW, V), dot(yv, dy), 1e-3)
    # against the exact-fp32 kernels on the same operands
    ref = N.conv1d(x, W)
    assert float((y - ref).abs().max()) <= 2e-4 * float(ref.abs().max())


@pytest.mark.parametrize("C,M,KW", SHAPES)
def test_f16_backward_kernels_adjoint_wgrad_fullsize(C, M, KW):
    """The fp16 kernels of the f16mx_hb backward at BASELINE's full sizes, fed exactly as the mode feeds them: a gradient as
    one fp16 plane under its loss scale (magnitude 1e-6: far below fp16's range), the other operand the H plane of an f16mx
    activation / packed weight.  Adjointness against the f16mx FORWARD launch (the pair the mode actually uses), the weight
    gradient against the forward's derivative in W, and both against the exact-fp32 kernels at fp16-product precision."""
    mag = 1e-6
    x, dy = rnd(B, C, L, seed=41), rnd(B, M, L, seed=42, scale=mag)
    W = rnd(M, C, KW, seed=43, scale=(C * KW) ** -0.5)
    V = rnd(M, C, KW, seed=44, scale=(C * KW) ** -0.5)
    gs = N.grad_scale(dy)
    xn = N.ncl_to_nlc(x, 2, "f16mx")                        # saved forward activation
    dyn = N.ncl_to_nlc(dy, 1, "f16", gs)                    # gradient entering the backward chain
    y = N.conv1d_bf16(xn, N.pack_weight(W, N.W_OIK, 3), out_ncl=True)             # f16mx forward
    dx = N.conv1d_bf16(dyn, N.pack_weight(W, N.W_IOK, 3), out_ncl=True)           # fp16 data gradient (H image of the weight)
    assert tuple(dx.shape) == (B, C, L)

    def close_scaled(lhs, rhs, a, b, tol):
        """the inner product of independent random operands is ~sqrt(n) |a| |b| / n: judge the difference on that scale (a
        relative bound on the inner product itself is at the mercy of how close to zero it happens to fall)"""
        scale = float(a.double().norm() * b.double().norm()) / a.numel() ** 0.5
        assert abs(lhs - rhs) <= tol * scale, (lhs, rhs, abs(lhs - rhs) / scale)

    close_scaled(dot(y, dy), dot(x, dx), y, dy, 3e-3)       # fp16 products: 2^-12 rms per operand rounding
    dW = N.conv1d_wgrad_bf16(dyn, xn, KW)                   # fp16 weight gradient (H plane of the activation)
    yv = N.conv1d_bf16(xn, N.pack_weight(V, N.W_OIK, 3), out_ncl=True)
    close_scaled(dot(dW, V), dot(yv, dy), yv, dy, 3e-3)
    ref_dx = N.conv1d(dy, W, w_layout=N.W_IOK)
    assert float((dx - ref_dx).abs().max()) <= 4e-3 * float(ref_dx.abs().max())
    ref_dW = N.conv1d_wgrad(dy, x, KW)
    assert float((dW - ref_dW).abs().max()) <= 4e-3 * float(ref_dW.abs().max())
    assert N.f16mx_range_flag() == 0                        # nothing saturated on the way


@pytest.mark.parametrize("KW", [1, 3])
def test_f16mx_conv_sample_independence_and_gap_rows_fullsize(KW):
    C = M = 1024
    x = rnd(B, C, L, seed=35)
    pk = N.pack_weight(rnd(M, C, KW, seed=36, scale=(C * KW) ** -0.5), N.W_OIK, 3)
    b = rnd(M, seed=37)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(38)).cuda()
    y = N.conv1d_bf16(N.ncl_to_nlc(x, 2, "f16mx"), pk, b, relu=True)
    yp = N.conv1d_bf16(N.ncl_to_nlc(x[perm].contiguous(), 2, "f16mx"), pk, b, relu=True)
    assert torch.equal(N.nlc_to_ncl(yp), N.nlc_to_ncl(y)[perm])
    gaps = torch.arange(0, B * (L + 1) + 1, L + 1, device="cuda")
    for plane in (0, 1):
        mat = y.matrix(plane).view(torch.int16)
        assert int(mat[gaps].abs().max()) == 0 and int(mat[B * (L + 1) + 1:].abs().max()) == 0


@pytest.mark.parametrize("KW", [1, 3])
def test_bf16_conv_sample_independence_and_gap_rows_fullsize(KW):
    C = M = 1024
    x = _bf(rnd(B, C, L, seed=15))
    pk = N.pack_weight(rnd(M, C, KW, seed=16, scale=(C * KW) ** -0.5), N.W_OIK)
    b = rnd(M, seed=17)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(18)).cuda()
    y = N.conv1d_bf16(N.ncl_to_nlc(x), pk, b, relu=True)
    yp = N.conv1d_bf16(N.ncl_to_nlc(x[perm].contiguous()), pk, b, relu=True)
    assert torch.equal(N.nlc_to_ncl(yp), N.nlc_to_ncl(y)[perm])
    # the zero gap rows (row 0 and the row after every sample) and the tail rows survive the fused epilogue
    mat = y.storage.view(-1, y.Cp)[y.guard:y.guard + y.rows]
    gaps = torch.arange(0, B * (L + 1) + 1, L + 1, device="cuda")
    assert float(mat[gaps].abs().max()) == 0.0 and float(mat[B * (L + 1) + 1:].abs().max()) == 0.0


def test_bf16_layout_roundtrip_fullsize():
    x = rnd(B, 201, L, seed=19)
    for planes, ref in ((1, _bf(x)), (2, None)):
        back = N.nlc_to_ncl(N.ncl_to_nlc(x, planes))
        if planes == 1:
            assert torch.equal(back, ref)
        else:
            assert float((back - x).abs().max()) <= 2.0 ** -16 * float(x.abs().max())


# ------------------------------------------------------------------------------------------------- quantiser
def test_vq_idempotence_optimality_checksum_fullsize():
    D, K = 128, 1024
    z = rnd(B, D, L, seed=21)                                  # (B, D, L) flattened WITHOUT permute, as the reference does
    flat = z.view(-1, D)
    E = rnd(K, D, seed=22, scale=0.7)
    idx = N.vq_argmin(flat, E)
    assert idx.dtype == torch.int64 and int(idx.min()) >= 0 and int(idx.max()) < K
    # optimality on a sample of rows, bit-exact against the CPU oracle's distance arithmetic
    rows = torch.arange(0, flat.shape[0], 61, device="cuda")
    want = O.vq_distances(flat[rows].cpu(), E.cpu()).argmin(dim=1)
    assert torch.equal(idx[rows].cpu(), want)
    # idempotence: quantising the code vectors returns the same codes and zero loss
    q = E[idx].contiguous()
    idx2 = N.vq_argmin(q, E)
    assert torch.equal(idx2, idx)
    q_st, out = N.vq_gather_loss(q, E, idx, 0.25)
    assert float(out[0]) == 0.0 and torch.equal(q_st, q)
    # perplexity is the exponential entropy of the code histogram (checksum over all 32 000 rows)
    _, out = N.vq_gather_loss(flat, E, idx, 0.25)
    p = torch.bincount(idx, minlength=K).double() / idx.numel()
    perp = float(torch.exp(-(p * torch.log(p + 1e-10)).sum()))
    assert abs(float(out[1]) - perp) <= 1e-4 * perp
    enc = N.onehot(idx, K)
    assert float(enc.sum()) == idx.numel() and torch.equal(enc.argmax(dim=1), idx)


def test_vq_stress_config_fullsize():
    """BASELINE configs[3]: codebook 4096 x 256, batch 512 x L 500 => N = 256 000 rows, one GPU.  Indices bit-exact
    against the CPU oracle's distance arithmetic on a strided sample of rows (every 499th: 514 rows spread over the
    whole range, ~2 s of CPU work), int64 in range everywhere, idempotence on the code vectors, and the histogram /
    perplexity as a checksum over all rows."""
    Nr, K, D = 256000, 4096, 256
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(Nr, D, device="cuda", generator=g)
    E = torch.randn(K, D, device="cuda", generator=g)
    idx = N.vq_argmin(x, E)
    assert idx.dtype == torch.int64 and idx.shape == (Nr,) and int(idx.min()) >= 0 and int(idx.max()) < K
    rows = torch.arange(0, Nr, 499, device="cuda")
    want = O.vq_distances(x[rows].cpu(), E.cpu()).argmin(dim=1)
    assert torch.equal(idx[rows].cpu(), want)
    # first and last row blocks too (edge tiles of the grid)
    for lo in (0, Nr - 256):
        assert torch.equal(idx[lo:lo + 256].cpu(), O.vq_distances(x[lo:lo + 256].cpu(), E.cpu()).argmin(dim=1))
    q = E[idx[:65536]].contiguous()
    assert torch.equal(N.vq_argmin(q, E), idx[:65536])                      # idempotence
    _, out = N.vq_gather_loss(x, E, idx, 0.25)
    p = torch.bincount(idx, minlength=K).double() / Nr
    perp = float(torch.exp(-(p * torch.log(p + 1e-10)).sum()))
    assert abs(float(out[1]) - perp) <= 1e-4 * perp


# ------------------------------------------------------------------------------------------------- elementwise pieces
def test_jitter_standardise_adam_fullsize():
    x = rnd(B, 128, L, seed=23)
    np.random.seed(3)
    from acoustic_locating_vq_vae import _ops
    src = torch.from_numpy(_ops.jitter_source_index(L, 0.25)).cuda()
    y = N.jitter_gather(x, src)
    assert torch.equal(y, x[:, :, src.long()])
    s = N.standardise(rnd(B, 201, L, seed=24, scale=3.0) + 1.5)
    assert float(s.mean(dim=1).abs().max()) < 1e-4                      # train_speech.py:63-64: moments over dim 1
    n = 16_836_937                                                     # the speech model's parameter count
    p = rnd(n, seed=25)
    p0, m, v = p.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    N.adam_step(p, torch.zeros(n, device="cuda"), m, v, 1)             # zero gradient from zero state: a fixed point
    assert torch.equal(p, p0) and float(m.abs().max()) == 0.0 and float(v.abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------- the whole step
@pytest.mark.parametrize("dtype", ["bf16", "x3mx_hb", "bf16x3_hb", "f16mx_hb"])
def test_train_step_invariants_fullsize(dtype):
    from acoustic_locating_vq_vae import _ops
    from acoustic_locating_vq_vae.train_step import Trainer
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    cfg = (201, 1024, 128, 3, 1024, 0.25, 1024)
    raw = rnd(B, 201, L, seed=26)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(27)).cuda()
    prev = _ops.get_compute_dtype()
    _ops.set_compute_dtype(dtype)
    try:
        outs = []
        for batch in (raw, raw, raw[perm].contiguous()):
            torch.manual_seed(5)
            m = ConvolutionalVQVAE(*cfg, use_jitter=False).cuda().train()
            tr = Trainer(m, "speech")
            loss, rec, perp = tr.step(batch)
            assert torch.isfinite(tr.buffers.grad).all() and torch.isfinite(tr.buffers.flat).all()
            outs.append((float(loss), float(rec), float(perp), tr.buffers.grad.clone()))
    finally:
        _ops.set_compute_dtype(prev)
    # identical state + identical batch -> identical losses; conv / bias gradients reduce in a fixed order
    assert outs[0][:3] == outs[1][:3]
    # the loss is a mean over samples: permuting the batch changes only the summation order
    for a, b in zip(outs[0][:3], outs[2][:3]):
        assert abs(a - b) <= 1e-5 * abs(a)
    g0, g2 = outs[0][3], outs[2][3]
    assert float((g0 - g2).norm() / g0.norm()) < (2e-2 if dtype == "bf16" else 2e-3 if dtype in ("bf16x3_hb", "f16mx_hd") else 1e-3)
