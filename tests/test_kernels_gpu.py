"""T2: every HIP kernel, called through the C ABI, against a plain fp32 PyTorch-CPU statement of the same op
(the oracle's ATen ops).  Tolerance: 1e-3 relative (north_star) -- in practice ~1e-6; indices bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from acoustic_locating_vq_vae import _native as N  # noqa: E402
from oracle import vqvae_oracle as O  # noqa: E402
from oracle import stft_oracle  # noqa: E402

TOL = 1e-3


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def dev(t):
    return t.cuda().contiguous()


CONV_SHAPES = [
    # B, C, M, L, KW
    (2, 7, 16, 13, 3), (2, 16, 7, 13, 3), (2, 8, 16, 13, 1), (1, 1, 1, 1, 3), (3, 5, 1, 201, 3),
    (2, 201, 1024, 500, 3), (2, 1024, 128, 500, 3), (2, 1024, 1024, 201, 1), (2, 500, 1024, 201, 3),
    (2, 192, 1024, 77, 3), (3, 64, 1024, 201, 1), (2, 1024, 201, 500, 3), (5, 130, 130, 129, 3),
]


@pytest.mark.parametrize("B,C,M,L,KW", CONV_SHAPES)
def test_conv_forward_oik(B, C, M, L, KW):
    torch.manual_seed(0)
    x, w, b = torch.randn(B, C, L), torch.randn(M, C, KW) / (C * KW) ** 0.5, torch.randn(M)
    ref = F.conv1d(x, w, b, padding=KW // 2)
    got = N.conv1d(dev(x), dev(w), dev(b))
    assert rel(got, ref) < TOL
    assert rel(got, ref) < 2e-5


@pytest.mark.parametrize("B,C,M,L,KW", CONV_SHAPES)
def test_conv_forward_iok_is_conv_transpose(B, C, M, L, KW):
    torch.manual_seed(1)
    x, w, b = torch.randn(B, C, L), torch.randn(C, M, KW) / (C * KW) ** 0.5, torch.randn(M)
    ref = F.conv_transpose1d(x, w, b, padding=KW // 2)
    got = N.conv1d(dev(x), dev(w), dev(b), w_layout=N.W_IOK)
    assert rel(got, ref) < 2e-5


def test_conv_epilogue_fusions():
    torch.manual_seed(2)
    B, C, M, L = 2, 24, 40, 50
    x, w, b = torch.randn(B, C, L), torch.randn(M, C, 3) / 8, torch.randn(M)
    s1, s2, mk, post = (torch.randn(B, M, L) for _ in range(4))
    acc = F.conv1d(x, w, b, padding=1) + s1 + s2
    v = F.relu(acc)
    v = torch.where(mk > 0, v, torch.zeros_like(v))
    y, y2 = N.conv1d(dev(x), dev(w), dev(b), dev(s1), dev(s2), dev(mk), dev(post), relu=True)
    assert rel(y, v) < 1e-5 and rel(y2, v + post) < 1e-5
    y = N.conv1d(dev(x), dev(w), None, dev(s1), relu=True)
    assert rel(y, F.relu(F.conv1d(x, w, None, padding=1) + s1)) < 1e-5


@pytest.mark.parametrize("B,C,M,L,KW", CONV_SHAPES)
def test_conv_dgrad_and_wgrad(B, C, M, L, KW):
    torch.manual_seed(3)
    x = torch.randn(B, C, L, requires_grad=True)
    w = (torch.randn(M, C, KW) / (C * KW) ** 0.5).requires_grad_(True)
    b = torch.randn(M, requires_grad=True)
    dy = torch.randn(B, M, L)
    F.conv1d(x, w, b, padding=KW // 2).backward(dy)
    dx = N.conv1d(dev(dy), dev(w.detach()), w_layout=N.W_IOK)            # conv dgrad == IOK forward
    assert rel(dx, x.grad) < 2e-5
    dw, db = N.conv1d_wgrad(dev(dy), dev(x.detach()), KW, N.W_OIK, want_bias=True)
    assert rel(dw, w.grad) < 2e-5
    # a bias grad is a plain sum of B*L terms that cancel: judge it against the magnitude summed
    assert float((db.cpu() - b.grad).abs().max()) < 2e-6 * float(dy.abs().sum(dim=(0, 2)).max())
    # accumulate (shared residual weights)
    dw2 = N.conv1d_wgrad(dev(dy), dev(x.detach()), KW, N.W_OIK, dw_out=dw.clone(), accumulate=True)
    assert rel(dw2, 2 * w.grad) < 2e-5


@pytest.mark.parametrize("B,C,M,L,KW", [(2, 16, 7, 13, 3), (2, 1024, 201, 500, 3), (2, 64, 64, 33, 1)])
def test_conv_transpose_grads(B, C, M, L, KW):
    torch.manual_seed(4)
    x = torch.randn(B, C, L, requires_grad=True)
    w = (torch.randn(C, M, KW) / (C * KW) ** 0.5).requires_grad_(True)
    dy = torch.randn(B, M, L)
    F.conv_transpose1d(x, w, None, padding=KW // 2).backward(dy)
    dx = N.conv1d(dev(dy), dev(w.detach()), w_layout=N.W_OIK)            # convT dgrad == plain conv with w as (O=C,I=M)
    assert rel(dx, x.grad) < 2e-5
    dw = N.conv1d_wgrad(dev(dy), dev(x.detach()), KW, N.W_IOK)
    assert rel(dw, w.grad) < 2e-5


@pytest.mark.parametrize("n,k,d,scale", [(2000, 1024, 128, 1.0), (600, 1024, 64, 1.0), (26, 16, 4, 1.0),
                                          (4096, 4096, 256, 1.0), (333, 1000, 100, 1.0)])
def test_vq_argmin_data_scale_bit_exact(n, k, d, scale):
    torch.manual_seed(5)
    x, e = torch.randn(n, d), torch.randn(k, d) * scale
    ref = torch.argmin(O.vq_distances(x, e), dim=1)
    got, dist = N.vq_argmin(dev(x), dev(e), want_dist=True)
    assert torch.equal(got.cpu(), ref)
    refd = O.vq_distances(x, e).gather(1, ref.view(-1, 1)).view(-1)
    assert float((dist.cpu() - refd).abs().max()) < 1e-3 * float(refd.abs().max())


def test_vq_argmin_exact_ties_lowest_index():
    torch.manual_seed(6)
    e = torch.randn(300, 32)
    e[17] = e[3]
    e[250] = e[3]
    e[299] = e[140]
    x = torch.cat([e[3:4] + 0.01 * torch.randn(50, 32), e[140:141] + 0.01 * torch.randn(50, 32), torch.randn(100, 32)])
    got = N.vq_argmin(dev(x), dev(e)).cpu()
    ref = torch.argmin(O.vq_distances(x, e), dim=1)
    assert torch.equal(got, ref)
    assert not np.isin(got.numpy(), [17, 250, 299]).any()


def test_vq_argmin_init_scale_within_2ulp(golden_dir):
    """Init-scale codebook U(+-1/K): the reference's own fp32 distances are rounding-dominated.
    Accept a mismatch only where the reference's distance at our index is within 2 ulp of its minimum."""
    x = torch.from_numpy(O.hashed_uniform(2000 * 128, 11, 1.7).reshape(2000, 128))
    e = torch.from_numpy(O.hashed_uniform(1024 * 128, 13, 1.0 / 1024).reshape(1024, 128))
    d = O.vq_distances(x, e)
    ref = torch.argmin(d, dim=1)
    got = N.vq_argmin(dev(x), dev(e)).cpu()
    bad = (got != ref).nonzero().view(-1)
    for r in bad.tolist():
        dm, dg = d[r, ref[r]], d[r, got[r]]
        ulp = float(torch.nextafter(dm.abs(), torch.tensor(float("inf"))) - dm.abs())
        assert float(dg - dm) <= 2 * ulp, (r, float(dm), float(dg))
    print("init-scale mismatches within 2 ulp: %d / 2000" % len(bad))
    assert len(bad) < 40


def test_vq_gather_loss_backward_onehot():
    torch.manual_seed(7)
    B, D, L, K, beta = 3, 8, 20, 32, 0.25
    z = torch.randn(B, D, L, requires_grad=True)
    cb = torch.randn(K, D, requires_grad=True)
    loss, q_st, perp, idx = O.vector_quantizer(z, cb, beta)
    g = torch.randn(B, D, L)
    (1.7 * loss + (q_st * g).sum()).backward()
    flat = dev(z.detach()).view(-1, D)
    gi = N.vq_argmin(flat, dev(cb.detach()))
    assert torch.equal(gi.cpu(), idx)
    q, out = N.vq_gather_loss(flat, dev(cb.detach()), gi, beta)
    assert rel(q.view(B, D, L), q_st) < 1e-6
    assert rel(out[0], loss) < 1e-5 and rel(out[1], perp) < 1e-5
    dx, dE = N.vq_backward(dev(g).view(-1, D), torch.tensor([1.7]).cuda(), flat, dev(cb.detach()), gi, beta)
    assert rel(dx.view(B, D, L), z.grad) < 1e-5
    assert rel(dE, cb.grad) < 1e-5
    enc = N.onehot(gi, K)
    assert torch.equal(enc.cpu(), O.onehot(idx, K))


def test_jitter_standardise_mse_add_transpose():
    torch.manual_seed(8)
    x = torch.randn(3, 5, 37)
    np.random.seed(0)
    src = O.jitter_source_index(37, 0.25)
    srcd = torch.from_numpy(src.astype(np.int32)).cuda()
    assert torch.equal(N.jitter_gather(dev(x), srcd).cpu(), x[:, :, torch.from_numpy(src)])
    keep = torch.from_numpy(src == np.arange(37)).view(1, 1, -1)
    assert torch.equal(N.jitter_gather(dev(x), srcd, backward=True).cpu(), torch.where(keep, x, torch.zeros_like(x)))
    big = torch.randn(4, 201, 500)
    assert rel(N.standardise(dev(big), take_abs=True), O.speech_preprocess(big)) < 1e-5
    assert rel(N.standardise(dev(big)), O.standardise(big)) < 1e-5
    a, b = torch.randn(4, 201, 500), torch.randn(4, 201, 500)
    assert rel(N.mse(dev(a), dev(b)), F.mse_loss(a, b)) < 1e-5
    gl = torch.tensor([0.7])
    assert rel(N.mse_backward(dev(a), dev(b), gl.cuda()), 0.7 * 2 * (a - b) / a.numel()) < 1e-5
    assert torch.equal(N.add(dev(a), dev(b)).cpu(), a + b)
    t = torch.randn(3, 201, 500)
    assert torch.equal(N.transpose12(dev(t)).cpu(), t.permute(0, 2, 1).contiguous())


def test_adam_matches_torch():
    torch.manual_seed(9)
    p0 = torch.randn(100003)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, amsgrad=False)
    p, m, v = dev(p0.clone()), torch.zeros(100003).cuda(), torch.zeros(100003).cuda()
    for step in range(1, 4):
        g = torch.randn(100003)
        ref.grad = g.clone()
        opt.step()
        N.adam_step(p, dev(g), m, v, step)
    assert float((p.cpu() - ref.detach()).abs().max()) < 2e-6


def test_adam_device_step_counter_matches_torch_without_host_syncs():
    """adam_advance + adam_step_dev (the trainers' form: step counter and bias corrections on the device) track
    torch.optim.Adam when all steps are queued back to back with no host synchronisation in between."""
    torch.manual_seed(11)
    n = 50001
    p0 = torch.randn(n)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, amsgrad=False)
    grads = [torch.randn(n) for _ in range(6)]
    gdev = [dev(g) for g in grads]
    p, m, v = dev(p0.clone()), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    scalars = torch.zeros(N.ADAM_SCALARS, device="cuda")
    torch.cuda.synchronize()
    for g in gdev:                                   # no sync inside: every step's scalars come from the device
        N.adam_advance(scalars, 1e-3, 0.9, 0.999, 1.0)
        N.adam_step_dev(p, g, m, v, scalars)
    for g in grads:
        ref.grad = g.clone()
        opt.step()
    assert float(scalars[3]) == 6.0
    assert abs(float(scalars[0]) - 1e-3 / (1 - 0.9 ** 6)) < 1e-9 and abs(float(scalars[1]) - (1 - 0.999 ** 6) ** 0.5) < 1e-8
    assert float((p.cpu() - ref.detach()).abs().max()) < 3e-6


def test_stft_power_unpinned(golden_dir):
    """PARITY UNPINNED (torchaudio absent): checked against the torch.stft restatement and a fp64 direct DFT."""
    import os
    g = np.load(os.path.join(golden_dir, "g6_stft_unpinned.npz"))
    wave = torch.from_numpy(g["wave"].astype(np.float32)).view(1, -1)
    got = N.stft_power(dev(wave))
    assert got.shape == (1, 201, 26)
    assert rel(got, torch.from_numpy(g["power_f64"])) < 1e-4
    torch.manual_seed(10)
    w2 = torch.randn(3, 16000)
    assert rel(N.stft_power(dev(w2)), stft_oracle.stft_power(w2)) < 1e-4
    w64 = torch.from_numpy(g["wave"]).view(1, -1)                     # float64, as the echoed signal is
    got64 = N.stft_power(dev(w64))
    assert got64.dtype == torch.float64 and rel(got64, torch.from_numpy(g["power_f64"])) < 1e-10


@pytest.mark.parametrize("n,K,D,hot", [(5000, 64, 128, True), (33000, 1024, 128, False), (70000, 16, 8, True), (4100, 8, 320, True), (3000, 7, 6, True)])
def test_vq_codebook_gradient_is_atomic_free_and_reproducible(n, K, D, hot):
    """dE from the one-workgroup-per-code gather (no atomics): equal to the index_add formulation, bitwise identical
    from run to run, including a code that owns more rows than the kernel's gather list holds (several flushes)."""
    torch.manual_seed(12)
    x = torch.randn(n, D)
    E = torch.randn(K, D) * 0.7
    idx = torch.randint(0, K, (n,))
    if hot:
        idx[: n - n // 8] = 3                      # one hot code
    gl = torch.tensor([0.37])
    g = torch.randn(n, D)
    beta = 0.25
    dxs, dEs = [], []
    for _ in range(2):
        dx, dE = N.vq_backward(dev(g), dev(gl), dev(x), dev(E), idx.cuda(), beta)
        dxs.append(dx)
        dEs.append(dE)
    assert torch.equal(dEs[0], dEs[1]) and torch.equal(dxs[0], dxs[1])
    nd = float(n * D)
    diff = (E[idx] - x).double()
    want_dE = torch.zeros(K, D, dtype=torch.float64).index_add_(0, idx, diff) * (0.37 * 2.0 / nd)
    want_dx = g.double() - 0.37 * (2.0 * beta / nd) * diff
    assert rel(dEs[0], want_dE) < 1e-5 and rel(dxs[0], want_dx) < 1e-6
    base = torch.randn(K, D, device="cuda") * float(want_dE.abs().max())     # accumulate form (gradient sinks)
    acc = base.clone()
    N.vq_backward(None, dev(gl), dev(x), dev(E), idx.cuda(), beta, want_dx=False, dE_out=acc)
    assert rel(acc.double() - base.double(), want_dE) < 1e-5


@pytest.mark.parametrize("n,K,D", [(32000, 1024, 128), (6432, 1024, 64), (777, 513, 100), (65, 16, 1), (4097, 1000, 31), (1000, 130, 200),
                                    (129, 4096, 256), (64000, 4096, 256), (300, 77, 4), (5, 3, 255)])
def test_vq_argmin_register_kernel_is_bit_identical_to_the_lds_kernel(n, K, D):
    """Round 4: for D <= 256 the search keeps a wave's x rows in registers and double-buffers the codebook tile (option vq_reg,
    default on).  Same accumulation order per distance: indices AND minimum distances must equal the LDS-stationary kernel's bit
    for bit -- ragged row / code / dim counts, init-scale and data-scale codebooks."""
    g = torch.Generator().manual_seed(n + K + D)
    x = torch.randn(n, D, generator=g).cuda()
    for scale in (0.8, 1.0 / K):
        e = (torch.rand(K, D, generator=g) * 2 - 1).cuda() * scale
        outs = {}
        prev = N.get_option("vq_reg")
        try:
            for v in (1, 0):
                N.set_option("vq_reg", v)
                idx, dist = N.vq_argmin(x, e, want_dist=True)
                outs[v] = (idx.clone(), dist.clone())
        finally:
            N.set_option("vq_reg", prev)
        assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1]), (scale, int((outs[1][0] != outs[0][0]).sum()))
