"""Weight-gradient scratch is caller-owned and sized by alvq_conv1d_wgrad_*_workspace_bytes (include/alvq.h).  A caller
that allocates EXACTLY that many bytes must be safe for every segment count: round 2 sized for 1 and 4 segments only and
the 3-segment launch of the speech model's shared residual weights (R = 3) wrote up to 2.75 MB past the end at some batch
sizes (advisor finding; e.g. width 1, 1024 <- 128 channels, B = 5: 60 partials needed, 54 sized).  Each launch here gets an
exactly sized workspace followed by a canary region, through the raw C ABI."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from acoustic_locating_vq_vae import _native as N  # noqa: E402

CANARY = 1 << 20
# (B, C, M, L, KW): shapes at which splits(3 * rows) > max(splits(rows), splits(4 * rows)) for at least one format
CASES = [(5, 128, 1024, 500, 1), (13, 128, 1024, 500, 1), (5, 64, 1024, 201, 1), (3, 1024, 1024, 95, 3), (7, 200, 520, 60, 3)]


def _enter(t, fmt):
    if fmt == "f16mx":
        return N.ncl_to_nlc(t, 2, "f16mx")
    return N.ncl_to_nlc(t, 2 if fmt == "bf16x3" else 1)


@pytest.mark.parametrize("fmt", ["bf16", "bf16x3", "f16mx"])
@pytest.mark.parametrize("B,C,M,L,KW", CASES)
@pytest.mark.parametrize("nseg", [1, 2, 3, 4])
def test_exactly_sized_workspace_is_enough(fmt, B, C, M, L, KW, nseg):
    lib = N.lib()
    torch.manual_seed(nseg)
    xs = [torch.randn(B, C, L, device="cuda") for _ in range(nseg)]
    dys = [torch.randn(B, M, L, device="cuda") for _ in range(nseg)]
    nbytes = getattr(lib, "alvq_conv1d_wgrad_%s_workspace_bytes" % fmt)(B, C, M, L, KW)
    buf = torch.full((nbytes + CANARY,), 0x5A, device="cuda", dtype=torch.uint8)
    xn, dyn = [_enter(x, fmt) for x in xs], [_enter(d, fmt) for d in dys]
    dw = torch.empty((M, C, KW), device="cuda")
    pd = (ctypes.c_void_p * nseg)(*[d.ptr for d in dyn])
    px = (ctypes.c_void_p * nseg)(*[x.ptr for x in xn])
    args = (pd, px, nseg, dw.data_ptr(), buf.data_ptr(), B, C, M, L, KW, N.W_OIK, 0)
    stream = torch.cuda.current_stream().cuda_stream
    if fmt == "f16mx":
        rc = lib.alvq_conv1d_wgrad_f16mx_multi(*args, None, stream)
    else:
        rc = getattr(lib, "alvq_conv1d_wgrad_%s_multi" % fmt)(*args, stream)
    assert rc == 0, lib.alvq_last_error()
    torch.cuda.synchronize()
    assert bool((buf[nbytes:] == 0x5A).all()), "the launch wrote past alvq_conv1d_wgrad_%s_workspace_bytes" % fmt
    want = torch.zeros(M, C, KW, device="cuda")
    for x, d in zip(xs, dys):
        w = torch.zeros(M, C, KW, device="cuda", requires_grad=True)
        F.conv1d(x, w, None, padding=KW // 2).backward(d)
        want += w.grad
    tol = 2e-2 if fmt == "bf16" else 2e-4
    assert float((dw - want).abs().max() / want.abs().max()) < tol


@pytest.mark.parametrize("fmt", ["bf16", "bf16x3", "f16mx"])
def test_exactly_sized_workspace_with_bias_gradient(fmt):
    lib = N.lib()
    B, C, M, L, KW = 5, 128, 1024, 500, 3
    torch.manual_seed(0)
    x, dy = torch.randn(B, C, L, device="cuda"), torch.randn(B, M, L, device="cuda")
    nbytes = getattr(lib, "alvq_conv1d_wgrad_%s_workspace_bytes" % fmt)(B, C, M, L, KW)
    buf = torch.full((nbytes + CANARY,), 0x5A, device="cuda", dtype=torch.uint8)
    xn, dyn = _enter(x, fmt), _enter(dy, fmt)
    dw, db = torch.empty((M, C, KW), device="cuda"), torch.empty((M,), device="cuda")
    args = (dyn.ptr, xn.ptr, dw.data_ptr(), db.data_ptr(), buf.data_ptr(), B, C, M, L, KW, N.W_OIK, 0)
    stream = torch.cuda.current_stream().cuda_stream
    rc = lib.alvq_conv1d_wgrad_f16mx(*args, None, stream) if fmt == "f16mx" else getattr(lib, "alvq_conv1d_wgrad_%s" % fmt)(*args, stream)
    assert rc == 0, lib.alvq_last_error()
    torch.cuda.synchronize()
    assert bool((buf[nbytes:] == 0x5A).all())
    tol = 2e-2 if fmt == "bf16" else 2e-4
    assert float((db - dy.sum(dim=(0, 2))).abs().max() / dy.sum(dim=(0, 2)).abs().max()) < tol
