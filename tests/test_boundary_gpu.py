"""The input boundary without transpositions (csrc/boundary.hip; SURVEY 8(b) "element strides of x"; callers
scripts/train_rir.py:42-49, scripts/train_echoed_speech.py:66): a model input handed over as `t.permute(0, 2, 1)` of a
contiguous tensor already has its channel axis contiguous, so it goes straight into the NLC compute layout, and the RIR
loop's standardisation is fused into that pass.  Both must be BIT-IDENTICAL to the round-3 route standardise ->
transpose -> ncl_to_nlc (same arithmetic, same summation order), in every NLC format."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FORMATS = [(1, "bf16"), (2, "bf16x3"), (2, "f16mx")]


def _planes_equal(a, b):
    assert (a.B, a.L, a.C, a.planes, a.fmt, a.rows, a.Cp) == (b.B, b.L, b.C, b.planes, b.fmt, b.rows, b.Cp)
    n = a.planes * (a.rows + 2 * a.guard) * a.Cp
    g = a.guard * a.Cp
    for p in range(a.planes):          # the matrix of every plane (the guard rows before / between / after are never written)
        lo = g + p * (a.rows + 2 * a.guard) * a.Cp
        if not torch.equal(a.storage[lo:lo + a.rows * a.Cp].view(torch.int16), b.storage[lo:lo + a.rows * a.Cp].view(torch.int16)):
            return False
    return n > 0


@pytest.mark.parametrize("planes,fmt", FORMATS)
@pytest.mark.parametrize("B,L,C", [(3, 201, 500), (1, 7, 5), (2, 64, 64), (5, 33, 130), (32, 201, 500)])
def test_rows_to_nlc_equals_transpose_then_convert(B, L, C, planes, fmt):
    from acoustic_locating_vq_vae import _native as N
    x = torch.randn(B, L, C, generator=torch.Generator().manual_seed(B * 1000 + L)).cuda() * 3.0
    direct = N.rows_to_nlc(x, planes, fmt)
    ref = N.ncl_to_nlc(N.transpose12(x), planes, fmt)
    assert _planes_equal(direct, ref)
    back = N.nlc_to_ncl(direct)                          # and it is a faithful image of x
    assert float((back - x.permute(0, 2, 1)).abs().max()) <= (3e-2 if fmt == "bf16" else 1e-4) * float(x.abs().max())


@pytest.mark.parametrize("planes,fmt", FORMATS)
@pytest.mark.parametrize("B,F,T,take_abs", [(4, 201, 500, False), (2, 5, 70, False), (3, 240, 64, True), (32, 201, 500, False)])
def test_fused_standardise_is_bit_identical_to_the_three_pass_route(B, F, T, take_abs, planes, fmt):
    """train_rir.py:42-45: standardise over dim 1, permute(0, 2, 1), enter the encoder."""
    from acoustic_locating_vq_vae import _native as N
    raw = (torch.randn(B, F, T, generator=torch.Generator().manual_seed(F + T)) * 2.0 + 0.5).cuda()
    fused = N.rows_to_nlc(raw, planes, fmt, standardise=True, take_abs=take_abs)
    ref = N.ncl_to_nlc(N.transpose12(N.standardise(raw, take_abs=take_abs)), planes, fmt)
    assert _planes_equal(fused, ref)


def test_fused_standardise_refuses_what_it_cannot_hold():
    from acoustic_locating_vq_vae import _native as N
    raw = torch.randn(1, 300, 64).cuda()
    assert not N.rows_to_nlc_supported("bf16", 300, True) and N.rows_to_nlc_supported("bf16", 300, False)
    with pytest.raises(RuntimeError):
        N.rows_to_nlc(raw, 1, "bf16", standardise=True)


@pytest.mark.parametrize("mode", ["x3mx_hb", "f16mx_hb", "bf16x3_hb", "bf16", "f32"])
def test_rir_trainer_step_is_unchanged_by_the_fused_boundary(mode, monkeypatch):
    """A Trainer step of the RIR loop with the fused boundary against ALVQ_ROWS_BOUNDARY=0 (standardise, transpose, convert
    as three launches): same losses and parameters, bit for bit, eager and from the graph -- and the module API fed the
    permuted VIEW (what train_rir.py:45 hands over) returns the same z as the materialised tensor."""
    from acoustic_locating_vq_vae import _ops
    from acoustic_locating_vq_vae.train_step import Trainer
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    raws = [torch.randn(4, 33, 50, generator=torch.Generator().manual_seed(s)).cuda() for s in range(3)]
    wien = [torch.randn(4, 33, generator=torch.Generator().manual_seed(10 + s)).cuda() for s in range(3)]
    _ops.set_compute_dtype(mode)
    try:
        outs = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("ALVQ_ROWS_BOUNDARY", flag)
            for graph in (False, True):
                torch.manual_seed(3)
                m = ConvolutionalVQVAE(50, 64, 8, 2, 32, 0.25, 32, use_jitter=False, out_channels=1).cuda().train()
                tr = Trainer(m, "rir", range_check_every=0)
                if graph:
                    tr.capture(raws[0], wien[0], warmup=1)
                losses = [float(tr.step(r, w)[0]) for r, w in zip(raws, wien)]
                torch.cuda.synchronize()
                outs[(flag, graph)] = (losses, tr.buffers.flat.clone())
        for graph in (False, True):
            assert outs[("1", graph)][0] == outs[("0", graph)][0]
            assert torch.equal(outs[("1", graph)][1], outs[("0", graph)][1])
        monkeypatch.setenv("ALVQ_ROWS_BOUNDARY", "1")
        m = ConvolutionalVQVAE(50, 64, 8, 2, 32, 0.25, 32, use_jitter=False, out_channels=1).cuda().eval()
        t = torch.randn(4, 33, 50).cuda()
        assert torch.equal(m._latent(t.permute(0, 2, 1)), m._latent(t.permute(0, 2, 1).contiguous()))
    finally:
        _ops.set_compute_dtype("f32")
