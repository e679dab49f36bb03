"""Fixed cost of a weight-gradient launch: the fp16 v3 kernel at 1024 x 1024 x 3 (B = 64, L = 500) over 1..4 segments (the R
uses of a shared weight in ONE launch) -- time against contraction length; a straight-line fit gives the per-launch overhead."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)
import torch
from acoustic_locating_vq_vae import _native as N

B, L, C = 64, 500, 1024
for KW in (3, 1):
    for fmt in ("f16", "bf16"):
        mk = (lambda t: N.ncl_to_nlc(t, 1, "f16")) if fmt == "f16" else (lambda t: N.ncl_to_nlc(t, 1))
        xs = [mk(torch.randn(B, C, L, device="cuda")) for _ in range(4)]
        dys = [mk(torch.randn(B, C, L, device="cuda")) for _ in range(4)]
        if fmt == "f16":
            sc = N.grad_scale(torch.randn(16, device="cuda"))
            for d in dys:
                d.gscale = sc
        for nseg in (1, 2, 3, 4):
            pairs = list(zip(dys[:nseg], xs[:nseg]))
            for want_bias in ((False, True) if nseg == 1 else (False,)):
                fn = (lambda: N.conv1d_wgrad_bf16(dys[0], xs[0], KW, want_bias=True)) if want_bias else (lambda: N.conv1d_wgrad_bf16_multi(pairs, KW))
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                gf = 2.0 * nseg * B * L * C * C * KW / 1e9
                print("KW=%d %s nseg=%d bias=%d  %.1f us (contraction + split reduction)  %.0f TFLOP/s" % (KW, fmt, nseg, want_bias, ms * 1e3, gf / ms), flush=True)
