"""A/B of the quantiser's argmin kernels on ONE box: x rows in registers (option vq_reg = 1) against the LDS-stationary
kernel (0), HIP events, at the speech / RIR / stress (BASELINE configs[3]) sizes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)
import torch
from acoustic_locating_vq_vae import _native as N

for n, K, D, name in ((32000, 1024, 128, "speech B=64"), (6432, 1024, 64, "rir B=32"), (256000, 4096, 256, "stress configs[3]")):
    g = torch.Generator(device="cuda").manual_seed(0)
    x, e = torch.randn(n, D, device="cuda", generator=g), torch.randn(K, D, device="cuda", generator=g)
    for rep in range(2):
        for v in (0, 1):
            N.set_option("vq_reg", v)
            for _ in range(3):
                N.vq_argmin(x, e)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                N.vq_argmin(x, e)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print("%-18s vq_reg=%d  %.4f ms  %.1f TFLOP/s (%.3f of 157.3; incl. the two norm launches)" % (name, v, ms, 2.0 * n * K * D / ms / 1e9, 2.0 * n * K * D / ms / 1e9 / 157.3), flush=True)
N.set_option("vq_reg", 1)
