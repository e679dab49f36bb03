"""Micro-benchmark of the split reduction / weight packing kernels at the speech config's shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)
import torch
from acoustic_locating_vq_vae import _native as N
from bench_kernels import timeit


def main():
    B, L = 64, 500
    for fmt, planes in (("f16mx", 2), ("bf16", 1)):
        for (C, M, KW) in [(1024, 1024, 3), (1024, 1024, 1), (1024, 201, 3)]:
            x = torch.randn(B, C, L, device="cuda")
            dy = torch.randn(B, M, L, device="cuda")
            xn, dyn = N.ncl_to_nlc(x, planes, fmt), N.ncl_to_nlc(dy, planes, fmt)
            for layout, name in ((N.W_OIK, "OIK"), (N.W_IOK, "IOK")):
                t = timeit(lambda: N.conv1d_wgrad_bf16(dyn, xn, KW, layout))
                print("%s C=%d M=%d KW=%d %s wgrad+reduce %.3f ms" % (fmt, C, M, KW, name, t), flush=True)


if __name__ == "__main__":
    main()
