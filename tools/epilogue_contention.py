"""How long does the f16mx conv epilogue take with N workgroups finishing at once?  (ALVQ_FX_DBG=256 prints the phase
stamps of every launch, 512 the same with the epilogue's stores removed; batch B -> ceil((B*501+1)/256)*4 workgroups of
the 1024-channel width-1 conv.)  Two epilogues: skip operand + ReLU (forward), sign-bit mask (data gradient)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)
import torch
from acoustic_locating_vq_vae import _native as N

for B in (4, 16, 32, 64):
    L, C, M = 500, 1024, 1024
    x = torch.randn(B, C, L, device="cuda")
    w = torch.randn(M, C, 1, device="cuda") / C ** 0.5
    xn = N.ncl_to_nlc(x, 2, "f16mx")
    sk = N.ncl_to_nlc(torch.randn(B, M, L, device="cuda"), 2, "f16mx")
    pk = N.pack_weight(w, N.W_OIK, 3)
    print("B=%d skip + relu" % B, file=sys.stderr, flush=True)
    for _ in range(2):
        t = N.conv1d_bf16(xn, pk, skip1=sk, relu=True)
    torch.cuda.synchronize()
    print("B=%d sign-bit mask" % B, file=sys.stderr, flush=True)
    for _ in range(2):
        N.conv1d_bf16(xn, pk, mask=t)
    torch.cuda.synchronize()
