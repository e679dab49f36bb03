"""BASELINE configs[3]: VQ argmin stress -- codebook 4096x256, N = 512*500 = 256000 rows (1 GPU).
Reports time / TFLOP/s (algorithmic 2*N*K*D) of alvq_vq_argmin_f32 and checks indices bit-exact vs the CPU oracle
on a 4096-row sample."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)
import torch
from acoustic_locating_vq_vae import _native as N
from oracle import vqvae_oracle as O

torch.manual_seed(0)
n, k, d = 256000, 4096, 256
x = torch.randn(n, d, device="cuda")
e = torch.randn(k, d, device="cuda")
for _ in range(2):
    idx = N.vq_argmin(x, e)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    idx = N.vq_argmin(x, e)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
ref = torch.argmin(O.vq_distances(x[:4096].cpu(), e.cpu()), dim=1)
print("vq stress N=%d K=%d D=%d: %.3f ms  %.1f TFLOP/s (fp32 MFMA peak 157.3)  idx bit-exact on sample: %s"
      % (n, k, d, ms, 2.0 * n * k * d / ms / 1e9, bool(torch.equal(idx[:4096].cpu(), ref))))
