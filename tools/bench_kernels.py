"""Micro-benchmark of the conv kernels at the speech config's dominant shapes (HIP events, random data)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)
import torch
from acoustic_locating_vq_vae import _native as N


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    B, L = (int(sys.argv[2]) if len(sys.argv) > 2 else 64), 500      # B=4 fills 32 of the 256 CUs: the un-throttled rate
    for (C, M, KW) in [(1024, 1024, 3), (1024, 1024, 1), (201, 1024, 3), (1024, 201, 3), (1024, 128, 3)]:
        flops = 2.0 * B * L * M * C * KW
        x = torch.randn(B, C, L, device="cuda")
        w = torch.randn(M, C, KW, device="cuda") / (C * KW) ** 0.5
        dy = torch.randn(B, M, L, device="cuda")
        line = "C=%4d M=%4d KW=%d  %6.1f GF |" % (C, M, KW, flops / 1e9)
        if which in ("all", "f32"):
            t = timeit(lambda: N.conv1d(x, w, relu=True))
            line += " f32 fwd %7.3f ms %6.1f TF |" % (t, flops / t / 1e9)
            t = timeit(lambda: N.conv1d_wgrad(dy, x, KW))
            line += " f32 wgrad %7.3f ms %6.1f TF |" % (t, flops / t / 1e9)
        if which in ("all", "bf16"):
            xn, dyn = N.ncl_to_nlc(x), N.ncl_to_nlc(dy)
            pk = N.pack_weight(w, N.W_OIK)
            t = timeit(lambda: N.conv1d_bf16(xn, pk, relu=True))
            line += " bf16 fwd %7.3f ms %6.1f TF |" % (t, flops / t / 1e9)
            t = timeit(lambda: N.conv1d_wgrad_bf16(dyn, xn, KW))
            line += " bf16 wgrad %7.3f ms %6.1f TF |" % (t, flops / t / 1e9)
            t = timeit(lambda: N.pack_weight(w, N.W_OIK))
            line += " pack %6.3f ms" % t
        if which in ("all", "bf16x3"):
            xn, dyn = N.ncl_to_nlc(x, 2), N.ncl_to_nlc(dy, 2)
            pk = N.pack_weight(w, N.W_OIK, 2)
            t = timeit(lambda: N.conv1d_bf16(xn, pk, relu=True))
            line += " x3 fwd %7.3f ms %6.1f TF |" % (t, flops / t / 1e9)
            t = timeit(lambda: N.conv1d_wgrad_bf16(dyn, xn, KW))
            line += " x3 wgrad %7.3f ms %6.1f TF |" % (t, flops / t / 1e9)
        if which in ("all", "f16mx"):
            xn, dyn = N.ncl_to_nlc(x, 2, "f16mx"), N.ncl_to_nlc(dy, 2, "f16mx")
            pk = N.pack_weight(w, N.W_OIK, 3)
            t = timeit(lambda: N.conv1d_bf16(xn, pk, relu=True))
            line += " f16mx fwd %7.3f ms %6.1f TF |" % (t, flops / t / 1e9)
            t = timeit(lambda: N.conv1d_wgrad_bf16(dyn, xn, KW))
            line += " f16mx wgrad %7.3f ms %6.1f TF |" % (t, flops / t / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
