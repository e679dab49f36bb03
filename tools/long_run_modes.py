"""Loss curves of the arithmetic modes over a few hundred train steps of the speech config (same initial weights,
same synthetic batches, jitter off so that the modes see identical inputs): do the split modes track fp32 beyond the
handful of steps the parity tests cover?  The yardstick is the CONTROL leg "f32+ulp": the f32 mode itself, with the
initial weights nudged by one part in 2^23 (a different-but-equally-exact fp32 run, e.g. another summation order) -- what
it loses against f32 is the trajectory's own sensitivity, not arithmetic.   python tools/long_run_modes.py [steps] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)
import torch
from acoustic_locating_vq_vae import _ops
from acoustic_locating_vq_vae.train_step import Trainer
from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    cfg = (201, 1024, 128, 3, 1024, 0.25, 1024)
    g = torch.Generator(device="cuda").manual_seed(1)
    pool = [torch.randn(B, 201, 500, device="cuda", generator=g).abs() * (1 + i % 3) for i in range(8)]
    curves = {}
    modes = ("f32", "f32+ulp", "x3mx_hb", "f16mx_hb", "bf16x3_hb", "bf16")
    for mode in modes:
        _ops.set_compute_dtype(mode.split("+")[0])
        torch.manual_seed(3)
        m = ConvolutionalVQVAE(*cfg, use_jitter=False).cuda().train()
        if mode.endswith("+ulp"):
            with torch.no_grad():
                for q in m.parameters():
                    q.mul_(1.0 + 2.0 ** -23)
        tr = Trainer(m, "speech")
        out = []
        for s in range(steps):
            loss, rec, perp = tr.step(pool[s % len(pool)])
            if s % 10 == 9 or s == 0:
                out.append((s + 1, float(loss), float(rec), float(perp)))
        curves[mode] = out
        print(mode, " ".join("%d:%.4f" % (a, b) for a, b, _, _ in out[:: max(1, len(out) // 8)]), flush=True)
    _ops.set_compute_dtype("f32")
    ref = curves["f32"]
    for mode in modes[1:]:
        worst = max(abs(a[1] - b[1]) / abs(b[1]) for a, b in zip(curves[mode], ref))
        wp = max(abs(a[3] - b[3]) / abs(b[3]) for a, b in zip(curves[mode], ref))
        # how long does the mode stay within 1e-3 of the fp32 loss curve?
        within = next((a[0] for a, b in zip(curves[mode], ref) if abs(a[1] - b[1]) > 1e-3 * abs(b[1])), steps + 1)
        print("%s vs f32: max relative loss deviation %.3e, perplexity %.3e over %d steps; within 1e-3 of the fp32 loss up to step %d"
              % (mode, worst, wp, steps, within - 1))


if __name__ == "__main__":
    main()
