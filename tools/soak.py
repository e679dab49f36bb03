"""Soak run of the default mode: N train steps of the speech config at the bench batch (hipGraph replay, fresh synthetic batch
every step from a pool, jitter on), checking what a short bench cannot: the loss stays finite, the fp16 range flag stays 0
(no activation or scaled gradient ever reached fp16's limit), the rate holds.   python tools/soak.py [steps] [batch] [mode]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)
import numpy as np
import torch
from acoustic_locating_vq_vae import _native as N
from acoustic_locating_vq_vae import _ops
from acoustic_locating_vq_vae.train_step import Trainer
from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    mode = sys.argv[3] if len(sys.argv) > 3 else _ops.DEFAULT_DTYPE
    _ops.set_compute_dtype(mode)
    torch.manual_seed(0)
    np.random.seed(0)
    m = ConvolutionalVQVAE(201, 1024, 128, 3, 1024, 0.25, 1024).cuda().train()
    tr = Trainer(m, "speech", range_check_every=0)
    g = torch.Generator(device="cuda").manual_seed(7)
    # spectrogram-like magnitudes over four decades (the Trainer standardises per frame, as the script does)
    pool = [torch.randn(B, 201, 500, device="cuda", generator=g).abs() * 10.0 ** (i % 4 - 2) for i in range(8)]
    tr.capture(pool[0])
    N.f16mx_range_flag(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for s in range(steps):
        out = tr.step(pool[s % len(pool)])
        if (s + 1) % max(1, steps // 10) == 0:
            torch.cuda.synchronize()
            marks.append((s + 1, float(out[0]), float(out[1]), float(out[2]), (s + 1) * B / (time.perf_counter() - t0)))
            print("step %5d  loss %.4f  recon %.4f  perplexity %.1f  %.0f spectrograms/s" % marks[-1], flush=True)
    flag = N.f16mx_range_flag(reset=True) if _ops.has_fp16_range(mode) else 0
    tr.opt.prepare(tr.grad_scale)            # the advance that counts the last step's verdict
    skipped = tr.opt.skipped_steps()
    ok = all(np.isfinite(v[1]) for v in marks) and flag == 0 and skipped == 0
    print("%s: %d steps at B=%d, final loss %.4f (first mark %.4f), fp16 range flag %d, skipped steps %d -> %s"
          % (mode, steps, B, marks[-1][1], marks[0][1], flag, skipped, "OK" if ok else "FAILED"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
