"""Soak of the saturation guard: N replayed train steps of the speech config at the bench batch with a NaN planted in the batch
every `period`-th step and a validation step (Trainer.evaluate) every 250th: every planted step must be skipped (device counter),
no other, parameters stay finite, the loss keeps falling, the rate holds.   python tools/soak_skips.py [steps] [batch] [period]"""
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
    period = int(sys.argv[3]) if len(sys.argv) > 3 else 97
    _ops.set_compute_dtype(_ops.DEFAULT_DTYPE)
    torch.manual_seed(0)
    np.random.seed(0)
    m = ConvolutionalVQVAE(201, 1024, 128, 3, 1024, 0.25, 1024).cuda().train()
    tr = Trainer(m, "speech", range_check_every=0)
    g = torch.Generator(device="cuda").manual_seed(7)
    pool = [torch.randn(B, 201, 500, device="cuda", generator=g).abs() * 10.0 ** (i % 4 - 2) for i in range(8)]
    bad = pool[3].clone()
    bad[5, 17, 123] = float("nan")
    tr.capture(pool[0])
    N.f16mx_range_flag(reset=True)
    tr.opt.skipped_steps(reset=True)
    applied0 = tr.opt.applied_steps()
    torch.cuda.synchronize()
    planted, t0, first, last, val = 0, time.perf_counter(), None, None, []
    for s in range(steps):
        if s % period == period - 1:
            tr.step(bad)
            planted += 1
        else:
            out = tr.step(pool[s % len(pool)])
            if s % 250 == 249:
                val.append(float(tr.evaluate(pool[(s + 1) % len(pool)])[0]))
                last = float(out[0])
                first = last if first is None else first
                print("step %5d  loss %.4f  validation %.4f  skipped so far %d  %.0f spectrograms/s"
                      % (s + 1, last, val[-1], tr.opt.skipped_steps(reset=False), (s + 1) * B / (time.perf_counter() - t0)), flush=True)
    torch.cuda.synchronize()
    # the device counter is advanced by the NEXT step's prepare: a skip verdict still sitting in the slot counts as well
    skipped = tr.opt.skipped_steps(reset=False) + int(float(tr.buffers.skip_slot) != 0.0)
    applied = tr.opt.applied_steps() - applied0
    finite = bool(torch.isfinite(tr.buffers.flat).all()) and bool(torch.isfinite(tr.opt.exp_avg_sq).all())
    ok = skipped == planted and applied == steps - planted and finite and np.isfinite(last) and last < first
    print("%s: %d steps at B=%d, %d NaN batches planted, %d steps skipped, %d applied, parameters finite %s, loss %.4f -> %.4f, "
          "sticky flag %d -> %s" % (_ops.get_compute_dtype(), steps, B, planted, skipped, applied, finite, first, last,
                                    N.f16mx_range_flag(reset=True), "OK" if ok else "FAILED"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
