"""Round-4 first GPU probe: x3mx_hb on every reference golden, its step time next to f16mx_hb's, and the skip guard."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from acoustic_locating_vq_vae import _native as N, _ops
from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
from acoustic_locating_vq_vae.train_step import Trainer
import g3_cases

out = {}
for mode in ("x3mx_hb", "f16mx_hb"):
    _ops.set_compute_dtype(mode)
    for tag in ("speech", "speech_b16", "speech_b64", "rir", "rir_b32", "echoed", "echoed_b32"):
        r = g3_cases.run(tag)
        keep = {k: r[k] for k in ("idx_total", "idx_mismatches", "z_rel_max", "recon_rel_max", "recon_rel_l2", "recon_error_rel",
                                  "grad_rel_max", "grad_rel_l2_median", "grad_rel_l2_max", "encoder_grad_rel_max") if k in r}
        print(mode, tag, json.dumps(keep), flush=True)
        out["%s/%s" % (mode, tag)] = keep

SPEECH = (201, 1024, 128, 3, 1024, 0.25, 1024)
for mode in ("x3mx_hb", "f16mx_hb", "x3mx_hb", "f16mx_hb"):
    _ops.set_compute_dtype(mode)
    torch.manual_seed(0)
    np.random.seed(1)
    m = ConvolutionalVQVAE(*SPEECH).cuda().train()
    tr = Trainer(m, "speech")
    raw = torch.randn(64, 201, 500, device="cuda")
    tr.capture(raw)
    for _ in range(5):
        tr.step(raw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        o = tr.step(raw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    print(mode, "ms/step %.3f  spec/s %.0f  loss %.5f  flag %d skipped %d" % (ms, 64e3 / ms, float(o[0]), N.f16mx_range_flag(), tr.opt.skipped_steps()), flush=True)
    out["time/" + mode] = ms
    del tr, m
    torch.cuda.empty_cache()

# skip guard: a huge input at step 3 must leave parameters untouched
_ops.set_compute_dtype("x3mx_hb")
torch.manual_seed(0)
m = ConvolutionalVQVAE(20, 48, 8, 2, 24, 0.25, 64).cuda().train()
tr = Trainer(m, "speech", range_check_every=0)
raw = torch.randn(4, 20, 40, device="cuda")
for i in range(2):
    tr.step(raw)
before = tr.buffers.flat.clone(); mom = tr.opt.exp_avg.clone()
bad = raw.clone(); bad[0, 0, 0] = float("nan")
tr.step(bad)
torch.cuda.synchronize()
same = bool(torch.equal(before, tr.buffers.flat)) and bool(torch.equal(mom, tr.opt.exp_avg))
tr.step(raw)
torch.cuda.synchronize()
moved = not torch.equal(before, tr.buffers.flat)
print("skip guard: untouched after bad step", same, "moved after clean step", moved, "scalars", tr.opt.scalars.tolist(), "flag", N.f16mx_range_flag(reset=False))
out["skip"] = [same, moved, tr.opt.scalars.tolist()]
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r4_probe.json"), "w"), indent=1)
