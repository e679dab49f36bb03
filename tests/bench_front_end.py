"""Utterances/s of the waveform front end: the dataset generator's per-sample arithmetic
(genereate_dataset.py:35-49) for 5 s utterances at 16 kHz with a 0.4 s impulse response, HIP kernels vs the CPU
restatement (scipy + torch.stft) on the host cores.    python tests/bench_front_end.py [batch=64]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)
import numpy as np
import torch

from acoustic_locating_vq_vae import front_end as FE
from oracle import front_end_oracle as FO


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S, Nh = 80000, 6400
    g = torch.Generator().manual_seed(0)
    wave = torch.randn(B, S, generator=g)
    h = torch.from_numpy(np.random.default_rng(0).standard_normal((B, Nh)) * np.exp(-np.arange(Nh) / 1000.0))
    wd, hd = wave.cuda(), h.cuda()
    for _ in range(2):
        FE.specs_from_waveform(wd, hd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        FE.specs_from_waveform(wd, hd)
    torch.cuda.synchronize()
    gpu = B * reps / (time.perf_counter() - t0)
    n_cpu = 4
    t0 = time.perf_counter()
    for b in range(n_cpu):
        FO.convert_speech_to_specs(wave[b:b + 1], h[b].numpy())
    cpu = n_cpu / (time.perf_counter() - t0)
    # algorithmic work per utterance: the FIR is 2*S*Nh fp64 flops; everything else is negligible beside it
    print(json.dumps({"batch": B, "gpu_utterances_per_s": gpu, "cpu_utterances_per_s": cpu, "cpu_threads": torch.get_num_threads(),
                      "gpu_fir_fp64_tflops": gpu * 2.0 * S * Nh / 1e12, "ratio": gpu / cpu}))


if __name__ == "__main__":
    main()
