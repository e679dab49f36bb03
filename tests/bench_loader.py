"""Batches/s of the dataset side on a synthetic dataset in the generator's file format:

  reference pattern  -- ``next(iter(DataLoader(ds, B, shuffle=True, collate_fn=spec_dataset_preprocessing)))`` per step,
                        then ``.to(device)`` (scripts/train_speech.py:59-62), restated here;
  DeviceLoader       -- streaming (thread pool + pinned staging + copy stream) and resident (samples parked in HBM).

    python tests/bench_loader.py [n_samples=256] [batch=64]
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)
import torch
from torch.utils.data import DataLoader

from acoustic_locating_vq_vae.data_preprocessing import spec_dataset_preprocessing
from acoustic_locating_vq_vae.rir_dataset_generator.device_loader import DeviceLoader
from acoustic_locating_vq_vae.rir_dataset_generator.specsdataset import SpecsDataset
from oracle import data_oracle as DO


def rate(fn, steps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    with tempfile.TemporaryDirectory() as root:
        DO.make_synthetic_dataset(root, n, [501, 520, 610, 505], seed=0)
        ds = SpecsDataset(root)
        ref_loader = DataLoader(ds, batch_size=B, shuffle=True, collate_fn=spec_dataset_preprocessing)

        def reference_step():
            x = next(iter(ref_loader))[0]
            return x.to("cuda")

        out = {"samples": n, "batch": B, "reference_pattern_batches_per_s": rate(reference_step, 5)}
        for name, resident in (("device_loader_streaming", False), ("device_loader_resident", True)):
            t0 = time.perf_counter()
            loader = DeviceLoader(ds, B, device="cuda", resident=resident, workers=8)
            setup = time.perf_counter() - t0
            out[name + "_batches_per_s"] = rate(lambda: next(iter(loader))[0], 20 if not resident else 200)
            out[name + "_setup_s"] = setup
            loader.close()
        print(json.dumps(out))


if __name__ == "__main__":
    main()
