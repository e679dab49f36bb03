"""Step time of the location head's training loop at the script's size (scripts/train_location.py:23-24,40-41: 201 x 1024
codes -> 1 angle, batch 16): the module API with torch.optim.Adam (what the unchanged script runs) against
train_step.LocationTrainer (flat HIP Adam, fc_1's gradient scattered into the flat buffer).   python tests/bench_location.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from acoustic_locating_vq_vae.train_step import LocationTrainer  # noqa: E402
from acoustic_locating_vq_vae.vq_vae.location_model.location_model import LocationModule  # noqa: E402


def main():
    L, K, B, steps = 201, 1024, 16, 10
    torch.manual_seed(0)
    idx = torch.randint(0, K, (B, L), device="cuda")
    theta = torch.rand(B, device="cuda") * 3
    m = LocationModule(L, K, 1).cuda().train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def script_step():
        opt.zero_grad()
        loss = F.mse_loss(m(idx), theta / torch.pi, reduction="mean")
        loss.backward()
        opt.step()

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    t_script = timeit(script_step)
    del opt, m
    torch.cuda.empty_cache()
    m2 = LocationModule(L, K, 1).cuda().train()
    tr = LocationTrainer(m2)
    t_flat = timeit(lambda: tr.step(idx, theta))
    print("location step, B=%d, %d x %d codes: module API + torch.optim.Adam %.2f ms | LocationTrainer (flat HIP Adam) %.2f ms"
          % (B, L, K, t_script, t_flat))


if __name__ == "__main__":
    main()
