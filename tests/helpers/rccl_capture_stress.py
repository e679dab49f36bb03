"""Child process of tests/test_rccl_gpu.py: ONE hipGraph capture of a Trainer while ANOTHER thread keeps calling
``hipEventQuery`` -- what ProcessGroupNCCL's watchdog thread does to the events of earlier collectives.  Under torch's default
capture error mode ("global") such a call from any thread while a capture is in progress is an error that invalidates the
capture (round 3: an intermittent hipErrorStreamCaptureInvalidated, about one capture in ten next to a live process group);
``Trainer.capture`` captures with "thread_local", under which only the capturing thread's own calls matter.  The spinner makes
the situation deterministic: it polls for the whole duration of the capture (round-3 verdict: replace the 24-capture repeat-
until-it-shows test).  A one-rank "nccl" group is opened as well, so the real watchdog thread is alive too.
Prints RCCL_CAPTURE_SPIN {json}."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from acoustic_locating_vq_vae import _ops
    from acoustic_locating_vq_vae.train_step import Trainer
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    _ops.set_compute_dtype("bf16")
    raw = torch.randn(4, 40, 60, device="cuda")
    torch.manual_seed(0)
    m = ConvolutionalVQVAE(40, 128, 16, 2, 64, 0.25, 64).cuda().train()
    tr = Trainer(m, "speech", force_collective=True, grad_buckets=2)
    np.random.seed(0)
    ev = torch.cuda.Event()
    ev.record()
    stop, polls, errors = threading.Event(), [0], []

    def spin():                                    # the watchdog's behaviour, without its 100 ms naps
        torch.cuda.set_device(0)
        while not stop.is_set():
            try:
                ev.query()
                polls[0] += 1
                time.sleep(2e-4)               # yield the GIL: the capture itself is Python-driven
            except Exception as exc:               # noqa: BLE001  (reported, not raised: the main thread asserts)
                errors.append(repr(exc)[:200])
                return

    th = threading.Thread(target=spin, daemon=True)
    th.start()
    while polls[0] < 100 and not errors:           # the spinner is demonstrably polling before the capture starts
        pass
    before = polls[0]
    tr.capture(raw, warmup=2)                      # warm-up steps issue collectives; the capture follows immediately
    during = polls[0] - before
    stop.set()
    th.join()
    loss = float(tr.step(raw)[0])
    torch.cuda.synchronize()
    print("RCCL_CAPTURE_SPIN " + json.dumps({"capture_mode": tr._capture_mode, "polls_during_capture": during,
                                              "spinner_errors": errors, "replayed_loss_finite": bool(np.isfinite(loss))}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
