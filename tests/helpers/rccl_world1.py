"""Child process of tests/test_rccl_gpu.py: a ONE-rank "nccl" (= RCCL on ROCm) process group on cuda:0, the Trainer's
all-reduce forced (it is the identity at world 1).  Prints one JSON line: per variant the number of all_reduce calls per
step and whether the parameters after the steps are bit-identical to the run without any collective."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "x3mx_hb"
    steps = 3
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from acoustic_locating_vq_vae import _ops
    from acoustic_locating_vq_vae.train_step import Trainer
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    _ops.set_compute_dtype(mode)
    calls = {"n": 0, "numel": []}
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls["n"] += 1
        calls["numel"].append(int(t.numel()))
        return real(t, *a, **k)

    dist.all_reduce = counting
    cfg = (40, 128, 16, 2, 64, 0.25, 64)
    raws = [torch.randn(4, 40, 60, generator=torch.Generator().manual_seed(10 + i)).cuda() for i in range(steps)]

    def run(force, graph, buckets):
        torch.manual_seed(0)
        model = ConvolutionalVQVAE(*cfg).cuda().train()
        tr = Trainer(model, "speech", force_collective=force, grad_buckets=buckets)
        np.random.seed(5)
        if graph:
            tr.capture(raws[0], warmup=1)          # one real step, then the capture
        calls["n"], calls["numel"] = 0, []
        losses = []
        for r in raws:
            losses.append(tr.step(r)[0])
        torch.cuda.synchronize()
        flat = tr.buffers.flat.detach().clone()
        return flat, calls["n"] / steps, list(calls["numel"][:2]), [float(v) for v in losses], int(tr.buffers.grad.numel())

    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "mode": mode, "variants": {}}
    for graph in (False, True):
        base, n0, _, l0, numel = run(False, graph, 1)
        assert n0 == 0
        for buckets in (1, 2):
            flat, n, sizes, losses, _ = run(True, graph, buckets)
            out["variants"]["%s_buckets%d" % ("graph" if graph else "eager", buckets)] = {
                "allreduce_calls_per_step": n, "first_call_numels": sizes, "flat_numel": numel,
                "params_bit_identical": bool(torch.equal(flat, base)), "losses_equal": losses == l0,
                "finite": bool(torch.isfinite(flat).all())}
    # and the collective really sums: 2 * grad after all_reduce of a doubled buffer?  At world 1 SUM is the identity;
    # check the RCCL kernel at least moved data: all_reduce on a scratch tensor with MAX / SUM returns it unchanged
    t = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    real(t, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    out["scratch_identity"] = bool(torch.equal(t, torch.arange(1 << 20, device="cuda", dtype=torch.float32)))
    print("RCCL_WORLD1 " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
