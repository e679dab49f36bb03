"""Child of tests/test_rccl_gpu.py::test_a_step_saturated_on_one_rank_is_skipped_on_every_rank -- launched by
torch.distributed.run with 2 ranks sharing cuda:0 over gloo.  Step 2's batch holds a NaN on rank 1 ONLY: the skip verdict
travels through the step's all-reduce (element 0 of the flat gradient buffer), so BOTH ranks must leave their parameters
untouched, count one skipped step, and stay bit-identical to each other afterwards.  Prints DDP_SKIP {json} on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    mode, buckets, graph = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "graph"
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from acoustic_locating_vq_vae import _ops
    from acoustic_locating_vq_vae.train_step import Trainer, shard_batch
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    _ops.set_compute_dtype(mode)
    torch.manual_seed(0)
    model = ConvolutionalVQVAE(40, 128, 16, 2, 64, 0.25, 64, use_jitter=False).cuda().train()
    tr = Trainer(model, "speech", grad_buckets=buckets, range_check_every=0)
    batch = lambda s: shard_batch(torch.randn(8, 40, 60, generator=torch.Generator().manual_seed(70 + s)).cuda(), rank, world)
    if graph:
        tr.capture(batch(0), warmup=1)

    def same_everywhere(t):
        chk = t.view(torch.int32).to(torch.int64).sum().reshape(1).cpu()
        got = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(got, chk)
        return len({int(g) for g in got}) == 1

    tr.step(batch(1))
    torch.cuda.synchronize()
    before = tr.buffers.flat.clone()
    bad = batch(2)
    if rank == 1:
        bad[0, 0, 0] = float("nan")
    tr.step(bad)
    torch.cuda.synchronize()
    untouched = bool(torch.equal(before, tr.buffers.flat))
    slot = float(tr.buffers.skip_slot)                       # summed over the ranks: 1.0 (one rank saturated)
    identical_after_skip = same_everywhere(tr.buffers.flat)
    tr.step(batch(3))
    tr.step(batch(4))
    torch.cuda.synchronize()
    moved = not torch.equal(before, tr.buffers.flat)
    out = {"rank": rank, "untouched": untouched, "slot": slot, "identical_after_skip": identical_after_skip, "moved": moved,
           "identical_at_end": same_everywhere(tr.buffers.flat), "finite": bool(torch.isfinite(tr.buffers.flat).all()),
           "skipped": tr.opt.skipped_steps(reset=False), "applied": tr.opt.applied_steps()}
    res = [None] * world
    dist.all_gather_object(res, out)
    if rank == 0:
        print("DDP_SKIP " + json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
