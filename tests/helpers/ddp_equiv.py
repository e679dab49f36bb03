"""Child of tests/test_rccl_gpu.py::test_two_rank_steps_equal_one_rank_on_the_concatenated_batch -- launched by
torch.distributed.run with 2 ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one device), or directly as ONE process
(no process group) on the whole batch.  Trains `steps` steps on this rank's shard and writes the flat parameter buffer."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    mode, out, buckets = sys.argv[1], sys.argv[2], int(sys.argv[3])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from acoustic_locating_vq_vae import _ops
    from acoustic_locating_vq_vae.train_step import Trainer, shard_batch
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    _ops.set_compute_dtype(mode)
    torch.manual_seed(0)
    model = ConvolutionalVQVAE(40, 128, 16, 2, 64, 0.25, 64, use_jitter=False).cuda().train()
    tr = Trainer(model, "speech", grad_buckets=buckets)
    losses = []
    for s in range(3):
        full = torch.randn(8, 40, 60, generator=torch.Generator().manual_seed(50 + s)).cuda()
        loss, rec, _ = tr.step(shard_batch(full, rank, world))
        t = torch.stack([loss, rec]).clone()
        if world > 1:
            dist.all_reduce(t)
            t /= world
        losses.append(t.cpu())
    if rank == 0:
        torch.save({"flat": tr.buffers.flat.detach().cpu(), "losses": torch.stack(losses)}, out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
