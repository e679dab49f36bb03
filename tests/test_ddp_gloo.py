"""T4 (CPU variant): the data-parallel plumbing with world_size 2 over gloo.

The HIP kernels cannot run here, so the model math is the CPU oracle; what is under test is the product's
FlatBuffers / shard_batch / sync_grads logic: N-rank grads after the ONE all-reduce (scaled by 1/world) equal
1-rank grads on the concatenated batch, and every rank ends a step with identical parameters."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup_paths():
    pkg = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
    for p in (ROOT, pkg, os.path.join(pkg, "src")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _loss(p, x, O):
    out = O.vqvae_forward(x, p, 2, 0.25, None)
    return F.mse_loss(out["recon"], x) + out["vq_loss"]


def _worker(rank, world, port, ret, spans=False):
    _setup_paths()
    from oracle import vqvae_oracle as O
    from acoustic_locating_vq_vae.train_step import FlatBuffers, shard_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    calls = {"n": 0}
    real_all_reduce = dist.all_reduce

    def counting(*a, **k):
        calls["n"] += 1
        return real_all_reduce(*a, **k)

    dist.all_reduce = counting
    shapes = O.vqvae_param_shapes(7, 16, 4, 8, 16)
    p = {k: v.clone().requires_grad_(True) for k, v in O.closed_form_params(shapes, 0.8).items()}
    if rank == 1:                                   # deliberately different init: broadcast must fix it
        with torch.no_grad():
            for v in p.values():
                v.add_(1.0)
    fb = FlatBuffers(p.values())
    fb.broadcast_params()
    xg = O.speech_preprocess(torch.from_numpy(O.hashed_uniform(4 * 7 * 13, 3, 2.0).reshape(4, 7, 13)))
    x = shard_batch(xg, rank, world)
    fb.zero_grad()
    _loss(p, x, O).backward()
    if spans:
        # the trainers' default: the same reduction issued as two contiguous spans (early = quantiser + decoder,
        # late = encoder + pre-VQ conv, a prefix of the buffer), each element reduced exactly once
        names = list(p)
        cut = next(i for i, k in enumerate(names) if not (k.startswith("_encoder") or k.startswith("_pre_vq")))
        late, early = [p[k] for k in names[:cut]], [p[k] for k in names[cut:]]
        (elo, ehi), (llo, lhi) = fb.span(early), fb.span(late)
        assert llo == 0 and lhi == elo and ehi == fb.grad.numel(), "the two spans partition the flat buffer"
        w_early = fb.sync_span(elo, ehi)
        w_late = fb.sync_span(llo, lhi)
        w_early.wait()
        w_late.wait()
        assert calls["n"] == 2
        scale = 1.0 / world
    else:
        scale = fb.sync_grads()
        assert calls["n"] == 1, "exactly one collective per step"
        assert abs(scale - 1.0 / world) < 1e-12
    grads = (fb.grad * scale).clone()
    with torch.no_grad():
        fb.flat.add_(grads, alpha=-0.1)             # any deterministic optimiser: ranks must stay identical
    gathered = [torch.zeros_like(fb.flat) for _ in range(world)]
    dist.all_gather(gathered, fb.flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0:
        ret["grads"] = grads.numpy()
        ret["same"] = same
        ret["views_alive"] = all(q.grad.data_ptr() == fb.grad.data_ptr() + 4 * o for q, o in zip(fb.params, fb.offsets))
    dist.destroy_process_group()


@pytest.mark.parametrize("spans", [False, True])
def test_two_rank_grads_equal_single_rank_on_full_batch(spans):
    _setup_paths()
    from oracle import vqvae_oracle as O
    from acoustic_locating_vq_vae.train_step import FlatBuffers
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), ret, spans), nprocs=2, join=True)
    assert ret["same"] and ret["views_alive"]
    shapes = O.vqvae_param_shapes(7, 16, 4, 8, 16)
    p = {k: v.clone().requires_grad_(True) for k, v in O.closed_form_params(shapes, 0.8).items()}
    fb = FlatBuffers(p.values())
    xg = O.speech_preprocess(torch.from_numpy(O.hashed_uniform(4 * 7 * 13, 3, 2.0).reshape(4, 7, 13)))
    fb.zero_grad()
    _loss(p, xg, O).backward()
    ref = fb.grad.numpy()
    err = np.abs(ret["grads"] - ref).max() / (np.abs(ref).max() + 1e-30)
    assert err < 1e-5, err


def test_flat_buffers_views_and_shard():
    _setup_paths()
    from acoustic_locating_vq_vae.train_step import FlatBuffers, shard_batch, unique_trainable
    a = torch.nn.Parameter(torch.randn(5, 3))
    b = torch.nn.Parameter(torch.randn(7))
    frozen = torch.nn.Parameter(torch.randn(2), requires_grad=False)
    fb = FlatBuffers([a, b, a, frozen])
    # a 64-float header first (element 0 of the gradient buffer = the step's skip slot), then 256-byte aligned slots
    assert len(fb.params) == 2 and fb.numel == 22 and fb.flat.numel() == 192
    assert a.data_ptr() == fb.flat.data_ptr() + 64 * 4 and b.data_ptr() == fb.flat.data_ptr() + 128 * 4
    assert fb.skip_slot.data_ptr() == fb.grad.data_ptr() and fb.skip_slot.numel() == 1
    (a.sum() * 2 + b.sum()).backward()
    assert float(fb.grad[64:79].sum()) == 30.0 and float(fb.grad[128:135].sum()) == 7.0 and float(fb.grad[:64].abs().sum()) == 0.0
    fb.zero_grad()
    assert float(fb.grad.abs().sum()) == 0.0 and a.grad.data_ptr() == fb.grad.data_ptr() + 64 * 4
    assert fb.span([a]) == (0, 128) and fb.span([b]) == (128, 192)      # a prefix span carries the header along
    x = torch.arange(8).view(8, 1)
    assert shard_batch(x, 1, 4).view(-1).tolist() == [2, 3]
    with pytest.raises(ValueError):
        shard_batch(x, 0, 3)
    assert fb.sync_grads() == 1.0                    # no process group: identity


# ------------------------------------------------------------------------------------------------------------------
# Trainer.step's collective pattern (north_star: "a single RCCL all-reduce of gradients ... per step").  The HIP
# kernels cannot run here, so a stand-in model with the product's attribute layout (_encoder / _pre_vq_conv / _vq /
# _decoder) computes in plain torch, and the optimiser launch is replaced; everything between -- bucket layout, the
# two-part backward, sync calls, their order -- is the product's own Trainer code.
class _TinyVQVAE(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self._encoder = torch.nn.Linear(6, 5)
        self._pre_vq_conv = torch.nn.Linear(5, 4)
        self._vq = torch.nn.Linear(4, 4, bias=False)
        self._decoder = torch.nn.Linear(4, 6)

    def forward(self, x):
        from acoustic_locating_vq_vae import _ops
        z = self._pre_vq_conv(torch.relu(self._encoder(x)))
        if _ops._LATENT_TAP is not None and z.requires_grad:
            z = _ops.tap_latent(z)
        q = self._vq(z)
        return (q - z.detach()).square().mean(), self._decoder(q), torch.zeros(())


def _trainer_worker(rank, world, port, ret, buckets):
    _setup_paths()
    import acoustic_locating_vq_vae.train_step as TS
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("ALVQ_GRAD_BUCKETS", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(t.numel())
        return real(t, *a, **k)

    dist.all_reduce = counting

    class CpuTrainer(TS.Trainer):
        def preprocess(self, raw, wiener=None):
            return raw, raw

        def forward_loss(self, x, target):
            vq_loss, recon, perp = self.model(x)
            err = F.mse_loss(recon, target)
            return err + vq_loss, err, perp

    class SGD:                                            # stands in for the one-launch flat Adam
        def __init__(self, b):
            self.b, self.step_count = b, 0

        def prepare(self, grad_scale=1.0):
            self.scale = grad_scale

        def apply(self, skip=()):
            with torch.no_grad():
                self.b.flat.add_(self.b.grad, alpha=-0.05 * self.scale)

    torch.manual_seed(3 + rank)                           # different init per rank: the broadcast must fix it
    model = _TinyVQVAE()
    tr = CpuTrainer(model, "speech", grad_buckets=buckets)
    tr.opt = SGD(tr.buffers)
    xg = torch.randn(8, 6, generator=torch.Generator().manual_seed(11))
    per_step = []
    for _ in range(3):
        n0 = len(calls)
        tr.step(TS.shard_batch(xg, rank, world))
        per_step.append(len(calls) - n0)
    gathered = [torch.zeros_like(tr.buffers.flat) for _ in range(world)]
    dist.all_gather(gathered, tr.buffers.flat)
    if rank == 0:
        ret["per_step"] = per_step
        ret["sizes"] = calls[-per_step[-1]:]
        ret["total"] = tr.buffers.grad.numel()
        ret["same"] = all(torch.equal(gathered[0], g) for g in gathered)
        ret["bucketed"] = tr._buckets is not None
    dist.destroy_process_group()


@pytest.mark.parametrize("buckets,expect", [(None, 1), (1, 1), (2, 2)])
def test_trainer_step_all_reduce_calls(buckets, expect):
    """Default (and grad_buckets=1): exactly ONE all-reduce of the whole flat buffer per Trainer.step.  grad_buckets=2:
    two calls whose spans partition the buffer.  Ranks end every step with identical parameters either way."""
    ret = mp.Manager().dict()
    mp.spawn(_trainer_worker, args=(2, _free_port(), ret, buckets), nprocs=2, join=True)
    assert ret["per_step"] == [expect] * 3, ret["per_step"]
    assert sum(ret["sizes"]) == ret["total"], "every gradient element is reduced exactly once"
    assert ret["bucketed"] == (expect == 2)
    assert ret["same"]
