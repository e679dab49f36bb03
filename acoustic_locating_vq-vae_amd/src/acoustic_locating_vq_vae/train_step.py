"""The callers' per-step arithmetic as a reusable driver, plus the data-parallel plumbing.

The reference inlines its training loops in scripts (scripts/train_speech.py:62-74,88-91;
train_rir.py:42-58,72-75; train_echoed_speech.py:62-75,89-92) and has no multi-GPU code at all.  This module
reproduces one step of each loop on the HIP path and adds what the north star asks for:

* ``FlatBuffers``   -- every trainable parameter (and its ``.grad``) is a view into one flat fp32 buffer;
* ``sync_grads`` / ``sync_span`` -- the step's all-reduce(sum) of the flat gradient buffer (RCCL over xGMI when
                       the process group backend is "nccl"; "gloo" on CPU for tests); the 1/world factor is folded
                       into the optimiser kernel.  The buffer is reduced exactly once per step: by default as ONE
                       all-reduce after the backward (what BASELINE.json's north_star specifies).  Optionally
                       (``ALVQ_GRAD_BUCKETS=2`` / ``Trainer(grad_buckets=2)``) as two contiguous spans -- quantiser +
                       decoder gradients as soon as they exist, encoder gradients at the end -- so the first overlaps
                       the encoder's backward; ``bench.py --gpus N`` times both and reports them side by side;
* ``FlatAdam``      -- torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, amsgrad=False) arithmetic
                       (train_speech.py:154) as one HIP launch over the flat buffer.

Batch sharding is by sample (rank r takes x[r*B/W:(r+1)*B/W]); with equal shards the mean of per-rank mean
losses is the global mean, including the VQ terms.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import _native as N
from . import _ops

_ALIGN = 64  # floats; keeps every parameter 256-byte aligned inside the flat buffer


def unique_trainable(params):
    seen, out = set(), []
    for p in params:
        if p.requires_grad and id(p) not in seen:
            seen.add(id(p))
            out.append(p)
    return out


class FlatBuffers:
    """Re-point ``params`` (and their grads) at slices of two flat buffers.  Device agnostic.

    Both buffers start with a header of ``_ALIGN`` floats.  Element 0 of the GRADIENT buffer is the step's SKIP SLOT: the
    last launch of a backward writes 1.0 there if the step saturated an fp16-range format (``N.range_flag_to_slot``), the
    step's all-reduce sums it over the ranks with the gradients -- so every rank reaches the same verdict -- and the
    optimiser launches leave everything untouched when it is non-zero (``FlatAdam``)."""

    def __init__(self, params):
        self.params = unique_trainable(params)
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.offsets, total = [], _ALIGN
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("all parameters must be fp32 on one device")
            self.offsets.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
        self.skip_slot = self.grad[0:1]

    def zero_grad(self):
        """Keeps the views alive (optimizer.zero_grad(set_to_none=True) would drop them)."""
        if self.grad.is_cuda:
            N.fill_(self.grad, 0.0)                  # the library's own fill: no ATen compute op on the step path
        else:
            self.grad.zero_()                        # CPU buffers exist only in the gloo plumbing tests
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view(p.shape)

    def sync_grads(self, group=None, force=False):
        """All-reduce(sum) of the whole flat gradient buffer, in place, as ONE collective.
        Returns the factor the optimiser must apply (1/world).  A one-rank group skips the call (the sum over one rank
        is the buffer itself) unless ``force`` -- the switch that lets a single-GPU box execute the RCCL path."""
        if not (dist.is_available() and dist.is_initialized()):
            return 1.0
        world = dist.get_world_size(group)
        if world == 1 and not force:
            return 1.0
        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / world

    def span(self, params):
        """[lo, hi) of the flat buffers covering ``params`` -- which must be a contiguous run of ``self.params``."""
        ids = {id(p) for p in params}
        idx = [i for i, p in enumerate(self.params) if id(p) in ids]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)) or len(idx) != len(ids):
            raise ValueError("bucket parameters are not a contiguous run of the flat buffer")
        last = idx[-1]
        hi = self.offsets[last + 1] if last + 1 < len(self.offsets) else self.grad.numel()
        return (0 if idx[0] == 0 else self.offsets[idx[0]]), hi      # a prefix span takes the header (the skip slot) along

    def sync_span(self, lo, hi, group=None, force=False):
        """Asynchronous all-reduce(sum) of grad[lo:hi); returns the work handle (None without a process group).  The
        caller ``wait()``s it -- a stream-level wait on the GPU -- before the optimiser reads the buffer."""
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
            return None
        return dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True)

    def broadcast_params(self, src=0, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.flat, src=src, group=group)


class FlatAdam:
    """Adam over FlatBuffers as HIP launches over the flat buffer.

    The step-dependent scalars (lr/bias_correction1, sqrt(bias_correction2), grad_scale) and the step counter live
    in a 4-float DEVICE buffer; ``prepare()`` advances them with a one-thread kernel in stream order.  Nothing is
    staged through host memory, so a host that queues many steps ahead of the device (graph replay does) cannot
    overwrite a step's scalars before that step's Adam launch has read them."""

    def __init__(self, buffers: FlatBuffers, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, guard=False):
        """``guard``: honour the buffers' skip slot -- a step whose slot is non-zero is not applied (parameters, moments,
        packed images untouched), does not advance the step count and is counted in ``scalars[4]``."""
        self.b = buffers
        self.lr, self.betas, self.eps = lr, betas, eps
        self.guard = bool(guard) and buffers.flat.is_cuda
        self.exp_avg = torch.zeros_like(buffers.flat)
        self.exp_avg_sq = torch.zeros_like(buffers.flat)
        self.step_count = 0                                   # steps ATTEMPTED on the host; scalars[3] = steps applied
        self.scalars = torch.zeros(N.ADAM_SCALARS, device=buffers.flat.device, dtype=torch.float32)

    def set_step(self, step):
        """Resume from a checkpointed step count."""
        self.step_count = int(step)
        self.scalars[3] = float(step)

    def prepare(self, grad_scale=1.0):
        self.step_count += 1
        N.adam_advance(self.scalars, self.lr, self.betas[0], self.betas[1], grad_scale,
                       prev_skip=self.b.skip_slot if self.guard else None)

    def applied_steps(self):
        """Optimiser steps actually applied so far (one host sync): the device's step number minus the step in flight if the
        guard skipped it (its retry will carry the same number)."""
        n = int(self.scalars[3].item())
        if self.guard and float(self.b.skip_slot.item()) != 0.0:
            n -= 1
        return n

    def skipped_steps(self, reset=True):
        """Steps the guard skipped since the counter was last reset (one host sync).  The step in flight is counted by the
        NEXT ``prepare``."""
        n = int(self.scalars[4].item())
        if reset and n:
            self.scalars[4] = 0.0
        return n

    def apply(self, skip=(), pack_pool=None):
        """One Adam launch over the flat buffer; ``skip`` = [lo, hi) element ranges that received no gradient this
        step (torch.optim.Adam leaves a parameter whose ``.grad`` is None untouched, moments included -- e.g. the
        codebook under ``set_train_vq(False)``): the launch is then split around them.

        ``pack_pool`` (a Trainer's ``_ops.PackPool``): the conv weights that have packed images in the pool are updated by
        ONE fused launch that also writes those images while the new weights are in registers (``alvq_adam_pack_batch``;
        round 2 re-read every weight in a separate packing launch at the start of the next step), everything else --
        biases, the codebook -- by one segmented launch.  Same arithmetic, bit for bit."""
        groups = pack_pool.adam_groups() if (pack_pool is not None and os.environ.get("ALVQ_ADAM_PACK", "1") != "0") else None
        skip_ptr = self.b.skip_slot if self.guard else None
        if groups:
            skipped = [(int(a), int(b)) for a, b in skip]
            by_planes, segments, fused = {}, [], set()
            for p, off in zip(self.b.params, self.b.offsets):
                n = p.numel()
                if any(a <= off and off + n <= b for a, b in skipped):
                    continue
                hit = groups.get(p.data_ptr()) if p.dim() == 3 else None
                if hit:
                    planes, imgs = hit
                    by_planes.setdefault(planes, []).append(
                        (p.data, p.grad, self.exp_avg[off:off + n].view(p.shape), self.exp_avg_sq[off:off + n].view(p.shape),
                         imgs.get(N.W_OIK), imgs.get(N.W_IOK)))
                    fused.add(p.data_ptr())
                else:
                    segments.append((off, off + n))
            for planes, entries in by_planes.items():       # one launch per weight format (per-role modes: two)
                N.adam_pack_batch(entries, planes, self.scalars, self.betas[0], self.betas[1], self.eps, skip=skip_ptr)
            N.adam_segments(self.b.flat, self.b.grad, self.exp_avg, self.exp_avg_sq, segments, self.scalars, self.betas[0],
                            self.betas[1], self.eps, skip=skip_ptr)
            pack_pool.mark_adam_packed(fused)
            _ops.bump_weight_epoch()
            return
        lo = 0
        for s_lo, s_hi in sorted(skip) + [(self.b.flat.numel(), self.b.flat.numel())]:
            if s_lo > lo:
                N.adam_step_dev(self.b.flat[lo:s_lo], self.b.grad[lo:s_lo], self.exp_avg[lo:s_lo],
                                self.exp_avg_sq[lo:s_lo], self.scalars, self.betas[0], self.betas[1], self.eps, skip=skip_ptr)
            lo = max(lo, s_hi)
        if pack_pool is not None:
            pack_pool.forget_adam_marks()   # the images were NOT written by this update: every one is stale (round-3 advisor)
        _ops.bump_weight_epoch()        # parameter memory changed behind torch's version counters: packed images are stale

    def step(self, grad_scale=1.0):
        self.prepare(grad_scale)
        self.apply()


class LocationTrainer:
    """One step of scripts/train_location.py:69-94 for the location head: codes -> LocationModule -> mse(theta / pi) ->
    backward -> Adam, with the optimiser as ONE HIP launch over a flat buffer instead of torch.optim.Adam's passes over
    fc_1's 211 M parameters (train_location.py:40), and fc_1's gradient scatter-added straight into that buffer.

    The arithmetic is torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, amsgrad=False)'s, dense: a column of fc_1 that no
    sample touched this step still decays its moments and moves by its first moment, exactly as in the reference loop.
    Per step: zero the 850 MB gradient buffer, forward / backward (a few MB), one pass over 4 x 850 MB."""

    def __init__(self, location_model, lr=1e-3, group=None):
        self.model, self.group = location_model, group
        self.buffers = FlatBuffers(location_model.parameters())
        if self.buffers.flat.is_cuda:
            _ops.register_grad_sinks(self.buffers.params)
        self.buffers.broadcast_params(group=group)
        self.opt = FlatAdam(self.buffers, lr=lr)
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.grad_scale = 1.0 / world

    def step(self, codes, theta):
        """codes: (B, L) int indices (``get_latent_indices(...)[3].view(B, L)``) or the dense one-hot (B, L, K) the script
        builds; theta: (B,) angles.  Returns the loss as a 0-dim device tensor."""
        import torch.nn.functional as F
        self.opt.prepare(self.grad_scale)
        self.buffers.zero_grad()
        with _ops.use_grad_sinks():
            location = self.model(codes)
            loss = F.mse_loss(location, torch.as_tensor(theta).float().to(location.device) / torch.pi, reduction="mean")
            loss.backward()
        self.buffers.sync_grads(self.group)
        self.opt.apply()
        return loss.detach()


def shard_batch(x, rank, world):
    """Rank's contiguous slice of the global batch (equal shards required for mean-of-means == global mean)."""
    b = x.shape[0]
    if b % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (b, world))
    per = b // world
    return x[rank * per:(rank + 1) * per]


class Trainer:
    """One train step of the speech / RIR / echoed loops on the HIP path.

    kind="speech":  x = standardise(|x_raw|);            loss = mse(recon, x) + vq_loss       (train_speech.py)
    kind="rir":     x = standardise(rir)^T; target = standardise(wiener)[:,None,:];  loss = mse + vq_loss (train_rir.py)
    kind="echoed":  x = standardise(echoed); x_rir = x^T; loss = mse(recon, x)                (train_echoed_speech.py)
    """

    def __init__(self, model, kind="speech", lr=1e-3, group=None, grad_buckets=None, force_collective=None,
                 range_check_every=None, strict_range=None):
        """``force_collective`` (or ALVQ_FORCE_COLLECTIVE=1): issue the step's all-reduce even in a one-rank process
        group, where it is the identity -- so that the RCCL call between the backward and the Adam launch can be
        executed (and is tested, tests/test_rccl_gpu.py) on a single-GPU box.

        fp16-range modes (x3mx_hb, f16mx_hb): a step in which any value saturated at 65504 (or was NaN) -- on ANY rank -- is
        SKIPPED on the device: the optimiser launches leave parameters, moments and packed weights untouched and the step
        does not count (no host sync; ALVQ_SKIP_SATURATED=0 restores apply-and-warn).  Every ``range_check_every`` steps
        (ALVQ_RANGE_CHECK_EVERY; default 200, 0 = never) the skipped-step counter and the sticky range flag are read back
        -- ONE host sync -- and reported: a RuntimeWarning, or with ``strict_range`` (ALVQ_RANGE_STRICT=1) a
        FloatingPointError when steps were skipped."""
        self.model, self.kind, self.group = model, kind, group
        self.range_check_every = int(os.environ.get("ALVQ_RANGE_CHECK_EVERY", "200")) if range_check_every is None \
            else int(range_check_every)
        self.strict_range = (os.environ.get("ALVQ_RANGE_STRICT", "0") != "0") if strict_range is None else bool(strict_range)
        self.skipped_steps = 0                # total reported by check_range so far
        self._steps_since_check = 0
        self.force_collective = (os.environ.get("ALVQ_FORCE_COLLECTIVE", "0") != "0") if force_collective is None \
            else bool(force_collective)
        # echoed loop: the reference hands Adam every parameter (train_echoed_speech.py:48) but detaches both encoder
        # outputs unless set_train_encoder(True) was called (echoed_speech_model.py:51-54), so only the decoder's
        # tensors ever receive gradients -- the flat buffers (and the all-reduce) hold exactly the tensors that do.
        # The caller's sub-models are left as they are (no requires_grad mutation).
        self._echoed_train_encoder = bool(getattr(model, "flag_train_encoder", False)) if kind == "echoed" else None
        params = model._decoder.parameters() if (kind == "echoed" and not self._echoed_train_encoder) else model.parameters()
        self.buffers = FlatBuffers(params)
        self._trainable = [p.requires_grad for p in self.buffers.params]
        if self.buffers.flat.is_cuda:
            _ops.register_grad_sinks(self.buffers.params)     # weight-grad launches accumulate straight into the flat buffer
        # every conv weight of the model (frozen sub-models of the echoed config included) keeps persistent packed
        # bf16 images, refreshed by one launch per step
        use_pool = self.buffers.flat.is_cuda and os.environ.get("ALVQ_PACK_POOL", "1") != "0"
        self.pack_pool = _ops.PackPool(list(model.parameters()), dynamic=self.buffers.params) if use_pool else None
        self.buffers.broadcast_params(group=group)
        self.opt = FlatAdam(self.buffers, lr=lr, guard=os.environ.get("ALVQ_SKIP_SATURATED", "1") != "0")
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.grad_scale = 1.0 / world
        self._graph = None
        # Two gradient buckets for the VQ-VAE loops: "late" = encoder + pre-VQ conv (a prefix of the flat buffer, its
        # gradients are produced last), "early" = quantiser + decoder.  The backward runs in two parts so the early
        # bucket's all-reduce overlaps the late part's kernels (see ``step``).  The echoed loop trains the decoder
        # only: one part, one bucket.
        self._buckets = None
        want_buckets = int(os.environ.get("ALVQ_GRAD_BUCKETS", "1")) if grad_buckets is None else grad_buckets
        if want_buckets >= 2 and kind != "echoed" and hasattr(model, "_encoder") and hasattr(model, "_pre_vq_conv"):
            late = unique_trainable(list(model._encoder.parameters()) + list(model._pre_vq_conv.parameters()))
            late_ids = {id(p) for p in late}
            early = [p for p in self.buffers.params if id(p) not in late_ids]
            if late and early:
                self._late_params, self._early_params = late, early
                self._buckets = (self.buffers.span(early), self.buffers.span(late))
        self._cut = None

    def preprocess(self, raw, wiener=None):
        if self.kind == "speech":
            x = N.standardise(_ops.dense(raw), take_abs=True)
            return x, x
        if self.kind == "rir":
            # (B, T, F): frames are channels -- standardise(raw).permute(0, 2, 1) (train_rir.py:42-45), left un-materialised: the
            # encoder's first launch converts raw into its compute layout with the standardisation fused in (csrc/boundary.hip)
            x = _ops.StandardisedT(_ops.dense(raw))
            w = wiener.float()
            tgt = N.standardise(w.unsqueeze(2).contiguous()).view(w.shape[0], 1, w.shape[1])
            return x, tgt
        x = N.standardise(_ops.dense(raw))
        return x, x

    def forward_loss(self, x, target):
        if self.kind == "echoed":
            recon, sp_perp, _ = self.model(x, x.permute(0, 2, 1))     # the view train_echoed_speech.py:66 passes; never materialised
            recon_error = _ops.MSEFn.apply(recon, target)
            return recon_error, recon_error, sp_perp
        vq_loss, recon, perplexity = self.model(x)
        recon_error = _ops.MSEFn.apply(recon, target)
        return recon_error + vq_loss, recon_error, perplexity

    def _check_frozen(self):
        """The flat buffers were laid out for the parameters that were trainable at construction; Adam walks the
        whole buffer, so a later ``requires_grad_(False)`` / ``set_train_encoder`` toggle would silently keep
        training (or never train) those tensors.  Fail loudly instead."""
        if [p.requires_grad for p in self.buffers.params] != self._trainable:
            raise RuntimeError("requires_grad of a parameter changed after the Trainer was built; build a new Trainer")
        if self.kind == "echoed" and bool(self.model.flag_train_encoder) != self._echoed_train_encoder:
            raise RuntimeError("set_train_encoder(%s) was called after the Trainer was built; build a new Trainer so "
                               "the encoders' parameters are (not) part of the optimiser" % self.model.flag_train_encoder)

    def _adam_skip(self):
        """Ranges of the flat buffer whose parameter gets no gradient this step: the codebook when its quantiser
        runs with ``set_train_vq(False)`` (vector_quantizer.py:46-50 detaches both MSE terms)."""
        skip = []
        ids = {id(p): i for i, p in enumerate(self.buffers.params)}
        for m in self.model.modules():
            if hasattr(m, "_train_vq") and not m._train_vq and id(m._embedding.weight) in ids:
                i = ids[id(m._embedding.weight)]
                skip.append((self.buffers.offsets[i], self.buffers.offsets[i] + m._embedding.weight.numel()))
        return skip

    def _body(self, raw, wiener):
        """Preprocess + forward + first part of the backward (everything above the encoder output, or the whole
        backward when the model has a single bucket).  ``_body`` and ``_body_late`` are the launch-bound part of a
        step (~130 launches) that hipGraphs capture; the collectives and the optimiser launch stay outside so no
        RCCL call is ever recorded into a graph."""
        if self.buffers.flat.is_cuda:
            N.arena_reset(self.buffers.flat.device)          # scratch of the deferred split reductions: a step's worth
        with _ops.use_pack_pool(self.pack_pool), _ops.use_grad_sinks(), _ops.deferred_reduce() as reductions:
            # one batched re-pack of every conv weight, then lookups; the weight-gradient launches leave their split
            # partials behind and ONE launch sums them all once the backward has been queued
            x, target = self.preprocess(raw, wiener)
            self.buffers.zero_grad()
            if getattr(self, "_one", None) is None or self._one.device != x.device:
                self._one = torch.ones((), device=x.device)   # the backward's root gradient, made once (autograd would fill a
                #                                               fresh ones_like(loss) every step: an ATen launch on the step path)
            if self._buckets is None:
                loss, recon_error, perplexity = self.forward_loss(x, target)
                loss.backward(self._one)
            else:
                with _ops.latent_tap() as tap:
                    loss, recon_error, perplexity = self.forward_loss(x, target)
                loss.backward(self._one)                     # decoder, quantiser -> early bucket + d(latent)
                self._cut = tap[0] if tap else None
            if reductions:
                N.wgrad_reduce_batch(reductions)
            if self._cut is None:
                self._post_verdict()
        return loss.detach(), recon_error.detach(), perplexity.detach()

    def _post_verdict(self):
        """Last launch of the step's backward: did this step saturate an fp16-range format?  -> the skip slot, which the
        step's all-reduce then sums over the ranks (it is element 0 of the flat gradient buffer)."""
        if getattr(self.opt, "guard", False) and _ops.has_fp16_range():
            N.range_flag_to_slot(self.buffers.skip_slot)

    def _body_late(self):
        """Second part of the backward: the encoder and the pre-VQ conv, from the latent's gradient."""
        if self._cut is None:
            return
        z, leaf = self._cut
        self._cut = None
        with _ops.use_pack_pool(self.pack_pool, refresh=False), _ops.use_grad_sinks(), _ops.deferred_reduce() as reductions:
            z.backward(leaf.grad)
            if reductions:
                N.wgrad_reduce_batch(reductions)
            self._post_verdict()

    def _sync_early(self):
        return self.buffers.sync_span(*self._buckets[0], group=self.group, force=self.force_collective) if self._buckets else None

    def _finish(self, early_work=None):
        if self._buckets is None:
            self.buffers.sync_grads(self.group, force=self.force_collective)   # single bucket: the step's one collective
        else:
            late_work = self.buffers.sync_span(*self._buckets[1], group=self.group, force=self.force_collective)
            for w in (early_work, late_work):
                if w is not None:
                    w.wait()                                   # stream-level wait: the Adam launch queues behind both
        if self.pack_pool is not None:                         # Adam + the re-pack of the conv weights' images in one launch
            self.opt.apply(self._adam_skip(), pack_pool=self.pack_pool)
        else:
            self.opt.apply(self._adam_skip())                  # one Adam launch over the flat buffer

    # ------------------------------------------------------------------------------------------- checkpoint / resume
    def state_dict(self):
        """Everything a resumed run needs beyond ``model.state_dict()``: the Adam moments (flat, in parameter order)
        and the step count.  The reference only ever saves the model (``torch.save(model, ...)``), so its resumed
        runs restart Adam from zero; with this a resumed step is bitwise the step that would have come next."""
        return {"model": self.model.state_dict(), "exp_avg": self.opt.exp_avg.clone(), "exp_avg_sq": self.opt.exp_avg_sq.clone(),
                "step": self.opt.applied_steps(),                 # steps APPLIED (a skipped step does not count)
                "numel": self.buffers.flat.numel(), "kind": self.kind}

    def load_state_dict(self, state):
        if state["numel"] != self.buffers.flat.numel() or state["kind"] != self.kind:
            raise ValueError("checkpoint is for a different trainer (kind %s, %d flat elements)" % (state["kind"], state["numel"]))
        self.model.load_state_dict(state["model"])           # copies INTO the flat-buffer views
        self.opt.exp_avg.copy_(state["exp_avg"])
        self.opt.exp_avg_sq.copy_(state["exp_avg_sq"])
        self.opt.set_step(state["step"])

    def _jitters(self):
        from .vq_vae.modules.jitter import Jitter
        if not self.model.training:
            return []
        return [m for m in self.model.modules() if isinstance(m, Jitter)]

    def check_range(self):
        """Read (and clear) the device's fp16 range state: the sticky flag -- 0, or bits {1: an input reached 65504, 2: an
        input was NaN, 4: a convolution produced such a value} -- and the number of steps the optimiser skipped because of
        it.  One host sync; ``step`` calls it every ``range_check_every`` steps in the fp16-range modes.  Returns the flag."""
        if not (self.buffers.flat.is_cuda and _ops.has_fp16_range()):
            return 0
        flag = N.f16mx_range_flag(reset=True, device=self.buffers.flat.device)
        skipped = self.opt.skipped_steps(reset=True) if self.opt.guard else 0
        self.skipped_steps += skipped
        if flag or skipped:
            msg = ("acoustic_locating_vq_vae: fp16 range flag %d in mode %s -- a value entering or produced inside the fp16-range "
                   "formats reached 65504 (or was NaN) since the last check; %s.  ALVQ_DTYPE=bf16x3_hb or f32 have fp32 range."
                   % (flag, _ops.get_compute_dtype(),
                      ("%d optimiser step(s) were SKIPPED (parameters untouched)" % skipped) if self.opt.guard
                      else "the results of those steps are saturated"))
            if self.strict_range and skipped:
                raise FloatingPointError(msg)
            import warnings
            warnings.warn(msg, RuntimeWarning)
        return flag

    def step(self, raw, wiener=None):
        """Returns (loss, recon_error, perplexity) as 0-dim device tensors -- no host sync in here (except the periodic
        range check of the f16mx modes, every ``range_check_every`` steps).

        Order on the stream:  part 1 (fwd + decoder/quantiser backward)  ->  all-reduce(early bucket) starts  ->
        part 2 (encoder backward) runs while it is in flight  ->  all-reduce(late bucket)  ->  Adam."""
        self._check_frozen()
        if self.range_check_every > 0:
            self._steps_since_check += 1
            if self._steps_since_check >= self.range_check_every:
                self._steps_since_check = 0
                self.check_range()
        if self.pack_pool is not None:
            self.pack_pool.refresh_static()        # frozen weights (echoed encoders): re-packed only if someone changed them
            if self._graph is not None and self.pack_pool.stale_dynamic():
                self.pack_pool.refresh()           # a parameter was modified behind the fused Adam's images (load_state_dict):
                #                                    the replayed graph holds no re-pack of them
        # a batch of another shape than the captured one (the ragged last batch of an epoch) runs as eager launches: the
        # graphs' static buffers have one shape, and copy_ would broadcast a one-sample batch into them without a word
        replay = self._graph is not None and tuple(raw.shape) == tuple(self._static_raw.shape) and \
            (wiener is None or tuple(wiener.shape) == tuple(self._static_wiener.shape))
        if not replay:
            if self._graph is not None:
                # the ragged batch of a captured trainer: the jitters are pinned to their static buffers (same L), which
                # only refresh() redraws -- without it the step would reuse the previous step's columns and leave the
                # np.random stream one draw behind the reference's from here on (round-3 advisor finding)
                for j in self._graph_jitters:
                    j.refresh()
            self.opt.prepare(self.grad_scale)
            out = self._body(raw, wiener)
            early = self._sync_early()
            self._body_late()
            self._finish(early)
            return out
        # replay: refresh the graphs' static inputs (batch, jitter columns, Adam scalars), then two launches
        self._static_raw.copy_(raw, non_blocking=True)
        if wiener is not None:
            self._static_wiener.copy_(wiener, non_blocking=True)
        for j in self._graph_jitters:
            j.refresh()
        self.opt.prepare(self.grad_scale)
        self._graph.replay()
        early = self._sync_early()
        if self._graph_late is not None:
            self._graph_late.replay()
        self._finish(early)
        return self._static_out

    def evaluate(self, raw, wiener=None):
        """The loops' validation step (train_speech.py:57-59,76-86; train_rir.py:36-40,60-70): ``model.eval()`` -- so no jitter and
        no draw from ``np.random`` -- the same preprocessing, forward and losses, no backward, no update; the model's mode is
        restored.  Returns (loss, recon_error, perplexity) as 0-dim device tensors.  (The reference runs it outside
        ``torch.no_grad()`` and throws the autograd graph away; nothing is recorded here.)"""
        was_training = self.model.training
        self.model.eval()
        try:
            if self.pack_pool is not None:
                self.pack_pool.refresh_static()
            with torch.no_grad(), _ops.use_pack_pool(self.pack_pool):
                x, target = self.preprocess(raw, wiener)
                loss, recon_error, perplexity = self.forward_loss(x, target)
        finally:
            self.model.train(was_training)
        return loss.detach(), recon_error.detach(), perplexity.detach()

    def capture(self, raw, wiener=None, warmup=3):
        """Capture the two launch-bound parts of a step into hipGraphs (one memory pool).  Runs ``warmup`` real
        training steps first (allocator + workspaces + packed-weight pool reach steady state)."""
        assert self._graph is None, "already captured"
        self._static_raw = raw.clone()
        self._static_wiener = wiener.clone() if wiener is not None else None
        L = self._static_raw.shape[1] if self.kind == "rir" else self._static_raw.shape[2]
        self._graph_jitters = self._jitters()
        for j in self._graph_jitters:
            j.pin_buffer(L, raw.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                for j in self._graph_jitters:
                    j.refresh()
                self.opt.prepare(self.grad_scale)
                self._body(self._static_raw, self._static_wiener)
                early = self._sync_early()
                self._body_late()
                self._finish(early)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # capture_error_mode="thread_local": with a process group alive, ProcessGroupNCCL's watchdog THREAD polls the events
        # of earlier collectives (hipEventQuery); under the default "global" mode such a call from another thread during
        # the capture is an error that invalidates it (seen as an intermittent hipErrorStreamCaptureInvalidated in
        # tests/test_rccl_gpu.py: about one capture in ten).  No collective is ever recorded into these graphs.
        mode = self._capture_mode = os.environ.get("ALVQ_CAPTURE_MODE", "thread_local")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode=mode):
            self._static_out = self._body(self._static_raw, self._static_wiener)
        graph_late = None
        if self._cut is not None:
            graph_late = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_late, pool=graph.pool(), capture_error_mode=mode):
                self._body_late()
        self._graph, self._graph_late = graph, graph_late
        return self
