"""Autograd glue: one ``torch.autograd.Function`` per reference module, each a hand-scheduled chain of
fused HIP launches (see include/alvq.h).  Forward keeps only post-ReLU activations: because the reference's
``nn.ReLU(True)`` rewrites every skip operand in place (residual.py:36,66; convolutional_encoder.py:42 --
SURVEY App. B.2) no pre-activation value is ever consumed, so bias/skip/ReLU live in the producing conv's
epilogue and the ReLU-backward masks live in the data-grad conv's epilogue.

Notation follows SURVEY App. A:  t_r = relu(h_{r-1}),  u_r = relu(W1 *3 t_r),  h_r = t_r + W2 *1 u_r.

Two precisions share these chains through a small "engine" object:

* ``f32``  -- activations (B,C,L) fp32 exactly as the reference lays them out, exact-fp32 MFMA (parity mode);
* ``bf16`` -- activations in the NLC-padded bf16 layout between convs, bf16 MFMA with fp32 accumulation
  (BASELINE configs[1]); module inputs/outputs, the quantiser, losses, gradients of parameters and the
  optimiser state stay fp32;
* ``bf16x3`` -- the same pipeline with every tensor split into two bf16 planes (hi, lo) and every product
  evaluated as hi*hi + hi*lo + lo*hi: fp32-grade parity (~1e-5) at a third of the bf16 MFMA rate, i.e. several
  times the exact-fp32 MFMA that gfx950 offers (it has no TF32/xf32).

* ``f16mx`` -- the split pipeline at TWO matrix-pipe units per product: every tensor is an fp16 plane plus an fp8
  (hi, lo) plane, products are fp16*fp16 (one fp16 MFMA) + hi8*lo8 + lo8*hi8 (one block-scaled fp8 MFMA at twice the
  rate); ~1.5e-5 per product, same parity class as bf16x3 at ~2/3 of its matrix time.  Gradients run under a
  power-of-two loss scale chosen on the device (csrc/f16mx_common.h).

* ``f16mx_hb`` -- f16mx forward + fp16 ("half") backward: the forward is f16mx's bit for bit; every backward product is
  one fp16 MFMA on the H planes of the saved activations / packed weights and of gradients kept as ONE fp16 plane under the
  loss scale.  Measured against f32 at the speech config (tests/analysis/gate_flips.py): gradient rel-L2 4.8e-4 median (f16mx:
  4.3e-4) -- in both modes the gradient error is set by the ~1e-5 forward noise flipping ReLU gates, not by the backward
  products -- at 0.72x the f16mx step time.

* ``bf16x3_hb`` -- bf16x3 forward + bf16 ("half") backward: the forward is bf16x3's bit for bit -- the most accurate split
  format: 6e-6, i.e. 2.3x fewer flipped near-ties than the f16mx family (expected 14 against 33 per million codebook rows,
  tests/analysis/near_ties.py; none in the 47 834 rows of the reference goldens, where f16mx flips one) -- and every backward product
  is ONE bf16 MFMA on the hi planes (no loss scale: bf16 has fp32's range).  Gradient rel-L2 ~2e-3 per tensor (the bf16
  operands), 0.94x f16mx_hb's rate.

* ``f16mx_hd`` -- (opt-in) f16mx_hb with the DECODER's forward on fp16 operands too (one fp16 plane per activation, one fp16
  MFMA per product; the H image of the same packed weights).  Encoder, pre-VQ convolution and quantiser are f16mx_hb's bit
  for bit, so the codebook indices stay bit-exact; the reconstruction carries fp16's operand rounding through ~10 layers:
  5e-4 ... 7e-4 of the fp32 result at the default configs, 1.3e-3 measured on a 48-channel model -- at the north star's 1e-3,
  not safely inside it -- and the decoder-side gradients see ~30x the ReLU gate flips.  0.87x the f16mx_hb step time.

* ``x3mx_hb`` -- (round 4; the DEFAULT) per-role arithmetic: everything the codebook indices depend on -- the encoder
  and the pre-VQ convolution -- runs the bf16x3 forward (6.6e-6: every index of every reference golden, the 1.8e-6 RIR
  near-tie included), the decoder's forward runs f16mx (2e-5 on the reconstruction), and every backward product is ONE 16-bit
  MFMA: bf16 on the hi planes of the encoder's saved activations, fp16 under the loss scale on the H planes of the decoder's.

Select with ``set_compute_dtype(...)`` or the environment variable ``ALVQ_DTYPE``.  User-selectable modes (``MODES``):
``x3mx_hb`` (default), ``f16mx_hb`` (2-3 % faster; resolves reference near-ties down to ~4e-6 relative), ``bf16x3_hb``,
``f32`` (exact-fp32 MFMA: strict gradient parity, 1e-6), ``bf16`` (throughput; ~1 % of the indices differ).  ``f16mx``,
``bf16x3`` and ``f16mx_hd`` are INTERNAL engines since round 4 (their forwards live on inside the ``_hb`` modes; the
kernel-level tests reach them with ``set_compute_dtype(name, internal=True)``).  The f16mx-range modes carry fp16's range:
activations must stay below 65504; a step whose values saturate is SKIPPED by the Trainer's optimiser launch (csrc/
pack_weights.hip) and counted (``_native.f16mx_range_state()``).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _native as N

OIK, IOK = N.W_OIK, N.W_IOK

DEFAULT_DTYPE = "x3mx_hb"
MODES = ("x3mx_hb", "f16mx_hb", "bf16x3_hb", "f32", "bf16")      # user-selectable (ALVQ_DTYPE / set_compute_dtype)
INTERNAL_MODES = ("f16mx", "bf16x3", "f16mx_hd")                 # engines kept for the kernel-level tests only
_DTYPE = os.environ.get("ALVQ_DTYPE", DEFAULT_DTYPE)
if _DTYPE not in MODES:
    raise ValueError("ALVQ_DTYPE must be one of %s, got %r" % (", ".join(repr(m) for m in MODES), _DTYPE))


def set_compute_dtype(name, internal=False):
    """Select the arithmetic of every later forward.  ``internal=True`` also admits the retired engines (tests)."""
    global _DTYPE
    if name not in MODES and not (internal and name in INTERNAL_MODES):
        raise ValueError("compute dtype must be one of %s, got %r" % (", ".join(repr(m) for m in MODES), name))
    _DTYPE = name


def has_fp16_range(mode=None):
    """Does the mode keep any tensor in an fp16-range format (f16mx planes, fp16 gradients)?"""
    return (mode or _DTYPE) in ("x3mx_hb", "f16mx_hb", "f16mx", "f16mx_hd")


def get_compute_dtype():
    return _DTYPE


# ---------------------------------------------------------------------------------------------------
# gradient sinks: train_step.FlatBuffers registers, per parameter, the slice of its flat gradient buffer.  A
# weight-gradient launch then accumulates straight into that slice (the split reduction does dw += sum) and the
# autograd node returns None for it -- no per-parameter temporary, no extra elementwise "grad += new" pass.
# ---------------------------------------------------------------------------------------------------
import weakref

_GRAD_SINKS = weakref.WeakValueDictionary()


def register_grad_sinks(params):
    for p in params:
        if p.grad is not None:
            _GRAD_SINKS[p.data_ptr()] = p.grad


_SINKS_ACTIVE = 0


class use_grad_sinks:
    """Scope of the gradient sinks: only inside ``with use_grad_sinks():`` (the body of a Trainer step) do the
    weight-gradient launches write into the flat buffer and the autograd nodes return None for those weights.
    Everywhere else -- ``torch.autograd.grad`` on the same model, a plain ``loss.backward()`` from user code --
    autograd sees ordinary gradient tensors."""

    def __enter__(self):
        global _SINKS_ACTIVE
        _SINKS_ACTIVE += 1

    def __exit__(self, *exc):
        global _SINKS_ACTIVE
        _SINKS_ACTIVE -= 1


# deferred split reductions: inside ``with deferred_reduce() as descs:`` (a Trainer step's backward) the weight-gradient
# launches that accumulate into a gradient sink leave their partials in arena scratch and append descriptors; the trainer
# sums them all with ONE launch when the backward has been queued (_native.wgrad_reduce_batch).
_DEFERRED = None


class deferred_reduce:
    def __enter__(self):
        global _DEFERRED
        self.prev, _DEFERRED = _DEFERRED, (N.DeferredReductions() if os.environ.get("ALVQ_DEFER_REDUCE", "1") != "0" else None)
        return _DEFERRED

    def __exit__(self, *exc):
        global _DEFERRED
        _DEFERRED = self.prev


# The module API without a Trainer (scripts/train_speech.py's own loop: model(x); loss.backward(); torch.optim.Adam.step()) has
# nobody watching the fp16 range flag: poll it every ALVQ_RANGE_CHECK_EVERY (200) training-mode forwards -- one host read,
# next to the script's own two .item() syncs per step (train_speech.py:93-94) -- and warn.  Nothing can be skipped here: the
# optimiser is torch's.
_API_FORWARDS = 0


def note_training_forward(device):
    global _API_FORWARDS
    if device.type != "cuda" or _SINKS_ACTIVE or not has_fp16_range() or torch.cuda.is_current_stream_capturing():
        return
    every = int(os.environ.get("ALVQ_RANGE_CHECK_EVERY", "200"))
    _API_FORWARDS += 1
    if every <= 0 or _API_FORWARDS < every:
        return
    _API_FORWARDS = 0
    flag = N.f16mx_range_flag(reset=True, device=device)
    if flag:
        import warnings
        warnings.warn("acoustic_locating_vq_vae: fp16 range flag %d in mode %s -- a value entering or produced inside the "
                      "fp16-range formats reached 65504 (or was NaN) during the last %d forward / backward passes; those steps "
                      "are saturated.  ALVQ_DTYPE=bf16x3_hb or f32 have fp32 range; train_step.Trainer skips such steps."
                      % (flag, _DTYPE, every), RuntimeWarning)


def _sink(t):
    if t is None or not _SINKS_ACTIVE or not len(_GRAD_SINKS) or not t.requires_grad:
        return None
    return _GRAD_SINKS.get(t.data_ptr())


# ---------------------------------------------------------------------------------------------------
# packed-weight pool: inside a train step (train_step.Trainer._body) the bf16 engines do not re-pack a parameter
# every time a layer uses it.  The pool keeps one persistent packed image per (parameter, layout); the trainer
# refreshes all of them with ONE batched launch at the start of the step (after the previous optimiser update) and
# the engines just look them up.  Outside a trainer step the pool is inactive and every use packs for itself, so
# a model evaluated after training never sees a stale image.
# ---------------------------------------------------------------------------------------------------
class PackPool:
    """``dynamic``: the parameters the trainer's optimiser updates every step -- their images are refreshed by the step's
    batched launch.  Every other conv weight of the model (the frozen encoders of the echoed config) is STATIC: packed at
    first use and again only when its version counter says it changed (``refresh_static``, called by the trainer outside
    any graph capture) -- round 2 re-packed ~30 MB of never-changing encoder weights on every echoed step."""

    def __init__(self, params, dynamic=None):
        self._eligible = {p.data_ptr(): p for p in params if p.dim() == 3}
        self._dynamic = None if dynamic is None else {p.data_ptr() for p in dynamic}
        self._entries = {}          # (data_ptr, layout, planes) -> (packed image, tag)
        self._static_version = {}   # key -> parameter version the static image was packed at
        self._adam_version = {}     # key -> parameter version at which the fused Adam + pack launch wrote the image

    def _is_static(self, ptr):
        return self._dynamic is not None and ptr not in self._dynamic

    def lookup(self, w, layout, planes):
        key = (w.data_ptr(), layout, planes)
        hit = self._entries.get(key)
        if hit is not None:
            return hit
        p = self._eligible.get(key[0])
        if p is None or p.shape != w.shape:
            return None
        # first use: allocate the persistent image and fill it now; later steps refresh it in the batched launch
        img, tag = N.packed_weight_alloc(p.detach(), layout, planes)
        N.pack_weights_batch([(p.detach(), img, layout)], planes)
        self._entries[key] = (img, tag)
        if self._is_static(key[0]):
            self._static_version[key] = p._version
        return self._entries[key]

    def _refresh(self, keys):
        by_planes = {}
        for key in keys:
            ptr, layout, planes = key
            by_planes.setdefault(planes, []).append((self._eligible[ptr].detach(), self._entries[key][0], layout))
        for planes, entries in by_planes.items():
            N.pack_weights_batch(entries, planes)

    def refresh(self):
        """The step's batched re-pack: every image of a parameter the optimiser updates -- EXCEPT the images the fused
        Adam + pack launch wrote at the end of the previous step (``adam_groups`` / ``mark_adam_packed``), which are
        already current unless someone has modified the parameter since (its version counter says so)."""
        keys = [k for k in self._entries if not self._is_static(k[0]) and
                self._adam_version.get(k) != self._eligible[k[0]]._version]
        if keys:
            self._refresh(keys)
            for k in keys:
                self._adam_version.pop(k, None)

    def refresh_static(self):
        """Host-side check, outside graph capture: re-pack a static image whose parameter was modified since."""
        stale = [k for k, v in self._static_version.items() if self._eligible[k[0]]._version != v]
        if stale:
            self._refresh(stale)
            for k in stale:
                self._static_version[k] = self._eligible[k[0]]._version

    def stale_dynamic(self):
        """Dynamic images that a captured graph (which holds no re-pack of them) would read stale: written by the fused
        Adam, parameter modified behind it since (load_state_dict, a broadcast).  The trainer re-packs them eagerly."""
        return [k for k, v in self._adam_version.items() if self._eligible[k[0]]._version != v]

    def adam_groups(self):
        """{data_ptr: (planes, {layout: image})} over the dynamic entries, or None when some parameter has images in two
        formats (a mode switch on a live Trainer: then the optimiser and the re-pack stay separate launches).  In the
        per-role modes the encoder's and the decoder's weights carry different formats -- one fused launch per format."""
        groups = {}
        for (ptr, layout, pl), (img, _) in self._entries.items():
            if self._is_static(ptr):
                continue
            hit = groups.setdefault(ptr, (pl, {}))
            if hit[0] != pl:
                return None
            hit[1][layout] = img
        return groups or None

    def forget_adam_marks(self):
        """The optimiser ran WITHOUT writing the images (unfused path): every image is stale until the next refresh."""
        self._adam_version.clear()

    def mark_adam_packed(self, ptrs):
        for key in self._entries:
            if key[0] in ptrs:
                self._adam_version[key] = self._eligible[key[0]]._version


_ACTIVE_POOL = None

# ---------------------------------------------------------------------------------------------------
# packed-weight cache for everything OUTSIDE a Trainer step (SURVEY 8b "Ownership": caller-owned re-layouts, cached and
# invalidated by the parameter's version counter): the script loop (model(x); loss.backward(); torch.optim.Adam.step())
# packs each (weight, layout) once per optimiser step instead of once per use, and a model whose weights do not change --
# evaluation, the frozen encoders of the echoed model -- packs them once.  An entry is valid only for the SAME tensor
# object (weak reference: a new parameter that reuses a freed one's address never hits) at the SAME ``_version`` (every
# in-place update torch knows about bumps it: optimizer.step, load_state_dict, copy_) and the same weight epoch
# (``bump_weight_epoch``: the HIP optimiser writes parameter memory behind torch's back).  Writes through ``.data`` are
# invisible to the version counter -- as they are to autograd -- call ``invalidate_packed_weights()`` after them.
# ---------------------------------------------------------------------------------------------------
_PACK_CACHE = {}
_WEIGHT_EPOCH = 0
PACK_CACHE_STATS = {"hits": 0, "packs": 0}


def bump_weight_epoch():
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


def invalidate_packed_weights():
    _PACK_CACHE.clear()


def _cached_pack(w, layout, wplanes):
    if os.environ.get("ALVQ_PACK_CACHE", "1") == "0":
        return N.pack_weight(w.detach(), layout, wplanes)
    key = (w.data_ptr(), layout, wplanes)
    hit = _PACK_CACHE.get(key)
    if hit is not None:
        ref, version, epoch, shape, packed = hit
        if ref() is w and version == w._version and epoch == _WEIGHT_EPOCH and shape == w.shape:
            PACK_CACHE_STATS["hits"] += 1
            return packed
    if len(_PACK_CACHE) > 512:                       # dead entries of discarded models
        for k in [k for k, v in _PACK_CACHE.items() if v[0]() is None]:
            del _PACK_CACHE[k]
    packed = N.pack_weight(w.detach(), layout, wplanes)
    PACK_CACHE_STATS["packs"] += 1
    try:
        _PACK_CACHE[key] = (weakref.ref(w), w._version, _WEIGHT_EPOCH, w.shape, packed)
    except TypeError:
        pass
    return packed

# two-part backward (train_step.Trainer): while a list is installed here, ConvolutionalVQVAE.forward hands the
# encoder output over as a detached leaf and records (graph output, leaf) so the trainer can run
# ``loss.backward()`` (fills leaf.grad) and later ``output.backward(leaf.grad)``.
_LATENT_TAP = None


def tap_latent(z):
    leaf = z.detach().requires_grad_(True)
    _LATENT_TAP.append((z, leaf))
    return leaf


class latent_tap:
    def __enter__(self):
        global _LATENT_TAP
        self.prev, _LATENT_TAP = _LATENT_TAP, []
        return _LATENT_TAP

    def __exit__(self, *exc):
        global _LATENT_TAP
        _LATENT_TAP = self.prev



class use_pack_pool:
    """with use_pack_pool(pool): ...  -- the body of a trainer step"""

    def __init__(self, pool, refresh=True):
        self.pool, self.do_refresh = pool, refresh

    def __enter__(self):
        global _ACTIVE_POOL
        self.prev, _ACTIVE_POOL = _ACTIVE_POOL, self.pool
        if self.pool is not None and self.do_refresh:
            self.pool.refresh()

    def __exit__(self, *exc):
        global _ACTIVE_POOL
        _ACTIVE_POOL = self.prev


def _wgrad(eng, dy, x, kw, layout, w, b=None, dw_prev=None):
    """(dw, db) of one conv use.  Sinked tensors come back as None (already accumulated in place); otherwise dw is
    accumulated onto ``dw_prev`` (shared residual weights) or freshly allocated.  A frozen weight
    (``requires_grad_(False)``) costs nothing: no launch, no sink write, gradient None -- as autograd would."""
    if b is not None and not b.requires_grad:
        b = None
    if not w.requires_grad:
        if b is None:
            return None, None
        raise RuntimeError("a convolution with a frozen weight and a trainable bias is not supported on the HIP path")
    sw, sb = _sink(w), _sink(b)
    if sw is not None and (b is None or sb is not None):
        if _DEFERRED is not None and getattr(eng, "can_defer", False):
            eng.wgrad(dy, x, kw, layout, want_bias=b is not None, dw_out=sw, dbias_out=sb, accumulate=True, defer=_DEFERRED)
        else:
            eng.wgrad(dy, x, kw, layout, want_bias=b is not None, dw_out=sw, dbias_out=sb, accumulate=True)
        return None, None
    if b is not None:
        dw, db = eng.wgrad(dy, x, kw, layout, want_bias=True, dw_out=dw_prev, accumulate=dw_prev is not None)
        return dw, db
    return eng.wgrad(dy, x, kw, layout, dw_out=dw_prev, accumulate=dw_prev is not None), None


def dense(x):
    """Materialise a strided input as dense (B,C,L) fp32 on the GPU (train_rir.py:45 hands over a permuted view)."""
    if isinstance(x, StandardisedT):
        return x.materialise()
    if x.dtype != torch.float32:
        x = x.float()
    if x.is_contiguous():
        return x
    if x.dim() == 3 and x.permute(0, 2, 1).is_contiguous():
        return N.transpose12(x.permute(0, 2, 1))
    return x.contiguous()


class StandardisedT:
    """The tensor ``standardise(raw).permute(0, 2, 1)`` of train_rir.py:42-45 -- raw (B, F, T) fp32 contiguous on the GPU,
    standardised over dim 1, seen as (B, T, F) -- NOT materialised: the Trainer hands it to the model, the first engine to
    ``enter`` it converts raw straight into its compute layout with the standardisation fused into that pass
    (``N.rows_to_nlc``), the fp32 engine (and any format the fused kernel does not serve) materialises it."""

    def __init__(self, raw):
        self.raw = raw
        b, f, t = raw.shape
        self.shape = torch.Size((b, t, f))
        self.device, self.is_cuda, self.dtype, self.requires_grad = raw.device, raw.is_cuda, raw.dtype, False

    def dim(self):
        return 3

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def materialise(self):
        return N.transpose12(N.standardise(self.raw))


def _as_blc(x):
    """x (B, C, L) that is the ``permute(0, 2, 1)`` view of a contiguous fp32 (B, L, C) tensor -> that tensor, else None."""
    if torch.is_tensor(x) and x.dim() == 3 and x.dtype == torch.float32 and not x.is_contiguous():
        xt = x.permute(0, 2, 1)
        if xt.is_contiguous():
            return xt
    return None


def _need_gpu(x, who):
    if not x.is_cuda:
        raise RuntimeError("%s: input is on %s -- this build runs the HIP path only (no CPU fallback); "
                           "move the model and data to the GPU" % (who, x.device))
    # one process per GPU: the launches go to the CURRENT device's stream, so a tensor that lives on another card would
    # be handed to a kernel running elsewhere -- refuse instead (torch's own ops switch devices under the hood; these do not)
    if x.device.index is not None and x.device.index != torch.cuda.current_device():
        raise RuntimeError("%s: input is on %s but the current device is cuda:%d -- call torch.cuda.set_device(%d) "
                           "(one process per GPU)" % (who, x.device, torch.cuda.current_device(), x.device.index))


# --------------------------------------------------------------------------------------------------
# precision engines
# --------------------------------------------------------------------------------------------------
class _F32Engine:
    name = "f32"

    def enter(self, x, grad=False):
        return dense(x)

    def leave(self, a):
        return a

    def conv(self, x, w, layout=OIK, bias=None, skip1=None, skip2=None, mask=None, post=None, relu=False, out_f32=False):
        return N.conv1d(x, w, bias, skip1, skip2, mask, post, relu, layout)

    def wgrad(self, dy, x, kw, layout, want_bias=False, dw_out=None, dbias_out=None, accumulate=False):
        return N.conv1d_wgrad(dy, x, kw, layout, want_bias=want_bias, dw_out=dw_out, dbias_out=dbias_out,
                              accumulate=accumulate)

    def relu_mask(self, dy, t):
        return N.relu_mask(dy, t)

    def pack(self, act):
        return act, None

    def unpack(self, tensor, meta):
        return tensor


class _BF16Engine:
    name = "bf16"
    planes = 1          # planes of an activation
    wplanes = 1         # format code of a packed weight (alvq_pack_weights_bf16_batch)
    fmt = "bf16"

    def _w(self, w, layout):
        if _ACTIVE_POOL is not None:
            hit = _ACTIVE_POOL.lookup(w, layout, self.wplanes)
            if hit is not None:
                return hit
        if torch.cuda.is_current_stream_capturing():
            # inside a user's own graph capture a cache hit would freeze a packed image into the graph while later
            # replays expect fresh weights: pack in the graph, every time
            return N.pack_weight(w.detach(), layout, self.wplanes)
        return _cached_pack(w, layout, self.wplanes)

    def _enter_rows(self, x, planes, fmt):
        """The input boundary without a transposition: a permuted view of a contiguous tensor (train_rir.py:45) or the
        Trainer's not-yet-standardised batch has its channel axis contiguous already -- one pass, straight into the layout."""
        if os.environ.get("ALVQ_ROWS_BOUNDARY", "1") == "0":
            return None
        if isinstance(x, StandardisedT):
            if N.rows_to_nlc_supported(fmt, x.raw.shape[1], True):
                return N.rows_to_nlc(x.raw, planes, fmt, standardise=True)
            return None
        blc = _as_blc(x)
        if blc is not None and blc.is_cuda and N.rows_to_nlc_supported(fmt, blc.shape[1]):
            return N.rows_to_nlc(blc, planes, fmt)
        return None

    def enter(self, x, grad=False):
        """fp32 (B,C,L) -> the engine's layout.  ``grad``: x is a gradient entering a backward chain (only the f16mx
        engine cares: it picks the chain's loss scale from x)."""
        if not grad:
            hit = self._enter_rows(x, self.planes, self.fmt)
            if hit is not None:
                return hit
        return N.ncl_to_nlc(dense(x), self.planes)

    def leave(self, a):
        return N.nlc_to_ncl(a)

    def conv(self, x, w, layout=OIK, bias=None, skip1=None, skip2=None, mask=None, post=None, relu=False, out_f32=False):
        return N.conv1d_bf16(x, self._w(w, layout), bias, skip1, skip2, mask, post, relu, out_ncl=out_f32)

    can_defer = True     # the launch defers only for the bf16 / fp16 operand formats; the others reduce at once

    def wgrad(self, dy, x, kw, layout, want_bias=False, dw_out=None, dbias_out=None, accumulate=False, defer=None):
        return N.conv1d_wgrad_bf16(dy, x, kw, layout, want_bias=want_bias, dw_out=dw_out, dbias_out=dbias_out,
                                   accumulate=accumulate, defer=defer)

    def relu_mask(self, dy, t):
        return N.relu_mask_bf16(dy, t)

    @property
    def wgrad_multi(self):
        """sum_i wgrad(dy_i, x_i) in one launch (every NLC mode)."""
        return N.conv1d_wgrad_bf16_multi

    def pack(self, act):
        return act.storage, (act.B, act.L, act.C, act.planes, act.has_bits, act.fmt)

    def unpack(self, tensor, meta):
        return N.NLC.wrap(tensor, *meta)


class _BF16x3Engine(_BF16Engine):
    name = "bf16x3"
    planes = 2
    wplanes = 2
    fmt = "bf16x3"


class _F16MXEngine(_BF16Engine):
    name = "f16mx"
    planes = 2
    wplanes = 3
    fmt = "f16mx"

    def enter(self, x, grad=False):
        if not grad:
            hit = self._enter_rows(x, 2, "f16mx")
            if hit is not None:
                return hit
        x = dense(x)
        return N.ncl_to_nlc(x, 2, "f16mx", N.grad_scale(x) if grad else None)

    @property
    def wgrad_multi(self):
        return N.conv1d_wgrad_bf16_multi


class _F16MXHBEngine(_F16MXEngine):
    """f16mx forward, fp16 ("half") backward: the forward pass -- everything the reference's outputs, codebook indices and
    reconstructions depend on -- is f16mx's, bit for bit; gradients enter their chains as ONE fp16 plane under the
    device-chosen loss scale and every backward product is fp16 x fp16 with fp32 accumulation (the H planes of the saved
    f16mx activations and packed weights are the operands: nothing is converted or stored twice).  ~5e-4 per backward
    product instead of 1.5e-5: mixed-precision-training gradients on top of an fp32-grade forward, at about 0.7x the
    f16mx step time.  The launches dispatch on their operands' format (_native.conv1d_bf16 / conv1d_wgrad_bf16)."""
    name = "f16mx_hb"

    def enter(self, x, grad=False):
        if not grad:
            hit = self._enter_rows(x, 2, "f16mx")
            if hit is not None:
                return hit
        x = dense(x)
        if grad:
            return N.ncl_to_nlc(x, 1, "f16", N.grad_scale(x))
        return N.ncl_to_nlc(x, 2, "f16mx", None)


class _BF16x3HBEngine(_BF16x3Engine):
    """bf16x3 forward, bf16 ("half") backward -- f16mx_hb's idea on the most accurate split format: the forward is bf16x3's bit
    for bit (6e-6: 2.3x fewer flipped near-ties than f16mx, none in the goldens' 47 834 rows), gradients enter
    their chains as ONE bf16 plane (fp32 range: no loss scale) and every backward product is one bf16 MFMA on the hi planes
    of the saved activations / packed weights."""
    name = "bf16x3_hb"

    def enter(self, x, grad=False):
        if not grad:
            hit = self._enter_rows(x, 2, "bf16x3")
            if hit is not None:
                return hit
        return N.ncl_to_nlc(dense(x), 1 if grad else 2)


class _F16MXHDEngine(_F16MXHBEngine):
    """f16mx_hb with the DECODER's forward in fp16 as well (opt-in; module docstring): encoder, pre-VQ convolution and
    quantiser -- everything the codebook indices depend on -- stay f16mx bit for bit; the decoder's activations are ONE fp16
    plane and its products one fp16 MFMA each (the H image of the same packed weights)."""
    name = "f16mx_hd"


class _F16DecoderEngine(_F16MXHBEngine):
    """The decoder's engine in the f16mx_hd mode: fp16 activations, fp16 gradients."""
    name = "f16dec"
    planes = 1
    fmt = "f16"

    def enter(self, x, grad=False):
        x = dense(x)
        return N.ncl_to_nlc(x, 1, "f16", N.grad_scale(x) if grad else None)


# x3mx_hb has no engine of its own: its parts run bf16x3_hb's (encoder side, and any module used on its own) and f16mx_hb's
# (decoder) engines, so a saved node re-wraps its activations by the name of the engine that made them
_ENGINES = {"f32": _F32Engine, "bf16": _BF16Engine, "bf16x3": _BF16x3Engine, "f16mx": _F16MXEngine, "f16mx_hb": _F16MXHBEngine,
            "f16mx_hd": _F16MXHDEngine, "bf16x3_hb": _BF16x3HBEngine, "x3mx_hb": _BF16x3HBEngine}
_ROLE_ENGINES = {("f16mx_hd", "decoder"): _F16DecoderEngine, ("x3mx_hb", "decoder"): _F16MXHBEngine}
_ENGINES_BY_NAME = dict(_ENGINES, f16dec=_F16DecoderEngine)
assert set(MODES) | set(INTERNAL_MODES) == set(_ENGINES)


def _engine(name=None, role=None):
    """The engine of the current mode (or, in a backward, the one that ran the node's forward); ``role`` lets a mode give
    one part of the model its own arithmetic."""
    if name is not None:
        return _ENGINES_BY_NAME[name]()
    return _ROLE_ENGINES.get((_DTYPE, role), _ENGINES[_DTYPE])()


_ACT_TAP = None     # analysis hook (tests/analysis/gate_flips.py): a list that receives (engine name, saved activations) per node


def _save(ctx, eng, tensors, acts):
    """save_for_backward(*tensors, *activation storages) + remember how to re-wrap the activations."""
    if _ACT_TAP is not None:
        _ACT_TAP.append((eng.name, list(acts)))
    packed = [eng.pack(a) for a in acts]
    ctx.eng_name = eng.name
    ctx.n_plain = len(tensors)
    ctx.act_meta = [m for _, m in packed]
    ctx.save_for_backward(*tensors, *[t for t, _ in packed])


def _load(ctx):
    eng = _engine(ctx.eng_name)
    saved = ctx.saved_tensors
    plain = saved[:ctx.n_plain]
    acts = [eng.unpack(t, m) for t, m in zip(saved[ctx.n_plain:], ctx.act_meta)]
    return eng, plain, acts


# --------------------------------------------------------------------------------------------------
# residual stack (shared W1, W2 used R times; residual_stack.py:40-46)
# --------------------------------------------------------------------------------------------------
def _stack_forward(eng, t1, w1, w2, R, post=None):
    """t1 = relu(h0).  Returns (ts[1..R+1], us[1..R], out) where out = ts[R+1] (+ post)."""
    ts, us = [t1], []
    out = None
    for r in range(R):
        u = eng.conv(ts[-1], w1, relu=True)
        us.append(u)
        if r == R - 1 and post is not None:
            t, out = eng.conv(u, w2, skip1=ts[-1], relu=True, post=post)
        else:
            t = eng.conv(u, w2, skip1=ts[-1], relu=True)
        ts.append(t)
    if out is None:
        out = ts[-1]
    return ts, us, out


def _stack_backward(eng, dh, ts, us, w1, w2, R, outer=None):
    """dh = grad wrt h_R, already masked by (t_{R+1} > 0).  outer = extra grad flowing into t_1 (encoder skip).
    Returns (dh0 masked by t_1>0, dW1, dW2) with the R uses of the shared weights summed in a fixed order."""
    dw1 = dw2 = None
    if w1.requires_grad != w2.requires_grad:
        raise RuntimeError("the two weights of the shared Residual must be frozen or trainable together")
    train = w1.requires_grad
    fused = train and getattr(eng, "wgrad_multi", None) is not None and 1 < R <= 4
    pairs1, pairs2 = [], []
    for r in range(R - 1, -1, -1):
        du = eng.conv(dh, w2, IOK, mask=us[r])                                 # k1 data-grad, * (u_r > 0)
        if fused:                                                              # one launch per shared weight, below
            pairs2.append((dh, us[r]))
            pairs1.append((du, ts[r]))
        elif train:
            dw2, _ = _wgrad(eng, dh, us[r], 1, OIK, w2, dw_prev=dw2)
            dw1, _ = _wgrad(eng, du, ts[r], 3, OIK, w1, dw_prev=dw1)
        dh = eng.conv(du, w1, IOK, skip1=dh, skip2=outer if r == 0 else None, mask=ts[r])
    if fused:
        s1, s2 = _sink(w1), _sink(w2)
        d2 = {"defer": _DEFERRED} if (_DEFERRED is not None and s2 is not None and getattr(eng, "can_defer", False)) else {}
        d1 = {"defer": _DEFERRED} if (_DEFERRED is not None and s1 is not None and getattr(eng, "can_defer", False)) else {}
        dw2 = eng.wgrad_multi(pairs2, 1, OIK, dw_out=s2, accumulate=s2 is not None, **d2)
        dw1 = eng.wgrad_multi(pairs1, 3, OIK, dw_out=s1, accumulate=s1 is not None, **d1)
        dw1, dw2 = (None if s1 is not None else dw1), (None if s2 is not None else dw2)
    return dh, dw1, dw2


def _check_layers(R):
    if R < 1:
        # with no residual layer the reference's in-place ReLU never runs, so its encoder returns relu(h0) + h0
        # (convolutional_encoder.py:42) -- a different function from the R >= 1 chain built here
        raise NotImplementedError("num_residual_layers=0 is not supported on the HIP path (no script uses it)")


def _encoder_forward(eng, x, wc, bc, w1, w2, R):
    _check_layers(R)
    xi = eng.enter(x)
    t1 = eng.conv(xi, wc, bias=bc, relu=True)
    ts, us, out = _stack_forward(eng, t1, w1, w2, R, post=t1)
    return xi, ts, us, out


def _encoder_backward(eng, d_out, xi, ts, us, wc, bc, w1, w2, R, need_dx):
    dh = eng.relu_mask(d_out, ts[R])                                             # * (h_R > 0)
    dh0, dw1, dw2 = _stack_backward(eng, dh, ts, us, w1, w2, R, outer=d_out)
    dwc, dbc = _wgrad(eng, dh0, xi, 3, OIK, wc, bc)
    dx = eng.conv(dh0, wc, IOK, out_f32=True) if need_dx else None
    return dx, dwc, dbc, dw1, dw2


class EncoderFn(torch.autograd.Function):
    """ConvolutionalEncoder.forward (convolutional_encoder.py:39-44): relu(h_R) + relu(h_0)."""

    @staticmethod
    def forward(ctx, x, wc, bc, w1, w2, R):
        eng = _engine()
        xi, ts, us, out = _encoder_forward(eng, x, wc, bc, w1, w2, R)
        ctx.R = R
        _save(ctx, eng, (wc, bc, w1, w2), [xi, *ts, *us])
        return eng.leave(out)

    @staticmethod
    def backward(ctx, d_out):
        R = ctx.R
        eng, (wc, bc, w1, w2), acts = _load(ctx)
        xi, ts, us = acts[0], acts[1:R + 2], acts[R + 2:]
        dx, dwc, dbc, dw1, dw2 = _encoder_backward(eng, eng.enter(d_out, grad=True), xi, ts, us, wc, bc, w1, w2, R,
                                                   ctx.needs_input_grad[0])
        return dx, dwc, dbc, dw1, dw2, None


class LatentFn(torch.autograd.Function):
    """Encoder + ``_pre_vq_conv`` (convolutional_vq_vae.py:94-95) as one node: z = W_pre *3 encoder(x) + b_pre,
    returned as dense (B,D,L) fp32 because the quantiser slices that buffer in memory order."""

    @staticmethod
    def forward(ctx, x, wc, bc, w1, w2, wp, bp, R):
        eng = _engine()
        xi, ts, us, out = _encoder_forward(eng, x, wc, bc, w1, w2, R)
        z = eng.conv(out, wp, bias=bp, out_f32=True)
        ctx.R = R
        _save(ctx, eng, (wc, bc, w1, w2, wp, bp), [xi, out, *ts, *us])
        return z

    @staticmethod
    def backward(ctx, dz):
        R = ctx.R
        eng, (wc, bc, w1, w2, wp, bp), acts = _load(ctx)
        xi, out, ts, us = acts[0], acts[1], acts[2:R + 3], acts[R + 3:]
        dzi = eng.enter(dz, grad=True)
        d_out = eng.conv(dzi, wp, IOK)
        dwp, dbp = _wgrad(eng, dzi, out, 3, OIK, wp, bp)
        dx, dwc, dbc, dw1, dw2 = _encoder_backward(eng, d_out, xi, ts, us, wc, bc, w1, w2, R, ctx.needs_input_grad[0])
        return dx, dwc, dbc, dw1, dw2, dwp, dbp, None


class StackFn(torch.autograd.Function):
    """ResidualStack.forward on its own (residual_stack.py:43-46): relu(h_R) from h_0."""

    @staticmethod
    def forward(ctx, h0, w1, w2, R):
        eng = _engine()
        _check_layers(R)
        hi = eng.enter(h0)
        t1 = eng.relu_mask(hi, hi)
        ts, us, out = _stack_forward(eng, t1, w1, w2, R)
        ctx.R = R
        _save(ctx, eng, (w1, w2), [*ts, *us])
        return eng.leave(out)

    @staticmethod
    def backward(ctx, d_out):
        R = ctx.R
        eng, (w1, w2), acts = _load(ctx)
        ts, us = acts[:R + 1], acts[R + 1:]
        dh = eng.relu_mask(eng.enter(d_out, grad=True), ts[R])
        dh0, dw1, dw2 = _stack_backward(eng, dh, ts, us, w1, w2, R)
        return eng.leave(dh0), dw1, dw2, None


class ResidualLayerFn(torch.autograd.Function):
    """One Residual on its own (residual.py:65-66): relu(x) + W2 *1 relu(W1 *3 relu(x)), no trailing ReLU."""

    @staticmethod
    def forward(ctx, x, w1, w2):
        eng = _engine()
        xi = eng.enter(x)
        t = eng.relu_mask(xi, xi)
        u = eng.conv(t, w1, relu=True)
        _save(ctx, eng, (w1, w2), [t, u])
        return eng.leave(eng.conv(u, w2, skip1=t))

    @staticmethod
    def backward(ctx, dy):
        eng, (w1, w2), (t, u) = _load(ctx)
        dyi = eng.enter(dy, grad=True)
        du = eng.conv(dyi, w2, IOK, mask=u)
        dw2, _ = _wgrad(eng, dyi, u, 1, OIK, w2)
        dw1, _ = _wgrad(eng, du, t, 3, OIK, w1)
        dx = eng.conv(du, w1, IOK, skip1=dyi, mask=t)
        return eng.leave(dx), dw1, dw2


class ConvFn(torch.autograd.Function):
    """Plain Conv1d / ConvTranspose1d (k in {1,3}, stride 1, same padding) with bias, fp32 in / fp32 out."""

    @staticmethod
    def forward(ctx, x, w, b, layout):
        eng = _engine()
        xi = eng.enter(x)
        ctx.layout = layout
        ctx.has_bias = b is not None
        _save(ctx, eng, (w,) + ((b,) if b is not None else ()), [xi])
        return eng.conv(xi, w, layout, bias=b, out_f32=True)

    @staticmethod
    def backward(ctx, dy):
        eng, plain, (xi,) = _load(ctx)
        w, b = plain[0], (plain[1] if ctx.has_bias else None)
        dyi = eng.enter(dy, grad=True)
        kw = w.shape[2]
        dw, db = _wgrad(eng, dyi, xi, kw, ctx.layout, w, b if ctx.needs_input_grad[2] else None)
        dx = eng.conv(dyi, w, IOK if ctx.layout == OIK else OIK, out_f32=True) if ctx.needs_input_grad[0] else None
        return dx, dw, db, None


class JitterFn(torch.autograd.Function):
    """Jitter (modules/jitter.py:42-70) with a host-drawn source-index vector (always fp32 (B,C,L))."""

    @staticmethod
    def forward(ctx, q, src):
        ctx.save_for_backward(src)
        return N.jitter_gather(dense(q), src)

    @staticmethod
    def backward(ctx, dy):
        (src,) = ctx.saved_tensors
        return N.jitter_gather(dy.contiguous(), src, backward=True), None


class DecoderFn(torch.autograd.Function):
    """DeconvolutionalDecoder.forward (deconvolutional_decoder.py:62-79): fp32 (B,D,L) in, fp32 (B,out,L) out."""

    @staticmethod
    def forward(ctx, q, src, wd, bd, w1, w2, wt1, bt1, wt2, bt2, wt3, bt3, R):
        eng = _engine(role="decoder")
        _check_layers(R)
        q = dense(q)
        qj = eng.enter(N.jitter_gather(q, src) if src is not None else q)
        t1 = eng.conv(qj, wd, bias=bd, relu=True)
        ts, us, top = _stack_forward(eng, t1, w1, w2, R)
        a1 = eng.conv(top, wt1, IOK, bias=bt1, relu=True)
        a2 = eng.conv(a1, wt2, IOK, bias=bt2, relu=True)
        y = eng.conv(a2, wt3, IOK, bias=bt3, out_f32=True)
        ctx.R = R
        ctx.src = src
        _save(ctx, eng, (wd, w1, w2, wt1, wt2, wt3, bd, bt1, bt2, bt3), [qj, a1, a2, *ts, *us])
        return y

    @staticmethod
    def backward(ctx, dy):
        R = ctx.R
        eng, (wd, w1, w2, wt1, wt2, wt3, bd, bt1, bt2, bt3), acts = _load(ctx)
        qj, a1, a2, ts, us = acts[0], acts[1], acts[2], acts[3:R + 4], acts[R + 4:]
        dyi = eng.enter(dy, grad=True)
        da2 = eng.conv(dyi, wt3, OIK, mask=a2)                                   # convT data-grad, * (a2 > 0)
        dwt3, dbt3 = _wgrad(eng, dyi, a2, 3, IOK, wt3, bt3)
        da1 = eng.conv(da2, wt2, OIK, mask=a1)
        dwt2, dbt2 = _wgrad(eng, da2, a1, 3, IOK, wt2, bt2)
        dh = eng.conv(da1, wt1, OIK, mask=ts[R])                                 # * (h_R > 0)
        dwt1, dbt1 = _wgrad(eng, da1, ts[R], 3, IOK, wt1, bt1)
        dh0, dw1, dw2 = _stack_backward(eng, dh, ts, us, w1, w2, R)
        dwd, dbd = _wgrad(eng, dh0, qj, 3, OIK, wd, bd)
        dq = None
        if ctx.needs_input_grad[0]:
            dq = eng.conv(dh0, wd, IOK, out_f32=True)
            if ctx.src is not None:
                dq = N.jitter_gather(dq, ctx.src, backward=True)
        return dq, None, dwd, dbd, dw1, dw2, dwt1, dbt1, dwt2, dbt2, dwt3, dbt3, None


class VQFn(torch.autograd.Function):
    """VectorQuantizer.forward (vector_quantizer.py:29-58) -> (loss, q_st, perplexity, idx).  Always fp32."""

    @staticmethod
    def forward(ctx, z, codebook, beta, train_vq):
        z = dense(z)
        d = codebook.shape[1]
        flat = z.view(-1, d)                       # memory-order rows, no permute (:32)
        idx = N.vq_argmin(flat, codebook)
        q_st, out = N.vq_gather_loss(flat, codebook, idx, beta)
        ctx.beta, ctx.train_vq, ctx.zshape = beta, train_vq, z.shape
        ctx.save_for_backward(flat, codebook, idx)
        ctx.mark_non_differentiable(idx)
        return out[0], q_st.view(z.shape), out[1], idx

    @staticmethod
    def backward(ctx, dloss, dq, dperp, _):
        flat, codebook, idx = ctx.saved_tensors
        d = codebook.shape[1]
        g = dq.contiguous().view(-1, d) if dq is not None else None
        if dloss is None:
            dloss = torch.zeros((), device=flat.device)
        want_de = ctx.train_vq and ctx.needs_input_grad[1]
        sink = _sink(codebook) if want_de else None
        dx, dE = N.vq_backward(g, dloss.reshape(1).contiguous(), flat, codebook, idx, ctx.beta,
                               want_dx=ctx.needs_input_grad[0], want_dE=want_de, dE_out=sink)
        if dx is not None:
            dx = dx.view(ctx.zshape)
        return dx, (None if sink is not None else dE), None, None


class MeanPoolFn(torch.autograd.Function):
    """torch.mean(z, dim=2, keepdim=True): the optional pooling of the latent (convolutional_vq_vae.py:96-97)."""

    @staticmethod
    def forward(ctx, z):
        z = dense(z)
        ctx.L = z.shape[2]
        return N.row_mean(z)

    @staticmethod
    def backward(ctx, dy):
        return N.row_mean(dy.contiguous(), backward_of=ctx.L)


class MSEFn(torch.autograd.Function):
    """F.mse_loss(a, b) (train_speech.py:74); b is a constant target."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = dense(a), dense(b)
        ctx.save_for_backward(a, b)
        return N.mse(a, b)[0]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return N.mse_backward(a, b, g.reshape(1).contiguous()), None


def jitter_source_index(length, probability):
    """Host side of Jitter: the index each output column copies from, consuming ``np.random`` exactly as
    modules/jitter.py:50-68 does -- one draw per column (``choice([1, 0], p=[p, 1-p])``; the column is replaced when
    the draw is 0, i.e. with probability 1 - p: the inversion at :55) and one more per replaced interior column
    (``choice([-1, 1], p=[.5, .5])``).

    ``RandomState.choice(a, p=...)`` with no size draws ONE ``random_sample()`` u and returns
    ``a[searchsorted(cumsum(p), u, side='right')]``, so the stream is reproduced from raw uniforms: replaced iff
    u >= p, neighbour = -1 iff u' < 0.5.  The reference's 500-iteration Python loop of ``choice`` calls costs ~9 ms
    per step on the host -- as long as the whole GPU step -- so the uniforms are drawn in two vectorised calls:
    a look-ahead block to learn how many the sequential rule consumes, then the state is rewound and exactly that
    many are drawn (identical values, identical final generator state)."""
    rng = np.random.mtrand._rand          # the global RandomState that np.random.choice uses
    p = float(probability)
    if length < 2:                        # no neighbour to copy from: jitter.py:57-60 indexes column 1 / -1 of a
        # one-column tensor (e.g. encoder_average_pooling=True in training mode) and raises
        raise IndexError("index 1 is out of bounds for dimension 2 with size %d (Jitter needs at least 2 columns)" % length)
    cdf0 = np.cumsum(np.array([p, 1.0 - p]))
    cdf0 /= cdf0[-1]
    thr = float(cdf0[0])                  # same normalisation as choice()
    state = rng.get_state()
    u = rng.random_sample(2 * length)     # upper bound on the draws
    src = np.arange(length, dtype=np.int32)
    ptr = 0
    last = length - 1
    for i in range(length):
        replaced = u[ptr] >= thr
        ptr += 1
        if replaced:
            if i == 0:
                src[i] = 1
            elif i == last:
                src[i] = i - 1
            else:
                src[i] = i + (-1 if u[ptr] < 0.5 else 1)
                ptr += 1
    rng.set_state(state)
    rng.random_sample(ptr)                # leave the generator exactly where the reference's loop would
    return src
