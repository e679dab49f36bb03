"""ctypes binding of libalvq.so (include/alvq.h) -- the only door from Python to the HIP kernels.

PyTorch is used for device memory and streams only: every function here takes torch CUDA(=HIP) tensors,
checks shape/dtype/contiguity on the host, and passes raw ``data_ptr()``s plus the current HIP stream to
the C ABI.  There is no CPU fallback: if the library is missing or a tensor is not on the GPU the call
raises.
"""
from __future__ import annotations

import ctypes
import os

import torch

_LIB = None
_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libalvq.so"))

W_OIK = 0
W_IOK = 1
VQ_PARTIALS = 1024
EW_PARTIALS = 1024

_c_void_p = ctypes.c_void_p
_i32 = ctypes.c_int
_i64 = ctypes.c_int64
_f32 = ctypes.c_float

_SIGNATURES = {
    "alvq_version": (ctypes.c_char_p, []),
    "alvq_last_error": (ctypes.c_char_p, []),
    "alvq_set_option": (_i32, [ctypes.c_char_p, _i64]),
    "alvq_get_option": (_i64, [ctypes.c_char_p]),
    "alvq_conv1d_f32": (_i32, [_c_void_p] * 9 + [_i32] * 7 + [_c_void_p]),
    "alvq_conv1d_wgrad_workspace_bytes": (_i64, [_i32] * 5),
    "alvq_conv1d_wgrad_f32": (_i32, [_c_void_p] * 5 + [_i32] * 7 + [_c_void_p]),
    "alvq_vq_argmin_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "alvq_vq_argmin_f32": (_i32, [_c_void_p] * 5 + [_i64, _i32, _i32, _c_void_p]),
    "alvq_vq_gather_loss_f32": (_i32, [_c_void_p] * 6 + [_i64, _i32, _i32, _c_void_p]),
    "alvq_vq_finalize_f32": (_i32, [_c_void_p] * 3 + [_i64, _i32, _i32, _f32, _c_void_p]),
    "alvq_vq_backward_workspace_bytes": (_i64, [_i32, _i32]),
    "alvq_vq_backward_f32": (_i32, [_c_void_p] * 8 + [_i64, _i32, _i32, _f32, _c_void_p]),
    "alvq_onehot_f32": (_i32, [_c_void_p, _c_void_p, _i64, _i32, _c_void_p]),
    "alvq_jitter_gather_f32": (_i32, [_c_void_p] * 3 + [_i64, _i32, _i32, _c_void_p]),
    "alvq_standardise_f32": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p]),
    "alvq_mse_f32": (_i32, [_c_void_p] * 4 + [_i64, _c_void_p]),
    "alvq_mse_backward_f32": (_i32, [_c_void_p] * 4 + [_i64, _c_void_p]),
    "alvq_fill_f32": (_i32, [_c_void_p, _f32, _i64, _c_void_p]),
    "alvq_add_f32": (_i32, [_c_void_p] * 3 + [_i64, _c_void_p]),
    "alvq_relu_mask_f32": (_i32, [_c_void_p] * 3 + [_i64, _c_void_p]),
    "alvq_row_mean_f32": (_i32, [_c_void_p] * 2 + [_i64, _i32, _c_void_p]),
    "alvq_row_mean_backward_f32": (_i32, [_c_void_p] * 2 + [_i64, _i32, _c_void_p]),
    "alvq_transpose_f32": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _c_void_p]),
    "alvq_adam_f32": (_i32, [_c_void_p] * 4 + [_i64, _i32, _f32, _f32, _f32, _f32, _f32, _c_void_p]),
    "alvq_adam_dev_f32": (_i32, [_c_void_p] * 4 + [_i64, _c_void_p, _f32, _f32, _f32, _c_void_p, _c_void_p]),
    "alvq_adam_advance_f32": (_i32, [_c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _c_void_p, _c_void_p]),
    "alvq_range_flag_to_slot": (_i32, [_c_void_p, _c_void_p]),
    "alvq_stft_power_f32": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p]),
    "alvq_stft_power_f64": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p]),
    "alvq_stft_complex_f32": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p]),
    "alvq_stft_complex_f64": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p]),
    "alvq_fir_same_f64": (_i32, [_c_void_p, _c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p]),
    "alvq_spec_rir_wiener_f64": (_i32, [_c_void_p] * 7 + [_i32, _i32, _i32, _c_void_p]),
    "alvq_rows_to_nlc": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _i32, _i32, _c_void_p]),
    "alvq_rows_to_nlc_max_std_rows": (_i32, []),
    "alvq_nlc_rows": (_i64, [_i32, _i32]),
    "alvq_nlc_channels": (_i32, [_i32]),
    "alvq_nlc_guard_rows": (_i32, []),
    "alvq_packed_weight_elems": (_i64, [_i32, _i32, _i32]),
    "alvq_pack_weight_bf16": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p]),
    "alvq_pack_weights_bf16_batch": (_i32, [_c_void_p, _i32, _i32, _c_void_p]),
    "alvq_adam_pack_batch": (_i32, [_c_void_p, _i32, _i32, _c_void_p, _f32, _f32, _f32, _c_void_p, _c_void_p]),
    "alvq_adam_segments_f32": (_i32, [_c_void_p] * 4 + [_c_void_p, _c_void_p, _i32, _c_void_p, _f32, _f32, _f32, _c_void_p, _c_void_p]),
    "alvq_ncl_to_nlc_bf16": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _c_void_p]),
    "alvq_nlc_to_ncl_f32": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _c_void_p]),
    "alvq_relu_mask_bf16": (_i32, [_c_void_p] * 3 + [_i64, _c_void_p]),
    "alvq_conv1d_bf16": (_i32, [_c_void_p] * 10 + [_i32] * 6 + [_c_void_p] * 3),
    "alvq_conv1d_wgrad_bf16_workspace_bytes": (_i64, [_i32] * 5),
    "alvq_conv1d_wgrad_bf16": (_i32, [_c_void_p] * 5 + [_i32] * 7 + [_c_void_p]),
    "alvq_conv1d_wgrad_bf16_multi": (_i32, [_c_void_p, _c_void_p, _i32, _c_void_p, _c_void_p] + [_i32] * 7 + [_c_void_p]),
    "alvq_nlc_plane_bytes": (_i64, [_i32, _i32, _i32]),
    "alvq_pack_weight_bf16x3": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p]),
    "alvq_ncl_to_nlc_bf16x3": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _c_void_p]),
    "alvq_nlc_to_ncl_bf16x3": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _c_void_p]),
    "alvq_relu_mask_bf16x3": (_i32, [_c_void_p] * 3 + [_i32, _i32, _i32, _c_void_p]),
    "alvq_conv1d_bf16x3": (_i32, [_c_void_p] * 10 + [_i32] * 6 + [_c_void_p]),
    "alvq_conv1d_wgrad_bf16x3_workspace_bytes": (_i64, [_i32] * 5),
    "alvq_conv1d_wgrad_bf16x3": (_i32, [_c_void_p] * 5 + [_i32] * 7 + [_c_void_p]),
    "alvq_grad_scale_f32": (_i32, [_c_void_p, _i64, _c_void_p, _c_void_p]),
    "alvq_f16mx_range_flag": (_i32, [_c_void_p, _i32, _c_void_p]),
    "alvq_ncl_to_nlc_f16mx": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _c_void_p, _c_void_p]),
    "alvq_nlc_to_ncl_f16mx": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _c_void_p, _c_void_p]),
    "alvq_relu_mask_f16mx": (_i32, [_c_void_p] * 3 + [_i32, _i32, _i32, _c_void_p]),
    "alvq_conv1d_f16mx": (_i32, [_c_void_p] * 10 + [_i32] * 6 + [_c_void_p] * 4),
    "alvq_conv1d_wgrad_f16mx_workspace_bytes": (_i64, [_i32] * 5),
    "alvq_conv1d_wgrad_f16mx": (_i32, [_c_void_p] * 5 + [_i32] * 7 + [_c_void_p, _c_void_p]),
    "alvq_conv1d_wgrad_bf16x3_multi": (_i32, [_c_void_p, _c_void_p, _i32, _c_void_p, _c_void_p] + [_i32] * 7 + [_c_void_p]),
    "alvq_conv1d_wgrad_f16mx_multi": (_i32, [_c_void_p, _c_void_p, _i32, _c_void_p, _c_void_p] + [_i32] * 7 + [_c_void_p, _c_void_p]),
    "alvq_conv1d_wgrad_bf16_splits": (_i32, [_i32] * 7),
    "alvq_conv1d_wgrad_bf16x3_splits": (_i32, [_i32] * 6),
    "alvq_conv1d_wgrad_f16mx_splits": (_i32, [_i32] * 6),
    "alvq_ncl_to_nlc_f16": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _c_void_p, _c_void_p]),
    "alvq_nlc_to_ncl_f16": (_i32, [_c_void_p, _c_void_p, _i32, _i32, _i32, _c_void_p, _c_void_p]),
    "alvq_conv1d_f16": (_i32, [_c_void_p] * 10 + [_i32] * 6 + [_c_void_p] * 4),
    "alvq_conv1d_wgrad_f16": (_i32, [_c_void_p] * 5 + [_i32] * 7 + [_c_void_p, _c_void_p]),
    "alvq_conv1d_wgrad_f16_multi": (_i32, [_c_void_p, _c_void_p, _i32, _c_void_p, _c_void_p] + [_i32] * 7 + [_c_void_p, _c_void_p]),
    "alvq_conv1d_wgrad_bf16_bias_offset": (_i64, [_i32] * 5),
    "alvq_wgrad_reduce_batch": (_i32, [_c_void_p, _i32, _c_void_p]),
    "alvq_onehot_to_index_f32": (_i32, [_c_void_p] * 3 + [_i64, _i32, _c_void_p]),
    "alvq_indices_to_i32": (_i32, [_c_void_p] * 3 + [_i64, _i32, _c_void_p]),
    "alvq_embedding_bag_fwd_f32": (_i32, [_c_void_p] * 4 + [_i32] * 4 + [_c_void_p] * 2),
    "alvq_embedding_bag_bwd_f32": (_i32, [_c_void_p] * 4 + [_i32] * 5 + [_c_void_p] * 2),
}

EXPORTS = tuple(_SIGNATURES)


def lib_path():
    return os.environ.get("ALVQ_LIB", _DEFAULT_LIB)


def lib():
    """Open libalvq.so once per process (module-global, so nn.Modules stay picklable)."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                "libalvq.so not found at %s -- build it with `python acoustic_locating_vq-vae_amd/build.py` "
                "(there is no CPU fallback for the HIP path)" % path)
        handle = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            if os.environ.get("ALVQ_LIB_ALLOW_MISSING") == "1" and not hasattr(handle, name):
                continue                             # tools/ab_bits.py against an OLDER build of the library
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def version():
    return lib().alvq_version().decode()


def set_option(name, value):
    """Dispatch option of the library (include/alvq.h: alvq_set_option); returns the previous value."""
    prev = get_option(name)
    _check(lib().alvq_set_option(name.encode(), int(value)), "alvq_set_option")
    return prev


def get_option(name):
    v = lib().alvq_get_option(name.encode())
    if v == -(1 << 63):
        raise KeyError(name)
    return v


def _check(rc, name):
    if rc != 0:
        raise RuntimeError("%s failed (rc=%d): %s" % (name, rc, lib().alvq_last_error().decode()))


def _ptr(t, dtype=torch.float32, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (got %s); the HIP path has no CPU fallback" % (name, t.device))
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional HIP-event timing of individual launches (bench.py's live roofline measurement).

    Events are recorded on the stream the kernels are launched on (torch's current stream).  Usage:
    ``with KernelTimer() as kt: ...``; then ``kt.summary()`` -> {family: (launches, seconds, flops)}.
    """
    active = None

    def __init__(self):
        self.records = []

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for family, flops, e0, e1 in self.records:
            n, t, f = out.get(family, (0, 0.0, 0.0))
            out[family] = (n + 1, t + e0.elapsed_time(e1) * 1e-3, f + flops)
        return out


class _timed:
    __slots__ = ("family", "flops", "e0")

    def __init__(self, family, flops):
        self.family, self.flops = family, flops

    def __enter__(self):
        if KernelTimer.active is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        kt = KernelTimer.active
        if kt is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            kt.records.append((self.family, self.flops, self.e0, e1))


# ----------------------------------------------------------------------------------------------- conv
def conv1d(x, w, bias=None, skip1=None, skip2=None, mask=None, post=None, relu=False, w_layout=W_OIK,
           want_y2=False):
    """Fused conv (see alvq_conv1d_f32).  x (B,C,L); w (M,C,KW) for W_OIK or (C,M,KW) for W_IOK.
    Returns y or (y, y2)."""
    B, C, L = x.shape
    if w_layout == W_OIK:
        M, Cw, KW = w.shape
    else:
        Cw, M, KW = w.shape
    if Cw != C:
        raise RuntimeError("conv1d: weight expects %d input channels, x has %d" % (Cw, C))
    y = torch.empty((B, M, L), device=x.device, dtype=torch.float32)
    y2 = torch.empty_like(y) if post is not None else None
    for t, nm in ((skip1, "skip1"), (skip2, "skip2"), (mask, "mask"), (post, "post")):
        if t is not None and tuple(t.shape) != (B, M, L):
            raise RuntimeError("conv1d: %s has shape %s, expected %s" % (nm, tuple(t.shape), (B, M, L)))
    if bias is not None and bias.numel() != M:
        raise RuntimeError("conv1d: bias has %d elements, expected %d" % (bias.numel(), M))
    with _timed("conv1d_f32_kernel", 2.0 * B * L * M * C * KW):
        rc = lib().alvq_conv1d_f32(_ptr(x, name="x"), _ptr(w, name="w"), _ptr(bias, name="bias"), _ptr(skip1, name="skip1"),
                                   _ptr(skip2, name="skip2"), _ptr(mask, name="mask"), _ptr(post, name="post"),
                                   _ptr(y), _ptr(y2), B, C, M, L, KW, w_layout, int(bool(relu)), _stream())
    _check(rc, "alvq_conv1d_f32")
    return (y, y2) if post is not None else y


_WS = {}
_WS_RETIRED = []


def _workspace(nbytes, device):
    """Grow-only scratch per (device, stream), caller-owned from the library's point of view.

    A buffer that has been handed out is never released: a captured hipGraph records the raw pointer of the
    scratch its launches used (weight-gradient split partials, VQ norms, codebook-gradient partials), so when a
    later, larger request outgrows the buffer the old one is parked in ``_WS_RETIRED`` instead of going back to
    the caching allocator, where a replay would scribble over whoever owns the memory next.  Keyed by stream as
    well: launches on different streams are not ordered against each other and must not share scratch."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _WS_RETIRED.append(buf)
        buf = torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)
        _WS[key] = buf
    return buf


# ---- deferred split reductions (alvq_wgrad_reduce_batch): every deferred launch needs scratch of its own until the batch
# launch at the end of the backward pass has summed it.  The arena is a bump allocator over grow-only buffers per
# (device, stream) -- reset at the start of a step, never released (a captured graph keeps the raw pointers).
WGRAD_DEFER = 2
_ARENA = {}


class ReduceDesc(ctypes.Structure):
    """struct alvq_reduce_desc (include/alvq.h)"""
    _fields_ = [("partial", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("splits", ctypes.c_int32),
                ("KW", ctypes.c_int32), ("M", ctypes.c_int32), ("C", ctypes.c_int32), ("w_layout", ctypes.c_int32),
                ("accumulate", ctypes.c_int32), ("stride", ctypes.c_int64)]


def arena_reset(device):
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    st = _ARENA.get(key)
    if st is not None:
        st["chunk"], st["off"] = 0, 0


def arena_alloc(nbytes, device):
    """256-byte aligned scratch that stays valid until the next ``arena_reset`` of this (device, stream)."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    st = _ARENA.setdefault(key, {"chunks": [], "chunk": 0, "off": 0})
    nbytes = (nbytes + 255) // 256 * 256
    while True:
        if st["chunk"] >= len(st["chunks"]):
            st["chunks"].append(torch.empty(max(nbytes, 256 << 20), device=device, dtype=torch.uint8))
            st["off"] = 0
        buf = st["chunks"][st["chunk"]]
        if st["off"] + nbytes <= buf.numel():
            ptr = buf.data_ptr() + st["off"]
            st["off"] += nbytes
            return ptr
        st["chunk"] += 1
        st["off"] = 0


class DeferredReductions(list):
    """The descriptors of a backward pass's deferred reductions, plus the tensors their raw pointers refer to that nobody
    else keeps alive until the batch launch -- the loss-scale state of a gradient chain is freed with the chain's last
    tensor, and the next chain's state would be allocated (and written) at the same address before the reduction ran."""

    def __init__(self):
        super().__init__()
        self.keep = []


def wgrad_reduce_batch(descs):
    """descs: list of ReduceDesc -- one launch sums them all (dst (+)= scale * sum_s partial[s], split order)."""
    if not descs:
        return
    arr = (ReduceDesc * len(descs))(*descs)
    _check(lib().alvq_wgrad_reduce_batch(ctypes.addressof(arr), len(descs), _stream()), "alvq_wgrad_reduce_batch")
    if isinstance(descs, DeferredReductions):
        descs.keep.clear()       # the launch is queued: stream order protects the memory from here on


def _defer_descs(defer, ws_ptr, dw, dbias, scale, nseg, B, C, M, L, KW, w_layout):
    """``scale``: the 4-float loss-scale state of the gradient chain (its 1/S is read by the reduction), or None."""
    L_ = lib()
    scale_ptr = _sptr(scale, 1)
    # the batch launch sums its descriptors in concurrent workgroups: two descriptors accumulating into ONE destination
    # (a shared residual weight used by more than four layers, i.e. outside the fused multi-segment launch) would race on
    # ``dst += sum``.  Sum what is pending first -- stream order then serialises the two accumulations, as the
    # per-launch reductions did (round-3 advisor finding).
    dsts = {dw.data_ptr()} | ({dbias.data_ptr()} if dbias is not None else set())
    if any(d.dst in dsts for d in defer):
        wgrad_reduce_batch(defer)
        del defer[:]
    if scale is not None and hasattr(defer, "keep"):
        defer.keep.append(scale)
    splits = L_.alvq_conv1d_wgrad_bf16_splits(B, C, M, L, KW, nseg, int(dbias is not None))
    defer.append(ReduceDesc(ws_ptr, dw.data_ptr(), scale_ptr, splits, KW, M, C, w_layout, 1, KW * M * C))
    if dbias is not None:
        off = L_.alvq_conv1d_wgrad_bf16_bias_offset(B, C, M, L, KW)
        defer.append(ReduceDesc(ws_ptr + off, dbias.data_ptr(), scale_ptr, splits, 1, 1, M, W_OIK, 1, (M + 63) // 64 * 64))


def conv1d_wgrad(dy, x, KW, w_layout=W_OIK, want_bias=False, dw_out=None, dbias_out=None, accumulate=False):
    """dw (and dbias) of the conv whose input was x (B,C,L) and output-grad is dy (B,M,L)."""
    B, C, L = x.shape
    Bd, M, Ld = dy.shape
    if (Bd, Ld) != (B, L):
        raise RuntimeError("conv1d_wgrad: dy %s does not match x %s" % (tuple(dy.shape), tuple(x.shape)))
    shape = (M, C, KW) if w_layout == W_OIK else (C, M, KW)
    if dw_out is None:
        dw_out = torch.empty(shape, device=x.device, dtype=torch.float32)
        accumulate = False
    elif tuple(dw_out.shape) != shape:
        raise RuntimeError("conv1d_wgrad: dw_out has shape %s, expected %s" % (tuple(dw_out.shape), shape))
    if want_bias and dbias_out is None:
        dbias_out = torch.empty((M,), device=x.device, dtype=torch.float32)
    nbytes = lib().alvq_conv1d_wgrad_workspace_bytes(B, C, M, L, KW)
    if nbytes < 0:
        raise RuntimeError("conv1d_wgrad: unsupported shape")
    ws = _workspace(nbytes, x.device)
    with _timed("conv1d_wgrad_f32_kernel", 2.0 * B * L * M * C * KW):
        rc = lib().alvq_conv1d_wgrad_f32(_ptr(dy, name="dy"), _ptr(x, name="x"), _ptr(dw_out, name="dw"),
                                         _ptr(dbias_out, name="dbias") if want_bias else None, ws.data_ptr(),
                                         B, C, M, L, KW, w_layout, int(bool(accumulate)), _stream())
    _check(rc, "alvq_conv1d_wgrad_f32")
    return (dw_out, dbias_out) if want_bias else dw_out


# ----------------------------------------------------------------------------------------------- VQ
def vq_argmin(flat, codebook, want_dist=False):
    N, D = flat.shape
    K, Dc = codebook.shape
    if D != Dc:
        raise RuntimeError("vq_argmin: row width %d != codebook dim %d" % (D, Dc))
    idx = torch.empty((N,), device=flat.device, dtype=torch.int64)
    dist = torch.empty((N,), device=flat.device, dtype=torch.float32) if want_dist else None
    ws = _workspace(lib().alvq_vq_argmin_workspace_bytes(N, K, D), flat.device)
    with _timed("vq_argmin_f32_kernel", 2.0 * N * K * D):
        rc = lib().alvq_vq_argmin_f32(_ptr(flat, name="x"), _ptr(codebook, name="codebook"), idx.data_ptr(),
                                      _ptr(dist), ws.data_ptr(), N, K, D, _stream())
    _check(rc, "alvq_vq_argmin_f32")
    return (idx, dist) if want_dist else idx


def vq_gather_loss(flat, codebook, idx, beta):
    """-> (q_st (N,D), out[2] = (loss, perplexity))."""
    N, D = flat.shape
    K = codebook.shape[0]
    q_st = torch.empty_like(flat)
    partials = torch.empty((VQ_PARTIALS,), device=flat.device, dtype=torch.float32)
    hist = torch.empty((K,), device=flat.device, dtype=torch.float32)
    fill_(hist, 0.0)                                 # the library's own fill (bit pattern 0 == int 0): no ATen op on the step path
    hist = hist.view(torch.int32)
    out = torch.empty((2,), device=flat.device, dtype=torch.float32)
    L = lib()
    _check(L.alvq_vq_gather_loss_f32(_ptr(flat, name="x"), _ptr(codebook, name="codebook"),
                                     _ptr(idx, torch.int64, "idx"), _ptr(q_st), _ptr(partials),
                                     _ptr(hist, torch.int32), N, K, D, _stream()), "alvq_vq_gather_loss_f32")
    _check(L.alvq_vq_finalize_f32(_ptr(partials), _ptr(hist, torch.int32), _ptr(out), N, K, D, float(beta), _stream()),
           "alvq_vq_finalize_f32")
    return q_st, out


def vq_backward(g, grad_loss, flat, codebook, idx, beta, want_dx=True, want_dE=True, dE_out=None):
    """dE_out: accumulate the codebook gradient into this (K,D) tensor instead of a fresh zeroed one."""
    N, D = flat.shape
    K = codebook.shape[0]
    dx = torch.empty_like(flat) if want_dx else None
    dE = (dE_out if dE_out is not None else torch.zeros_like(codebook)) if want_dE else None
    ws = _workspace(lib().alvq_vq_backward_workspace_bytes(K, D), flat.device).data_ptr() if want_dE else None
    _check(lib().alvq_vq_backward_f32(_ptr(g, name="g"), _ptr(grad_loss, name="grad_loss"), _ptr(flat, name="x"),
                                      _ptr(codebook, name="codebook"), _ptr(idx, torch.int64, "idx"), _ptr(dx), _ptr(dE),
                                      ws, N, K, D, float(beta), _stream()), "alvq_vq_backward_f32")
    return dx, dE


def onehot(idx, K):
    N = idx.numel()
    enc = torch.empty((N, K), device=idx.device, dtype=torch.float32)
    _check(lib().alvq_onehot_f32(_ptr(idx, torch.int64, "idx"), _ptr(enc), N, K, _stream()), "alvq_onehot_f32")
    return enc


def onehot_to_index(enc):
    """(rows, K) fp32 one-hot rows -> (idx int32 [rows], flag int32 [1]); flag != 0 iff some row is not exactly one-hot."""
    rows, K = enc.shape
    idx = torch.empty((rows,), device=enc.device, dtype=torch.int32)
    flag = torch.empty((1,), device=enc.device, dtype=torch.float32)
    fill_(flag, 0.0)                                 # bit pattern 0 == int 0
    flag = flag.view(torch.int32)
    _check(lib().alvq_onehot_to_index_f32(_ptr(enc, name="encodings"), idx.data_ptr(), flag.data_ptr(), rows, K, _stream()),
           "alvq_onehot_to_index_f32")
    return idx, flag


BAG_MAX_INDICES = 16384          # indices per embedding-bag launch (its 64 KB LDS table); larger batches are chunked over B


def device_flag(device):
    """A zeroed sticky int32 flag on the device (out-of-range indices, ...)."""
    flag = torch.empty((1,), device=device, dtype=torch.float32)
    fill_(flag, 0.0)                                 # bit pattern 0 == int 0
    return flag.view(torch.int32)


def indices_to_i32(idx, K, flag):
    """int64 indices -> int32 with the range check a plain cast lacks; ORs 1 into ``flag`` for any value outside [0, K)."""
    out = torch.empty(idx.shape, device=idx.device, dtype=torch.int32)
    _check(lib().alvq_indices_to_i32(_ptr(idx, torch.int64, "idx"), out.data_ptr(), flag.data_ptr(), idx.numel(), K, _stream()),
           "alvq_indices_to_i32")
    return out


def embedding_bag_fwd(W, bias, idx, L, K, flag=None):
    """out (B, M) = bias + sum_l W[:, l*K + idx[b, l]];  W (M, L*K) fp32, idx (B, L) int32.  ``flag``: sticky device int
    that receives 1 if an index lies outside [0, K) (such terms contribute nothing).  Any B: chunked over samples."""
    B = idx.shape[0]
    M = W.shape[0]
    if W.shape[1] != L * K or tuple(idx.shape) != (B, L):
        raise RuntimeError("embedding_bag_fwd: W %s / idx %s do not match L=%d, K=%d" % (tuple(W.shape), tuple(idx.shape), L, K))
    if L > BAG_MAX_INDICES:
        raise RuntimeError("embedding_bag_fwd: L = %d exceeds the %d-index table" % (L, BAG_MAX_INDICES))
    out = torch.empty((B, M), device=W.device, dtype=torch.float32)
    step = max(1, BAG_MAX_INDICES // L)
    for b0 in range(0, B, step):
        nb = min(step, B - b0)
        _check(lib().alvq_embedding_bag_fwd_f32(_ptr(W, name="W"), _ptr(bias, name="bias"), _ptr(idx[b0:b0 + nb], torch.int32, "idx"),
                                                _ptr(out[b0:b0 + nb]), nb, L, K, M, flag.data_ptr() if flag is not None else None,
                                                _stream()), "alvq_embedding_bag_fwd_f32")
    return out


def embedding_bag_bwd(dz, idx, L, K, want_bias=True, flag=None, dW_out=None, db_out=None):
    """(dW (M, L*K) dense with the touched columns filled, dbias (M,)) from dz (B, M) and idx (B, L) int32.
    ``dW_out`` / ``db_out``: accumulate into these (a zeroed gradient sink) instead of fresh tensors."""
    B, M = dz.shape
    if dW_out is not None:
        if tuple(dW_out.shape) != (M, L * K):
            raise RuntimeError("embedding_bag_bwd: dW_out has shape %s, expected %s" % (tuple(dW_out.shape), (M, L * K)))
        dW = dW_out
    else:
        dW = torch.empty((M, L * K), device=dz.device, dtype=torch.float32)
        fill_(dW, 0.0)
    db = None
    if want_bias:
        db = db_out if db_out is not None else torch.empty((M,), device=dz.device, dtype=torch.float32)
    step = max(1, BAG_MAX_INDICES // L)
    for b0 in range(0, B, step):                     # later chunks accumulate into the same dW / dbias
        nb = min(step, B - b0)
        _check(lib().alvq_embedding_bag_bwd_f32(_ptr(dz[b0:b0 + nb], name="dz"), _ptr(idx[b0:b0 + nb], torch.int32, "idx"), _ptr(dW), _ptr(db),
                                                nb, L, K, M, int(b0 > 0 or db_out is not None),
                                                flag.data_ptr() if flag is not None else None, _stream()),
               "alvq_embedding_bag_bwd_f32")
    return dW, db


# ----------------------------------------------------------------------------------------------- misc
def jitter_gather(x, src, backward=False):
    """x (B,C,L) contiguous, src int32[L] on device."""
    L = x.shape[-1]
    y = torch.empty_like(x)
    _check(lib().alvq_jitter_gather_f32(_ptr(x, name="x"), _ptr(src, torch.int32, "src"), _ptr(y), x.numel() // L, L,
                                        int(bool(backward)), _stream()), "alvq_jitter_gather_f32")
    return y


def standardise(x, take_abs=False):
    B, C, L = x.shape
    y = torch.empty_like(x)
    _check(lib().alvq_standardise_f32(_ptr(x, name="x"), _ptr(y), B, C, L, int(bool(take_abs)), _stream()),
           "alvq_standardise_f32")
    return y


def mse(a, b):
    if a.shape != b.shape:
        raise RuntimeError("mse: shapes differ %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    loss = torch.empty((1,), device=a.device, dtype=torch.float32)
    ws = torch.empty((EW_PARTIALS,), device=a.device, dtype=torch.float32)
    _check(lib().alvq_mse_f32(_ptr(a, name="a"), _ptr(b, name="b"), _ptr(loss), _ptr(ws), a.numel(), _stream()), "alvq_mse_f32")
    return loss


def mse_backward(a, b, grad_loss):
    grad = torch.empty_like(a)
    _check(lib().alvq_mse_backward_f32(_ptr(a, name="a"), _ptr(b, name="b"), _ptr(grad_loss, name="grad_loss"), _ptr(grad),
                                       a.numel(), _stream()), "alvq_mse_backward_f32")
    return grad


def fill_(t, value=0.0):
    """t[...] = value in place (fp32, contiguous, 16-byte aligned start)."""
    if t.numel():
        _check(lib().alvq_fill_f32(_ptr(t, name="t"), float(value), t.numel(), _stream()), "alvq_fill_f32")
    return t


def add(a, b):
    out = torch.empty_like(a)
    _check(lib().alvq_add_f32(_ptr(a, name="a"), _ptr(b, name="b"), _ptr(out), a.numel(), _stream()), "alvq_add_f32")
    return out


def relu_mask(dy, t):
    """t > 0 ? dy : 0."""
    out = torch.empty_like(dy)
    _check(lib().alvq_relu_mask_f32(_ptr(dy, name="dy"), _ptr(t, name="t"), _ptr(out), dy.numel(), _stream()), "alvq_relu_mask_f32")
    return out


def row_mean(x, backward_of=None):
    """mean over the last dimension, keepdim: (B, D, L) -> (B, D, 1).  ``backward_of`` = L: the adjoint, (B, D, 1) -> (B, D, L)."""
    if backward_of is None:
        B, D, L = x.shape
        y = torch.empty((B, D, 1), device=x.device, dtype=torch.float32)
        _check(lib().alvq_row_mean_f32(_ptr(x, name="x"), _ptr(y), B * D, L, _stream()), "alvq_row_mean_f32")
        return y
    B, D, _ = x.shape
    dx = torch.empty((B, D, backward_of), device=x.device, dtype=torch.float32)
    _check(lib().alvq_row_mean_backward_f32(_ptr(x, name="dy"), _ptr(dx), B * D, backward_of, _stream()), "alvq_row_mean_backward_f32")
    return dx


def transpose12(x):
    """(B,R,C) -> (B,C,R) dense."""
    B, R, C = x.shape
    y = torch.empty((B, C, R), device=x.device, dtype=torch.float32)
    _check(lib().alvq_transpose_f32(_ptr(x, name="x"), _ptr(y), B, R, C, _stream()), "alvq_transpose_f32")
    return y


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    _check(lib().alvq_adam_f32(_ptr(param, name="param"), _ptr(grad, name="grad"), _ptr(exp_avg, name="exp_avg"),
                               _ptr(exp_avg_sq, name="exp_avg_sq"), param.numel(), int(step), float(lr), float(beta1),
                               float(beta2), float(eps), float(grad_scale), _stream()), "alvq_adam_f32")


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, scalars, beta1=0.9, beta2=0.999, eps=1e-8, skip=None):
    """Adam with {lr/bc1, sqrt(bc2), grad_scale} read from the first 3 floats of the device tensor ``scalars``
    (graph-replayable; ``adam_advance`` maintains them).  ``skip``: the skip slot (a device float tensor; non-zero = leave
    everything untouched), or None."""
    _check(lib().alvq_adam_dev_f32(_ptr(param, name="param"), _ptr(grad, name="grad"), _ptr(exp_avg, name="exp_avg"),
                                   _ptr(exp_avg_sq, name="exp_avg_sq"), param.numel(), _ptr(scalars, name="scalars"),
                                   float(beta1), float(beta2), float(eps), _ptr(skip, name="skip"), _stream()), "alvq_adam_dev_f32")


ADAM_SCALARS = 8     # floats of the device state: {lr/bc1, sqrt(bc2), grad_scale, step, skipped steps, -, -, -}


def adam_advance(scalars, lr, beta1=0.9, beta2=0.999, grad_scale=1.0, prev_skip=None):
    """Device-side ``step += 1`` on the 8-float tensor ``scalars`` = {lr/bc1, sqrt(bc2), grad_scale, step, skipped, ...}.
    ``prev_skip``: the skip slot still holding the previous step's verdict -- a skipped step does not count."""
    if scalars.numel() != ADAM_SCALARS:
        raise RuntimeError("adam_advance: scalars must hold %d floats" % ADAM_SCALARS)
    _check(lib().alvq_adam_advance_f32(_ptr(scalars, name="scalars"), float(lr), float(beta1), float(beta2),
                                       float(grad_scale), _ptr(prev_skip, name="prev_skip"), _stream()), "alvq_adam_advance_f32")


def range_flag_to_slot(slot):
    """slot[0] = 1.0 if the fp16-range flag holds a bit raised since the step's guarded ``adam_advance``, else 0.0."""
    _check(lib().alvq_range_flag_to_slot(_ptr(slot, name="slot"), _stream()), "alvq_range_flag_to_slot")


def stft_power(wave, n_fft=400, hop=160):
    """Power spectrogram of (B,S) waveforms, fp32 or fp64 (the reference's echoed signal is float64)."""
    B, S = wave.shape
    power = torch.empty((B, n_fft // 2 + 1, 1 + S // hop), device=wave.device, dtype=wave.dtype)
    if wave.dtype == torch.float64:
        _check(lib().alvq_stft_power_f64(_ptr(wave, torch.float64, "wave"), _ptr(power, torch.float64), B, S, n_fft, hop,
                                         _stream()), "alvq_stft_power_f64")
    else:
        _check(lib().alvq_stft_power_f32(_ptr(wave, name="wave"), _ptr(power), B, S, n_fft, hop, _stream()), "alvq_stft_power_f32")
    return power


def stft_complex(wave, n_fft=400, hop=160):
    """Complex STFT (B, n_fft/2+1, T) of (B,S) waveforms: complex64 for fp32 input, complex128 for fp64."""
    B, S = wave.shape
    F, T = n_fft // 2 + 1, 1 + S // hop
    out = torch.empty((B, F, T, 2), device=wave.device, dtype=wave.dtype)
    if wave.dtype == torch.float64:
        _check(lib().alvq_stft_complex_f64(_ptr(wave, torch.float64, "wave"), _ptr(out, torch.float64), B, S, n_fft, hop,
                                           _stream()), "alvq_stft_complex_f64")
    else:
        _check(lib().alvq_stft_complex_f32(_ptr(wave, name="wave"), _ptr(out), B, S, n_fft, hop, _stream()), "alvq_stft_complex_f32")
    return torch.view_as_complex(out)


def fir_same(wave, h):
    """scipy.signal.convolve(wave, h, mode='same') in float64: wave (B,S) fp32, h (Nh,) or (B,Nh) fp64."""
    B, S = wave.shape
    if h.dim() == 1:
        stride, Nh = 0, h.shape[0]
    else:
        if h.shape[0] != B:
            raise RuntimeError("fir_same: %d impulse responses for %d waveforms" % (h.shape[0], B))
        stride, Nh = h.shape[1], h.shape[1]
    out = torch.empty((B, S), device=wave.device, dtype=torch.float64)
    _check(lib().alvq_fir_same_f64(_ptr(wave, name="wave"), _ptr(h, torch.float64, "h"), _ptr(out, torch.float64), B, S, Nh,
                                   stride, _stream()), "alvq_fir_same_f64")
    return out


def spec_rir_wiener(speech_spec, echoed_spec):
    """complex64 S and complex128 E, both (B,F,T) -> (speech_pow fp32, echoed_pow fp64, rir_pow fp64, wiener (B,F) fp64)."""
    if speech_spec.dtype != torch.complex64 or echoed_spec.dtype != torch.complex128 or speech_spec.shape != echoed_spec.shape:
        raise RuntimeError("spec_rir_wiener: expected complex64 and complex128 tensors of one shape")
    B, F, T = speech_spec.shape
    dev = speech_spec.device
    sr, er = torch.view_as_real(speech_spec), torch.view_as_real(echoed_spec)
    speech_pow = torch.empty((B, F, T), device=dev, dtype=torch.float32)
    echoed_pow = torch.empty((B, F, T), device=dev, dtype=torch.float64)
    rir_pow = torch.empty((B, F, T), device=dev, dtype=torch.float64)
    wiener = torch.empty((B, F), device=dev, dtype=torch.float64)
    ws = torch.empty((B, F), device=dev, dtype=torch.float64)
    _check(lib().alvq_spec_rir_wiener_f64(_ptr(sr, name="speech_spec"), _ptr(er, torch.float64, "echoed_spec"), _ptr(speech_pow),
                                          _ptr(echoed_pow, torch.float64), _ptr(rir_pow, torch.float64),
                                          _ptr(wiener, torch.float64), _ptr(ws, torch.float64), B, F, T, _stream()),
           "alvq_spec_rir_wiener_f64")
    return speech_pow, echoed_pow, rir_pow, wiener


# ----------------------------------------------------------------------------------------------- bf16 path
class NLC:
    """An activation in the NLC-padded layout (see include/alvq.h): storage = guard rows + matrix + guard rows.

    ``fmt``: "bf16" (one plane), "bf16x3" (hi + lo bf16 planes), "f16mx" (fp16 H plane + fp8 Q plane; same bytes and
    geometry as bf16x3) or "f16" (one fp16 plane: the gradients of the f16mx_hb mode; an f16mx tensor serves wherever an
    "f16" operand is expected -- its H plane is one).  ``gscale``: for a gradient in the f16mx / f16 formats, the 4-float
    device state of its loss scale ({S, 1/S, ...}, alvq_grad_scale_f32) -- inherited by everything computed from it and
    divided out where the chain leaves the format; None for forward tensors."""
    __slots__ = ("storage", "B", "L", "C", "Cp", "rows", "guard", "planes", "has_bits", "fmt", "gscale")

    def __init__(self, B, L, C, device, planes=1, fmt=None, gscale=None):
        """planes=2: the split forms (hi / H plane, then the lo / Q plane at +alvq_nlc_plane_bytes).
        bf16 and f16mx buffers carry a tail of rows*Cp/8 bytes for the sign bits a ReLU'd convolution can leave behind
        (see alvq_conv1d_bf16 / alvq_conv1d_f16mx); ``has_bits`` says whether they are valid."""
        L_ = lib()
        self.B, self.L, self.C, self.planes = B, L, C, planes
        self.fmt = fmt or ("bf16x3" if planes == 2 else "bf16")
        self.gscale = gscale
        self.Cp = L_.alvq_nlc_channels(C)
        self.rows = L_.alvq_nlc_rows(B, L)
        self.guard = L_.alvq_nlc_guard_rows()
        self.has_bits = False
        n = planes * (self.rows + 2 * self.guard) * self.Cp + (self.rows * self.Cp // 16 if self.fmt != "bf16x3" else 0)
        self.storage = torch.empty((n,), device=device, dtype=torch.bfloat16)

    @classmethod
    def wrap(cls, storage, B, L, C, planes=1, has_bits=False, fmt=None):
        self = cls.__new__(cls)
        L_ = lib()
        self.B, self.L, self.C, self.planes = B, L, C, planes
        self.fmt = fmt or ("bf16x3" if planes == 2 else "bf16")
        self.gscale = None
        self.Cp, self.rows, self.guard = L_.alvq_nlc_channels(C), L_.alvq_nlc_rows(B, L), L_.alvq_nlc_guard_rows()
        self.storage = storage
        self.has_bits = bool(has_bits) and self.fmt != "bf16x3" and \
            storage.numel() >= planes * (self.rows + 2 * self.guard) * self.Cp + self.rows * self.Cp // 16
        return self

    @property
    def ptr(self):
        return self.storage.data_ptr() + self.guard * self.Cp * 2

    @property
    def bits_ptr(self):
        """Sign-bit area behind the plane(s) (bf16 and f16mx)."""
        return self.storage.data_ptr() + self.planes * (self.rows + 2 * self.guard) * self.Cp * 2

    def matrix(self, plane=0):
        g = (self.guard + plane * (self.rows + 2 * self.guard)) * self.Cp
        return self.storage[g:g + self.rows * self.Cp].view(self.rows, self.Cp)

    def to_ncl(self):
        """(B,C,L) fp32 copy -- test/debug helper (torch indexing, not on the hot path)."""
        if self.fmt in ("f16mx", "f16"):
            return nlc_to_ncl(self)
        m = self.matrix(0).float()
        if self.planes == 2:
            m = m + self.matrix(1).float()
        m = m[1:1 + self.B * (self.L + 1)].view(self.B, self.L + 1, self.Cp)
        return m[:, :self.L, :self.C].permute(0, 2, 1).contiguous()


def nlc_like(ref, C):
    return NLC(ref.B, ref.L, C, ref.storage.device, ref.planes, ref.fmt, ref.gscale)


def _sptr(gscale, which):
    """Device pointer to S (which = 0) or 1/S (which = 1) of a loss-scale state, or None."""
    return None if gscale is None else gscale.data_ptr() + 4 * which


def grad_scale(x):
    """Loss scale of a backward chain in the f16mx format: 4-float device state {S, 1/S, -, -} with S the power of two
    that puts amax|x| in [2^7, 2^8) -- chosen on the device, no host round trip (graph-capturable)."""
    state = torch.empty((4,), device=x.device, dtype=torch.float32)
    fill_(state, 0.0)
    _check(lib().alvq_grad_scale_f32(_ptr(x, name="x"), x.numel(), state.data_ptr(), _stream()), "alvq_grad_scale_f32")
    return state


def f16mx_range_flag(reset=True, device="cuda"):
    """Sticky range flag of the f16mx / fp16 formats on the current device (one host sync): bit 0 = a value >= 65504 in
    magnitude entered the format (it was stored saturated), bit 1 = a NaN did, bit 2 = a convolution PRODUCED such a value
    (an activation, or a scaled gradient that outgrew its headroom).  The formats carry fp16's range: standardised
    spectrograms and Kaiming-scale weights stay orders of magnitude below; gradients are brought into range by the
    device-chosen loss scale.  0 = nothing saturated since the last reset."""
    out = device_flag(torch.device(device))
    _check(lib().alvq_f16mx_range_flag(out.data_ptr(), int(bool(reset)), _stream()), "alvq_f16mx_range_flag")
    return int(out.item())


def _fmt_serves(t, ref):
    """Can ``t`` be read as an operand of ``ref``'s format?  Same format, or an f16mx tensor read through its H plane."""
    return (t.planes, t.fmt) == (ref.planes, ref.fmt) or (ref.fmt == "f16" and t.fmt == "f16mx") or \
        (ref.fmt == "bf16" and t.fmt == "bf16x3")           # ... or a bf16x3 tensor through its hi plane (bf16x3_hb's backward)


def _nlc_ptr(t, ref, C, name):
    if t is None:
        return None
    if not isinstance(t, NLC) or (t.B, t.L, t.C) != (ref.B, ref.L, C) or not _fmt_serves(t, ref):
        raise RuntimeError("%s: expected an NLC activation of (B=%d, L=%d, C=%d, %s)" % (name, ref.B, ref.L, C, ref.fmt))
    return t.ptr


def ncl_to_nlc(x, planes=1, fmt=None, gscale=None):
    """(B,C,L) fp32 dense -> NLC bf16 (planes=2: split hi/lo; fmt="f16mx": fp16 + fp8 planes, multiplied by the loss
    scale S of ``gscale`` when given)."""
    B, C, L = x.shape
    out = NLC(B, L, C, x.device, planes, fmt, gscale)
    if out.fmt == "f16":
        _check(lib().alvq_ncl_to_nlc_f16(_ptr(x, name="x"), out.ptr, B, C, L, _sptr(gscale, 0), _stream()), "alvq_ncl_to_nlc_f16")
    elif out.fmt == "f16mx":
        _check(lib().alvq_ncl_to_nlc_f16mx(_ptr(x, name="x"), out.ptr, B, C, L, _sptr(gscale, 0), _stream()), "alvq_ncl_to_nlc_f16mx")
    elif planes == 2:
        _check(lib().alvq_ncl_to_nlc_bf16x3(_ptr(x, name="x"), out.ptr, B, C, L, _stream()), "alvq_ncl_to_nlc_bf16x3")
    else:
        _check(lib().alvq_ncl_to_nlc_bf16(_ptr(x, name="x"), out.ptr, B, C, L, _stream()), "alvq_ncl_to_nlc_bf16")
    return out


def rows_to_nlc_supported(fmt, L, standardise=False):
    return fmt in ("bf16", "bf16x3", "f16mx") and (not standardise or 2 <= L <= lib().alvq_rows_to_nlc_max_std_rows())


def rows_to_nlc(x_blc, planes=1, fmt=None, standardise=False, take_abs=False):
    """(B, L, C) fp32 contiguous -- the model input (B, C, L) seen through ``permute(0, 2, 1)`` -- straight into the NLC
    layout (channels are already contiguous: no transposition either way), optionally standardised over L per (b, c) in the
    same pass (train_rir.py:43-44; bit-identical to standardise -> transpose -> ncl_to_nlc)."""
    B, L, C = x_blc.shape
    out = NLC(B, L, C, x_blc.device, planes, fmt)
    if not rows_to_nlc_supported(out.fmt, L, standardise):
        raise RuntimeError("rows_to_nlc: format %s / L = %d not supported" % (out.fmt, L))
    code = {"bf16": 1, "bf16x3": 2, "f16mx": 3}[out.fmt]
    _check(lib().alvq_rows_to_nlc(_ptr(x_blc, name="x"), out.ptr, B, C, L, code, int(bool(standardise)), int(bool(take_abs)), _stream()),
           "alvq_rows_to_nlc")
    return out


def nlc_to_ncl(a):
    """NLC bf16 -> (B,C,L) fp32 dense."""
    y = torch.empty((a.B, a.C, a.L), device=a.storage.device, dtype=torch.float32)
    if a.fmt == "f16":
        _check(lib().alvq_nlc_to_ncl_f16(a.ptr, _ptr(y), a.B, a.C, a.L, _sptr(a.gscale, 1), _stream()), "alvq_nlc_to_ncl_f16")
    elif a.fmt == "f16mx":
        _check(lib().alvq_nlc_to_ncl_f16mx(a.ptr, _ptr(y), a.B, a.C, a.L, _sptr(a.gscale, 1), _stream()), "alvq_nlc_to_ncl_f16mx")
    elif a.planes == 2:
        _check(lib().alvq_nlc_to_ncl_bf16x3(a.ptr, _ptr(y), a.B, a.C, a.L, _stream()), "alvq_nlc_to_ncl_bf16x3")
    else:
        _check(lib().alvq_nlc_to_ncl_f32(a.ptr, _ptr(y), a.B, a.C, a.L, _stream()), "alvq_nlc_to_ncl_f32")
    return y


def pack_weight(w, w_layout, planes=1):
    """fp32 weight (M,C,KW) [OIK] or (C,M,KW) [IOK] -> packed bf16 image(s) + (M, C, KW, planes)."""
    if w_layout == W_OIK:
        M, C, KW = w.shape
    else:
        C, M, KW = w.shape
    n = lib().alvq_packed_weight_elems(M, C, KW)
    wp = torch.empty((min(planes, 2) * n,), device=w.device, dtype=torch.bfloat16)
    if planes == 3:                                  # f16mx: H + Q images
        pack_weights_batch([(w, wp, w_layout)], 3)
    elif planes == 2:
        _check(lib().alvq_pack_weight_bf16x3(_ptr(w, name="w"), wp.data_ptr(), M, C, KW, w_layout, _stream()), "alvq_pack_weight_bf16x3")
    else:
        _check(lib().alvq_pack_weight_bf16(_ptr(w, name="w"), wp.data_ptr(), M, C, KW, w_layout, _stream()), "alvq_pack_weight_bf16")
    return wp, (M, C, KW, planes)


class PackDesc(ctypes.Structure):
    """struct alvq_pack_desc (include/alvq.h)"""
    _fields_ = [("w", ctypes.c_void_p), ("wp", ctypes.c_void_p), ("M", ctypes.c_int32), ("C", ctypes.c_int32),
                ("KW", ctypes.c_int32), ("w_layout", ctypes.c_int32)]


def packed_weight_alloc(w, w_layout, planes=1):
    """Uninitialised packed image(s) for ``w`` + the (M, C, KW, planes) tag ``pack_weight`` returns."""
    if w_layout == W_OIK:
        M, C, KW = w.shape
    else:
        C, M, KW = w.shape
    n = lib().alvq_packed_weight_elems(M, C, KW)
    return torch.empty((min(planes, 2) * n,), device=w.device, dtype=torch.bfloat16), (M, C, KW, planes)


def pack_weights_batch(entries, planes=1):
    """entries: [(w fp32 cuda tensor, packed image from packed_weight_alloc, w_layout)] -- one launch for all."""
    if not entries:
        return
    arr = (PackDesc * len(entries))()
    for d, (w, wp, w_layout) in zip(arr, entries):
        if w_layout == W_OIK:
            M, C, KW = w.shape
        else:
            C, M, KW = w.shape
        need = min(planes, 2) * lib().alvq_packed_weight_elems(M, C, KW)
        if wp.numel() != need or wp.dtype != torch.bfloat16:
            raise RuntimeError("pack_weights_batch: packed image has %d elements, expected %d" % (wp.numel(), need))
        d.w, d.wp, d.M, d.C, d.KW, d.w_layout = _ptr(w, name="w"), wp.data_ptr(), M, C, KW, w_layout
    _check(lib().alvq_pack_weights_bf16_batch(ctypes.addressof(arr), len(entries), planes, _stream()),
           "alvq_pack_weights_bf16_batch")


class AdamPackDesc(ctypes.Structure):
    """struct alvq_adam_pack_desc (include/alvq.h)"""
    _fields_ = [("w", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("wp_oik", ctypes.c_void_p), ("wp_iok", ctypes.c_void_p), ("dim0", ctypes.c_int32), ("dim1", ctypes.c_int32),
                ("KW", ctypes.c_int32)]


def adam_pack_batch(entries, planes, scalars, beta1=0.9, beta2=0.999, eps=1e-8, skip=None):
    """entries: [(w, g, m, v fp32 views of one conv weight and its Adam state, packed OIK image or None, packed IOK image or
    None)] -- Adam's update and the re-pack of every image in one launch."""
    if not entries:
        return
    arr = (AdamPackDesc * len(entries))()
    for d, (w, g, m, v, oik, iok) in zip(arr, entries):
        d.w, d.g, d.m, d.v = _ptr(w, name="w"), _ptr(g, name="g"), _ptr(m, name="m"), _ptr(v, name="v")
        d.wp_oik = oik.data_ptr() if oik is not None else None
        d.wp_iok = iok.data_ptr() if iok is not None else None
        d.dim0, d.dim1, d.KW = w.shape
    _check(lib().alvq_adam_pack_batch(ctypes.addressof(arr), len(entries), planes, _ptr(scalars, name="scalars"), float(beta1),
                                      float(beta2), float(eps), _ptr(skip, name="skip"), _stream()), "alvq_adam_pack_batch")


def adam_segments(param, grad, exp_avg, exp_avg_sq, segments, scalars, beta1=0.9, beta2=0.999, eps=1e-8, skip=None):
    """Adam over the element ranges [(lo, hi), ...] of the flat buffers, one launch."""
    segments = [(int(a), int(b)) for a, b in segments if b > a]
    if not segments:
        return
    lo = (ctypes.c_int64 * len(segments))(*[a for a, _ in segments])
    hi = (ctypes.c_int64 * len(segments))(*[b for _, b in segments])
    _check(lib().alvq_adam_segments_f32(_ptr(param, name="param"), _ptr(grad, name="grad"), _ptr(exp_avg, name="exp_avg"),
                                        _ptr(exp_avg_sq, name="exp_avg_sq"), lo, hi, len(segments), _ptr(scalars, name="scalars"),
                                        float(beta1), float(beta2), float(eps), _ptr(skip, name="skip"), _stream()),
           "alvq_adam_segments_f32")


def relu_mask_bf16(dy, t):
    out = nlc_like(dy, dy.C)
    if dy.fmt == "f16":                  # t: fp16, or an f16mx activation through its H plane (sign test on the int16 pattern)
        n = dy.rows * dy.Cp
        _check(lib().alvq_relu_mask_bf16(dy.ptr, _nlc_ptr(t, dy, dy.C, "t"), out.ptr, n, _stream()), "alvq_relu_mask_bf16")
    elif dy.fmt == "f16mx":
        _check(lib().alvq_relu_mask_f16mx(dy.ptr, _nlc_ptr(t, dy, dy.C, "t"), out.ptr, dy.B, dy.C, dy.L, _stream()),
               "alvq_relu_mask_f16mx")
    elif dy.planes == 2:
        _check(lib().alvq_relu_mask_bf16x3(dy.ptr, _nlc_ptr(t, dy, dy.C, "t"), out.ptr, dy.B, dy.C, dy.L, _stream()),
               "alvq_relu_mask_bf16x3")
    else:
        n = dy.rows * dy.Cp
        _check(lib().alvq_relu_mask_bf16(dy.ptr, _nlc_ptr(t, dy, dy.C, "t"), out.ptr, n, _stream()), "alvq_relu_mask_bf16")
    return out


USE_SIGN_BITS = os.environ.get("ALVQ_SIGN_BITS", "1") != "0"


def conv1d_bf16(x, packed, bias=None, skip1=None, skip2=None, mask=None, post=None, relu=False, out_ncl=False):
    """x: NLC; packed = pack_weight(...).  Returns NLC y, (y, y2) with post, or a (B,M,L) fp32 tensor if out_ncl."""
    wp, (M, C, KW, wplanes) = packed
    if C != x.C:
        raise RuntimeError("conv1d_bf16: weight expects %d input channels, x has %d" % (C, x.C))
    # an f16 launch reads the H image of an f16mx packed weight, a bf16 launch may read the hi image of a bf16x3 one
    if wplanes != (3 if x.fmt in ("f16mx", "f16") else x.planes) and not (x.fmt == "bf16" and wplanes == 2):
        raise RuntimeError("conv1d_bf16: weight packed for format %d, activation is %s" % (wplanes, x.fmt))
    split = x.planes == 2
    if bias is not None and bias.numel() != M:
        raise RuntimeError("conv1d_bf16: bias has %d elements, expected %d" % (bias.numel(), M))
    y = y2 = y_ncl = None
    if out_ncl:
        y_ncl = torch.empty((x.B, M, x.L), device=wp.device, dtype=torch.float32)
    else:
        y = nlc_like(x, M)
        y2 = nlc_like(x, M) if post is not None else None
    # KernelTimer names follow rocprofv3's kernel names -- function plus its leading template arguments -- so that bench.py's
    # per-kernel figures can be laid beside `rocprofv3 --kernel-trace --stats` line by line: conv1d_f16mx_kernel<OUT, KW, ...>,
    # conv1d_bf16x3_kernel<OUT, KW, ...>, conv1d_bf16_k3_kernel<OUT, F16>, conv1d_bf16_v2_kernel<OUT, F16>,
    # conv1d_bf16_kernel<KW, OUT, F16> (OUT: 0 = NLC output, 1 = fp32 NCL; F16: the fp16 opcodes of the bf16 kernels)
    o_ = int(bool(out_ncl))
    if x.fmt == "f16mx":
        family, fn = "conv1d_f16mx_kernel<%d, %d, ...>" % (o_, KW), lib().alvq_conv1d_f16mx
    elif split:
        family, fn = "conv1d_bf16x3_kernel<%d, %d, ...>" % (o_, KW), lib().alvq_conv1d_bf16x3
    else:
        # mirrors the dispatch in csrc/conv1d_bf16.hip: wide layers go to the 256x256-tile kernels (k3 for width 3)
        f16 = int(x.fmt == "f16")
        min_tiles = get_option("wide_min_tiles")
        wide = ((M + 255) // 256 * 256 - M) <= 32 and (x.rows // 256) * ((M + 255) // 256) >= min_tiles
        if wide:
            family = ("conv1d_bf16_k3_kernel<%d, %d>" if KW == 3 else "conv1d_bf16_v2_kernel<%d, %d>") % (o_, f16)
        else:
            family = "conv1d_bf16_kernel<%d, %d, %d>" % (KW, o_, f16)
        fn = lib().alvq_conv1d_f16 if f16 else lib().alvq_conv1d_bf16
    # sign bits: a ReLU'd bf16 output records them; a mask operand that carries valid bits is passed as bits
    extra = ()
    mask_ptr = _nlc_ptr(mask, x, M, "mask")
    with_bits = x.fmt != "bf16x3"
    if with_bits:
        mask_bits = None
        if mask is not None and mask.has_bits and USE_SIGN_BITS:
            mask_bits, mask_ptr = mask.bits_ptr, None
        bits_out = y.bits_ptr if (y is not None and relu and USE_SIGN_BITS) else None
        extra = (mask_bits, bits_out)
    if x.fmt in ("f16mx", "f16"):
        extra += (_sptr(x.gscale, 1) if out_ncl else None,)     # a gradient leaving the format: divide the loss scale out
    with _timed(family, 2.0 * x.B * x.L * M * C * KW):
        rc = fn(x.ptr, wp.data_ptr(), _ptr(bias, name="bias"), _nlc_ptr(skip1, x, M, "skip1"),
                                    _nlc_ptr(skip2, x, M, "skip2"), mask_ptr,
                                    _nlc_ptr(post, x, M, "post"), y.ptr if y is not None else None,
                                    y2.ptr if y2 is not None else None, _ptr(y_ncl), x.B, C, M, x.L, KW,
                                    int(bool(relu)), *extra, _stream())
    if with_bits and y is not None and relu and USE_SIGN_BITS:
        y.has_bits = True
    _check(rc, "alvq_conv1d_bf16")
    if out_ncl:
        return y_ncl
    return (y, y2) if post is not None else y


def _wgrad16_name(KW, with_bias, f16):
    """rocprofv3's name of the 16-bit weight-gradient kernel a launch gets (mirrors csrc/conv1d_wgrad_bf16_v2.hip: the v3
    kernels serve the launches without a bias gradient, option wgrad_v3)."""
    sel = get_option("wgrad_v3")
    if not with_bias and ((KW == 1 and sel & 1) or (KW == 3 and sel & 2)):
        return "conv1d_wgrad_bf16_v3_kernel<%s, %d>" % ("3, 1, 2" if KW == 3 else "1, 2, 4", f16)
    return "conv1d_wgrad_bf16_v2_kernel<%s, %d>" % ("3, 2" if KW == 3 else "1, 4", f16)


def conv1d_wgrad_bf16(dy, x, KW, w_layout=W_OIK, want_bias=False, dw_out=None, dbias_out=None, accumulate=False, defer=None):
    """dy, x: NLC.  fp32 dw in the weight's native layout (and dbias).  ``defer`` (a list; bf16 / fp16 formats, with
    ``accumulate`` into caller-owned dw_out / dbias_out): launch the contraction only and append the reduction's descriptors
    -- the caller sums them all later with ``wgrad_reduce_batch``."""
    M, C = dy.C, x.C
    shape = (M, C, KW) if w_layout == W_OIK else (C, M, KW)
    dev = x.storage.device
    if dw_out is None:
        dw_out = torch.empty(shape, device=dev, dtype=torch.float32)
        accumulate = False
    elif tuple(dw_out.shape) != shape:
        raise RuntimeError("conv1d_wgrad_bf16: dw_out has shape %s, expected %s" % (tuple(dw_out.shape), shape))
    if want_bias and dbias_out is None:
        dbias_out = torch.empty((M,), device=dev, dtype=torch.float32)
    if not _fmt_serves(x, dy):
        raise RuntimeError("conv1d_wgrad_bf16: dy and x differ in format")
    extra = ()
    if dy.fmt == "f16":                  # x: fp16, or the H plane of a saved f16mx activation
        family, fn, wsfn = _wgrad16_name(KW, want_bias, 1), lib().alvq_conv1d_wgrad_f16, lib().alvq_conv1d_wgrad_bf16_workspace_bytes
        extra = (_sptr(dy.gscale, 1),)
    elif x.fmt == "f16mx":
        family, fn, wsfn = "conv1d_wgrad_f16mx_kernel", lib().alvq_conv1d_wgrad_f16mx, lib().alvq_conv1d_wgrad_f16mx_workspace_bytes
        extra = (_sptr(dy.gscale, 1),)
    elif x.planes == 2 and dy.planes == 2:
        family, fn, wsfn = "conv1d_wgrad_bf16x3_kernel", lib().alvq_conv1d_wgrad_bf16x3, lib().alvq_conv1d_wgrad_bf16x3_workspace_bytes
    else:
        family, fn, wsfn = _wgrad16_name(KW, want_bias, 0), lib().alvq_conv1d_wgrad_bf16, lib().alvq_conv1d_wgrad_bf16_workspace_bytes
    deferred = defer is not None and accumulate and dy.fmt in ("bf16", "f16")
    ws_ptr = arena_alloc(wsfn(x.B, C, M, x.L, KW), dev) if deferred else _workspace(wsfn(x.B, C, M, x.L, KW), dev).data_ptr()
    with _timed(family, 2.0 * x.B * x.L * M * C * KW):
        rc = fn(dy.ptr, x.ptr, _ptr(dw_out, name="dw"),
                                          _ptr(dbias_out, name="dbias") if want_bias else None, ws_ptr,
                                          x.B, C, M, x.L, KW, w_layout, WGRAD_DEFER if deferred else int(bool(accumulate)), *extra,
                                          _stream())
    _check(rc, "alvq_conv1d_wgrad_bf16")
    if deferred:
        _defer_descs(defer, ws_ptr, dw_out, dbias_out if want_bias else None, dy.gscale if dy.fmt == "f16" else None, 1, x.B, C, M,
                     x.L, KW, w_layout)
    return (dw_out, dbias_out) if want_bias else dw_out


def conv1d_wgrad_bf16_multi(pairs, KW, w_layout=W_OIK, dw_out=None, accumulate=False, defer=None):
    """dw (+)= sum_i wgrad(dy_i, x_i) in one launch (shared residual weights).  pairs: [(dy NLC, x NLC), ...] (1..4)."""
    dy0, x0 = pairs[0]
    M, C = dy0.C, x0.C
    fx, x3, h16 = dy0.fmt == "f16mx", dy0.fmt == "bf16x3", dy0.fmt == "f16"
    for dy, x in pairs:
        if (dy.B, dy.L, dy.C, dy.fmt, dy.planes, x.B, x.L, x.C) != (dy0.B, dy0.L, M, dy0.fmt, dy0.planes, x0.B, x0.L, C) or \
                not _fmt_serves(x, dy):
            raise RuntimeError("conv1d_wgrad_bf16_multi: all segments must share one shape and format")
        if (fx or h16) and dy.gscale is not dy0.gscale:
            raise RuntimeError("conv1d_wgrad_bf16_multi: the segments belong to different loss-scale chains")
    shape = (M, C, KW) if w_layout == W_OIK else (C, M, KW)
    dev = x0.storage.device
    if dw_out is None:
        dw_out = torch.empty(shape, device=dev, dtype=torch.float32)
        accumulate = False
    elif tuple(dw_out.shape) != shape:
        raise RuntimeError("conv1d_wgrad_bf16_multi: dw_out has shape %s, expected %s" % (tuple(dw_out.shape), shape))
    n = len(pairs)
    dys = (ctypes.c_void_p * n)(*[dy.ptr for dy, _ in pairs])
    xs = (ctypes.c_void_p * n)(*[x.ptr for _, x in pairs])
    deferred = defer is not None and accumulate and dy0.fmt in ("bf16", "f16")
    if deferred or h16 or not (fx or x3):
        nbytes = lib().alvq_conv1d_wgrad_bf16_workspace_bytes(x0.B, C, M, x0.L, KW)
        ws_ptr = arena_alloc(nbytes, dev) if deferred else _workspace(nbytes, dev).data_ptr()
        acc = WGRAD_DEFER if deferred else int(bool(accumulate))
    if h16:
        with _timed(_wgrad16_name(KW, False, 1), 2.0 * n * x0.B * x0.L * M * C * KW):
            rc = lib().alvq_conv1d_wgrad_f16_multi(dys, xs, n, _ptr(dw_out, name="dw"), ws_ptr, x0.B, C, M, x0.L, KW,
                                                   w_layout, acc, _sptr(dy0.gscale, 1), _stream())
        _check(rc, "alvq_conv1d_wgrad_f16_multi")
        if deferred:
            _defer_descs(defer, ws_ptr, dw_out, None, dy0.gscale, n, x0.B, C, M, x0.L, KW, w_layout)
        return dw_out
    if fx:
        ws = _workspace(lib().alvq_conv1d_wgrad_f16mx_workspace_bytes(x0.B, C, M, x0.L, KW), dev)
        with _timed("conv1d_wgrad_f16mx_kernel", 2.0 * n * x0.B * x0.L * M * C * KW):
            rc = lib().alvq_conv1d_wgrad_f16mx_multi(dys, xs, n, _ptr(dw_out, name="dw"), ws.data_ptr(), x0.B, C, M, x0.L, KW,
                                                     w_layout, int(bool(accumulate)), _sptr(dy0.gscale, 1), _stream())
        _check(rc, "alvq_conv1d_wgrad_f16mx_multi")
        return dw_out
    if x3:
        ws = _workspace(lib().alvq_conv1d_wgrad_bf16x3_workspace_bytes(x0.B, C, M, x0.L, KW), dev)
        with _timed("conv1d_wgrad_bf16x3_kernel", 2.0 * n * x0.B * x0.L * M * C * KW):
            rc = lib().alvq_conv1d_wgrad_bf16x3_multi(dys, xs, n, _ptr(dw_out, name="dw"), ws.data_ptr(), x0.B, C, M, x0.L, KW,
                                                      w_layout, int(bool(accumulate)), _stream())
        _check(rc, "alvq_conv1d_wgrad_bf16x3_multi")
        return dw_out
    with _timed(_wgrad16_name(KW, False, 0), 2.0 * n * x0.B * x0.L * M * C * KW):
        rc = lib().alvq_conv1d_wgrad_bf16_multi(dys, xs, n, _ptr(dw_out, name="dw"), ws_ptr, x0.B, C, M, x0.L, KW,
                                                w_layout, acc, _stream())
    _check(rc, "alvq_conv1d_wgrad_bf16_multi")
    if deferred:
        _defer_descs(defer, ws_ptr, dw_out, None, None, n, x0.B, C, M, x0.L, KW, w_layout)
    return dw_out
