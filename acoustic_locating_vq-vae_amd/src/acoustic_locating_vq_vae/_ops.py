"""Autograd glue: one ``torch.autograd.Function`` per reference module, each a hand-scheduled chain of
fused HIP launches (see include/alvq.h).  Forward keeps only post-ReLU activations: because the reference's
``nn.ReLU(True)`` rewrites every skip operand in place (residual.py:36,66; convolutional_encoder.py:42 --
SURVEY App. B.2) no pre-activation value is ever consumed, so bias/skip/ReLU live in the producing conv's
epilogue and the ReLU-backward masks live in the data-grad conv's epilogue.

Notation follows SURVEY App. A:  t_r = relu(h_{r-1}),  u_r = relu(W1 *3 t_r),  h_r = t_r + W2 *1 u_r.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native as N

OIK, IOK = N.W_OIK, N.W_IOK


def dense(x):
    """Materialise a strided input as dense (B,C,L) fp32 on the GPU (train_rir.py:45 hands over a permuted view)."""
    if x.dtype != torch.float32:
        x = x.float()
    if x.is_contiguous():
        return x
    if x.dim() == 3 and x.permute(0, 2, 1).is_contiguous():
        return N.transpose12(x.permute(0, 2, 1))
    return x.contiguous()


def _need_gpu(x, who):
    if not x.is_cuda:
        raise RuntimeError("%s: input is on %s -- this build runs the HIP path only (no CPU fallback); "
                           "move the model and data to the GPU" % (who, x.device))


# --------------------------------------------------------------------------------------------------
# residual stack (shared W1, W2 used R times; residual_stack.py:40-46)
# --------------------------------------------------------------------------------------------------
def _stack_forward(t1, w1, w2, R, post=None):
    """t1 = relu(h0).  Returns (ts[1..R+1], us[1..R], out) where out = ts[R+1] (+ post)."""
    ts, us = [t1], []
    out = None
    for r in range(R):
        u = N.conv1d(ts[-1], w1, relu=True)
        us.append(u)
        if r == R - 1 and post is not None:
            t, out = N.conv1d(u, w2, skip1=ts[-1], relu=True, post=post)
        else:
            t = N.conv1d(u, w2, skip1=ts[-1], relu=True)
        ts.append(t)
    if out is None:
        out = ts[-1]
    return ts, us, out


def _stack_backward(dh, ts, us, w1, w2, R, outer=None):
    """dh = grad wrt h_R, already masked by (t_{R+1} > 0).  outer = extra grad flowing into t_1 (encoder skip).
    Returns (dh0 masked by t_1>0, dW1, dW2)."""
    dw1 = dw2 = None
    for r in range(R - 1, -1, -1):
        du = N.conv1d(dh, w2, mask=us[r], w_layout=IOK)                       # k1 data-grad, * (u_r > 0)
        dw2 = N.conv1d_wgrad(dh, us[r], 1, OIK, dw_out=dw2, accumulate=dw2 is not None)
        dw1 = N.conv1d_wgrad(du, ts[r], 3, OIK, dw_out=dw1, accumulate=dw1 is not None)
        dh = N.conv1d(du, w1, skip1=dh, skip2=outer if r == 0 else None, mask=ts[r], w_layout=IOK)
    return dh, dw1, dw2


class EncoderFn(torch.autograd.Function):
    """ConvolutionalEncoder.forward (convolutional_encoder.py:39-44): relu(h_R) + relu(h_0)."""

    @staticmethod
    def forward(ctx, x, wc, bc, w1, w2, R):
        x = dense(x)
        t1 = N.conv1d(x, wc, bc, relu=True)
        ts, us, out = _stack_forward(t1, w1, w2, R, post=t1)
        ctx.R = R
        ctx.save_for_backward(x, wc, w1, w2, *ts, *us)
        return out

    @staticmethod
    def backward(ctx, d_out):
        R = ctx.R
        saved = ctx.saved_tensors
        x, wc, w1, w2 = saved[:4]
        ts, us = saved[4:4 + R + 1], saved[5 + R:5 + 2 * R]
        d_out = d_out.contiguous()
        dh = N.relu_mask(d_out, ts[R])                                          # * (h_R > 0)
        dh0, dw1, dw2 = _stack_backward(dh, ts, us, w1, w2, R, outer=d_out)
        dwc, dbc = N.conv1d_wgrad(dh0, x, 3, OIK, want_bias=True)
        dx = N.conv1d(dh0, wc, w_layout=IOK) if ctx.needs_input_grad[0] else None
        return dx, dwc, dbc, dw1, dw2, None


class StackFn(torch.autograd.Function):
    """ResidualStack.forward on its own (residual_stack.py:43-46): relu(h_R) from h_0."""

    @staticmethod
    def forward(ctx, h0, w1, w2, R):
        h0 = dense(h0)
        t1 = N.relu_mask(h0, h0)
        ts, us, out = _stack_forward(t1, w1, w2, R)
        ctx.R = R
        ctx.save_for_backward(w1, w2, *ts, *us)
        return out

    @staticmethod
    def backward(ctx, d_out):
        R = ctx.R
        saved = ctx.saved_tensors
        w1, w2 = saved[:2]
        ts, us = saved[2:2 + R + 1], saved[3 + R:3 + 2 * R]
        dh = N.relu_mask(d_out.contiguous(), ts[R])
        dh0, dw1, dw2 = _stack_backward(dh, ts, us, w1, w2, R)
        return dh0, dw1, dw2, None


class ResidualLayerFn(torch.autograd.Function):
    """One Residual on its own (residual.py:65-66): relu(x) + W2 *1 relu(W1 *3 relu(x)), no trailing ReLU."""

    @staticmethod
    def forward(ctx, x, w1, w2):
        x = dense(x)
        t = N.relu_mask(x, x)
        u = N.conv1d(t, w1, relu=True)
        ctx.save_for_backward(w1, w2, t, u)
        return N.conv1d(u, w2, skip1=t)

    @staticmethod
    def backward(ctx, dy):
        w1, w2, t, u = ctx.saved_tensors
        dy = dy.contiguous()
        du = N.conv1d(dy, w2, mask=u, w_layout=IOK)
        dw2 = N.conv1d_wgrad(dy, u, 1, OIK)
        dw1 = N.conv1d_wgrad(du, t, 3, OIK)
        dx = N.conv1d(du, w1, skip1=dy, mask=t, w_layout=IOK)
        return dx, dw1, dw2


class ConvFn(torch.autograd.Function):
    """Plain Conv1d / ConvTranspose1d (k in {1,3}, stride 1, same padding) with bias
    (convolutional_vq_vae.py:32-37,95)."""

    @staticmethod
    def forward(ctx, x, w, b, layout):
        x = dense(x)
        ctx.layout = layout
        ctx.save_for_backward(x, w)
        return N.conv1d(x, w, b, w_layout=layout)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        kw = w.shape[2]
        if ctx.needs_input_grad[2]:
            dw, db = N.conv1d_wgrad(dy, x, kw, ctx.layout, want_bias=True)
        else:
            dw, db = N.conv1d_wgrad(dy, x, kw, ctx.layout), None
        dx = N.conv1d(dy, w, w_layout=IOK if ctx.layout == OIK else OIK) if ctx.needs_input_grad[0] else None
        return dx, dw, db, None


class JitterFn(torch.autograd.Function):
    """Jitter (modules/jitter.py:42-70) with a host-drawn source-index vector."""

    @staticmethod
    def forward(ctx, q, src):
        ctx.save_for_backward(src)
        return N.jitter_gather(dense(q), src)

    @staticmethod
    def backward(ctx, dy):
        (src,) = ctx.saved_tensors
        return N.jitter_gather(dy.contiguous(), src, backward=True), None


class DecoderFn(torch.autograd.Function):
    """DeconvolutionalDecoder.forward (deconvolutional_decoder.py:62-79)."""

    @staticmethod
    def forward(ctx, q, src, wd, bd, w1, w2, wt1, bt1, wt2, bt2, wt3, bt3, R):
        q = dense(q)
        qj = N.jitter_gather(q, src) if src is not None else q
        t1 = N.conv1d(qj, wd, bd, relu=True)
        ts, us, top = _stack_forward(t1, w1, w2, R)
        a1 = N.conv1d(top, wt1, bt1, relu=True, w_layout=IOK)
        a2 = N.conv1d(a1, wt2, bt2, relu=True, w_layout=IOK)
        y = N.conv1d(a2, wt3, bt3, w_layout=IOK)
        ctx.R = R
        ctx.has_src = src is not None
        extra = (src,) if src is not None else ()
        ctx.save_for_backward(qj, wd, w1, w2, wt1, wt2, wt3, a1, a2, *ts, *us, *extra)
        return y

    @staticmethod
    def backward(ctx, dy):
        R = ctx.R
        saved = ctx.saved_tensors
        qj, wd, w1, w2, wt1, wt2, wt3, a1, a2 = saved[:9]
        ts, us = saved[9:9 + R + 1], saved[10 + R:10 + 2 * R]
        dy = dy.contiguous()
        da2 = N.conv1d(dy, wt3, mask=a2, w_layout=OIK)                         # convT data-grad, * (a2 > 0)
        dwt3, dbt3 = N.conv1d_wgrad(dy, a2, 3, IOK, want_bias=True)
        da1 = N.conv1d(da2, wt2, mask=a1, w_layout=OIK)
        dwt2, dbt2 = N.conv1d_wgrad(da2, a1, 3, IOK, want_bias=True)
        dh = N.conv1d(da1, wt1, mask=ts[R], w_layout=OIK)                      # * (h_R > 0)
        dwt1, dbt1 = N.conv1d_wgrad(da1, ts[R], 3, IOK, want_bias=True)
        dh0, dw1, dw2 = _stack_backward(dh, ts, us, w1, w2, R)
        dwd, dbd = N.conv1d_wgrad(dh0, qj, 3, OIK, want_bias=True)
        dq = None
        if ctx.needs_input_grad[0]:
            dq = N.conv1d(dh0, wd, w_layout=IOK)
            if ctx.has_src:
                dq = N.jitter_gather(dq, saved[-1], backward=True)
        return dq, None, dwd, dbd, dw1, dw2, dwt1, dbt1, dwt2, dbt2, dwt3, dbt3, None


class VQFn(torch.autograd.Function):
    """VectorQuantizer.forward (vector_quantizer.py:29-58) -> (loss, q_st, perplexity, idx)."""

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
        dx, dE = N.vq_backward(g, dloss.reshape(1).contiguous(), flat, codebook, idx, ctx.beta,
                               want_dx=ctx.needs_input_grad[0], want_dE=want_de)
        if dx is not None:
            dx = dx.view(ctx.zshape)
        return dx, dE, None, None


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
    """Host side of Jitter: same ``np.random`` call order as modules/jitter.py:50-68 (one draw per column, one
    more per replaced interior column; a column is replaced with probability 1 - p, the inversion at :55)."""
    src = np.arange(length, dtype=np.int32)
    choice = np.random.choice
    for i in range(length):
        replace = [True, False][choice([1, 0], p=[probability, 1 - probability])]
        if replace:
            if i == 0:
                src[i] = 1
            elif i == length - 1:
                src[i] = i - 1
            else:
                src[i] = i + choice([-1, 1], p=[0.5, 0.5])
    return src
