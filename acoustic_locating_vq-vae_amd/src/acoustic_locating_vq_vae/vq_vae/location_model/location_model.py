"""``LocationModule`` -- the 5-layer MLP that regresses the source angle from the RIR encoder's codes.

Reference: vq_vae/location_model/location_model.py:5-29; caller scripts/train_location.py:69-77, which feeds it the
dense one-hot ``encodings.reshape(B, 201, 1024)`` of ``get_latent_representation``.  Same constructor, attributes,
``state_dict`` keys and pickled layout.  What changes is how ``fc_1`` -- Linear(201*1024 -> 1024), 211 M of the model's
212 M parameters -- is evaluated: on one-hot input ``fc_1(flatten(x))`` is the sum of 201 selected weight columns per
sample, so it runs as an embedding-bag gather over the code indices (``alvq_embedding_bag_fwd_f32``; backward = the
matching scatter-add) instead of a (B x 205 824) x (205 824 x 1024) GEMM against a 99.9 %-zero operand.

``forward`` accepts
  * the dense one-hot float tensor ``(B, L, K)`` exactly as the reference is called -- verified on the device to be
    one-hot (one flag read back; the caller's loop syncs every step anyway, train_location.py:95), and evaluated by
    the plain dense product when it is not, so arbitrary inputs keep the reference's semantics;
  * the indices themselves, ``(B, L)`` int32/int64 (``get_latent_indices(...)[3].view(B, L)``): no one-hot is ever
    materialised.
The four small layers behind it (1024-512-512-64-out) are plain library GEMMs through ``nn.Linear``.
"""
import torch
from torch import nn
from torch.nn import functional as F

from ... import _native as N
from ... import _ops


class _EmbeddingBagLinearFn(torch.autograd.Function):
    """fc_1 on one-hot rows: out = bias + sum_l W[:, l*K + idx[b, l]]."""

    @staticmethod
    def forward(ctx, idx, weight, bias, L, K, flag=None):
        ctx.save_for_backward(idx)
        ctx.L, ctx.K, ctx.has_bias = L, K, bias is not None
        ctx.weight_ref, ctx.bias_ref = weight, bias               # looked up in the gradient sinks by the backward
        return N.embedding_bag_fwd(weight, bias, idx, L, K, flag)

    @staticmethod
    def backward(ctx, dz):
        (idx,) = ctx.saved_tensors
        dW = db = None
        if ctx.needs_input_grad[1]:
            # inside a train_step.LocationTrainer step the scatter-add goes straight into fc_1's slice of the (already
            # zeroed) flat gradient buffer: no 843 MB temporary, no dense "grad += dW" pass by autograd
            sink_w, sink_b = _ops._sink(ctx.weight_ref), (_ops._sink(ctx.bias_ref) if ctx.has_bias else None)
            if sink_w is not None and (not ctx.has_bias or sink_b is not None):
                N.embedding_bag_bwd(dz.contiguous(), idx, ctx.L, ctx.K, want_bias=ctx.has_bias, dW_out=sink_w, db_out=sink_b)
                return None, None, None, None, None, None
            dW, db = N.embedding_bag_bwd(dz.contiguous(), idx, ctx.L, ctx.K,
                                         want_bias=ctx.has_bias and ctx.needs_input_grad[2])
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = dz.sum(dim=0)
        return None, dW, db, None, None, None


class LocationModule(nn.Module):

    def __init__(self, encoder_output_dim: int, num_hiddens: int, output_dim: int):
        super(LocationModule, self).__init__()
        self.encoder_output_dim = encoder_output_dim
        self.fc_1 = nn.Linear(encoder_output_dim * num_hiddens, 1024)
        self.relu1 = nn.ReLU()
        self.fc_2 = nn.Linear(1024, 512)
        self.relu2 = nn.ReLU()
        self.fc_3 = nn.Linear(512, 512)
        self.relu3 = nn.ReLU()
        self.fc_4 = nn.Linear(512, 64)
        self.relu4 = nn.ReLU()
        self.fc_5 = nn.Linear(64, output_dim)

    def _fc_1(self, x):
        L = self.encoder_output_dim
        K = self.fc_1.in_features // L
        w, b = self.fc_1.weight, self.fc_1.bias
        if not torch.is_floating_point(x):                         # (B, L) code indices
            if x.dim() != 2 or x.shape[1] != L:
                raise RuntimeError("LocationModule: an index input must be (B, %d), got %s" % (L, tuple(x.shape)))
            # caller-supplied indices are range-checked on the device (the kernels skip and flag an index outside
            # [0, K) instead of reading / writing out of bounds; a 64-bit value is checked BEFORE it is narrowed) and the
            # flag is read back here -- one sync, like the one-hot path's -- so that bad input raises as
            # torch.nn.functional.embedding_bag does
            flag = N.device_flag(x.device)
            if x.dtype == torch.int64:
                idx = N.indices_to_i32(x.contiguous(), K, flag)
            elif x.dtype == torch.int32:
                idx = x.contiguous()
            else:
                raise RuntimeError("LocationModule: index input must be int32 or int64, got %s" % x.dtype)
            out = _EmbeddingBagLinearFn.apply(idx, w, b, L, K, flag)
            if int(flag.item()) != 0:
                raise IndexError("LocationModule: a code index lies outside [0, %d)" % K)
            return out
        flat = torch.flatten(x, start_dim=1)
        if flat.shape[1] != L * K:
            return self.fc_1(flat)                                 # raises the reference's own shape error
        if not x.requires_grad:
            idx, flag = N.onehot_to_index(_ops.dense(flat).view(-1, K))
            if int(flag.item()) == 0:                              # every row exactly one-hot: the sparse evaluation
                return _EmbeddingBagLinearFn.apply(idx.view(-1, L), w, b, L, K)
        return F.linear(flat, w, b)                                # anything else: the dense product, as the reference

    def forward(self, x):
        _ops._need_gpu(x, "LocationModule")
        z = self._fc_1(x)
        z = self.relu1(z)
        z = self.fc_2(z)
        z = self.relu2(z)
        z = self.fc_3(z)
        z = self.relu3(z)
        z = self.fc_4(z)
        z = self.relu4(z)
        return self.fc_5(z)
