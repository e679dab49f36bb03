"""``VectorQuantizer`` -- nearest-codebook lookup with straight-through gradient.

Reference: vq_vae/vector_quantizer.py:8-58.  Rows are the D-float chunks of the (B,D,L) buffer in memory
order (no permute, :32).  The HIP path keeps the codebook as an index problem: argmin -> gather -> loss /
histogram; the dense one-hot ``encodings`` (N,K) the reference builds (:39-40) is only materialised for the
value this method returns, never used for the matmul (:43 is a gather: SURVEY App. A.4).
"""
import torch
import torch.nn as nn

from .. import _native, _ops

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class VectorQuantizer(nn.Module):
    def __init__(self, num_embeddings, embedding_dim, commitment_cost, flag_flatten=True):
        super().__init__()
        self._embedding_dim = embedding_dim
        self._num_embeddings = num_embeddings
        self._embedding = nn.Embedding(num_embeddings, embedding_dim)
        self._embedding.weight.data.uniform_(-1 / num_embeddings, 1 / num_embeddings)
        self._commitment_cost = commitment_cost
        self._train_vq = True
        self._flag_flatten = flag_flatten   # stored, never read -- as in the reference (:21)

    def get_embedding_dim(self):
        return self._embedding_dim

    def set_train_vq(self, train_vq):
        self._train_vq = train_vq

    def quantize(self, inputs):
        """(loss, quantized_st, perplexity, indices[N] int64) without the dense one-hot."""
        _ops._need_gpu(inputs, "VectorQuantizer")
        if inputs.numel() % self._embedding_dim != 0:
            raise RuntimeError("shape '[-1, %d]' is invalid for input of size %d" % (self._embedding_dim, inputs.numel()))
        return _ops.VQFn.apply(inputs, self._embedding.weight, float(self._commitment_cost), bool(self._train_vq))

    def forward(self, inputs):
        loss, quantized, perplexity, idx = self.quantize(inputs)
        encodings = _native.onehot(idx, self._num_embeddings)
        return loss, quantized, perplexity, encodings
