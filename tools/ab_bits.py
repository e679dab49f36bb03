"""Bitwise fingerprint of the f16mx / bf16 / bf16x3 kernels' outputs on a fixed set of deterministic cases: run once per
library build (ALVQ_LIB=...), diff the JSON lines.  A kernel rewrite that claims "bit-identical" must leave every hash
unchanged.     ALVQ_LIB=$PWD/acoustic_locating_vq-vae_amd/lib/libalvq_base.so python3 tools/ab_bits.py > a.json
               python3 tools/ab_bits.py > b.json && diff a.json b.json"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from acoustic_locating_vq_vae import _native as N  # noqa: E402


def digest(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        if isinstance(t, N.NLC):     # the defined parts only: the planes' matrices and, if valid, the sign bits (guard rows hold garbage)
            for pl in range(t.planes):
                h.update(t.matrix(pl).contiguous().view(torch.uint8).cpu().numpy().tobytes())
            if t.has_bits:
                off = (t.bits_ptr - t.storage.data_ptr())
                h.update(t.storage.view(torch.uint8)[off:off + t.rows * t.Cp // 8].cpu().numpy().tobytes())
            continue
        h.update(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def main():
    modes = sys.argv[1:] or ["f16mx", "bf16", "bf16x3"]
    out = {}
    shapes = [(2, 7, 16, 13, 3), (3, 72, 136, 95, 1), (2, 201, 1024, 500, 3), (2, 1024, 128, 500, 3), (2, 1024, 1024, 201, 1),
              (5, 130, 130, 129, 3), (4, 1024, 1024, 500, 1), (4, 1024, 1024, 500, 3), (3, 1024, 201, 500, 3), (2, 500, 1024, 201, 3)]
    for mode in modes:
        planes, wpl = {"f16mx": (2, 3), "bf16": (1, 1), "bf16x3": (2, 2)}[mode]
        fmt = "f16mx" if mode == "f16mx" else None
        for (B, C, M, L, KW) in shapes:
            g = torch.Generator(device="cuda").manual_seed(B * 1000 + C + M + L + KW)
            x = torch.randn(B, C, L, device="cuda", generator=g)
            # small / huge magnitudes ride along: denormal fp16 halves, saturation
            x[0, 0, :4] = torch.tensor([3e-8, -2e-7, 7e4, 1e-5], device="cuda")[:min(4, L)]
            w = torch.randn(M, C, KW, device="cuda", generator=g) / (C * KW) ** 0.5
            b = torch.randn(M, device="cuda", generator=g)
            s1, s2, post = (torch.randn(B, M, L, device="cuda", generator=g) for _ in range(3))
            mk = torch.randn(B, M, L, device="cuda", generator=g)
            dy = torch.randn(B, M, L, device="cuda", generator=g)
            enter = (lambda t, gs=None: N.ncl_to_nlc(t, planes, fmt, gs)) if fmt else (lambda t, gs=None: N.ncl_to_nlc(t, planes))
            xn = enter(x)
            key = "%s:%s" % (mode, (B, C, M, L, KW))
            res = {"enter": digest(xn), "leave": digest(N.nlc_to_ncl(xn))}
            pk = N.pack_weight(w, N.W_OIK, wpl)
            pki = N.pack_weight(w, N.W_IOK, wpl)
            res["pack"] = digest(pk[0], pki[0])
            t = N.conv1d_bf16(xn, pk, b, relu=True)                                   # bias + relu (+ sign bits)
            res["relu"] = digest(t)
            y, y2 = N.conv1d_bf16(xn, pk, b, enter(s1), enter(s2), enter(mk), enter(post), relu=True)
            res["all"] = digest(y, y2)
            res["skip_relu"] = digest(N.conv1d_bf16(xn, pk, None, enter(s1), relu=True))
            res["ncl"] = digest(N.conv1d_bf16(xn, pk, b, out_ncl=True))
            gs = N.grad_scale(dy) if fmt else None
            dyn = enter(dy, gs)
            res["dgrad_maskbits"] = digest(N.conv1d_bf16(dyn, N.pack_weight(w, N.W_IOK, wpl), mask=enter(torch.randn(B, C, L, device="cuda", generator=g)) if mode == "bf16x3" else xn_relu(xn, N, mode, B, C, L, g, enter)))
            res["dgrad_skip_mask"] = digest(N.conv1d_bf16(dyn, N.pack_weight(w, N.W_IOK, wpl), skip1=enter(torch.randn(B, C, L, device="cuda", generator=g), gs), mask=enter(torch.randn(B, C, L, device="cuda", generator=g))))
            dw, db = N.conv1d_wgrad_bf16(dyn, xn, KW, N.W_OIK, want_bias=True)
            res["wgrad"] = digest(dw, db)
            res["wgrad_multi3"] = digest(N.conv1d_wgrad_bf16_multi([(dyn, xn)] * 3, KW, N.W_OIK))
            res["relu_mask"] = digest(N.relu_mask_bf16(dyn, enter(mk)))
            out[key] = res
    for k in sorted(out):
        print(json.dumps({k: out[k]}, sort_keys=True))


def xn_relu(xn, N, mode, B, C, L, g, enter):
    """A ReLU'd tensor of the input's shape that carries sign bits (the mask operand of a data-gradient launch)."""
    t = enter(torch.randn(B, C, L, device="cuda", generator=g))
    out = N.relu_mask_bf16(t, t)        # relu via mask; no sign bits here
    pk1 = N.pack_weight(torch.eye(C, device="cuda").view(C, C, 1).contiguous(), N.W_OIK, {"f16mx": 3, "bf16": 1, "bf16x3": 2}[mode])
    return N.conv1d_bf16(out, pk1, relu=True)   # identity conv with ReLU: leaves the sign bits behind


if __name__ == "__main__":
    main()
