"""How often can a mode's codebook index differ from the reference's?  A row flips when the mode's error in the difference of
its two smallest distances exceeds the reference's own gap between them.  This tool measures, on the B = 64 speech golden
(32 000 rows), per mode: e = [(d(i2) - d(i1)) in the mode - the same in fp64 from the f32-mode latent] / d(i1) per row, with
(i1, i2) the two nearest codes, and combines mean|e| with the golden's density of near-ties rho (P(gap < g) ~ rho * g, counted
from the reference's stored top-2 distances) into the expected flips per million rows, rho * mean|e| / 2.
    python tests/analysis/near_ties.py [modes...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from acoustic_locating_vq_vae import _ops
from g3_cases import O, _build


def main():
    modes = sys.argv[1:] or ["f32", "x3mx_hb", "f16mx_hb", "bf16"]
    g = np.load(os.path.join(ROOT, "tests", "golden", "g3_speech_b64.npz"))
    cfg = (201, 1024, 128, 3, 1024, 0.25, 1024)
    p = O.closed_form_params(O.vqvae_param_shapes(201, 1024, 128, 1024, 1024), float(g["cb_scale"]), float(g["gain"]))
    x = O.speech_preprocess(torch.from_numpy(O.hashed_uniform(64 * 201 * 500, 21, 2.0).reshape(64, 201, 500))).cuda()
    cb = p["_vq._embedding.weight"].double().cuda()
    gap = (g["top2_val"][:, 1] - g["top2_val"][:, 0]) / np.abs(g["top2_val"][:, 0])
    n = gap.size
    rho = {t: float((gap < t).sum()) / n / t for t in (3e-5, 1e-4, 3e-4)}
    print("reference near-tie density: P(gap < g) / g = %s  (rows %d, smallest gap %.2e)"
          % ({k: round(v, 1) for k, v in rho.items()}, n, gap.min()))
    zs = {}
    for mode in ["f32"] + [m for m in modes if m != "f32"]:
        _ops.set_compute_dtype(mode, internal=True)
        m = _build(cfg, p).eval()
        with torch.no_grad():
            zs[mode] = m._latent(x).double().view(-1, 128)       # rows in memory order, as the quantiser slices them
    _ops.set_compute_dtype("f32")

    def dist(z):
        return (z * z).sum(1, keepdim=True) + (cb * cb).sum(1)[None, :] - 2.0 * z @ cb.t()
    d0 = dist(zs["f32"])
    v, i = torch.topk(d0, 2, dim=1, largest=False)
    ref_diff = (v[:, 1] - v[:, 0])
    for mode in modes:
        d = dist(zs[mode])
        diff = d.gather(1, i[:, 1:2]).view(-1) - d.gather(1, i[:, 0:1]).view(-1)
        e = ((diff - ref_diff) / v[:, 0].abs()).cpu().numpy()
        zerr = float((zs[mode] - zs["f32"]).abs().max() / zs["f32"].abs().max())
        print("%-8s z error vs the f32 mode %.2e   mean|e| %.2e  rms %.2e  max %.2e   expected flips per million rows %.1f"
              % (mode, zerr, np.abs(e).mean(), np.sqrt((e * e).mean()), np.abs(e).max(), 1e6 * rho[1e-4] * np.abs(e).mean() / 2))


if __name__ == "__main__":
    main()
