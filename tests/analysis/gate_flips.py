"""Why the split modes' GRADIENTS differ from fp32 by 1e-2 in max-norm while their forward agrees to 2e-5.

Every ReLU of the model is a gate on the backward path.  Forward noise of ~1e-5 relative flips the gates whose
pre-activation lies within that noise of zero; a flipped gate switches ONE of the B*L terms of every weight-gradient
element it feeds fully on or off.  This tool measures, at the speech config with the goldens' closed-form weights:

  * how many of the saved post-ReLU activations differ in sign pattern between the f32 mode and a split mode,
    and how large the values at the flipped positions are (they must be at the noise level);
  * the gradient error of the split mode against the f32 mode, per batch size -- a flipped gate is one term in B*L,
    so the max-norm error must fall roughly as 1/sqrt(B) while the flip RATE stays constant.

    python3 tests/analysis/gate_flips.py [modes ...]      (GPU box; prints one JSON line per (mode, batch))
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from acoustic_locating_vq_vae import _native as N  # noqa: E402
from acoustic_locating_vq_vae import _ops  # noqa: E402
from oracle import vqvae_oracle as O  # noqa: E402  (weight / input generators only)
import g3_cases  # noqa: E402

CFG = (201, 1024, 128, 3, 1024, 0.25, 1024)


def dense(a):
    return a if torch.is_tensor(a) else a.to_ncl()


def one(mode, B, p):
    _ops.set_compute_dtype(mode, internal=True)
    m = g3_cases._build(CFG, p).train()
    x = O.speech_preprocess(torch.from_numpy(O.hashed_uniform(B * 201 * 500, 21, 2.0).reshape(B, 201, 500))).cuda()
    np.random.seed(9)
    _ops._ACT_TAP = tap = []
    vq_loss, recon, _ = m(x)
    _ops._ACT_TAP = None
    (F.mse_loss(recon, x) + vq_loss).backward()
    _, _, _, idx = m.get_latent_indices(x)
    masks = [[(dense(a) > 0) for a in acts[1:]] for _, acts in tap]            # acts[0] is the node's input (not a gate)
    vals = [[dense(a) for a in acts[1:]] for _, acts in tap]
    grads = {k: q.grad.detach().clone() for k, q in m.named_parameters()}
    return masks, vals, grads, idx, recon.detach()


def main():
    modes = sys.argv[1:] or ["x3mx_hb", "f16mx_hb", "bf16x3_hb"]
    p = O.closed_form_params(O.vqvae_param_shapes(201, 1024, 128, 1024, 1024), 1.0, 0.5)
    for B in (2, 8, 32):
        ref_masks, ref_vals, ref_grads, ref_idx, ref_recon = one("f32", B, p)
        for mode in modes:
            masks, vals, grads, idx, recon = one(mode, B, p)
            flips = gates = 0
            worst_val = 0.0
            typical = []
            for node_r, node_m, node_vr, node_vm in zip(ref_masks, masks, ref_vals, vals):
                for mr, mm, vr, vm in zip(node_r, node_m, node_vr, node_vm):
                    d = mr != mm
                    flips += int(d.sum())
                    gates += d.numel()
                    if bool(d.any()):
                        worst_val = max(worst_val, float(torch.maximum(vr, vm)[d].max() / vr.abs().mean()))
            gmax, gl2 = {}, {}
            for k in grads:
                a, b = grads[k].double(), ref_grads[k].double()
                gmax[k] = float((a - b).abs().max() / b.abs().max())
                gl2[k] = float((a - b).norm() / b.norm())
            kw = max(gmax, key=gmax.get)
            print(json.dumps({"mode": mode, "B": B, "rows_B_times_L": B * 500, "gates": gates, "flipped_gates": flips,
                              "flip_rate": flips / gates, "largest_flipped_value_over_mean_activation": worst_val,
                              "idx_mismatches_vs_f32": int((idx != ref_idx).sum()),
                              "recon_rel_max": float((recon - ref_recon).abs().max() / ref_recon.abs().max()),
                              "grad_rel_max_worst": gmax[kw], "grad_rel_max_worst_key": kw,
                              "grad_rel_max_median": float(np.median(list(gmax.values()))),
                              "grad_rel_l2_worst": max(gl2.values()), "grad_rel_l2_median": float(np.median(list(gl2.values())))}),
                  flush=True)
    _ops.set_compute_dtype("f32")


if __name__ == "__main__":
    main()
