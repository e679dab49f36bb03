"""Every compute mode at the DEFAULT configs (speech / RIR / echoed ctor sizes) against the goldens made by the real
reference (tests/golden/g3_*.npz).  The f32 mode's strict version lives in tests/test_modules_gpu.py; here the
throughput mode (bf16 -- what bench.py's headline runs) and the split parity mode are held to measured floors and
ceilings, every number is printed, and no assertion is conditional on another one passing.

Measured on MI355X (B=2 goldens): bf16x3  z 6e-6, recon 7e-6, 0 code mismatches, grads <= 7e-3;
                                  f16mx   z 1.3e-5 - 1.8e-5, recon 2.2e-5, 0 code mismatches, grads <= 7e-3;
                                  bf16    z 3-4e-3, 99.0-99.1 % of codes agree (every flip a reference near-tie with
                                          relative top-2 gap < 1e-3), recon rel-L2 0.05-0.11 (flipped codes), losses 1e-4."""
import json

import pytest

pytestmark = pytest.mark.gpu

from g3_cases import run  # noqa: E402  (tests/ is on sys.path under pytest's rootdir/conftest)


@pytest.fixture
def mode(request):
    from acoustic_locating_vq_vae import _ops
    _ops.set_compute_dtype(request.param)
    yield request.param
    _ops.set_compute_dtype("f32")


@pytest.mark.parametrize("tag", ["speech", "rir", "echoed"])
@pytest.mark.parametrize("mode", ["bf16"], indirect=True)
def test_bf16_default_configs_against_reference_golden(mode, tag, golden_dir):
    r = run(tag, golden_dir)
    print("g3-%s %s: %s" % (tag, mode, json.dumps(r)))
    if tag != "echoed":
        assert r["idx_agree"] >= 0.97, r                  # floor on codebook-index agreement
        assert r["mismatch_gap_max"] < 5e-3, r            # only codes whose top-2 reference distances nearly tie flip
        assert r["z_rel_l2"] < 1.5e-2 and r["z_rel_max"] < 1.5e-2, r
        assert r["vq_loss_rel"] < 2e-3, r
    assert r["recon_error_rel"] < 1e-2, r                 # the loss the loop optimises
    assert r["recon_rel_l2"] < 0.25, r                    # local differences where a code flipped
    assert r["grad_rel_l2_median"] < 0.6, r               # bf16 storage flips ~0.3 % of ReLU gates per layer


@pytest.mark.parametrize("tag", ["speech", "rir", "echoed"])
@pytest.mark.parametrize("mode", ["bf16x3", "f16mx"], indirect=True)
def test_split_modes_default_configs_against_reference_golden(mode, tag, golden_dir):
    r = run(tag, golden_dir)
    print("g3-%s %s: %s" % (tag, mode, json.dumps(r)))
    flips = 0
    if tag != "echoed":
        flips = r["idx_mismatches"]
        assert flips <= 1 and r["mismatch_gap_max"] < 1e-4, r      # indices bit-exact up to one reference near-tie
        assert r["z_rel_max"] < 1e-4, r
        assert r["vq_loss_rel"] < (1e-4 if flips == 0 else 5e-3), r
    assert r["recon_error_rel"] < (1e-4 if flips == 0 else 5e-3), r
    assert r["recon_rel_max"] < (1e-3 if flips == 0 else 2e-1), r   # north-star tolerance when the codes agree
    assert r["grad_rel_max"] < (1.5e-2 if flips == 0 else 5e-2), r
    if tag == "echoed":
        assert r["encoders_grad_free"]
