"""Every user-selectable compute mode at the DEFAULT configs (speech / RIR / echoed ctor sizes) against the goldens made by
the real reference (tests/golden/g3_*.npz).  The f32 mode's strict version lives in tests/test_modules_gpu.py; here the
throughput mode (bf16) is held to measured floors and ceilings and the parity modes to the north star's bar itself; every
number is printed, and no assertion is conditional on another one passing.

Round 4: the DEFAULT mode x3mx_hb (bf16x3 forward for everything the codebook indices depend on, f16mx decoder forward,
16-bit backward) must return EVERY index of EVERY golden -- == 0, no allowance -- including the RIR golden's 1.8e-6
near-tie that the f16mx family flips; the allowance survives, named, for f16mx_hb only.  The retired engines f16mx / bf16x3
(their forwards are the _hb modes' bit for bit: tests/test_f16mx_hb_gpu.py, tests/test_bf16x3_hb_gpu.py) left the golden matrices.

Measured on MI355X (B=2 goldens): x3mx_hb   z 6.4e-6, recon 2.2e-5, 0 code mismatches, grads max-norm 4e-2 (rel-L2 median 1.6e-3);
                                  bf16x3_hb z 6e-6, recon 7e-6, 0 code mismatches;
                                  f16mx_hb  z 1.3e-5 - 1.8e-5, recon 2.2e-5, 0 code mismatches, grads <= 7e-3 rel-L2;
                                  bf16      z 3-4e-3, 99.0-99.1 % of codes agree (every flip a reference near-tie with
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
@pytest.mark.parametrize("mode", ["x3mx_hb", "bf16x3_hb", "f16mx_hb"], indirect=True)
def test_split_modes_default_configs_against_reference_golden(mode, tag, golden_dir):
    """The north star's bar, unconditionally: codebook indices BIT-EXACT (0 of 1000 / 402 differ -- the goldens' smallest
    relative top-2 gap is 3e-5, so there is no near-tie to excuse), outputs within 1e-3 (measured 2e-5) on 4096-element
    slices AND on the fp64 checksums of the whole tensors (measured 1e-6), losses to 1e-5, gradients of EVERY parameter
    (encoder side included) compared on 2048-element slices and checksums.
    A regression to a single flipped code fails this test."""
    r = run(tag, golden_dir)
    print("g3-%s %s: %s" % (tag, mode, json.dumps(r)))
    if tag != "echoed":
        assert r["idx_mismatches"] == 0, r
        assert r["slice_elems"] >= 4096
        assert r["z_rel_max"] < 1e-4 and r["z_sum_rel"] < 1e-5, r
        assert r["vq_loss_rel"] < 1e-5 and r["perplexity_rel"] < 1e-5, r
    assert r["recon_error_rel"] < 1e-5, r
    assert r["recon_rel_max"] < 1e-3 and r["recon_sum_rel"] < 1e-5, r       # north-star tolerance (measured 2e-5 / 1e-6)
    # Gradients (outside the north star's wording; stated separately in DESIGN section 6).  The ~1e-5 forward noise flips
    # a few dozen of the ~1e7 ReLU gates (pre-activations within rounding of zero) and a flipped gate is a full-size
    # error in ONE of the B * L = 1000 (402) terms of a weight-gradient element: ~1/sqrt(1000) of that element at these
    # B = 2 goldens, shrinking with the batch (tests/analysis/gate_flips.py measures both).  Measured on the 2048-element slices:
    # max-norm 1.5e-2 ... 6e-2, per-tensor rel-L2 <= 8e-3 (median 3e-4 ... 2.6e-3), whole-tensor checksums <= 1.4e-3.
    # Strict gradient parity (1e-4 max-norm here) is the f32 mode's, tests/test_modules_gpu.py.
    assert r["grad_rel_max"] < 0.1 and r["encoder_grad_rel_max"] < 0.1, r
    assert r["grad_rel_l2_max"] < 1.5e-2 and r["grad_rel_l2_median"] < 5e-3 and r["grad_sum_rel_max"] < 3e-3, r
    if tag == "echoed":
        assert r["encoders_grad_free"]


@pytest.mark.parametrize("mode", ["f32", "x3mx_hb", "bf16x3_hb", "f16mx_hb", "bf16"], indirect=True)
def test_speech_config_at_a_training_batch_against_reference_golden(mode, golden_dir):
    """G3-speech at B = 16 (round 3; made by the real reference): 8 000 codebook rows, 22 of them with a relative top-2
    distance gap below 1e-4 and the smallest at 6.9e-6 -- every parity mode must still return ALL 8 000 indices of the
    reference -- and gradients where a flipped ReLU gate is one of 8 000 terms instead of one of 1 000: the split modes'
    max-norm error falls from 4e-2 (B = 2) to 1e-2, as tests/analysis/gate_flips.py's 1/sqrt(B) says.
    Measured: f32 0 / 8000, z 1.6e-6, grads 9.8e-4; bf16x3 0, 7e-6, 2.7e-2 (rel-L2 median 1.4e-4); f16mx = f16mx_hb 0, 1.5e-5,
    recon 2.3e-5, grads 1.0e-2 (median 1.6e-4 / 5.9e-4); f16mx_hd recon 6.7e-4; bf16 68 differ (99.15 %)."""
    r = run("speech_b16", golden_dir)
    print("g3-speech_b16 %s: %s" % (mode, json.dumps(r)))
    assert r["idx_total"] == 8000 and r["slice_elems"] >= 4096
    if mode == "bf16":
        assert r["idx_agree"] >= 0.985 and r["mismatch_gap_max"] < 5e-3, r
        assert r["z_rel_l2"] < 1.5e-2 and r["recon_error_rel"] < 1e-2 and r["grad_rel_l2_median"] < 0.3, r
        return
    assert r["idx_mismatches"] == 0, r
    assert r["z_rel_max"] < 1e-4 and r["z_sum_rel"] < 1e-5 and r["vq_loss_rel"] < 1e-5 and r["perplexity_rel"] < 1e-5, r
    assert r["recon_error_rel"] < 1e-5, r
    assert r["recon_rel_max"] < 1e-4 and r["recon_sum_rel"] < 1e-5, r
    if mode == "f32":
        assert r["grad_rel_max"] < 3e-3 and r["grad_rel_l2_max"] < 1e-3 and r["grad_sum_rel_max"] < 1e-4, r
    else:
        # bf16 operands in a backward (bf16x3_hb everywhere, x3mx_hb on the encoder side: 2^-9) set a floor of ~2e-3 under
        # those tensors' rel-L2
        b16bwd = mode in ("bf16x3_hb", "x3mx_hb")
        assert r["grad_rel_max"] < 5e-2 and r["grad_rel_l2_max"] < 3e-2 and r["grad_rel_l2_median"] < (5e-3 if b16bwd else 2e-3), r
        assert r["grad_sum_rel_max"] < (3e-3 if b16bwd else 5e-4) and r["encoder_grad_rel_max"] < 2e-2, r


@pytest.mark.parametrize("mode", ["f32", "x3mx_hb", "bf16x3_hb", "f16mx_hb", "bf16"], indirect=True)
def test_the_bench_workload_against_reference_golden(mode, golden_dir):
    """G3-speech at B = 64: BASELINE configs[1] itself -- the batch bench.py times -- run by the real reference on the CPU
    (tests/golden/g3_speech_b64.npz: 32 000 codebook rows, 92 of them with a relative top-2 gap below 1e-4, the smallest
    4.2e-6).  Every parity mode must return ALL 32 000 reference indices, outputs at the north star's bar, and gradients at
    the level a training batch gives (a flipped ReLU gate is one of 32 000 terms).
    Measured: f32 0 / 32000, z 1.5e-6, grads max-norm 1.7e-3; bf16x3 0, 6.6e-6, 6.2e-3; f16mx = f16mx_hb 0, 1.8e-5, recon
    1.9e-5, grads 5.0e-3 / 4.4e-3 (rel-L2 median 1.3e-4 / 3.6e-4); bf16x3_hb = bf16x3's forward, grads 6.3e-3 (median 1.9e-3: bf16
    operands in the backward); f16mx_hd recon 6.2e-4, grads 2.3e-2; bf16 264 differ."""
    r = run("speech_b64", golden_dir)
    print("g3-speech_b64 %s: %s" % (mode, json.dumps(r)))
    assert r["idx_total"] == 32000 and r["slice_elems"] >= 4096
    if mode == "bf16":
        assert r["idx_agree"] >= 0.985 and r["mismatch_gap_max"] < 5e-3, r
        assert r["z_rel_l2"] < 1.5e-2 and r["recon_error_rel"] < 1e-2 and r["grad_rel_l2_median"] < 0.2, r
        return
    assert r["idx_mismatches"] == 0, r
    assert r["z_rel_max"] < 1e-4 and r["z_sum_rel"] < 1e-5 and r["vq_loss_rel"] < 1e-5 and r["perplexity_rel"] < 1e-5, r
    assert r["recon_error_rel"] < 1e-5, r
    assert r["recon_rel_max"] < 1e-4 and r["recon_sum_rel"] < 1e-5, r
    if mode == "f32":
        assert r["grad_rel_max"] < 5e-3 and r["grad_rel_l2_max"] < 1e-3 and r["grad_sum_rel_max"] < 1e-4, r
    else:
        b16bwd = mode in ("bf16x3_hb", "x3mx_hb")
        assert r["grad_rel_max"] < 2e-2 and r["grad_rel_l2_max"] < 1.5e-2 and r["grad_rel_l2_median"] < (5e-3 if b16bwd else 1e-3), r
        assert r["grad_sum_rel_max"] < (3e-3 if b16bwd else 5e-4) and r["encoder_grad_rel_max"] < 1e-2, r


F16MX_FAMILY_NEAR_TIE_ALLOWANCE = 1      # f16mx_hb ONLY: the RIR golden's 1.8e-6 reference near-tie (see the docstring below)


@pytest.mark.parametrize("tag", ["rir_b32", "echoed_b32"])
@pytest.mark.parametrize("mode", ["f32", "x3mx_hb", "bf16x3_hb", "f16mx_hb", "bf16"], indirect=True)
def test_rir_and_echoed_at_their_per_gpu_batch_against_reference_golden(mode, tag, golden_dir):
    """BASELINE configs[2] / [4] at their per-GPU share (B = 32), run by the real reference.  The RIR golden holds 6 432
    codebook rows of which TWO are reference near-ties of 1.3e-6 and 1.8e-6 relative (11 and 15 fp32 ulps between the two
    nearest codes).  The DEFAULT mode x3mx_hb -- like f32 and bf16x3_hb -- must return ALL 6 432 reference indices, the
    reconstruction inside 1e-4 and the echoed model's (which embeds this RIR encoder) likewise: == 0, no allowance.
    f16mx_hb (z within 1.5e-5) returns 6 431: it flips the 1.8e-6 row -- the ONE flipped index in the 47 834 rows of all the
    goldens -- and a flipped code is a different decoder input at one position (reconstruction rel-max 6e-2 over ~11
    positions, rel-L2 7e-3; the echoed model inherits it).  That is the resolution limit of 3-bit-mantissa cross terms and
    the reason f16mx_hb stopped being the default in round 4; its bars below are that golden's measured behaviour, named."""
    r = run(tag, golden_dir)
    print("g3-%s %s: %s" % (tag, mode, json.dumps(r)))
    fx = mode == "f16mx_hb"
    if tag == "rir_b32":
        assert r["idx_total"] == 6432
        if mode == "bf16":
            assert r["idx_agree"] >= 0.985 and r["mismatch_gap_max"] < 5e-3 and r["z_rel_l2"] < 1.5e-2, r
        else:
            assert r["idx_mismatches"] <= (F16MX_FAMILY_NEAR_TIE_ALLOWANCE if fx else 0) and r["mismatch_gap_max"] < 5e-6, r
            assert r["z_rel_max"] < 1e-4 and r["z_sum_rel"] < 1e-5 and r["vq_loss_rel"] < 1e-5, r
    if mode == "bf16":
        assert r["recon_error_rel"] < 1e-2 and r["recon_rel_l2"] < 0.25 and r["grad_rel_l2_median"] < 0.3, r
        return
    assert r["recon_error_rel"] < 5e-5, r
    if fx:      # one flipped code: local reconstruction difference, gradients of the layers next to it
        assert r["recon_rel_l2"] < 2e-2 and r["recon_sum_rel"] < 2e-4, r
        assert r["grad_rel_max"] < 0.15 and r["grad_rel_l2_median"] < 3e-2 and r["grad_sum_rel_max"] < 1e-2, r
    else:
        assert r["recon_rel_max"] < 1e-4 and r["recon_sum_rel"] < 1e-5, r
        assert r["grad_rel_max"] < (5e-3 if mode == "f32" else 3e-2) and r["grad_rel_l2_max"] < (3e-3 if mode == "f32" else 1.5e-2), r
    if tag == "echoed_b32":
        assert r["encoders_grad_free"]


@pytest.fixture
def wide_min_tiles(request):
    from acoustic_locating_vq_vae import _native as N
    prev = N.set_option("wide_min_tiles", request.param)
    yield request.param
    N.set_option("wide_min_tiles", prev)


@pytest.mark.parametrize("tag", ["rir", "echoed", "speech"])
@pytest.mark.parametrize("wide_min_tiles", [192, 1], ids=["default_dispatch", "wide_forced"], indirect=True)
@pytest.mark.parametrize("mode", ["bf16"], indirect=True)
def test_bf16_production_dispatch(mode, wide_min_tiles, tag, golden_dir):
    """The bf16 dispatch a USER gets (option wide_min_tiles = 192: at the goldens' B = 2 every 1024-channel layer has
    fewer than 192 tiles of 256 x 256 and runs the 128 x 128 kernel with its skip / mask / sign-bit epilogues) and the one
    the rest of the test session forces (1: the 256 x 256 kernels), both against the reference goldens at the same bars.
    Round-2 verdict: under pytest the production path of configs[2] / [4] in bf16 was never compared with a golden."""
    from acoustic_locating_vq_vae import _native as N
    assert N.get_option("wide_min_tiles") == wide_min_tiles
    r = run(tag, golden_dir)
    print("g3-%s bf16 wide_min_tiles=%d: %s" % (tag, wide_min_tiles, json.dumps(r)))
    if tag != "echoed":
        assert r["idx_agree"] >= 0.97 and r["mismatch_gap_max"] < 5e-3, r
        assert r["z_rel_l2"] < 1.5e-2 and r["vq_loss_rel"] < 2e-3, r
    assert r["recon_error_rel"] < 1e-2 and r["recon_rel_l2"] < 0.25, r
    assert r["grad_rel_l2_median"] < 0.6, r
