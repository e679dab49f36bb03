"""Host-side logic that needs no GPU: module surface, state_dict keys, aliasing, pickling, jitter stream,
and the loud failure of the HIP path on CPU tensors."""
import io
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import vqvae_oracle as O


def make(cfg=(7, 16, 4, 2, 8, 0.25, 16), **kw):
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    return ConvolutionalVQVAE(*cfg, **kw)


def test_state_dict_keys_and_shapes_match_reference_layout():
    m = make((201, 32, 8, 3, 16, 0.25, 64))
    sd = m.state_dict()
    expect = O.vqvae_param_shapes(201, 32, 8, 16, 64)
    for k, shape in expect.items():
        for r in range(3):
            kk = k.replace("_layers.0.", "_layers.%d." % r)
            assert kk in sd and tuple(sd[kk].shape) == shape, kk
    assert len(sd) == 17 + 2 * 2 * 2      # 17 unique + (2 stacks x 2 weights x layers 1..2) aliases
    assert len(list(m.parameters())) == 17
    # shared residual object (residual_stack.py:40-41)
    layers = m._encoder._residual_stack._layers
    assert layers[0] is layers[1] is layers[2]
    assert m._decoder._use_jitter and hasattr(m._decoder, "_jitter")
    assert not hasattr(make(use_jitter=False)._decoder, "_jitter")
    assert make(out_channels=1)._decoder._conv_trans_3.weight.shape == (16, 1, 3)
    assert m.get_embedding_dim() == 8


def test_init_distributions():
    torch.manual_seed(0)
    m = make((64, 256, 16, 2, 128, 0.25, 512))
    w = m._encoder._conv_1.weight
    assert float(w.abs().max()) <= (6.0 / (64 * 3)) ** 0.5 + 1e-6
    cb = m._vq._embedding.weight
    assert float(cb.abs().max()) <= 1.0 / 512
    k1 = m._encoder._residual_stack._layers[0]._block[3].weight     # never Kaiming-initialised (residual.py:55)
    assert float(k1.abs().max()) <= (1.0 / 128) ** 0.5 + 1e-6
    wt = m._decoder._conv_trans_3.weight                             # fan_in = Cout*k for ConvTranspose
    assert float(wt.abs().max()) <= (6.0 / (64 * 3)) ** 0.5 + 1e-6


def test_src_alias_is_same_module_object():
    import src.acoustic_locating_vq_vae.vq_vae.modules.residual as a
    import acoustic_locating_vq_vae.vq_vae.modules.residual as b
    assert a is b and a.Residual is b.Residual
    import importlib
    c = importlib.import_module("src.acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae")
    from acoustic_locating_vq_vae.vq_vae import convolutional_vq_vae as d
    assert c is d


def test_whole_module_pickle_roundtrip_keeps_aliasing():
    m = make()
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert m2._encoder._residual_stack._layers[0] is m2._encoder._residual_stack._layers[1]
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    pickle.loads(pickle.dumps(m))


def test_load_reference_style_state_dict():
    m = make()
    p = O.closed_form_params(O.vqvae_param_shapes(7, 16, 4, 8, 16), codebook_scale=0.8)
    sd = {}
    for k, v in p.items():
        for r in range(2):
            sd[k.replace("_layers.0.", "_layers.%d." % r)] = v
    m.load_state_dict(sd)
    assert torch.equal(m._decoder._conv_trans_1.weight, p["_decoder._conv_trans_1.weight"])


def test_jitter_host_stream_matches_golden(golden_dir):
    from acoustic_locating_vq_vae import _ops
    g = np.load(os.path.join(golden_dir, "g5_jitter.npz"))
    for length in (13, 201, 500):
        for seed in (0, 1):
            np.random.seed(seed)
            assert np.array_equal(_ops.jitter_source_index(length, 0.25), g["L%d_s%d" % (length, seed)])


def test_forward_on_cpu_fails_loudly():
    m = make()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(2, 7, 13))
    from acoustic_locating_vq_vae.vq_vae.vector_quantizer import VectorQuantizer
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        VectorQuantizer(16, 4, 0.25)(torch.randn(2, 4, 6))


def test_tensor_on_another_card_is_refused(monkeypatch):
    """One process per GPU: the kernels are launched on the current device's stream, so an input living on another card
    must raise instead of being handed to a kernel that runs elsewhere (stand-in tensor: this test needs no GPU)."""
    from acoustic_locating_vq_vae import _ops

    class OnCard1:
        is_cuda = True
        device = torch.device("cuda", 1)

    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    with pytest.raises(RuntimeError, match=r"set_device\(1\)"):
        _ops._need_gpu(OnCard1(), "ConvolutionalVQVAE")
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 1)
    _ops._need_gpu(OnCard1(), "ConvolutionalVQVAE")


def test_echoed_model_surface():
    from acoustic_locating_vq_vae.vq_vae.echoed_speech_model import EchoedSpeechReconModel
    rir = make((20, 16, 4, 2, 8, 0.25, 16), use_jitter=False, out_channels=1)
    sp = make((9, 16, 6, 3, 16, 0.25, 32))
    if torch.cuda.is_available():
        pytest.skip("constructor moves sub-models to the GPU; covered by the gpu tests")
    e = EchoedSpeechReconModel(rir, sp, 9, 16, 2, 16, True)
    assert e.embedding_dim == 10 and not e.rir_model._vq._train_vq and not e.speech_model._vq._train_vq
    e.set_train_encoder(True)
    assert e.flag_train_encoder
    assert e._decoder._conv_1.weight.shape == (16, 10, 3)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/acoustic_locating_vq_vae"),
                    reason="overlay check needs the reference checkout (build container only)")
def test_overlay_resolves_non_hot_path_modules_from_the_reference():
    """With the reference LATER on sys.path, every module this build provides (hot path, dataset side, location head)
    comes from this build and everything else (visualization, ...) keeps resolving from the reference."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "acoustic_locating_vq-vae_amd")
    code = (
        "import acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae as a;"
        "import acoustic_locating_vq_vae.vq_vae.location_model.location_model as b;"
        "import acoustic_locating_vq_vae.rir_dataset_generator.specsdataset as c;"
        "from src.acoustic_locating_vq_vae.vq_vae.modules.residual import Residual;"
        "import importlib.util as u;"
        "print(a.__file__); print(b.__file__); print(c.__file__); print(Residual.__module__);"
        "print(u.find_spec('acoustic_locating_vq_vae.visualization').origin)"
    )
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([pkg, os.path.join(pkg, "src"), "/root/reference", "/root/reference/src"]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    # hot-path modules, the dataset side and the location head (rank 4) come from this build; a module
    # this build does not provide (plotting) is found in the reference checkout
    assert lines[0].startswith(pkg) and lines[1].startswith(pkg) and lines[2].startswith(pkg)
    assert lines[4].startswith("/root/reference")


def test_package_default_mode_is_the_parity_holding_fast_mode():
    """A user who sets nothing (scripts/train_speech.py unchanged) gets x3mx_hb; ALVQ_DTYPE overrides; junk -- and the
    engines retired to internal use in round 4 -- are refused."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "acoustic_locating_vq-vae_amd")
    code = "from acoustic_locating_vq_vae import _ops; print(_ops.get_compute_dtype(), _ops.DEFAULT_DTYPE)"
    env = {k: v for k, v in os.environ.items() if k != "ALVQ_DTYPE"}
    env["PYTHONPATH"] = os.pathsep.join([pkg, os.path.join(pkg, "src")])
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.split() == ["x3mx_hb", "x3mx_hb"], out.stdout + out.stderr
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, ALVQ_DTYPE="f32"), capture_output=True, text=True, timeout=300)
    assert out.stdout.split() == ["f32", "x3mx_hb"], out.stdout + out.stderr
    for junk in ("fp64", "f16mx", "bf16x3", "f16mx_hd"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(env, ALVQ_DTYPE=junk), capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "ALVQ_DTYPE" in out.stderr
    code = ("from acoustic_locating_vq_vae import _ops\n"
            "assert _ops.MODES == ('x3mx_hb', 'f16mx_hb', 'bf16x3_hb', 'f32', 'bf16')\n"
            "try:\n    _ops.set_compute_dtype('f16mx')\n    raise SystemExit(3)\nexcept ValueError:\n    pass\n"
            "_ops.set_compute_dtype('f16mx', internal=True)\nprint(_ops.get_compute_dtype())")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.split() == ["f16mx"], out.stdout + out.stderr
