"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/alvq.h declares.
No compute entry point is called (there is no GPU here)."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    from acoustic_locating_vq_vae import _native
    return _native


def test_header_symbols_exported(native):
    hdr = open(os.path.join(ROOT, "include", "alvq.h")).read()
    declared = set(re.findall(r"\b(alvq_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = native.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "libalvq.so does not export %s" % name
    assert declared == set(native.EXPORTS)
    assert native.version().startswith("alvq")


def test_argument_errors_do_not_launch(native):
    lib = native.lib()
    rc = lib.alvq_conv1d_f32(None, None, None, None, None, None, None, None, None, 1, 1, 1, 1, 3, 0, 0, None)
    assert rc == -1 and b"null" in lib.alvq_last_error()
    assert lib.alvq_conv1d_wgrad_workspace_bytes(2, 7, 16, 13, 3) > 0
    assert lib.alvq_conv1d_wgrad_workspace_bytes(2, 7, 16, 13, 5) == -1
    assert lib.alvq_vq_argmin_workspace_bytes(10, 4, 3) == (10 + 4) * 4


def test_missing_library_fails_loudly(native, monkeypatch):
    import acoustic_locating_vq_vae._native as n
    monkeypatch.setattr(n, "_LIB", None)
    monkeypatch.setenv("ALVQ_LIB", "/nonexistent/libalvq.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        n.lib()


def test_cpu_tensor_rejected(native):
    import torch
    with pytest.raises(RuntimeError, match="GPU"):
        native.conv1d(torch.zeros(1, 1, 4), torch.zeros(1, 1, 3))
