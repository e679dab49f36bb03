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


def test_wgrad_workspace_covers_every_segment_count(native):
    """Round-2 advisor finding: ceil(n / ceil(n / want)) is not monotone in n, so a workspace sized for 1 and 4 segments
    under-sized some 3-segment launches (e.g. f16mx, 1024 x 128 width 1, B = 5: 60 splits needed, 54 sized).  The split
    count of every launch (nseg = 1..4, with / without the fused bias gradient) must fit what *_workspace_bytes sizes."""
    lib = native.lib()
    shapes = [(1024, 128, 1), (128, 1024, 1), (1024, 1024, 1), (1024, 1024, 3), (64, 1024, 3), (1024, 64, 1), (201, 1024, 3),
              (1024, 201, 3), (16, 7, 3), (500, 1024, 3), (192, 1024, 3), (1, 1024, 3)]
    bias_floats = {"bf16": 64, "bf16x3": 64, "f16mx": 128}
    worst = 0
    for C, M, KW in shapes:
        for L in (201, 500, 13):
            for B in list(range(1, 81)) + [96, 128, 256]:
                per = KW * M * C * 4
                pad_m = (M + 63) // 64 * 64
                for fmt in ("bf16", "bf16x3", "f16mx"):
                    ws = getattr(lib, "alvq_conv1d_wgrad_%s_workspace_bytes" % fmt)(B, C, M, L, KW)
                    sized = (ws - bias_floats[fmt] * pad_m * 4) // per
                    for nseg in (1, 2, 3, 4):
                        if fmt == "bf16":
                            used = max(lib.alvq_conv1d_wgrad_bf16_splits(B, C, M, L, KW, nseg, 0),
                                       lib.alvq_conv1d_wgrad_bf16_splits(B, C, M, L, KW, 1, 1))
                        else:
                            used = getattr(lib, "alvq_conv1d_wgrad_%s_splits" % fmt)(B, C, M, L, KW, nseg)
                        assert 1 <= used <= min(sized, 64), (fmt, B, C, M, L, KW, nseg, used, sized)
                        worst = max(worst, used)
    assert worst == 64                                       # the sweep reaches the cap
    assert lib.alvq_conv1d_wgrad_f16mx_splits(5, 128, 1024, 500, 1, 3) == 60   # the advisor's case: needs 60 ...
    assert lib.alvq_conv1d_wgrad_f16mx_workspace_bytes(5, 128, 1024, 500, 1) >= 60 * 1024 * 128 * 4   # ... and gets them
    assert lib.alvq_conv1d_wgrad_f16mx_splits(5, 128, 1024, 500, 1, 5) == -1


def test_every_entry_point_rejects_null_arguments_before_any_launch(native):
    """The error convention of the boundary: every entry point validates its arguments BEFORE any launch and
    returns a negative status with a message -- called here with null pointers and zero sizes on a machine without a GPU:
    no crash, no HIP error (a launch attempt would return a positive hipError_t), a message from the library."""
    import ctypes
    lib = native.lib()
    checked = 0
    for name, (res, args) in native._SIGNATURES.items():
        if res is not native._i32 or not any(a is ctypes.c_void_p for a in args):
            continue
        vals = [0.0 if a in (ctypes.c_float, ctypes.c_double) else (0 if a in (ctypes.c_int, ctypes.c_int64) else None) for a in args]
        rc = getattr(lib, name)(*vals)
        msg = lib.alvq_last_error() or b""
        assert rc < 0, (name, rc)
        assert msg.startswith(b"alvq_"), (name, msg)
        checked += 1
    assert checked >= 55, checked
