"""CPU oracle for the VQ-VAE train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: it may be
imported only by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker (or the
timed CPU baseline), never as a fallback for the HIP path.

The reference (guy3540/Acoustic_Locating_VQ-VAE) is pure Python on PyTorch, so
the restatement is a functional PyTorch-CPU program that issues the same ATen
op sequence (conv1d / conv_transpose1d / matmul / argmin).  There is no C
restatement to compile.  It is pinned two ways:

* ``oracle/check_against_reference.py`` imports the real reference from
  ``/root/reference`` (build container only) and checks bit-identical outputs,
  indices and gradients;
* ``tests/golden/*.npz`` hold vectors produced by the real reference
  (``tests/golden/make_goldens.py``) and are checked on any machine.

The STFT front end follows torchaudio semantics; torchaudio is not installed
and the reference holds no fixture for it, so that one function is
"parity unpinned" (see ``stft_oracle.py``).
"""
