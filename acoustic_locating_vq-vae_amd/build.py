"""Build libalvq.so (the C-ABI HIP library) in-tree for gfx950.

    python build.py                    # incremental
    python build.py --force
    python build.py --debug-kernels    # lib/libalvq_dbg.so: also the timing-ablation / phase-stamp instantiations of the
                                       # f16mx kernels (ALVQ_FX_DBG; tools/ablate_f16mx.sh loads it through ALVQ_LIB).
                                       # The shipped libalvq.so never contains them.

hipcc cross-compiles without a GPU.  Output: <this dir>/lib/libalvq.so (git-ignored, but it
travels to the GPU box with the gpurun snapshot).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result", "-I", os.path.join(os.path.dirname(HERE), "include")]


def _stamp(path, only=None):
    """Hash of every header + (all sources | the one source ``only``) + the flags."""
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)) + ["../../include/alvq.h"]:
        if f.endswith(".h") or (f.endswith(".hip") and (only is None or f == only)):
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True, debug_kernels=False):
    global OBJ
    os.makedirs(OUT, exist_ok=True)
    flags = FLAGS + (["-DALVQ_DEBUG_KERNELS"] if debug_kernels else [])
    obj_dir = OBJ + ("_dbg" if debug_kernels else "")
    os.makedirs(obj_dir, exist_ok=True)
    lib = os.path.join(OUT, "libalvq_dbg.so" if debug_kernels else "libalvq.so")
    stamp_file = os.path.join(obj_dir, "stamp")
    stamp = _stamp(CSRC) + ("dbg" if debug_kernels else "")
    if not force and os.path.exists(lib) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return lib
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))

    def cc(src):
        # per-object stamp: an edit to one .hip recompiles that file only (a header edit recompiles everything)
        obj = os.path.join(obj_dir, src[:-4] + ".o")
        ostamp_file, ostamp = obj + ".stamp", _stamp(CSRC, only=src) + " ".join(flags)
        if not force and os.path.exists(obj) and os.path.exists(ostamp_file) and open(ostamp_file).read() == ostamp:
            return obj
        cmd = [HIPCC] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(ostamp_file, "w") as fh:
            fh.write(ostamp)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, debug_kernels="--debug-kernels" in sys.argv))
