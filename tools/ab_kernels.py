"""A/B two builds of libalvq.so on the same device in one process-sequence: interleaved rounds of bench_kernels.
Box-to-box spread (2-3 %) exceeds most kernel-level changes, so comparisons must share a device.  Keep a copy of the
build to compare against as lib/libalvq_old.so (cp lib/libalvq.so lib/libalvq_old.so before editing a kernel)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {"old": os.path.join(ROOT, "acoustic_locating_vq-vae_amd", "lib", "libalvq_old.so"),
        "new": os.path.join(ROOT, "acoustic_locating_vq-vae_amd", "lib", "libalvq.so")}
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if not os.path.exists(libs["old"]):
    raise SystemExit("no baseline build: cp %s %s first" % (libs["new"], libs["old"]))
for r in range(rounds):
    for name, lib in libs.items():
        env = dict(os.environ, ALVQ_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_kernels.py"), "bf16"], env=env,
                             capture_output=True, text=True).stdout
        for line in out.splitlines():
            if "M=1024" in line:
                print(name, line[:118])
