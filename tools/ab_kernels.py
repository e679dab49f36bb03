"""A/B two builds of libalvq.so on the same device in one process-sequence: interleaved rounds of bench_kernels."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {"old": os.path.join(ROOT, "acoustic_locating_vq-vae_amd", "lib", "libalvq_old.so"),
        "new": os.path.join(ROOT, "acoustic_locating_vq-vae_amd", "lib", "libalvq.so")}
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for r in range(rounds):
    for name, lib in libs.items():
        env = dict(os.environ, ALVQ_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_kernels.py"), "bf16"], env=env,
                             capture_output=True, text=True).stdout
        for line in out.splitlines():
            if "M=1024" in line:
                print(name, line[:118])
