"""Build libcvmhip.so in-tree for gfx950:  python -m cvmatrix_amd.build"""

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "cvmhip.hip")
OUT = os.path.join(HERE, "libcvmhip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC"]


def build(force: bool = False, verbose: bool = True) -> str:
    csrc = os.path.join(HERE, "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(HERE, "..", "include", "cvmhip.h")]
    if (not force and os.path.exists(OUT)
            and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *FLAGS, "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
