"""Build libcvmhip.so in-tree for gfx950:  python -m cvmatrix_amd.build

The library carries a hash of the sources it was built from (cvm_source_hash()); `build()` rebuilds
whenever that hash differs from the sources on disk, and the loader (cvmatrix_amd/_lib.py) checks
the same thing, so the binary under test is always the committed source."""

import ctypes
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "cvmhip.hip")
OUT = os.path.join(HERE, "libcvmhip.so")
HEADER = os.path.join(HERE, "..", "include", "cvmhip.h")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC"]


def source_files():
    csrc = os.path.join(HERE, "csrc")
    return sorted(os.path.join(csrc, f) for f in os.listdir(csrc)
                  if f.endswith((".hip", ".hpp", ".h"))) + [HEADER]


def source_hash() -> str:
    """sha256 over (name, bytes) of every source of the library, first 16 hex digits; the build
    flags are part of it."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in source_files():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_hash(path: str = OUT):
    """Hash embedded in an existing library (None if it cannot be read)."""
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        lib.cvm_source_hash.restype = ctypes.c_char_p
        return lib.cvm_source_hash().decode()
    except (OSError, AttributeError):
        return None


def _embedded_hash_without_loading(path: str = OUT):
    """The same, read from the file (no dlopen: a library loaded once stays mapped even after
    it has been rebuilt)."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        blob = f.read()
    tag = b"cvmhip 0.3.0 (gfx950) src "
    i = blob.find(tag)
    if i < 0:
        return None
    return blob[i + len(tag): i + len(tag) + 16].decode(errors="replace")


def build(force: bool = False, verbose: bool = True) -> str:
    import fcntl

    want = source_hash()
    if not force and _embedded_hash_without_loading() == want:
        return OUT
    # several processes (ranks of one job, pytest workers) may find the library stale at the same
    # time: one builds, the others wait and find it current
    with open(OUT + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if _embedded_hash_without_loading() == want and not force:
            return OUT
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        tmp = OUT + f".tmp{os.getpid()}"
        cmd = [hipcc, *FLAGS, f'-DCVM_SRC_SHA="{want}"', "-o", tmp, SRC]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
