"""Builds libaaerec_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
OUT = os.path.join(_HERE, "libaaerec_hip.so")
SOURCES = ["aae_abi.hip"]


def _headers():
    """Every header the library is compiled from: csrc/*.h and the public C-ABI header."""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "aaerec_hip.h")]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(f) > t for f in [os.path.join(CSRC, f) for f in SOURCES] + _headers())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-o", OUT] + \
          [os.path.join(CSRC, f) for f in SOURCES] + ["-Wl,-rpath,/opt/rocm/lib"] + \
          os.environ.get("AAE_HIPCC_FLAGS", "").split()        # (experiments: extra -D switches)
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force=True)
