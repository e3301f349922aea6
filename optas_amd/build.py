"""In-tree build of liboptas_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["oh_kernels.hip", "oh_free.hip", "oh_pointmass.hip", "oh_ik.hip", "oh_qp.hip", "oh_api.hip"]
HEADERS = ["oh_device.h", "oh_kernels.h", "oh_figure8.h", os.path.join("..", "..", "include", "optas_hip.h")]
OUT = os.path.join(HERE, "liboptas_hip.so")


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-shared", "-fPIC", "-o", OUT] + [
        os.path.join(CSRC, s) for s in SOURCES
    ]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
