"""Build the HIP library in-tree: vslam_amd/libvslam_amd.so (gfx950 only).

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
-ffp-contract=off is load-bearing: the RANSAC and corner kernels must execute the same IEEE
operations, in the same order, as the CPU oracle (see DESIGN.md "Numerics").
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvslam_amd.so")
SOURCES = ["capi.hip", "match.hip", "ransac.hip", "kdtree.hip", "extract.hip"]
HEADERS = ["ctx.h", "introselect.h", os.path.join("..", "..", "include", "vslam_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
