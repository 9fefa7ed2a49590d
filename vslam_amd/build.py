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
SOURCES = ["capi.hip", "match.hip", "ransac.hip", "kdtree.hip", "gray.hip", "response.hip", "select.hip", "blur.hip",
           "brief.hip", "orb_grid.hip", "pose.hip", "assoc.hip"]
HEADERS = ["ctx.h", "introselect.h", "image_common.h", os.path.join("..", "..", "include", "vslam_amd.h")]
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


HOST_LIB = os.path.join(HERE, "libvslam_host.so")
HOST_SRCS = [os.path.join(HERE, "host", f) for f in ("adapters.cpp", "kdtree_nodes.cpp", "ingest.cpp")]
HOST_HDRS = [os.path.join(HERE, "host", "host_internal.h")]
INCLUDE = os.path.join(HERE, "..", "include")


def build_host(force=False, verbose=False):
    """The C++ drop-in layer (include/vslam/*.h): plain g++, links against libvslam_amd.so."""
    build(force=False)
    deps = HOST_SRCS + HOST_HDRS + [LIB] + [os.path.join(INCLUDE, "vslam", f) for f in os.listdir(os.path.join(INCLUDE, "vslam"))]
    if not force and os.path.exists(HOST_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB) for d in deps):
        return HOST_LIB
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-pthread", "-o", HOST_LIB] + HOST_SRCS + [
        "-L" + HERE, "-lvslam_amd", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return HOST_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_host(force="--force" in sys.argv, verbose=True)
    print(LIB)
    print(HOST_LIB)
