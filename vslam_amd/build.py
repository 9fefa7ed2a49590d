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
           "brief.hip", "orb_grid.hip", "pose.hip", "assoc.hip", "multi.hip", "pipeline.hip"]
HEADERS = ["ctx.h", "introselect.h", "image_common.h", os.path.join("..", "..", "include", "vslam_amd.h"),
           os.path.join("..", "..", "include", "vslam_brief_pattern_31.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall"]
# Per-file additions.  response.hip: the SLP vectoriser pairs the stencil's float operations into v_pk_* forms whose
# operands it then has to assemble with v_mov (a packed op issues for two slots, so the copies are a net loss: min_eigen
# 1.131 -> 1.116 ms without it); the RANSAC kernels, whose packed math is written out by hand, are faster with it on.
EXTRA_FLAGS = {"response.hip": ["-fno-slp-vectorize"]}
OBJDIR = os.path.join(HERE, "_obj")


# The experiments build: the same sources with -DVSLAM_EXPERIMENTS (ctx.h), i.e. WITH the environment-variable A/B switches and
# the kernel variants that were measured and not chosen.  tools/ab_*.py and the variant tests load it through VSLAM_AMD_LIB /
# capi.load_library(EXP_LIB); the product (LIB) reads no environment variable and carries one kernel per stage.
EXP_LIB = os.path.join(HERE, "libvslam_amd_exp.so")
EXP_OBJDIR = os.path.join(HERE, "_obj_exp")


def _compile(hipcc, src, verbose, objdir=None, extra=()):
    objdir = objdir or OBJDIR
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
        return obj, None
    cmd = [hipcc] + FLAGS + list(extra) + EXTRA_FLAGS.get(src, []) + ["-c", "-o", obj, src]
    if verbose:
        print(" ".join(cmd))
    return obj, subprocess.Popen(cmd, cwd=CSRC)


def build(force=False, verbose=False, experiments=False):
    """Every source is its own translation unit (no device code crosses files), compiled in parallel, then linked.
    experiments=True builds libvslam_amd_exp.so (-DVSLAM_EXPERIMENTS) instead of the product."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir, lib, extra = (EXP_OBJDIR, EXP_LIB, ("-DVSLAM_EXPERIMENTS",)) if experiments else (OBJDIR, LIB, ())
    return _build(hipcc, objdir, lib, extra, force, verbose)


def _build(hipcc, OBJDIR, LIB, extra, force, verbose):
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    jobs = [_compile(hipcc, src, verbose, OBJDIR, extra) for src in SOURCES]
    failed = [obj for obj, proc in jobs if proc is not None and proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c (" + ", ".join(os.path.basename(f) for f in failed) + ")")
    objs = [obj for obj, _ in jobs]
    if os.path.exists(LIB) and all(proc is None for _, proc in jobs) and all(
            os.path.getmtime(o) <= os.path.getmtime(LIB) for o in objs):
        return LIB
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
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
    build(force="--force" in sys.argv, verbose=True, experiments=True)
    print(LIB)
    print(EXP_LIB)
    print(HOST_LIB)
