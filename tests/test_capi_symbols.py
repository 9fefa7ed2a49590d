"""The C-ABI library builds, loads without a GPU, exports every symbol include/vslam_amd.h
declares, and refuses to run without a device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from vslam_amd import build, capi
    build.build()
    return capi.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vslam_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vslam_[A-Za-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from vslam_amd import capi
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vslam_amd.h but not exported"
    assert sorted(capi.SYMBOLS) == names


def test_no_cpu_fallback_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    h = ctypes.c_void_p()
    assert lib.vslam_ctx_create(0, ctypes.byref(h)) == -3   # VSLAM_ERR_NO_DEVICE
    from vslam_amd import Context, VslamError
    with pytest.raises(VslamError):
        Context(0)


def test_product_does_not_reference_oracle():
    """Nothing under vslam_amd/ or include/ may include, link or import the oracle."""
    bad = []
    for base in ("vslam_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".pyc")):
                    continue
                s = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r'#include\s+"[^"]*(vso|oracle)', s) or "liboracle" in s or "oracle_lib" in s:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_host_library_exports_its_c_entry_points():
    """libvslam_host.so (the C++ drop-in surfaces) exports the extern "C" entry points include/vslam/Ingest.h declares."""
    from vslam_amd import build
    host = ctypes.CDLL(build.build_host())
    text = open(os.path.join(ROOT, "include", "vslam", "Ingest.h")).read()
    names = sorted(set(re.findall(r'extern "C" int (vslam_host_[a-z_]+)\(', text)))
    assert names == ["vslam_host_run_sequence", "vslam_host_run_sequence_devices"]
    assert hasattr(host, "vslam_host_last_batches_redone")
    for n in names:
        assert hasattr(host, n), n


def test_product_library_reads_no_environment_and_carries_no_variants():
    """The A/B switches and the kernel variants that were measured and not chosen live in the experiments build only
    (-DVSLAM_EXPERIMENTS -> libvslam_amd_exp.so): the product holds no VSLAM_* environment name, no getenv import, and none of
    the variant kernels; the experiments build holds them."""
    import subprocess
    from vslam_amd import build
    prod = subprocess.run(["strings", "-a", build.build()], capture_output=True, text=True, check=True).stdout
    for name in ("VSLAM_MATCH_", "VSLAM_RANSAC_", "VSLAM_KD_", "VSLAM_RBRIEF_", "VSLAM_STREAM_", "VSLAM_OVERLAP_", "VSLAM_SHARED_",
                 "VSLAM_LAZY_", "VSLAM_SETS_", "VSLAM_TREE_", "VSLAM_NO_GRAY", "VSLAM_CORNER_EXACT"):
        assert name not in prod, name
    for kernel in ("match_knn2_mfma_kernel", "ransac_close_kernel"):
        assert kernel not in prod, kernel
    undefined = subprocess.run(["nm", "-D", "--undefined-only", build.LIB], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    exp = subprocess.run(["strings", "-a", build.build(experiments=True)], capture_output=True, text=True, check=True).stdout
    assert "VSLAM_MATCH_POPCOUNT" in exp and "match_knn2_mfma_kernel" in exp and "ransac_close_kernel" in exp
    assert os.path.getsize(build.LIB) < os.path.getsize(build.EXP_LIB)
