import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle, build_oracle
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        build_oracle()
    return Oracle()


@pytest.fixture(scope="session")
def native_bin(tmp_path_factory):
    """Compile a tests/native/*.cpp helper against the oracle and return a runner."""
    out_dir = tmp_path_factory.mktemp("native")

    def build(name, link_oracle=True):
        src = os.path.join(ROOT, "tests", "native", name + ".cpp")
        exe = str(out_dir / name)
        cmd = ["g++", "-O2", "-std=c++17", "-o", exe, src]
        if link_oracle:
            odir = os.path.join(ROOT, "oracle")
            if not os.path.exists(os.path.join(odir, "liboracle.so")):
                subprocess.run(["make", "-s", "-C", odir], check=True)
            cmd += ["-L" + odir, "-loracle", "-Wl,-rpath," + odir]
        subprocess.run(cmd, check=True)
        return exe

    return build


@pytest.fixture(scope="session")
def ctx():
    """A live vslam_ctx on cuda:0.  No fallback: absence of the GPU or the .so is a failure."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test needs a HIP device"
    from vslam_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx_exp():
    """A context of the EXPERIMENTS build (libvslam_amd_exp.so, -DVSLAM_EXPERIMENTS): the kernel variants that the product
    library does not carry (the matcher's int8 form and other workgroup shape) are held to the oracle through it."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test needs a HIP device"
    from vslam_amd import Context, capi
    c = Context(0, lib=capi.load_library(capi.EXP_LIB_PATH))
    yield c
    c.close()
