"""Sanitizer builds of the host-side C++ that can run without a GPU (CPU only; never on the GPU box):

* AddressSanitizer + UndefinedBehaviourSanitizer: the oracle over awkward inputs (tests/native/oracle_san_driver.cpp), the
  reference's k-d test procedure against the oracle (kdtree_replay), the introselect replay checker, the public node-pointer
  k-d functions of the drop-in layer (vslam_amd/host/kdtree_nodes.cpp has no device call in it);
* ThreadSanitizer and ASan + UBSan: the capture loop of vslam_amd/host/ingest.cpp -- reader pool, page-locked double
  buffer, the slot threads of run_sequence_devices -- against a CPU stand-in for the C-ABI calls it makes
  (tests/native/capi_stub.cpp: asynchronous uploads on a worker thread, a checksum per frame pair for results).
A finding is a non-zero exit (halt_on_error) and fails the test."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")
ORACLE_SRCS = [os.path.join(ROOT, "oracle", f) for f in
               ("vso_kdtree.cpp", "vso_match.cpp", "vso_svd.cpp", "vso_ransac.cpp", "vso_extract.cpp", "vso_orb.cpp", "vso_pose.cpp")]
ASAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
TSAN = ["-fsanitize=thread", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="halt_on_error=1:detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
           TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")


def build(tmp_path, name, sources, flags, extra=()):
    exe = str(tmp_path / name)
    # the oracle's own flags (-ffp-contract=off, popcnt) with -O1 -g for readable reports
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-mpopcnt", "-pthread", "-Wall"] + flags + ["-o", exe] + sources + \
          ["-I" + os.path.join(ROOT, "include")] + list(extra)
    subprocess.run(cmd, check=True)
    return exe


def run(exe, *args, timeout=600):
    r = subprocess.run([exe, *args], env=ENV, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{os.path.basename(exe)} exit {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-6000:]}"
    return r.stdout


def test_oracle_under_asan_ubsan(tmp_path):
    exe = build(tmp_path, "oracle_san", [os.path.join(NATIVE, "oracle_san_driver.cpp")] + ORACLE_SRCS, ASAN)
    out = run(exe).split()
    assert len(out) == 1 and len(out[0]) == 16
    # the same driver without sanitizers computes the same bytes: the sanitizer build is the code that is tested elsewhere
    plain = build(tmp_path, "oracle_plain", [os.path.join(NATIVE, "oracle_san_driver.cpp")] + ORACLE_SRCS, [])
    assert run(plain).split() == out


def test_kdtree_reference_procedure_under_asan_ubsan(tmp_path):
    exe = build(tmp_path, "kdtree_replay_san", [os.path.join(NATIVE, "kdtree_replay.cpp")] + ORACLE_SRCS, ASAN)
    nn_ok, rad_ok, trials = map(int, run(exe, "60").split())
    assert (nn_ok, rad_ok, trials) == (60, 60, 60)


def test_introselect_replay_under_asan_ubsan(tmp_path):
    exe = build(tmp_path, "introselect_san", [os.path.join(NATIVE, "introselect_check.cpp")], ASAN)
    ok, total, heap_calls = map(int, run(exe).split())
    assert ok == total and heap_calls > 0


def test_node_pointer_kdtree_under_asan_ubsan(tmp_path):
    """The public recursive helpers of include/vslam/KDTree.h (host code: vslam_amd/host/kdtree_nodes.cpp) on trees the
    driver links by hand, against brute force."""
    exe = build(tmp_path, "kdnodes_san", [os.path.join(NATIVE, "kdtree_nodes_san_driver.cpp"),
                                          os.path.join(ROOT, "vslam_amd", "host", "kdtree_nodes.cpp")], ASAN)
    assert int(run(exe).split()[0]) > 1000


@pytest.mark.parametrize("flags", [TSAN, ASAN], ids=["tsan", "asan_ubsan"])
def test_capture_loop_threads_against_a_stub_device(tmp_path, flags):
    exe = build(tmp_path, "ingest_san", [os.path.join(NATIVE, "ingest_san_driver.cpp"), os.path.join(NATIVE, "capi_stub.cpp"),
                                         os.path.join(ROOT, "vslam_amd", "host", "ingest.cpp")], flags)
    work = tmp_path / "work"
    work.mkdir()
    assert int(run(exe, str(work)).split()[0]) > 2000
