"""Pins the oracle's KDTree against the reference's own test procedure.

/root/reference/tests/test_kdtree.cpp:148-151 runs test_nearest_neighbor(1000, 2500, 3000) and
test_radius_search(1000, 10, 100, 2500, 3000) on an unseeded glibc rand() stream and prints
"1000 successes out of 1000 trials" twice.  tests/native/kdtree_replay.cpp follows the same
procedure (same stream, same sizes, same acceptance rules) against oracle/vso_kdtree.cpp.
"""
import subprocess


def test_reference_kdtree_procedure_1000_of_1000(native_bin):
    exe = native_bin("kdtree_replay")
    out = subprocess.run([exe, "1000"], check=True, capture_output=True, text=True).stdout.split()
    nn_ok, rad_ok, trials = map(int, out)
    assert trials == 1000
    assert nn_ok == 1000, f"nearest: {nn_ok} successes out of 1000 trials"
    assert rad_ok == 1000, f"radius_search: {rad_ok} successes out of 1000 trials"
