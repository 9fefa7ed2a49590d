"""The device k-d build replays libstdc++'s introselect; this checks that replay on the host:
vslam_amd/csrc/introselect.h must leave exactly the permutation std::nth_element leaves."""
import subprocess


def test_introselect_matches_std_nth_element(native_bin):
    exe = native_bin("introselect_check", link_oracle=False)
    ok, total, heap_calls = map(int, subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split())
    assert ok == total
    assert heap_calls > 0, "the depth-limit / heap_select fallback was never exercised"
