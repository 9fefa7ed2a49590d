"""vslam_shard_range (the C ABI's split of a batch over devices / ranks) is the split the Python harness and bench.py use."""
import ctypes

from vslam_amd import build, capi, shard


def test_c_and_python_shard_ranges_agree():
    build.build()
    lib = capi.load_library()
    lo, hi = ctypes.c_int(), ctypes.c_int()
    for items in (0, 1, 7, 8, 255, 256, 2048, 4097):
        for world in (1, 2, 3, 8, 16):
            covered = 0
            for rank in range(world):
                assert lib.vslam_shard_range(items, rank, world, ctypes.byref(lo), ctypes.byref(hi)) == 0
                assert (lo.value, hi.value) == shard.shard_range(items, rank, world)
                assert lo.value == covered
                covered = hi.value
            assert covered == items
    assert lib.vslam_shard_range(8, 3, 3, ctypes.byref(lo), ctypes.byref(hi)) == -1
    assert lib.vslam_shard_range(8, 0, 0, ctypes.byref(lo), ctypes.byref(hi)) == -1
