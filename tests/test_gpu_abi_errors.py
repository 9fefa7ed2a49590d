"""The C ABI fails loudly and specifically: bad arguments, capacity limits, undefined-in-the-reference inputs."""
import ctypes as C

import numpy as np
import pytest
import torch

from vslam_amd import VslamError, synth

pytestmark = pytest.mark.gpu


def test_null_and_range_arguments_are_rejected(ctx):
    lib, h = ctx.lib, ctx.handle
    z = C.c_void_p(0)
    assert lib.vslam_match_knn2_ratio(h, z, z, z, z, 1, 16, z, z, z) == -1          # VSLAM_ERR_INVALID
    assert lib.vslam_kdtree_build(h, z, z, 1, 16, z) == -1
    assert lib.vslam_ransac_sets(h, z, z, 1, 16, z, z) == -1
    assert lib.vslam_match_knn2_ratio(C.c_void_p(0), z, z, z, z, 1, 16, z, z, z) == -1   # null context
    assert b"requirement failed" in lib.vslam_last_error(h)
    d = torch.zeros((1, 20000, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros((1,), dtype=torch.int32, device="cuda")
    with pytest.raises(VslamError, match="CAPACITY"):                                # > VSLAM_MAX_KP slots
        ctx.match_knn2_ratio(d, n, d, n)


def test_undefined_reference_inputs_are_defined_here(ctx):
    """< 2 train rows (src/Frame.cpp:91 reads m[1]) and < 8 matches (src/RansacFilter.cpp:24) are UB in the
    reference; here they produce 'no matches' / 'no model' without touching the outputs."""
    K = 64
    d1, d2, _ = synth.descriptors_pair(1, K, K)
    t = lambda a: torch.from_numpy(a).cuda()
    n1 = torch.tensor([K], dtype=torch.int32).cuda()
    pairs, m = ctx.match_knn2_ratio(t(d1[None]), n1, t(d2[None]), torch.tensor([1], dtype=torch.int32).cuda())
    assert int(m[0]) == 0
    xy = torch.rand((1, K, 2), device="cuda") * 100
    seeds = torch.tensor([5], dtype=torch.int32).cuda()
    F0 = torch.full((1, 9), 7.0, device="cuda")
    out = dict(matches=torch.zeros((1, K, 2), dtype=torch.int32, device="cuda"), best=torch.zeros((1, 4), dtype=torch.int32, device="cuda"),
               F=F0.clone(), prelim_m=torch.zeros((1,), dtype=torch.int32, device="cuda"))
    # random descriptors against each other: (almost) nothing survives the ratio test -> fewer than 8 matches
    r1 = torch.randint(0, 256, (1, K, 32), dtype=torch.uint8, device="cuda")
    r2 = torch.randint(0, 256, (1, K, 32), dtype=torch.uint8, device="cuda")
    ctx.match_features(xy, r1, n1, xy, r2, n1, seeds, 32, 10.0, out=out)
    ctx.synchronize()
    assert int(out["prelim_m"][0]) < 8
    assert out["best"][0].tolist() == [-1, 0, 0, 0] and torch.equal(out["F"], F0)     # `fundamental` left untouched


def test_sequence_and_upload_arguments(ctx):
    """vslam_frontend_sequence needs at least two frames; the upload helpers reject null pointers; page-locked
    buffers round-trip through the copy stream."""
    lib, h = ctx.lib, ctx.handle
    z = C.c_void_p(0)
    bgr = torch.zeros((1, 64, 64, 3), dtype=torch.uint8, device="cuda")
    pat = torch.from_numpy(synth.brief_pattern()).cuda()
    ca, sa = synth.keypoint_rotation()
    p = ctx._params(50, ca, sa, pat)
    assert lib.vslam_frontend_sequence(h, C.c_void_p(bgr.data_ptr()), 1, 64, 64, 192, C.byref(p), 50, z, 8, C.c_float(10.0),
                                       z, z, z, z, z, z, z) == -1
    assert lib.vslam_upload_async(h, z, z, C.c_size_t(16)) == -1
    assert lib.vslam_host_alloc(h, C.c_size_t(16), z) == -1
    # a real round trip: pinned -> device (copy stream) -> fence -> read back on the compute stream
    hp = C.c_void_p()
    assert lib.vslam_host_alloc(h, C.c_size_t(4096), C.byref(hp)) == 0
    src = (C.c_uint8 * 4096).from_address(hp.value)
    for i in range(4096):
        src[i] = (i * 7) & 0xFF
    dst = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    assert lib.vslam_upload_async(h, C.c_void_p(dst.data_ptr()), hp, C.c_size_t(4096)) == 0
    assert lib.vslam_upload_fence(h) == 0
    assert lib.vslam_upload_wait(h) == 0
    ctx.synchronize()
    assert np.array_equal(dst.cpu().numpy(), (np.arange(4096) * 7 & 0xFF).astype(np.uint8))
    assert lib.vslam_host_free(h, hp) == 0


def test_pack_records_kernel_equals_the_python_packing(ctx):
    """vslam_pack_records writes the record layout shard.pack_records / unpack_records define."""
    from vslam_amd import shard
    P, K = 5, 333
    g = torch.Generator().manual_seed(3)
    F = torch.randn((P, 9), generator=g).cuda()
    F[1, 2] = float("nan")
    best = torch.randint(-1, 5000, (P, 4), generator=g, dtype=torch.int32).cuda()
    m = torch.randint(0, 16384, (P, K, 2), generator=g, dtype=torch.int32).cuda()
    rec = ctx.pack_records(F, best, m)
    ctx.synchronize()
    assert torch.equal(rec, shard.pack_records(F, best, m))
    F2, b2, m2 = shard.unpack_records(rec, K)
    assert torch.equal(F2.view(torch.int32), F.view(torch.int32)) and torch.equal(b2, best) and torch.equal(m2, m)
