"""Several devices behind the C ABI, rehearsed on the one GPU there is: N contexts on device 0.

* vslam_multi_frontend_pairs over 1, 2, 3 and 5 slots (even and uneven slices): the records equal the single-context
  result and the oracle's, bit for bit -- per-pair seeds are base ^ GLOBAL pair index, so the split cannot be seen.
* the same through the C++ surface (include/vslam/MultiDevice.h, tests/native/multi_demo.cpp).
* vslam_gather_records (RCCL all-gather on the context's stream) with the one rank a one-GPU box allows.
Nothing here says anything about N > 1 real devices: that has not been measured anywhere."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from vslam_amd import build, capi, shard, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, MAXC, HYP, THR, SEED, P = 320, 240, 300, 64, 10.0, 0xBEEF, 7


@pytest.fixture(scope="module")
def batch():
    bgr = synth.frames_numpy(123, P, W, H)
    return np.ascontiguousarray(bgr[:P]), np.ascontiguousarray(bgr[P:])


@pytest.fixture(scope="module")
def reference(batch, oracle):
    last, cur = batch
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    out = []
    for i in range(P):
        a = oracle.extract_features(last[i], MAXC, ca, sa, pat)
        b = oracle.extract_features(cur[i], MAXC, ca, sa, pat)
        r = oracle.match_features(a["xy"], a["desc"], b["xy"], b["desc"], SEED ^ i, HYP, THR)
        out.append((a["n"], b["n"], r))
    return out


def check_records(rec, n, reference):
    Fm, best, matches = shard.unpack_records(torch.from_numpy(rec), MAXC)
    Fm, best, matches = Fm.numpy(), best.numpy(), matches.numpy()
    for i, (na, nb, r) in enumerate(reference):
        k = len(r["matches"])
        assert (n[i], n[P + i]) == (na, nb), i
        assert best[i, 3] == k and np.array_equal(matches[i, :k], r["matches"]), i
        if r["rc"] == 0:
            assert np.array_equal(Fm[i].view(np.uint32), r["F"].view(np.uint32)), i


@pytest.mark.parametrize("slots", [[0], [0, 0], [0, 0, 0], [0, 0, 0, 0, 0]])
def test_multi_device_records_do_not_depend_on_the_split(batch, reference, slots):
    last, cur = batch
    ca, sa = synth.keypoint_rotation()
    md = capi.MultiDevice(slots)
    try:
        assert md.size() == len(slots)
        rec, n = md.frontend_pairs(last, cur, MAXC, ca, sa, None, SEED, HYP, THR)
        check_records(rec, n, reference)
        rec2, n2 = md.frontend_pairs(last, cur, MAXC, ca, sa, synth.brief_pattern(), SEED, HYP, THR)   # explicit table, reused buffers
        assert np.array_equal(rec[:, :13], rec2[:, :13]) and np.array_equal(n, n2)
        check_records(rec2, n2, reference)
    finally:
        md.close()


def test_more_slots_than_pairs(batch, reference):
    """Three pairs over five slots: two slices are empty and their contexts stay idle."""
    last, cur = batch
    ca, sa = synth.keypoint_rotation()
    md = capi.MultiDevice([0] * 5)
    try:
        rec, n = md.frontend_pairs(last[:3], cur[:3], MAXC, ca, sa, None, SEED, HYP, THR)
    finally:
        md.close()
    Fm, best, matches = shard.unpack_records(torch.from_numpy(rec), MAXC)
    for i in range(3):
        r = reference[i][2]
        k = len(r["matches"])
        assert int(best[i, 3]) == k and np.array_equal(matches[i, :k].numpy(), r["matches"]), i
        assert np.array_equal(Fm[i].numpy().view(np.uint32), r["F"].view(np.uint32)), i


@pytest.mark.parametrize("slots", [[0], [0, 0, 0], [0] * 9])
def test_multi_device_resident_frames(batch, reference, slots):
    """Frames already on the slots' devices (vslam_multi_frontend_pairs_resident): same records as from host memory; with
    nine slots over seven pairs two slices are empty and their pointers NULL."""
    last, cur = batch
    ca, sa = synth.keypoint_rotation()
    md = capi.MultiDevice(slots)
    try:
        slices = []
        for r in range(len(slots)):
            lo, hi = shard.shard_range(P, r, len(slots))
            slices.append(torch.from_numpy(np.concatenate([last[lo:hi], cur[lo:hi]])).cuda() if hi > lo else None)
        torch.cuda.synchronize()
        rec, n = md.frontend_pairs_resident(slices, P, MAXC, ca, sa, None, SEED, HYP, THR)
        check_records(rec, n, reference)
        rec_h, n_h = md.frontend_pairs(last, cur, MAXC, ca, sa, None, SEED, HYP, THR)
        assert np.array_equal(rec_h[:, :13], rec[:, :13]) and np.array_equal(n, n_h)
        if len(slots) > 1:
            slices[1] = None                                      # a slot with pairs but no frames is refused
            with pytest.raises(capi.VslamError, match="INVALID"):
                md.frontend_pairs_resident(slices, P, MAXC, ca, sa, None, SEED, HYP, THR)
    finally:
        md.close()


def test_multi_device_host_frames_in_chunks(oracle):
    """More pairs per slot than one upload chunk (64): chunk k + 1 uploads while chunk k computes; the records are those of
    one call over the whole slice (a pair's result does not depend on its batch)."""
    w, h, maxc, n_pairs = 160, 120, 100, 150
    base = synth.frames_numpy(321, 6, w, h)
    idx = np.arange(n_pairs) % 6
    last, cur = np.ascontiguousarray(base[:6][idx]), np.ascontiguousarray(base[6:][idx])
    ca, sa = synth.keypoint_rotation()
    pat = synth.brief_pattern()
    md = capi.MultiDevice([0, 0])
    try:
        rec, n = md.frontend_pairs(last, cur, maxc, ca, sa, None, 5, HYP, THR)
    finally:
        md.close()
    Fm, best, matches = shard.unpack_records(torch.from_numpy(rec), maxc)
    feats = [(oracle.extract_features(base[i], maxc, ca, sa, pat), oracle.extract_features(base[6 + i], maxc, ca, sa, pat)) for i in range(6)]
    for i in list(range(0, n_pairs, 7)) + [63, 64, 74, 75, 138, 139, 149]:       # chunk and slice borders included
        a, b = feats[idx[i]]
        r = oracle.match_features(a["xy"], a["desc"], b["xy"], b["desc"], 5 ^ i, HYP, THR)
        k = len(r["matches"])
        assert (n[i], n[n_pairs + i]) == (a["n"], b["n"]), i
        assert int(best[i, 3]) == k and np.array_equal(matches[i, :k].numpy(), r["matches"]), i
        if r["rc"] == 0:
            assert np.array_equal(Fm[i].numpy().view(np.uint32), r["F"].view(np.uint32)), i


def test_multi_device_rejects_bad_arguments(batch):
    last, cur = batch
    with pytest.raises(capi.VslamError):
        capi.MultiDevice([0, 99])
    md = capi.MultiDevice([0])
    try:
        with pytest.raises(capi.VslamError):
            md.frontend_pairs(last[:, :, :0], cur[:, :, :0], MAXC, 1.0, 0.0, None, 1, HYP, THR)
    finally:
        md.close()


def test_device_pool_cpp_surface(batch, reference, tmp_path):
    last, cur = batch
    build.build_host()
    exe = str(tmp_path / "multi_demo")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "native", "multi_demo.cpp"),
                    "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "vslam_amd"), "-lvslam_amd",
                    "-Wl,-rpath," + os.path.join(ROOT, "vslam_amd")], check=True)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    slots = [0, 0, 0]
    with open(fin, "wb") as f:
        f.write(struct.pack("7i", W, H, MAXC, HYP, SEED, P, len(slots)))
        f.write(struct.pack(f"{len(slots)}i", *slots))
        f.write(last.tobytes())
        f.write(cur.tobytes())
    subprocess.run([exe, fin, fout], check=True, timeout=120)
    buf = open(fout, "rb").read()
    off = 0
    for i, (_, _, r) in enumerate(reference):
        winner, inl, k = struct.unpack_from("3i", buf, off); off += 12
        Fm = np.frombuffer(buf, np.float32, 9, off); off += 36
        m = np.frombuffer(buf, np.int32, 2 * k, off).reshape(k, 2); off += 8 * k
        assert k == len(r["matches"]) and np.array_equal(m, r["matches"]), i
        if r["rc"] == 0:
            assert np.array_equal(Fm.view(np.uint32), r["F"].view(np.uint32)) and inl == k, i
    assert off == len(buf)


def test_gather_records_over_rccl_single_rank(ctx):
    """ncclAllGather through the C ABI on the context's stream; world = 1 is what one GPU allows (two ranks on one device are
    refused by RCCL), so this checks the plumbing -- library found, communicator made, call enqueued on the right stream --
    not a transfer between devices."""
    lib = ctx.lib
    uid = (ctypes.c_ubyte * 128)()
    assert lib.vslam_comm_unique_id(uid) == 0
    comm = ctypes.c_void_p()
    rc = lib.vslam_comm_create(ctx.handle, uid, 1, 0, ctypes.byref(comm))
    assert rc == 0, lib.vslam_last_error(ctx.handle)
    try:
        rec = torch.arange(5 * (13 + 40), dtype=torch.int32, device="cuda").reshape(5, 53) * 7 - 3
        out = torch.zeros_like(rec)
        assert lib.vslam_gather_records(ctx.handle, comm, ctypes.c_void_p(rec.data_ptr()), ctypes.c_size_t(rec.numel()),
                                        ctypes.c_void_p(out.data_ptr())) == 0
        ctx.synchronize()
        assert torch.equal(out, rec)
        assert lib.vslam_gather_records(ctx.handle, comm, None, ctypes.c_size_t(4), ctypes.c_void_p(out.data_ptr())) == -1
        world, rank = ctypes.c_int(-1), ctypes.c_int(-1)
        assert lib.vslam_comm_info(comm, ctypes.byref(world), ctypes.byref(rank)) == 0 and (world.value, rank.value) == (1, 0)
        # the form with per-rank counts, all-gather (root -1) and rooted (root 0): with one rank the own block is all there is
        words = (ctypes.c_size_t * 1)(rec.numel() - 11)
        for root in (-1, 0):
            out.zero_()
            assert lib.vslam_gather_records_v(ctx.handle, comm, ctypes.c_void_p(rec.data_ptr()), words, ctypes.c_int(root),
                                              ctypes.c_void_p(out.data_ptr())) == 0
            ctx.synchronize()
            assert torch.equal(out.view(-1)[:rec.numel() - 11], rec.view(-1)[:rec.numel() - 11]) and int(out.view(-1)[-11:].abs().sum()) == 0
        assert lib.vslam_gather_records_v(ctx.handle, comm, ctypes.c_void_p(rec.data_ptr()), words, ctypes.c_int(1),
                                          ctypes.c_void_p(out.data_ptr())) == -1       # no such root
    finally:
        assert lib.vslam_comm_destroy(comm) == 0
