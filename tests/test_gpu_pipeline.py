"""Batches in flight behind the C ABI (vslam_pipeline_*, include/vslam_amd.h; C++: include/vslam/Pipeline.h).

* the pipeline's output for every batch == the same batch on a single context == the oracle, bit for bit, for queues whose
  length is not a multiple of the number of contexts and batches of different sizes (workspaces regrow between tickets);
* a batch that overflows (VSLAM_ERR_CAPACITY) reports it on ITS ticket; the tickets before and behind it -- the next batch
  of the same context included -- are clean and exact; a bad argument is refused at once and costs nothing;
* the consecutive-frames form, poll(), drain();
* the C++ surface through tests/native/pipeline_demo.cpp."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from vslam_amd import build, capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, MAXC, HYP, THR, SEED = 320, 240, 300, 64, 10.0, 0xFACE
SIZES = [2, 1, 3, 1, 4, 2, 1]          # seven batches, 14 pairs


@pytest.fixture(scope="module")
def batches():
    out, first = [], 0
    for i, n in enumerate(SIZES):
        bgr = synth.frames_numpy(500 + i, n, W, H)
        seeds = (np.uint32(SEED) ^ np.arange(first, first + n, dtype=np.uint32))
        out.append((bgr, seeds))
        first += n
    return out


@pytest.fixture(scope="module")
def reference(batches, oracle):
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    ref = []
    for bgr, seeds in batches:
        n = bgr.shape[0] // 2
        rows = []
        for i in range(n):
            a = oracle.extract_features(bgr[i], MAXC, ca, sa, pat)
            b = oracle.extract_features(bgr[n + i], MAXC, ca, sa, pat)
            rows.append((a, b, oracle.match_features(a["xy"], a["desc"], b["xy"], b["desc"], int(seeds[i]), HYP, THR)))
        ref.append(rows)
    return ref


def check_batch(out, rows):
    n = len(rows)
    o = {k: v.cpu().numpy() for k, v in out.items()}
    for i, (a, b, r) in enumerate(rows):
        k = len(r["matches"])
        assert (o["n"][i], o["n"][n + i]) == (a["n"], b["n"]), i
        assert np.array_equal(o["xy"][i, :a["n"]], a["xy"]) and np.array_equal(o["desc"][n + i, :b["n"]], b["desc"]), i
        assert np.array_equal(o["nodes"][i, :a["n"]], a["nodes"]) and np.array_equal(o["nodes"][n + i, :b["n"]], b["nodes"]), i
        assert o["best"][i, 3] == k and np.array_equal(o["matches"][i, :k], r["matches"]), i
        if r["rc"] == 0:
            assert np.array_equal(o["F"][i].view(np.uint32), r["F"].view(np.uint32)), i


@pytest.mark.parametrize("n_ctx", [1, 2, 3, 5])
def test_pipeline_equals_single_context_and_oracle(ctx, batches, reference, n_ctx):
    pat = torch.from_numpy(synth.brief_pattern()).cuda()
    ca, sa = synth.keypoint_rotation()
    dev = [(torch.from_numpy(b).cuda(), torch.from_numpy(s.view(np.int32)).cuda()) for b, s in batches]
    pipe = capi.Pipeline(0, n_ctx)
    try:
        assert pipe.size() == n_ctx
        outs, recs, tickets = [], [], []
        for bgr, seeds in dev:
            n = bgr.shape[0] // 2
            o = capi.Pipeline.alloc_outputs(torch, 2 * n, n, MAXC, bgr.device)
            r = torch.zeros((n, 13 + MAXC), dtype=torch.int32, device=bgr.device)
            torch.cuda.synchronize()       # the pipeline's streams do not wait for torch's fill kernels
            tickets.append(pipe.submit_pairs(bgr, n, MAXC, ca, sa, pat if len(outs) % 2 else None, seeds, HYP, THR, o, records=r))
            outs.append(o)
            recs.append(r)
        assert tickets == list(range(len(SIZES)))
        pipe.wait(tickets[3])               # out of order on purpose
        assert pipe.poll(tickets[3]) and pipe.poll(tickets[0])
        pipe.drain()
        assert all(pipe.poll(t) for t in tickets)
        for (bgr, seeds), o, r, rows in zip(dev, outs, recs, reference):
            n = bgr.shape[0] // 2
            check_batch(o, rows)
            single = ctx.frontend_pairs(bgr, n, MAXC, ca, sa, pat, seeds, HYP, THR)
            ctx.synchronize()
            for k in ("n", "best", "F", "xy", "desc", "nodes"):
                assert torch.equal(single[k], o[k]), k
            assert torch.equal(ctx.pack_records(o["F"], o["best"], o["matches"]), r)
        assert pipe.workspace_bytes() > 0
    finally:
        pipe.close()


def test_an_error_in_one_batch_does_not_poison_the_next(oracle):
    """Ticket 1 overflows the corner lists on more frames than the fallback pool has sets (a bound of 40 entries, 8 frames,
    4 sets): its wait() says VSLAM_ERR_CAPACITY.  Tickets 0, 2 and 3 -- 3 runs on the same context right behind it -- are
    clean and exact, and so is ticket 1's own repeat with the default bound."""
    w, h, maxc = 320, 240, 60
    ca, sa = synth.keypoint_rotation()
    bgr_np = synth.frames_numpy(77, 4, w, h)                  # 4 pairs = 8 textured frames
    bgr = torch.from_numpy(bgr_np).cuda()
    seeds_np = np.arange(4, dtype=np.uint32) + 9
    seeds = torch.from_numpy(seeds_np.view(np.int32)).cuda()
    pat = synth.brief_pattern()
    rows = []
    for i in range(4):
        a = oracle.extract_features(bgr_np[i], maxc, ca, sa, pat)
        b = oracle.extract_features(bgr_np[4 + i], maxc, ca, sa, pat)
        rows.append((a, b, oracle.match_features(a["xy"], a["desc"], b["xy"], b["desc"], int(seeds_np[i]), HYP, THR)))

    def check(o):
        o = {k: v.cpu().numpy() for k, v in o.items()}
        for i, (a, b, r) in enumerate(rows):
            k = len(r["matches"])
            assert (o["n"][i], o["n"][4 + i]) == (a["n"], b["n"]), i
            assert o["best"][i, 3] == k and np.array_equal(o["matches"][i, :k], r["matches"]), i

    pipe = capi.Pipeline(0, 2)
    try:
        outs = [capi.Pipeline.alloc_outputs(torch, 8, 4, maxc, bgr.device) for _ in range(5)]
        torch.cuda.synchronize()
        t0 = pipe.submit_pairs(bgr, 4, maxc, ca, sa, None, seeds, HYP, THR, outs[0])
        t1, c1 = pipe.acquire()                                # slot 1, by hand: the bound applies to this batch only
        c1.set_option(c1.OPT_CORNER_LIST_CAP, 40)
        c1.frontend_pairs(bgr, 4, maxc, ca, sa, None, seeds, HYP, THR, out=outs[1])
        c1.set_option(c1.OPT_CORNER_LIST_CAP, 0)
        pipe.commit(t1)
        t2 = pipe.submit_pairs(bgr, 4, maxc, ca, sa, None, seeds, HYP, THR, outs[2])
        t3 = pipe.submit_pairs(bgr, 4, maxc, ca, sa, None, seeds, HYP, THR, outs[3])   # same context as t1 (retires t1 first)
        assert (t0, t1, t2, t3) == (0, 1, 2, 3)
        # a bad argument is refused at once, takes no ticket's worth of state with it
        with pytest.raises(capi.VslamError, match="INVALID"):
            pipe.submit_pairs(bgr[:0], 0, maxc, ca, sa, None, seeds, HYP, THR, outs[4])
        t5 = pipe.submit_pairs(bgr, 4, maxc, ca, sa, None, seeds, HYP, THR, outs[4])
        assert pipe.wait_status(t3)[0] == 0 and pipe.wait_status(t0)[0] == 0 and pipe.wait_status(t2)[0] == 0
        rc, msg = pipe.wait_status(t1)
        assert rc == -4 and "ticket 1" in msg and "overflowed" in msg, (rc, msg)
        assert pipe.wait_status(t1)[0] == 0                   # collected: reported once
        assert pipe.wait_status(t5)[0] == 0
        pipe.drain()                                           # nothing left to report
        for i in (0, 2, 3, 4):
            check(outs[i])
        n1 = outs[1]["n"].cpu().numpy()
        assert (n1 == 0).sum() == 4                            # the four frames that found no whole-image set
    finally:
        pipe.close()


def test_drain_reports_the_first_uncollected_failure():
    w, h, maxc = 320, 240, 60
    ca, sa = synth.keypoint_rotation()
    bgr = torch.from_numpy(synth.frames_numpy(78, 4, w, h)).cuda()
    seeds = torch.arange(4, dtype=torch.int32).cuda()
    pipe = capi.Pipeline(0, 3)
    try:
        pipe.set_option(capi.Context.OPT_CORNER_LIST_CAP, 40)
        outs = [capi.Pipeline.alloc_outputs(torch, 8, 4, maxc, bgr.device) for _ in range(2)]
        for o_ in outs:   # tickets built by hand: the pipeline cannot queue these again, their overflow is reported
            t_, c_ = pipe.acquire()
            c_.frontend_pairs(bgr, 4, maxc, ca, sa, None, seeds, HYP, THR, out=o_)
            pipe.commit(t_)
        with pytest.raises(capi.VslamError, match=r"CAPACITY: ticket 0: .*\+1 more"):
            pipe.drain()
        pipe.drain()
        t, c = pipe.acquire()
        with pytest.raises(capi.VslamError, match="open"):
            pipe.drain()
        pipe.commit(t)
        with pytest.raises(capi.VslamError, match="INVALID"):
            pipe.commit(t)
        pipe.drain()
    finally:
        pipe.close()


def test_pipeline_sequence_form(ctx):
    pat = torch.from_numpy(synth.brief_pattern()).cuda()
    ca, sa = synth.keypoint_rotation()
    clips = [torch.from_numpy(np.ascontiguousarray(synth.frames_numpy(600 + i, n, W, H)[:n + 1])).cuda() for i, n in enumerate((3, 5, 2, 4))]
    pipe = capi.Pipeline(0, 3)
    try:
        outs = []
        for i, clip in enumerate(clips):
            f = clip.shape[0]
            seeds = torch.arange(10 * i, 10 * i + f - 1, dtype=torch.int32).cuda()
            o = capi.Pipeline.alloc_outputs(torch, f, f - 1, MAXC, clip.device)
            torch.cuda.synchronize()
            pipe.submit_sequence(clip, MAXC, ca, sa, pat, seeds, HYP, THR, o)
            outs.append((clip, seeds, o))
        pipe.drain()
        for clip, seeds, o in outs:
            single = ctx.frontend_sequence(clip, MAXC, ca, sa, pat, seeds, HYP, THR)
            ctx.synchronize()
            for k in ("n", "best", "F", "xy", "desc", "nodes"):
                assert torch.equal(single[k], o[k]), k
    finally:
        pipe.close()


def test_pipeline_cpp_surface(batches, reference, tmp_path):
    build.build_host()
    exe = str(tmp_path / "pipeline_demo")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "native", "pipeline_demo.cpp"),
                    "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "vslam_amd"), "-lvslam_amd",
                    "-Wl,-rpath," + os.path.join(ROOT, "vslam_amd")], check=True)
    # one contiguous run of pairs; the demo cuts it into batches of 1, 2, 3, 1, 2, 3 ... with seeds SEED ^ global pair index
    last = np.concatenate([b[:b.shape[0] // 2] for b, _ in batches])
    cur = np.concatenate([b[b.shape[0] // 2:] for b, _ in batches])
    rows = [r for rr in reference for r in rr]
    P = last.shape[0]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("7i", W, H, MAXC, HYP, SEED, P, 3))
        f.write(last.tobytes())
        f.write(cur.tobytes())
    subprocess.run([exe, fin, fout], check=True, timeout=120)
    buf = open(fout, "rb").read()
    off = 0
    for i, (_, _, r) in enumerate(rows):
        winner, inl, k = struct.unpack_from("3i", buf, off); off += 12
        Fm = np.frombuffer(buf, np.float32, 9, off); off += 36
        m = np.frombuffer(buf, np.int32, 2 * k, off).reshape(k, 2); off += 8 * k
        assert k == len(r["matches"]) and np.array_equal(m, r["matches"]), i
        if r["rc"] == 0:
            assert np.array_equal(Fm.view(np.uint32), r["F"].view(np.uint32)) and inl == k, i
    assert off == len(buf)


def test_the_cpp_example_runs(tmp_path):
    """examples/batches_in_flight.cpp builds against the installed headers and library and collects every record."""
    exe = str(tmp_path / "batches_in_flight")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "examples", "batches_in_flight.cpp"),
                    "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "vslam_amd"), "-lvslam_amd",
                    "-Wl,-rpath," + os.path.join(ROOT, "vslam_amd")], check=True)
    r = subprocess.run([exe, "7", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "21 records" in r.stdout
    first = [ln for ln in r.stdout.splitlines() if ln.startswith("pair 0:")]
    assert first and "hypothesis -1" not in first[0], r.stdout       # the shifted scene gives a model


def test_submitted_batch_that_exhausts_the_corner_pool_is_done_again(oracle):
    """More frames of one batch need the corner detector's whole-image fallback than its pool holds (here: pure noise, every
    frame): a batch that came through submit_pairs is queued once more with whole-image lists when its status is collected,
    reports VSLAM_OK, and its outputs are those of an unbounded run -- and of the oracle."""
    w, h, maxc, P = 320, 240, 200, 12
    ca, sa = synth.keypoint_rotation()
    rng = np.random.default_rng(11)
    bgr_np = rng.integers(0, 256, (2 * P, h, w, 3), dtype=np.uint8)
    bgr = torch.from_numpy(bgr_np).cuda()
    seeds_np = np.arange(P, dtype=np.int32) + 77
    seeds = torch.from_numpy(seeds_np).cuda()
    pipe = capi.Pipeline(0, 2)
    try:
        out = capi.Pipeline.alloc_outputs(torch, 2 * P, P, maxc, bgr.device)
        ref = capi.Pipeline.alloc_outputs(torch, 2 * P, P, maxc, bgr.device)
        torch.cuda.synchronize()
        pipe.set_option(capi.Context.OPT_CORNER_LIST_CAP, 40)   # (the test knob: lists of 40 entries, so every frame of this small batch overflows)
        t = pipe.submit_pairs(bgr, P, maxc, ca, sa, None, seeds, HYP, THR, out)
        assert pipe.wait_status(t)[0] == 0
        assert pipe.batches_redone() == 1
        pipe.set_option(capi.Context.OPT_CORNER_LIST_CAP, 0)
        # the same batch with nothing bounded, by hand
        t2, c2 = pipe.acquire()
        c2.set_option(c2.OPT_CORNER_LIST_CAP, -1)
        c2.frontend_pairs(bgr, P, maxc, ca, sa, None, seeds, HYP, THR, out=ref)
        c2.set_option(c2.OPT_CORNER_LIST_CAP, 0)
        pipe.commit(t2)
        assert pipe.wait_status(t2)[0] == 0 and pipe.batches_redone() == 1
        o = {k: v.cpu().numpy() for k, v in out.items()}
        r = {k: v.cpu().numpy() for k, v in ref.items()}
        for k in ("n", "xy", "desc", "nodes", "best", "matches"):
            assert np.array_equal(o[k], r[k]), k
        assert np.array_equal(o["F"].view(np.uint32), r["F"].view(np.uint32))
        assert (o["n"] > 50).all()            # every frame has its corners: none came back empty
        pat = synth.brief_pattern()
        a = oracle.extract_features(bgr_np[0], maxc, ca, sa, pat)
        assert o["n"][0] == a["n"] and np.array_equal(o["xy"][0, :a["n"]], a["xy"]) and np.array_equal(o["desc"][0, :a["n"]], a["desc"])
    finally:
        pipe.close()


def test_pose_chain_as_tickets(ctx):
    """vslam_pipeline_submit_pairs_pose: every batch's pose outputs (R, t, c2, triangulated points, reprojection filter) equal the
    same call on a single context, with three batches in flight and a queue longer than the pipeline; a batch that exhausts the
    corner pool is done again, pose stages included."""
    w, h, maxc, P = 320, 240, 300, 3
    ca, sa = synth.keypoint_rotation()
    Kmat = np.array([[525.0, 0, w // 2], [0, 525.0, h // 2], [0, 0, 1]], np.float32)
    frames = [torch.from_numpy(synth.frames_numpy(900 + i, P, w, h)).cuda() for i in range(5)]
    seeds = [torch.from_numpy((np.arange(P, dtype=np.int32) + 31 * i)).cuda() for i in range(5)]
    refs = []
    for fr, sd in zip(frames, seeds):
        o = ctx.frontend_pairs_pose(fr, P, maxc, ca, sa, None, sd, HYP, THR, Kmat)
        ctx.synchronize()
        refs.append({k: v.cpu().numpy() for k, v in o.items()})
    assert sum(int(r["n_inliers"].sum()) for r in refs) > 50, "the scenes should triangulate"

    def same(out, ref, tag):   # what a batch defines: rows up to its counts (the buffers are reused by later tickets)
        g = {k: v.cpu().numpy() for k, v in out.items()}
        bits = lambda a: np.ascontiguousarray(a).view(np.uint8)
        for k in ("n", "best", "F", "R", "t", "c2", "n_inliers", "error"):
            assert np.array_equal(bits(g[k]), bits(ref[k])), (tag, k)
        for f in range(2 * P):
            n = int(ref["n"][f])
            for k in ("xy", "desc", "nodes"):
                assert np.array_equal(bits(g[k][f, :n]), bits(ref[k][f, :n])), (tag, k, f)
        for b in range(P):
            m, ni = int(ref["best"][b, 3]), int(ref["n_inliers"][b])
            assert np.array_equal(g["matches"][b, :m], ref["matches"][b, :m]), (tag, "matches", b)
            if ref["best"][b, 0] >= 0:
                assert np.array_equal(bits(g["points4d"][b, :m]), bits(ref["points4d"][b, :m])), (tag, "points4d", b)
                assert np.array_equal(g["inlier_idx"][b, :ni], ref["inlier_idx"][b, :ni]), (tag, "inlier_idx", b)

    pipe = capi.Pipeline(0, 3)
    try:
        outs = [capi.Pipeline.alloc_pose_outputs(torch, 2 * P, P, maxc, frames[0].device) for _ in range(3)]
        tickets = []
        for i, (fr, sd) in enumerate(zip(frames, seeds)):
            if i >= 3:
                assert pipe.wait_status(tickets[i - 3])[0] == 0
                same(outs[(i - 3) % 3], refs[i - 3], i - 3)
            tickets.append(pipe.submit_pairs_pose(fr, P, maxc, ca, sa, None, sd, HYP, THR, Kmat, outs[i % 3]))
        for i in (2, 3, 4):
            assert pipe.wait_status(tickets[i])[0] == 0
            same(outs[i % 3], refs[i], i)
        # the overflow case: lists of 40 entries, every frame overflows, the pool holds fewer -> queued again, pose included
        pipe.set_option(capi.Context.OPT_CORNER_LIST_CAP, 40)
        t = pipe.submit_pairs_pose(frames[0], P, maxc, ca, sa, None, seeds[0], HYP, THR, Kmat, outs[0])
        assert pipe.wait_status(t)[0] == 0
        assert pipe.batches_redone() == 1      # six frames overflow, the pool of this batch holds four
        same(outs[0], refs[0], "redone")
    finally:
        pipe.close()


def test_the_python_pose_chain_example_runs():
    """examples/pose_chain.py: pose tickets on a pipeline, then the association block, end to end."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "pose_chain.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert sum(ln.startswith("batch ") for ln in lines) == 12 and sum(ln.startswith("association, pair") for ln in lines) == 4, r.stdout
    found = [int(ln.split(",")[-1].split()[0]) for ln in lines if ln.startswith("association")]
    assert max(found) > 10, r.stdout          # the shifted scenes triangulate and their points find their keypoints again
