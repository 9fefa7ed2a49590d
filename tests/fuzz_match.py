"""Randomised parity of the matcher: ragged batches of random descriptor sets with planted copies, near copies, repeated
train rows (distance ties: the lower index must win), all-zero / all-one / complementary rows, sizes from 0 to a few
hundred rows, every workgroup shape and matrix-core form; indices, distances and the ratio-test survivors must equal the
oracle's (src/Frame.cpp:83-94).

`python tests/fuzz_match.py SEED SECONDS` runs it for a wall-clock budget; tests/test_gpu_fuzz.py runs a fixed slice."""
import sys
import time

import numpy as np
import torch


def run(ctx, o, seed, cases=None, seconds=None, variants=True):
    """variants: draw a workgroup shape and a matrix-core form per case (a context of the EXPERIMENTS build; the product
    library carries one matcher and refuses the options)."""
    rng = np.random.default_rng(seed)
    t0, done = time.time(), 0
    while (cases is None or done < cases) and (seconds is None or time.time() - t0 < seconds):
        B = int(rng.integers(1, 5))
        K = int(rng.choice([40, 130, 300, 700]))
        items = []
        for _ in range(B):
            n1 = int(rng.choice([0, 1, 2, 3, 31, 32, 33, 64, 100, K]))
            n2 = int(rng.choice([0, 1, 2, 3, 31, 32, 33, 63, 65, 100, K]))
            n1, n2 = min(n1, K), min(n2, K)
            a = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
            t = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
            for _k in range(int(rng.integers(0, 12))):           # plant structure
                if n1 == 0 or n2 == 0:
                    break
                i, j = int(rng.integers(n1)), int(rng.integers(n2))
                kind = int(rng.integers(0, 7))
                if kind == 0:
                    t[j] = a[i]                                   # exact copy
                elif kind == 1:
                    t[j] = a[i]
                    t[j, rng.integers(32)] ^= np.uint8(1 << rng.integers(8))   # one bit off
                elif kind == 2:
                    t[j] = ~a[i]                                  # distance 256
                elif kind == 3:
                    t[j] = t[int(rng.integers(n2))]               # repeated train row: a tie between indices
                elif kind == 4:
                    a[i] = 0 if rng.random() < 0.5 else 255
                elif kind == 5:
                    t[j] = 0 if rng.random() < 0.5 else 255
                else:
                    flips = rng.random(256) < rng.uniform(0.02, 0.4)
                    t[j] = a[i] ^ np.packbits(flips)
            items.append((a, t))
        d1 = np.zeros((B, K, 32), np.uint8); d2 = np.zeros((B, K, 32), np.uint8)
        n1 = np.zeros(B, np.int32); n2 = np.zeros(B, np.int32)
        for b, (a, t) in enumerate(items):
            d1[b, :len(a)] = a; d2[b, :len(t)] = t
            n1[b], n2[b] = len(a), len(t)
        shape_, form_ = int(rng.integers(0, 3)), int(rng.integers(0, 3))   # (drawn either way: the case stream does not depend on `variants`)
        if variants:
            ctx.set_option(ctx.OPT_MATCH_SHAPE, shape_)
            ctx.set_option(ctx.OPT_MATCH_FORM, form_)
        try:
            pairs, m, knn = ctx.match_knn2_ratio(torch.from_numpy(d1).cuda(), torch.from_numpy(n1).cuda(), torch.from_numpy(d2).cuda(),
                                                 torch.from_numpy(n2).cuda(), want_knn=True)
            ctx.synchronize()
        finally:
            if variants:
                ctx.set_option(ctx.OPT_MATCH_SHAPE, 0)
                ctx.set_option(ctx.OPT_MATCH_FORM, 0)
        pairs, m, knn = pairs.cpu().numpy(), m.cpu().numpy(), knn.cpu().numpy()
        for b, (a, t) in enumerate(items):
            if len(t) >= 2 and len(a) >= 1:
                i0, e0, i1, e1 = o.match_knn2(a, t)
                g = knn[b, :len(a)]
                assert np.array_equal(g[:, 0], i0) and np.array_equal(g[:, 1], e0), ("knn best", done, b, len(a), len(t))
                assert np.array_equal(g[:, 2], i1) and np.array_equal(g[:, 3], e1), ("knn second", done, b, len(a), len(t))
                ref, rc = o.match_knn2_ratio(a, t)
                assert rc == 0 and m[b] == len(ref) and np.array_equal(pairs[b, :m[b]], ref), ("pairs", done, b)
            else:
                assert m[b] == 0, ("degenerate", done, b, len(a), len(t))
        done += 1
        if seconds is not None and done % 200 == 0:
            print(f"fuzz_match: {done} cases ok, {time.time() - t0:.0f} s", flush=True)
    return done


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_lib import Oracle
    from vslam_amd import Context, capi
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    print("fuzz ok (product):", run(Context(0), Oracle(), seed, seconds=secs / 2, variants=False), "cases")
    print("fuzz ok (experiments build, all variants):",
          run(Context(0, lib=capi.load_library(capi.EXP_LIB_PATH)), Oracle(), seed + 1, seconds=secs / 2), "cases")
