"""Randomised parity of the RANSAC kernels: coordinate scales from 1e-12 to 1e12, duplicated / collinear / zero
points, small and large match counts, thresholds from 1e-6 to 1e6 — per-hypothesis F, counts, sums, winner, mask and
final F must equal the oracle's bit for bit.  Extreme scales push the Jacobi rotation out of the range where the
short f64 sqrt / division sequences apply and the convergence test into its tie branch, so both alternatives of
every wave-uniform branch get exercised.

`python tests/fuzz_ransac.py SEED SECONDS` runs for a wall-clock budget; tests/test_gpu_fuzz.py runs a fixed slice."""
import sys
import time

import numpy as np
import torch


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def run(ctx, o, seed, cases=None, seconds=None):
    """Cases alternate between the two scoring paths (default: sums of the maximum-count hypotheses only;
    VSLAM_OPT_RANSAC_ALL_SUMS: every sum)."""
    rng = np.random.default_rng(seed)
    t0, done = time.time(), 0
    while (cases is None or done < cases) and (seconds is None or time.time() - t0 < seconds):
        B = int(rng.integers(1, 5))
        K = int(rng.choice([16, 64, 200, 200, 700, 1600]))   # up to several 256-match sub-blocks of the counting kernel
        Hy = int(rng.choice([8, 64, 130]))
        thr = float(10.0 ** rng.uniform(-6, 6)) if rng.random() < 0.5 else 10.0
        xy1 = np.zeros((B, K, 2), np.float32)
        xy2 = np.zeros((B, K, 2), np.float32)
        m = np.zeros(B, np.int32)
        for b in range(B):
            n = int(rng.integers(8, K + 1))
            m[b] = n
            scale = np.float32(10.0 ** rng.uniform(-12, 12)) if rng.random() < 0.4 else np.float32(1.0)
            kind = int(rng.integers(0, 6))
            p = rng.uniform(0, 1000, (K, 2))
            if kind == 0:      # integer pixels, small motion + noise (the usual case)
                a = np.rint(p); c = np.rint(p * 1.01 + rng.normal(0, 1.5, (K, 2)) + 5)
            elif kind == 1:    # collinear
                s = rng.uniform(0, 1000, K); a = np.stack([s, 2 * s + 1], 1); c = a + rng.normal(0, 0.5, (K, 2))
            elif kind == 2:    # heavy duplication
                a = np.rint(p[rng.integers(0, 4, K)]); c = np.rint(a + rng.integers(-1, 2, (K, 2)))
            elif kind == 3:    # identical frames
                a = np.rint(p); c = a.copy()
            elif kind == 4:    # zeros and a few points
                a = np.zeros((K, 2)); c = np.zeros((K, 2)); a[: K // 4] = p[: K // 4]; c[: K // 4] = p[: K // 4] + 1
            else:              # unrelated point sets, real-valued
                a = p; c = rng.uniform(0, 1000, (K, 2))
            xy1[b] = (a * scale).astype(np.float32)
            xy2[b] = (c * scale).astype(np.float32)
        pairs = np.tile(np.stack([np.arange(K), np.arange(K)], 1)[None], (B, 1, 1)).astype(np.int32)
        for b in range(B):
            pairs[b, :, 1] = rng.permutation(K) if rng.random() < 0.3 else pairs[b, :, 1]
        sets = np.stack([o.ransac_sets(int(rng.integers(0, 2 ** 31)), int(m[b]), Hy) for b in range(B)])
        t = lambda a: torch.from_numpy(a).cuda()
        all_sums = done % 2 == 1
        ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, all_sums)
        out = ctx.ransac_fundamental(t(xy1), t(xy2), t(pairs), t(m), t(sets), thr)
        ctx.synchronize()
        out = {k: v.cpu().numpy() for k, v in out.items()}
        for b in range(B):
            n = int(m[b])
            ref = o.find_fundamental(xy1[b], xy2[b], pairs[b, :n], sets[b], thr)
            tag = (done, b, n, Hy, thr)
            assert np.array_equal(bits(out["hypF"][b]), bits(ref["hypF"])), ("hypF",) + tag
            from test_gpu_ransac import check_counts, check_sums
            check_counts(out["hyp_count"][b], ref["hyp_count"], "all" if all_sums else "ties", ("count",) + tag, ref["hyp_sum"])
            check_sums(out["hyp_sum"][b], ref["hyp_count"], ref["hyp_sum"], "all" if all_sums else "ties", ("sum",) + tag,
                       out["hyp_count"][b])
            assert out["best"][b, 0] == ref["winner"], ("winner",) + tag
            if ref["winner"] >= 0:
                assert out["best"][b, 1] == ref["count"], ("best count",) + tag
                assert np.array_equal(bits(out["F"][b]), bits(ref["F"])), ("F",) + tag
                assert np.array_equal(out["mask"][b, :n], ref["mask"]), ("mask",) + tag
        done += 1
        if seconds is not None and done % 200 == 0:   # a long run says so as it goes
            print("fuzz_ransac: %d cases ok" % done, flush=True)
    ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, False)
    return done


if __name__ == "__main__":
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    from oracle_lib import Oracle
    from vslam_amd import Context
    n = run(Context(0), Oracle(), int(sys.argv[1]) if len(sys.argv) > 1 else 1, seconds=float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
    print("fuzz ok:", n, "cases")
