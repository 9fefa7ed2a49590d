import numpy as np

from vslam_amd import synth


def test_ratio_identity_exhaustive():
    """`m0.distance < m1.distance * 0.7` (float vs float*double, src/Frame.cpp:91) equals the
    integer test 10*d0 < 7*d1 for every pair of Hamming distances the kernel can see."""
    for d0 in range(257):
        for d1 in range(d0, 257):
            ref = bool(np.float32(d0) < np.float64(np.float32(d1)) * 0.7)
            assert ref == (10 * d0 < 7 * d1), (d0, d1)


def test_knn2_against_numpy_bruteforce(oracle):
    d1, d2, _ = synth.descriptors_pair(11, 150, 170)
    d2[5] = d2[17]          # force distance ties between train rows
    d2[40] = d2[17]
    i0, e0, i1, e1 = oracle.match_knn2(d1, d2)
    bits1 = np.unpackbits(d1, axis=1).astype(np.int32)
    bits2 = np.unpackbits(d2, axis=1).astype(np.int32)
    D = (bits1[:, None, :] != bits2[None, :, :]).sum(-1)
    order = np.lexsort((np.arange(D.shape[1])[None, :].repeat(D.shape[0], 0), D), axis=1)   # by (dist, idx)
    assert np.array_equal(i0, order[:, 0]) and np.array_equal(i1, order[:, 1])
    assert np.array_equal(e0, D[np.arange(150), order[:, 0]])
    assert np.array_equal(e1, D[np.arange(150), order[:, 1]])


def test_ratio_pairs_are_in_query_order_and_recover_truth(oracle):
    d1, d2, truth = synth.descriptors_pair(3, 400, 420)
    pairs, rc = oracle.match_knn2_ratio(d1, d2)
    assert rc == 0 and len(pairs) > 150
    assert np.all(np.diff(pairs[:, 0]) > 0)
    assert np.mean(truth[pairs[:, 0]] == pairs[:, 1]) > 0.99


def test_degenerate_train_set_is_rejected(oracle):
    d1, d2, _ = synth.descriptors_pair(4, 10, 10)
    _, rc = oracle.match_knn2_ratio(d1, d2[:1])
    assert rc != 0
