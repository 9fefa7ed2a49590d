"""The invariant behind the device's padded rows (vslam_ctx::img_pitch, HISTORY.md round 6): an image whose rows are continued
by their BORDER_REFLECT_101 mirror, filtered as a wider image, carries the original image's result in its first `w` columns.
Checked on the oracle alone (no GPU): the 7 x 7 Gaussian for every pad the device can choose, and the corner response, where
the one column next to the pad differs (a mirrored x-derivative has the opposite sign: the device negates that column's xy
product, response.hip) and every other column agrees."""
import numpy as np
import pytest


def _padded(img, pitch):
    h, w = img.shape
    out = np.zeros((h, pitch), img.dtype)
    out[:, :w] = img
    for k in range(pitch - w):
        out[:, w + k] = img[:, w - 2 - k]
    return out


@pytest.mark.parametrize("w", [65, 66, 67, 253, 254, 255, 257])
def test_blur_of_mirrored_rows_is_the_blur_of_the_image(oracle, w):
    rng = np.random.default_rng(w)
    img = rng.integers(0, 256, (40, w), dtype=np.uint8)
    pitch = (w + 3 + 15) & ~15                       # what vslam_extract_features chooses
    for p in (pitch, w + 3, w + 4):                  # the bound itself: three mirrored columns are enough
        wide = oracle.gaussian7(_padded(img, p))
        assert np.array_equal(wide[:, :w], oracle.gaussian7(img)), (w, p)
    short = oracle.gaussian7(_padded(img, w + 2))    # two are not: the last column reads w + 2
    assert not np.array_equal(short[:, :w], oracle.gaussian7(img))


@pytest.mark.parametrize("w", [65, 66, 67])
def test_response_of_mirrored_rows_differs_in_the_last_column_only(oracle, w):
    rng = np.random.default_rng(100 + w)
    img = rng.integers(0, 256, (48, w), dtype=np.uint8)
    ref = oracle.min_eigen(img)
    wide = oracle.min_eigen(_padded(img, (w + 3 + 15) & ~15))[:, :w]
    assert np.array_equal(wide[:, :w - 1].view(np.uint32), ref[:, :w - 1].view(np.uint32))
    assert not np.array_equal(wide[:, w - 1], ref[:, w - 1])   # the box filter mirrors PRODUCTS: dx dy changes sign there
