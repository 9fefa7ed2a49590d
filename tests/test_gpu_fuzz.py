"""Fixed slices of the randomised parity runs (tests/fuzz_extract.py, fuzz_match.py, fuzz_ransac.py, fuzz_grid.py, fuzz_assoc.py, fuzz_pose.py)."""
import pytest

import fuzz_extract

pytestmark = pytest.mark.gpu


def test_extraction_stages_on_random_shapes(ctx, oracle):
    assert fuzz_extract.run(ctx, oracle, seed=20261004, cases=150) == 150


def test_matcher_on_random_ragged_batches(ctx, ctx_exp, oracle):
    import fuzz_match
    assert fuzz_match.run(ctx, oracle, seed=20261006, cases=60, variants=False) == 60      # the product's one matcher
    assert fuzz_match.run(ctx_exp, oracle, seed=20261007, cases=60, variants=True) == 60   # its variants (experiments build)


def test_ransac_kernels_on_random_and_degenerate_inputs(ctx, oracle):
    import fuzz_ransac
    assert fuzz_ransac.run(ctx, oracle, seed=20261005, cases=400) == 400


def test_grid_extractor_on_random_shapes_grids_and_content(ctx, oracle):
    import fuzz_grid
    assert fuzz_grid.run(ctx, oracle, seed=20261008, cases=60) == 60


def test_fuzz_assoc_slice(ctx, oracle):
    import fuzz_assoc
    assert fuzz_assoc.run(ctx, oracle, seed=20261009, cases=40) == 40


def test_fuzz_pose_slice(ctx, oracle):
    import fuzz_pose
    assert fuzz_pose.run(ctx, oracle, seed=20261010, cases=60) == 60
