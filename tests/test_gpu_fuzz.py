"""A fixed slice of the randomised extraction parity run (tests/fuzz_extract.py)."""
import pytest

import fuzz_extract

pytestmark = pytest.mark.gpu


def test_extraction_stages_on_random_shapes(ctx, oracle):
    assert fuzz_extract.run(ctx, oracle, seed=20261004, cases=150) == 150
