"""MPCController.__init__ against the reference constructor's recorded behaviour (mpc.py:76-252): the same exception
type and message for every rejected input, the same stored shapes / defaults / aliasing for every accepted one
(tests/golden/ctor_outcomes.json, produced by tests/golden/make_errors.py from the imported reference class)."""
import json
import os

import pytest

import error_cases
from util import GOLDEN_DIR

with open(os.path.join(GOLDEN_DIR, 'ctor_outcomes.json')) as f:
    GOLDEN = json.load(f)


def test_table_and_golden_file_agree():
    assert sorted(GOLDEN) == sorted(error_cases.CASES)
    assert sum(1 for v in GOLDEN.values() if v[0] == 'error') >= 20        # every ValueError branch of mpc.py:82-223 + the Qx=None quirk


@pytest.mark.parametrize('name', sorted(error_cases.CASES))
def test_constructor_behaves_like_the_reference(name):
    from pympc_amd import MPCController
    assert error_cases.outcome(MPCController, error_cases.CASES[name]) == GOLDEN[name]
