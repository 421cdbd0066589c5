"""The C-ABI library loads and exports every symbol include/mpcqp.h declares (no compute without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'mpcqp.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mpcqp_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported():
    import __graft_entry__ as g
    from pympc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    L = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.SYMBOLS) == names


def test_no_gpu_fails_loudly():
    from pympc_amd import _lib
    L = _lib.load()
    if L.mpcqp_device_count() > 0:
        pytest.skip('GPU present')
    from pympc_amd import MPCController, fixtures
    K = MPCController(**fixtures.point_mass())
    with pytest.raises(RuntimeError):
        K.setup()


def test_status_strings_and_defaults():
    from pympc_amd import _lib
    L = _lib.load()
    s = _lib.Settings()
    L.mpcqp_default_settings(s)
    assert (s.rho, s.sigma, s.alpha, s.eps_abs, s.eps_rel) == (0.1, 1e-6, 1.6, 1e-3, 1e-3)
    assert (s.max_iter, s.check_termination, s.scaling, s.adaptive_rho, s.warm_start) == (4000, 25, 10, 1, 1)
    assert L.mpcqp_status_string(1) == b'solved'
    assert L.mpcqp_status_string(-3) == b'primal infeasible'
    assert L.mpcqp_status_string(-2) == b'maximum iterations reached'
    assert L.mpcqp_status_string(2) == b'solved inaccurate'
