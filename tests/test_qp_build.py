"""Host QP builder vs the structural golden vectors captured from the reference
(tests/golden/make_golden.py; reference code pyMPC/mpc.py:386-608)."""
import numpy as np
import pytest

from pympc_amd.controller import MPCController
from util import golden_names, load_golden, golden_kwargs, golden_csc, update_steps, apply_attrs


class _NullProb:
    """Stands in for the solver so that only the host logic runs (like the capture stub)."""
    def __init__(self):
        self.n = None
    def setup(self, P, q, A, l, u, **kw):
        self.n = P.shape[0]
        self.kw = kw
    def update(self, **kw):
        self.last = kw
    def solve(self):
        class R: pass
        r = R(); r.x = np.zeros(self.n); r.info = R(); r.info.status = 'solved'; r.info.obj_val = 0.0
        return r


@pytest.mark.parametrize('name', golden_names())
def test_build_matches_reference(name):
    g = load_golden(name)
    K = apply_attrs(MPCController(**golden_kwargs(g)), golden_kwargs(g))
    K.prob = _NullProb()
    K.setup(solve=False)
    for which, M in (('P', K.P), ('A', K.A)):
        R = golden_csc(g, which)
        assert M.shape == R.shape
        assert M.format == 'csc'
        assert np.array_equal(M.toarray(), R.toarray()), which + ' values differ'
        assert np.array_equal(M.indptr, R.indptr) and np.array_equal(M.indices, R.indices), which + ' pattern differs'
        assert np.array_equal(M.data, R.data)
    assert np.array_equal(K.q, g['q'])
    assert np.array_equal(K.l, g['l'])
    assert np.array_equal(K.u, g['u'])
    # swapped eps kwargs of mpc.py:266 reach the solver
    assert K.prob.kw['eps_abs'] == K.eps_rel and K.prob.kw['eps_rel'] == K.eps_abs
    assert K.prob.kw['warm_start'] is True and K.prob.kw['verbose'] is False


@pytest.mark.parametrize('name', golden_names())
def test_update_matches_reference(name):
    g = load_golden(name)
    K = apply_attrs(MPCController(**golden_kwargs(g)), golden_kwargs(g))
    K.prob = _NullProb()
    K.setup(solve=False)
    for s, st in enumerate(update_steps(g)):
        K.update(st['x'], u=st['u'], xref=st['xref'], solve=False)
        assert np.array_equal(K.q, st['q']), 'q step %d' % s
        assert np.array_equal(K.l, st['l'])
        assert np.array_equal(K.u, st['u_bound'])
        if s == 0:
            K.solve()
            u0 = K.output()
            assert np.array_equal(u0, g['upd0_output_u'])
            assert K.uminus1_rh is u0
