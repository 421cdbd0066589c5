"""examples/cart_pole_c_abi.c: the C ABI of include/mpcqp.h driven from plain C (no Python, no torch) -- what a binder in any other language
does.  CPU: it compiles against the header and links and runs against the CPU twin of the ABI (oracle/libmpcqp_cpu.so, test infrastructure).
GPU: the same source linked against libmpcqp_hip.so must print, step by step, the inputs the Python drop-in class computes for the same
closed loop (pympc_amd.MPCController on fixtures.cart_pole())."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'examples', 'cart_pole_c_abi.c')


def _build(tmp_path, libdir, lib):
    exe = str(tmp_path / ('cart_pole_' + lib))
    subprocess.run(['gcc', '-O2', '-I' + os.path.join(ROOT, 'include'), SRC, '-o', exe, '-L' + libdir, '-l' + lib, '-Wl,-rpath,' + libdir, '-lm'],
                   check=True, capture_output=True, text=True)
    return exe


def _run(exe, nsteps):
    out = subprocess.run([exe, str(nsteps)], check=True, capture_output=True, text=True, timeout=300).stdout.strip().splitlines()
    assert len(out) == nsteps
    u = np.array([float(ln.split()[2]) for ln in out])
    its = [int(ln.split()[ln.split().index('iters') + 1]) for ln in out]
    assert all(ln.rstrip().endswith('solved') for ln in out)
    return u, its


def _python_loop(K, kw, nsteps):
    import warnings
    us, its = [], []
    x = np.array(kw['x0'], dtype=float)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup()
        for _ in range(nsteps):
            u = K.output(); us.append(float(u[0])); its.append(int(K.res.info.iter))
            x = kw['Ad'] @ x + kw['Bd'] @ u
            K.update(x)
    return np.array(us), its


def test_c_example_runs_against_the_cpu_twin(tmp_path):
    """(no GPU needed) header and source agree, every symbol the example uses resolves, and the loop it prints is the oracle's."""
    libdir = os.path.join(ROOT, 'oracle')
    if not os.path.exists(os.path.join(libdir, 'libmpcqp_cpu.so')):
        subprocess.run(['make', '-C', libdir, 'libmpcqp_cpu.so'], check=True, capture_output=True)
    u, its = _run(_build(tmp_path, libdir, 'mpcqp_cpu'), 30)
    from pympc_amd import MPCController, fixtures
    from oracle.osqp_oracle import OSQP
    kw = fixtures.cart_pole()
    K = MPCController(**kw); K.prob = OSQP()
    uo, itso = _python_loop(K, kw, 30)
    assert its == itso
    assert np.abs(u - uo).max() <= 1e-9 * max(1.0, np.abs(uo).max())


@pytest.mark.gpu
def test_c_example_on_the_gpu_matches_the_python_class(tmp_path):
    u, its = _run(_build(tmp_path, os.path.join(ROOT, 'pympc_amd'), 'mpcqp_hip'), 40)
    from pympc_amd import MPCController, fixtures
    kw = fixtures.cart_pole()
    up, itsp = _python_loop(MPCController(**kw), kw, 40)
    assert its == itsp
    # same library and kernels underneath; the plant step is summed in a different order in C than by numpy, so not bit for bit
    assert np.abs(u - up).max() <= 1e-9 * max(1.0, np.abs(up).max())
