"""ctypes binding of libmpcqp_hip.so (C ABI declared in include/mpcqp.h).

There is deliberately no fallback: if the shared library has not been built
(``python -c 'import __graft_entry__ as g; g.build()'`` or ``pympc_amd/csrc/build.sh``)
loading raises, and creating a problem without a GPU raises ``RuntimeError``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmpcqp_hip.so')     # the in-tree build, nothing else (no environment override)

# every symbol declared in include/mpcqp.h
SYMBOLS = [
    'mpcqp_default_settings', 'mpcqp_status_string', 'mpcqp_last_error', 'mpcqp_device_count',
    'mpcqp_create', 'mpcqp_destroy', 'mpcqp_set_stream', 'mpcqp_synchronize',
    'mpcqp_setup', 'mpcqp_setup_qp', 'mpcqp_create_csc', 'mpcqp_setup_csc', 'mpcqp_update', 'mpcqp_update_vectors', 'mpcqp_warm_start', 'mpcqp_update_settings', 'mpcqp_solve',
    'mpcqp_mpc_step', 'mpcqp_step_host', 'mpcqp_mpc_run', 'mpcqp_mpc_loop', 'mpcqp_get_solution', 'mpcqp_get_u0', 'mpcqp_get_shape', 'mpcqp_get_dims', 'mpcqp_get_stream_bytes', 'mpcqp_get_work', 'mpcqp_get_occupancy', 'mpcqp_kernel_name', 'mpcqp_get_stats', 'mpcqp_get_launch_times', 'mpcqp_profile',
    'mpcqp_export_qp', 'mpcqp_get_scaling', 'mpcqp_debug_kkt_solve', 'mpcqp_get_iterate', 'mpcqp_iterate', 'mpcqp_refactor', 'mpcqp_share_factor', 'mpcqp_eq_solve',
]


class Settings(C.Structure):
    _fields_ = [('rho', C.c_double), ('sigma', C.c_double), ('alpha', C.c_double),
                ('eps_abs', C.c_double), ('eps_rel', C.c_double),
                ('eps_prim_inf', C.c_double), ('eps_dual_inf', C.c_double),
                ('adaptive_rho_tolerance', C.c_double),
                ('max_iter', C.c_int32), ('check_termination', C.c_int32), ('scaling', C.c_int32),
                ('adaptive_rho', C.c_int32), ('adaptive_rho_interval', C.c_int32), ('warm_start', C.c_int32),
                ('soft_constraints', C.c_int32), ('backend', C.c_int32), ('tuning', C.c_int32)]


# enum mpcqp_backend / mpcqp_tuning of include/mpcqp.h
BACKEND_AUTO, BACKEND_SWEEPS, BACKEND_DENSE, BACKEND_BCR, BACKEND_BCR8, BACKEND_BCRT = 0, 1, 2, 3, 4, 5
TUNE_NO_BALANCE, TUNE_NO_LSTAGE, TUNE_NO_GROUPING, TUNE_NO_W8, TUNE_NO_QUEUE, TUNE_NO_PARTS = 1, 2, 4, 8, 16, 32
TUNE_NO_SHARE = 1 << 30    # every instance keeps to its own factor (mpcqp_share_factor)
TUNE_PACE_SHIFT = 8        # development: tuning bits 8..15 = pacing units (include/mpcqp.h)


class Info(C.Structure):
    _fields_ = [('status', C.c_int32), ('iter', C.c_int32), ('rho_updates', C.c_int32), ('reserved', C.c_int32),
                ('obj_val', C.c_double), ('pri_res', C.c_double), ('dua_res', C.c_double), ('rho', C.c_double)]


_pd = C.POINTER(C.c_double)


class Model(C.Structure):
    _fields_ = [(k, _pd) for k in ('Ad', 'Bd', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax',
                                   'Dumin', 'Dumax', 'uref', 'eps_feas')]


class Loop(C.Structure):
    """mpcqp_loop (include/mpcqp.h): optional inputs/outputs of the device-side closed loop."""
    _fields_ = [('w', C.c_void_p), ('Ap', C.c_void_p), ('Bp', C.c_void_p), ('xref_traj', C.c_void_p),
                ('ny', C.c_int32), ('xref_rows', C.c_int32),
                ('C', C.c_void_p), ('Lgain', C.c_void_p), ('v', C.c_void_p), ('x_true', C.c_void_p),
                ('x_traj', C.c_void_p), ('xhat_traj', C.c_void_p), ('y_traj', C.c_void_p), ('u_traj', C.c_void_p),
                ('status_traj', C.c_void_p), ('iter_traj', C.c_void_p)]


_lib = None


def load():
    """Load the HIP library (once) and declare the prototypes of include/mpcqp.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libmpcqp_hip.so is missing (%s). Build it with pympc_amd/csrc/build.sh; "
            "pympc_amd has no CPU fallback." % LIB_PATH)
    # One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 with the same sonames as /opt/rocm's, so
    # whichever is mapped first serves both: this library works on top of torch's, torch on top of the system's reports "No HIP GPUs are
    # available" at its first CUDA call.  Where torch is installed it therefore loads first (it is the package's plumbing for device
    # buffers, streams and torch.distributed anyway).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    L.mpcqp_default_settings.argtypes = [C.POINTER(Settings)]
    L.mpcqp_default_settings.restype = None
    L.mpcqp_status_string.argtypes = [C.c_int]
    L.mpcqp_status_string.restype = C.c_char_p
    L.mpcqp_last_error.restype = C.c_char_p
    L.mpcqp_device_count.restype = C.c_int
    L.mpcqp_create.argtypes = [C.POINTER(H), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Settings)]
    L.mpcqp_destroy.argtypes = [H]
    L.mpcqp_destroy.restype = None
    L.mpcqp_set_stream.argtypes = [H, C.c_void_p]
    L.mpcqp_synchronize.argtypes = [H]
    L.mpcqp_setup.argtypes = [H, C.POINTER(Model), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.mpcqp_update.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.mpcqp_setup_qp.argtypes = [H, C.POINTER(Model), C.c_void_p, C.c_void_p, C.c_void_p]
    L.mpcqp_update_vectors.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mpcqp_create_csc.argtypes = [C.POINTER(H), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Settings)]
    L.mpcqp_setup_csc.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mpcqp_warm_start.argtypes = [H, C.c_void_p, C.c_void_p]
    L.mpcqp_update_settings.argtypes = [H, C.POINTER(Settings)]
    L.mpcqp_solve.argtypes = [H]
    L.mpcqp_mpc_run.argtypes = [H, C.c_int] + [C.c_void_p] * 7
    L.mpcqp_mpc_step.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.mpcqp_mpc_loop.argtypes = [H, C.c_int, C.POINTER(Loop)]
    L.mpcqp_step_host.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mpcqp_get_solution.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mpcqp_get_u0.argtypes = [H, C.c_void_p]
    L.mpcqp_get_shape.argtypes = [H] + [C.POINTER(C.c_int)] * 4
    L.mpcqp_get_dims.argtypes = [H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.mpcqp_get_stream_bytes.argtypes = [H, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.mpcqp_get_work.argtypes = [H, C.POINTER(C.c_int64)]
    L.mpcqp_get_occupancy.argtypes = [H] + [C.POINTER(C.c_int)] * 3
    L.mpcqp_kernel_name.argtypes = [H, C.c_int, C.c_char_p, C.c_int]
    L.mpcqp_get_stats.argtypes = [H, C.POINTER(C.c_uint64), C.c_int]
    L.mpcqp_get_launch_times.argtypes = [H, C.c_void_p, C.c_int]
    L.mpcqp_profile.argtypes = [H, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    L.mpcqp_export_qp.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mpcqp_get_scaling.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mpcqp_debug_kkt_solve.argtypes = [H, C.c_void_p, C.c_void_p]
    L.mpcqp_get_iterate.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mpcqp_iterate.argtypes = [H, C.c_int]
    L.mpcqp_refactor.argtypes = [H]
    L.mpcqp_share_factor.argtypes = [H, C.POINTER(C.c_int)]
    L.mpcqp_eq_solve.argtypes = [H, C.c_int, C.c_int, C.c_double, C.c_void_p]
    for name in SYMBOLS:
        getattr(L, name)           # AttributeError here = header and library out of sync
        if name not in ('mpcqp_default_settings', 'mpcqp_status_string', 'mpcqp_last_error', 'mpcqp_destroy'):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = load().mpcqp_last_error().decode()
        raise RuntimeError('%s failed (%d): %s' % (what, rc, msg))
