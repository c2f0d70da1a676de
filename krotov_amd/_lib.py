"""ctypes binding of libkrotov_hip.so (the C ABI declared in include/krotov_hip.h).

There is no CPU fallback: if the shared library is missing or fails to load,
:func:`load` raises ``RuntimeError`` and so does everything that needs the
engine.  ``python __graft_entry__.py`` (or ``krotov_amd.build.build()``)
compiles it in-tree with hipcc for gfx950.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libkrotov_hip.so')

KH_OK = 0
KH_ERR_INVALID = -1
KH_ERR_HIP = -2
KH_ERR_UNSUPPORTED = -3
KH_ERR_TIMEOUT = -4
KH_ERR_NOMEM = -5


class kh_problem(ctypes.Structure):
    _fields_ = [
        ('K', ctypes.c_int32),
        ('N', ctypes.c_int32),
        ('L', ctypes.c_int32),
        ('nt', ctypes.c_int32),
        ('is_super', ctypes.c_int32),
        ('reserved', ctypes.c_int32),
        ('dt', ctypes.POINTER(ctypes.c_double)),
        ('ops', ctypes.POINTER(ctypes.c_void_p)),
        ('op_norms', ctypes.POINTER(ctypes.c_double)),
        ('tol', ctypes.c_double),
        ('theta_max', ctypes.c_double),
    ]


class kh_csr(ctypes.Structure):
    _fields_ = [
        ('nnz', ctypes.c_int64),
        ('indptr', ctypes.c_void_p),
        ('indices', ctypes.c_void_p),
        ('data', ctypes.c_void_p),
    ]


class kh_problem_csr(ctypes.Structure):
    _fields_ = [
        ('K', ctypes.c_int32),
        ('N', ctypes.c_int32),
        ('L', ctypes.c_int32),
        ('nt', ctypes.c_int32),
        ('is_super', ctypes.c_int32),
        ('reserved', ctypes.c_int32),
        ('dt', ctypes.POINTER(ctypes.c_double)),
        ('ops', ctypes.POINTER(kh_csr)),
        ('ops_adj', ctypes.POINTER(kh_csr)),
        ('op_norms', ctypes.POINTER(ctypes.c_double)),
        ('tol', ctypes.c_double),
        ('theta_max', ctypes.c_double),
    ]


# every symbol include/krotov_hip.h declares: name -> (restype, argtypes)
_P = ctypes.c_void_p
SYMBOLS = {
    'kh_last_error': (ctypes.c_char_p, []),
    'kh_version': (ctypes.c_char_p, []),
    'kh_engine_create': (ctypes.c_int, [ctypes.POINTER(kh_problem), ctypes.POINTER(_P)]),
    'kh_engine_create_csr': (ctypes.c_int, [ctypes.POINTER(kh_problem_csr), ctypes.POINTER(_P)]),
    'kh_engine_destroy': (None, [_P]),
    'kh_engine_kernel': (ctypes.c_char_p, [_P]),
    'kh_forward_store': (ctypes.c_int, [_P, _P, _P, _P, _P, _P]),
    'kh_backward_store': (ctypes.c_int, [_P, _P, _P, _P, _P]),
    'kh_forward_update': (ctypes.c_int, [_P] * 11),
    'kh_set_second_order': (ctypes.c_int, [_P, _P, _P, _P]),
    'kh_update_begin': (ctypes.c_int, [_P] * 9),
    'kh_update_step': (ctypes.c_int, [_P, ctypes.c_int32] + [_P] * 9),
    'kh_update_step_dev': (ctypes.c_int, [_P] * 11),
    'kh_update_end': (ctypes.c_int, [_P, _P, _P]),
    'kh_p2p_create_window': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, _P]),
    'kh_p2p_open_peers': (ctypes.c_int, [_P, _P]),
    'kh_p2p_selftest': (ctypes.c_int, [_P, ctypes.c_int32, _P]),
    'kh_p2p_disable': (ctypes.c_int, [_P]),
    'kh_tau': (ctypes.c_int, [_P, _P, _P, _P, _P]),
    'kh_chi_boundary': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    'kh_check': (ctypes.c_int, [_P]),
    'kh_last_stats': (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double)]),
    'kh_debug_occupy': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_double, _P]),
    'kh_p2p_stats': (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double)]),
    'kh_set_update_workgroups': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]),
    'kh_debug_launched': (ctypes.c_int, [ctypes.c_int32, ctypes.c_char_p, ctypes.c_int32]),
    'kh_series_tables': (ctypes.c_int, [ctypes.c_int32, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double)]),
    'kh_series_tables_defect': (ctypes.c_int, [ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                               ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    'kh_ell_rows_of': (ctypes.c_int32, [ctypes.c_int32]),
    'kh_ell_layout': (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, _P, ctypes.POINTER(ctypes.c_int32),
                                     ctypes.POINTER(ctypes.c_int32), _P, _P, ctypes.c_int32]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "krotov_amd: %s is missing. Build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH
        )
    # PyTorch-ROCm owns the device memory the engine works on: import it first so
    # that the library binds to the HIP runtime torch has loaded (two copies of
    # libamdhip64 in one process do not see each other's device context).
    import torch  # noqa: F401

    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise RuntimeError("krotov_amd: cannot load %s: %s" % (LIB_PATH, exc))
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def forget_launched_kernels():
    """Empty the list ``kernel_instantiations(launched_only=True)`` reports (tests that check which instantiation a
    case dispatches to)."""
    load().kh_debug_launched(2, None, 0)


def kernel_instantiations(launched_only=False):
    """Names of the sweep-kernel template instantiations some dispatch of the library can select (``kh_debug_launched``;
    needs no GPU), or of those this process has launched so far."""
    lib = load()
    which = 0 if launched_only else 1
    size = lib.kh_debug_launched(which, None, 0)
    buf = ctypes.create_string_buffer(size)
    lib.kh_debug_launched(which, buf, size)
    return [line for line in buf.value.decode().split('\n') if line]


class KrotovHipError(RuntimeError):
    """An entry point of libkrotov_hip.so returned ``code`` (one of the ``KH_ERR_*`` values)."""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


def check(rc):
    if rc != KH_OK:
        msg = load().kh_last_error().decode('utf-8', 'replace')
        raise KrotovHipError("libkrotov_hip error %d: %s" % (rc, msg), rc)
