"""The C-ABI library loads and exports every symbol include/krotov_hip.h declares
(no compute calls: runs without a GPU)."""
import os
import re

from krotov_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'krotov_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(kh_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_exports_every_declared_symbol():
    build.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == declared
    assert b'gfx950' in lib.kh_version()


def test_bad_arguments_are_reported_not_crashed():
    lib = _lib.load()
    import ctypes

    out = ctypes.c_void_p()
    rc = lib.kh_engine_create(None, ctypes.byref(out))
    assert rc == -1 and b'null' in lib.kh_last_error()
    pr = _lib.kh_problem()
    pr.K, pr.N, pr.L, pr.nt = 0, 2, 1, 5
    rc = lib.kh_engine_create(ctypes.byref(pr), ctypes.byref(out))
    assert rc == -1 and b'bad sizes' in lib.kh_last_error()
    assert lib.kh_check(None) == -1
    # the other entry points validate before touching the device
    assert lib.kh_engine_create_csr(None, ctypes.byref(out)) == -1
    pc = _lib.kh_problem_csr()
    pc.K, pc.N, pc.L, pc.nt = 1, 2, 0, 3
    assert lib.kh_engine_create_csr(ctypes.byref(pc), ctypes.byref(out)) == -1
    assert b'required' in lib.kh_last_error()
    for call in (lambda: lib.kh_forward_store(None, None, None, None, None, None),
                 lambda: lib.kh_backward_store(None, None, None, None, None),
                 lambda: lib.kh_forward_update(None, None, None, None, None, None, None, None, None, None, None),
                 lambda: lib.kh_set_second_order(None, None, None, None),
                 lambda: lib.kh_chi_boundary(None, None, None, None, None, None, None, None),
                 lambda: lib.kh_tau(None, None, None, None, None)):
        assert call() == -1
        assert lib.kh_last_error() != b''
