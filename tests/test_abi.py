"""CPU-side checks of the drop-in boundary: the shared library builds, loads and exports every symbol that
include/eigensolver_amd.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "eigensolver_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(es_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_declared_symbols():
    from eigensolver_amd import build
    lib_path = build.build()
    assert os.path.exists(lib_path)
    lib = ctypes.CDLL(lib_path)
    syms = declared_symbols()
    assert len(syms) >= 8
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    lib.es_abi_version.restype = ctypes.c_int
    assert lib.es_abi_version() == 1
    lib.es_status_string.restype = ctypes.c_char_p
    assert lib.es_status_string(0) == b"success"


def test_no_cpu_fallback_without_device():
    """Without a HIP device the product must refuse to run instead of computing on the host."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from eigensolver_amd import _lib
    with pytest.raises(_lib.EsError):
        _lib.Context(0)
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.es_context_create(0, None, ctypes.byref(h)) != 0


def test_ctypes_struct_layouts_match_the_library():
    """The ctypes mirrors in eigensolver_amd/_lib.py (and the oracle's mirror of es_shoot_desc) have the size the C
    compiler gave the structs of include/eigensolver_amd.h."""
    from eigensolver_amd import _lib, build
    from oracle import port
    lib = ctypes.CDLL(build.build())
    lib.es_abi_sizeof.restype = ctypes.c_int
    lib.es_abi_sizeof.argtypes = [ctypes.c_int]
    mirrors = [_lib.SlabAnalyticParams, _lib.ShootDesc, _lib.Profiles, _lib.RootTable, _lib.WorkerSpec,
               _lib.CylUniformParams, _lib.ComplexRootTable]
    for i, m in enumerate(mirrors):
        assert lib.es_abi_sizeof(i) == ctypes.sizeof(m), (i, m.__name__, lib.es_abi_sizeof(i), ctypes.sizeof(m))
    assert lib.es_abi_sizeof(1) == ctypes.sizeof(port.ShootDesc) and lib.es_abi_sizeof(2) == ctypes.sizeof(port.Profiles)
    assert lib.es_abi_sizeof(99) == -1
