"""The C-ABI library loads on a CPU-only host and exports every symbol include/orbfe.h
declares; without a GPU the entry points fail loudly (no compute, no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "orbfe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orbfe_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_binding_binds():
    import orbfe
    assert sorted(orbfe.EXPORTS) == declared_symbols()


def test_library_exports_every_declared_symbol():
    import orbfe
    if not os.path.exists(orbfe.LIB_PATH):  # clean checkout: hipcc cross-compiles without a GPU
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    lib = orbfe.lib()  # raises if liborbfe.so has not been built
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.orbfe_version() == 1
    assert lib.orbfe_load_pattern() == orbfe.OK


def test_struct_layouts_match_the_header():
    import orbfe
    assert C.sizeof(orbfe.Config) == 40
    assert C.sizeof(orbfe.PyramidLevel) == 48  # size_t x3, ptr, size_t, ptr
    assert C.sizeof(orbfe.Soa) == 48
    assert orbfe.KEYPOINT_DTYPE.itemsize == 52
    assert orbfe.KEYPOINT_DTYPE.fields["desc"][1] == 20


def test_no_cpu_fallback_without_a_device():
    import orbfe
    lib = orbfe.lib()
    if lib.orbfe_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(orbfe.OrbfeError) as e:
        orbfe.Context(640, 480)
    assert e.value.code == orbfe.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_argument_validation_needs_no_device():
    import orbfe
    lib = orbfe.lib()
    assert lib.orbfe_gaussian_blur_3x3(None, 0, None, 0, 0, 0, None) == orbfe.ERR_INVALID_ARG
    assert lib.orbfe_fast_calculate_lut(None, 12, None) == orbfe.ERR_INVALID_ARG
    assert lib.orbfe_fast_calculate_lut(1, 17, None) == orbfe.ERR_INVALID_ARG
    assert b"invalid argument" in lib.orbfe_last_error(None)
    h = C.c_void_p()
    cfg = orbfe.Config(640, 480, 1, 24, 13, 12, 0, 0, 1, 0)  # cell 24 is not a power of two
    assert lib.orbfe_create(C.byref(cfg), C.byref(h)) == orbfe.ERR_UNSUPPORTED
    cfg = orbfe.Config(640, 480, 1, 32, 13, 8, 0, 0, 1, 0)   # arc 8 outside 9..12
    assert lib.orbfe_create(C.byref(cfg), C.byref(h)) == orbfe.ERR_UNSUPPORTED
    assert lib.orbfe_num_cells(None) == 0


def test_missing_library_fails_loudly(monkeypatch):
    import orbfe
    monkeypatch.setattr(orbfe, "_lib", None)
    monkeypatch.setattr(orbfe, "LIB_PATH", "/nonexistent/liborbfe.so")
    with pytest.raises(orbfe.OrbfeError):
        orbfe.lib()
