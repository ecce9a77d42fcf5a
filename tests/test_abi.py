"""The C-ABI library loads on a CPU-only host and exports every symbol include/orbfe.h
declares; without a GPU the entry points fail loudly (no compute, no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="orbfe.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
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
    assert lib.orbfe_version() == 2
    assert lib.orbfe_load_pattern() == orbfe.OK


def test_ingest_header_is_bound_and_exported():
    """include/orbfe_ingest.h (the staging ring): every declared symbol is bound and exported, the config
    struct is 8 x int32, and the entry points refuse bad arguments before any HIP call."""
    import orbfe
    assert sorted(orbfe.INGEST_EXPORTS) == declared_symbols("orbfe_ingest.h")
    lib = orbfe.lib()
    for name in orbfe.INGEST_EXPORTS:
        assert hasattr(lib, name), name
    assert C.sizeof(orbfe.IngestConfig) == 32
    cfg = orbfe.IngestConfig()
    lib.orbfe_ingest_default_config(C.byref(cfg), 64)
    assert (cfg.slots, cfg.frames_per_slot, cfg.channels, cfg.match_mode, cfg.download_matches) == (3, 64, 1, -1, 0)
    h = C.c_void_p()
    assert lib.orbfe_ingest_create(None, C.byref(cfg), C.byref(h)) == orbfe.ERR_INVALID_ARG
    assert b"orbfe_ingest_create" in lib.orbfe_ingest_last_error(None)
    assert lib.orbfe_ingest_submit(None, 0, 1) == orbfe.ERR_INVALID_ARG
    assert lib.orbfe_ingest_wait(None, 0, None, None, None, None) == orbfe.ERR_INVALID_ARG
    assert lib.orbfe_ingest_slots(None) == 0 and lib.orbfe_ingest_frame_bytes(None) == 0
    assert not lib.orbfe_ingest_host_frames(None, 0)
    lib.orbfe_ingest_destroy(None)


def test_struct_layouts_match_the_header():
    import orbfe
    assert C.sizeof(orbfe.Config) == 44  # 11 int32: ... max_batch, device, descriptor_level (ORBFE_VERSION 2)
    assert orbfe.Config.descriptor_level.offset == 40
    assert C.sizeof(orbfe.PyramidLevel) == 48  # size_t x3, ptr, size_t, ptr
    assert C.sizeof(orbfe.Soa) == 48
    assert orbfe.KEYPOINT_DTYPE.itemsize == 52
    assert orbfe.KEYPOINT_DTYPE.fields["desc"][1] == 20
    # rs2_intrinsics / rs2_extrinsics as the reference's kernels read them (cuda-align.cu:57-119): 12 x 4 bytes each
    assert C.sizeof(orbfe.Intrinsics) == 48 and orbfe.Intrinsics.model.offset == 24 and orbfe.Intrinsics.coeffs.offset == 28
    assert C.sizeof(orbfe.Extrinsics) == 48 and orbfe.Extrinsics.translation.offset == 36


def test_align_depth_argument_validation_needs_no_device():
    """orbfe_align_depth_to_other / _batch reject bad arguments and the models the reference cannot run before any HIP
    call (host-side checks), so the contract is testable without a GPU."""
    import orbfe
    lib = orbfe.lib()
    ok = orbfe.Intrinsics(64, 48, 32.0, 24.0, 50.0, 50.0, 0, (C.c_float * 5)())
    ex = orbfe.Extrinsics((C.c_float * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1), (C.c_float * 3)())
    fake = 0x1000  # never dereferenced: every case below is refused first
    call = lambda scale, d, o: lib.orbfe_align_depth_to_other(fake, fake, None, scale, 64, 48, C.byref(d), C.byref(o), C.byref(ex), None)
    assert lib.orbfe_align_depth_to_other(None, fake, None, 0.001, 64, 48, C.byref(ok), C.byref(ok), C.byref(ex), None) == orbfe.ERR_INVALID_ARG
    assert call(float("inf"), ok, ok) == orbfe.ERR_INVALID_ARG
    for dm, om in ((1, 0), (3, 0), (1, 3)):  # forward-distorted DEPTH models; f-theta on the other camera is supported
        d = orbfe.Intrinsics(64, 48, 32.0, 24.0, 50.0, 50.0, dm, (C.c_float * 5)())
        o = orbfe.Intrinsics(64, 48, 32.0, 24.0, 50.0, 50.0, om, (C.c_float * 5)())
        assert call(0.001, d, o) == orbfe.ERR_UNSUPPORTED
        assert b"align_depth_to_other" in lib.orbfe_last_error(None)
    big = orbfe.Intrinsics(40000, 48, 32.0, 24.0, 50.0, 50.0, 0, (C.c_float * 5)())
    assert call(0.001, ok, big) == orbfe.ERR_INVALID_ARG  # output coordinates travel as int16 pairs
    assert lib.orbfe_align_depth_batch(fake, 10, fake, 64 * 48, 2, 0.001, C.byref(ok), C.byref(ok), C.byref(ex), None) == orbfe.ERR_INVALID_ARG  # stride < frame
    assert lib.orbfe_align_depth_batch(None, 0, None, 0, 0, 0.001, C.byref(ok), C.byref(ok), C.byref(ex), None) == orbfe.OK  # nothing to do


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


def test_library_keeps_no_mutable_global_state():
    """include/orbfe.h promises "no global mutable state" (a process-global LUT-pointer -> arc table sat
    behind orbfe_detect).  Every writable data symbol of liborbfe.so must be toolchain / HIP-runtime plumbing
    (kernel handles and the fat-binary handle, written once at load) or the thread-local error text."""
    import subprocess
    import orbfe
    if not os.path.exists(orbfe.LIB_PATH):
        pytest.skip("library not built")
    out = subprocess.run(["nm", "-C", orbfe.LIB_PATH], capture_output=True, text=True, check=True).stdout
    bad = []
    for line in out.splitlines():
        parts = line.split(None, 2)
        if len(parts) < 3 or parts[1] not in "bBdD":
            continue
        name = parts[2]
        plumbing = (name.startswith(("__hip", "_DYNAMIC", "_GLOBAL_OFFSET_TABLE_", "DW.ref", "__dso_handle", "completed.",
                                     "__TMC_END__", "__bss_start", "_edata", "_end", "__data_start", "__frame_dummy",
                                     "__do_global", "guard variable", "__do_init", "__do_fini", "__init", "__fini"))  # HIP module ctor / dtor
                    or "_kernel" in name           # device-stub handles hipcc emits per __global__ function
                    or "t_err" in name)            # thread_local text behind orbfe_last_error(NULL)
        if not plumbing:
            bad.append(line)
    assert not bad, "writable static storage in liborbfe.so:\n" + "\n".join(bad)
