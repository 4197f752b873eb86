"""CPU tests of the boundary: the in-tree library loads, exports every symbol include/lasso_hip.h
declares, the ctypes mirrors of the ABI structs have the C sizes, and compute fails loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "lasso_hip.h")).read()


def declared_symbols():
    return sorted(set(re.findall(r"\b(lh_[a-z0-9_]+)\s*\(", HEADER)))


def test_library_exports_every_declared_symbol(hl):
    from halo2_lasso_amd import _ffi
    lib = _ffi.load()
    syms = declared_symbols()
    assert len(syms) >= 40
    for name in syms:
        assert hasattr(lib, name), "liblasso_hip.so does not export %s" % name
    assert set(_ffi.SIGNATURES) == set(syms), "ctypes binding and header disagree"


def test_struct_layouts(hl):
    from halo2_lasso_amd import _ffi
    assert C.sizeof(_ffi.lh_fr) == 32 and C.sizeof(_ffi.lh_g1) == 64
    assert C.sizeof(_ffi.lh_evaluation) == 40
    assert C.sizeof(_ffi.lh_sop) == 8 + 32 * 48 + 48 + 48 * 4
    assert C.sizeof(_ffi.lh_lasso_table) == 12 + 64 + 64 + 4 + 32 * 16 + 16 + 16 * 4 + 0 \
        or C.sizeof(_ffi.lh_lasso_table) % 8 == 0
    assert C.sizeof(_ffi.lh_transcript) == 8 * C.sizeof(C.c_void_p)
    assert C.sizeof(_ffi.lh_g2) == 128


def test_marshalling(hl):
    for v in (0, 1, 2 ** 200 + 17, hl.R_MOD - 1):
        assert hl.fr_from_bytes(hl.fr_to_bytes(v)) == v
    # Montgomery form of 1 is R mod r
    assert hl.fr_to_bytes(1) == bytes.fromhex("fbffff4f1c3496ac29cd609f9576fc362e4679786fa36e662fdf079ac1770a0e")
    assert hl.g1_from_bytes(hl.g1_to_bytes((1, 2))) == (1, 2)
    assert hl.g1_from_bytes(hl.g1_to_bytes(None)) is None
    t = hl.LassoTable.range(2, 16).to_c()
    assert (t.num_chunks, t.chunk_bits, t.num_memories, t.num_terms) == (2, 16, 2, 2)
    assert hl.fr_from_bytes(bytes(t.g_coeff[1])) == 1 << 16


def test_keccak_transcript_host_side(hl):
    """The built-in transcript is host code: usable (and checkable against the oracle) without a GPU."""
    from oracle.pyref.transcript import Keccak256Transcript as OT
    t, ot = hl.Keccak256Transcript(), OT()
    for v in (0, 1, hl.R_MOD - 1, 1 << 255):
        t.write_field_element(v), ot.write_field_element(v)
        assert t.squeeze_challenge() == ot.squeeze_challenge()
    t.common_field_element(7), ot.common_field_element(7)
    t.write_commitment((1, 2)), ot.write_commitment((1, 2))
    assert t.squeeze_challenges(3) == ot.squeeze_challenges(3)
    assert t.into_proof() == ot.into_proof()
    with pytest.raises(hl.TranscriptError):
        t.write_commitment(None)


def test_no_gpu_means_loud_failure(hl):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hl.DeviceError):
        hl.Context(0)


def test_cpu_list_format(hl):
    """the kernel's CPU list format of sysfs local_cpulist, as Context.host_cpus reads it"""
    assert hl.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert hl.parse_cpulist("") == set() and hl.parse_cpulist("5") == {5}
    assert len(hl.parse_cpulist("0-63,128-191")) == 128


@pytest.mark.gpu
def test_host_binding_next_to_the_device(hl, ctx):
    """lh_ctx_host_cpus names the device's PCI address and the CPUs on its NUMA node; Context.bind_host narrows the calling
    thread's affinity to them (never widens it, never leaves it empty) and hands back what it replaced."""
    bus, local = ctx.host_cpus()
    assert re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.[0-9a-f]", bus), bus
    assert os.path.isdir("/sys/bus/pci/devices/" + bus)
    before = os.sched_getaffinity(0)
    prev = ctx.bind_host()
    try:
        now = os.sched_getaffinity(0)
        if prev is None:
            assert now == before and (not (local & before) or (local & before) == before)
        else:
            assert prev == before and now == (local & before) and now and now < before
    finally:
        os.sched_setaffinity(0, before)


@pytest.mark.gpu
def test_null_arguments_are_errors_not_crashes(hl, ctx):
    """Every pointer an entry point dereferences is checked at the boundary (capi.cpp NEED / NEED_N): NULL is LH_ERR_ARG with
    a message naming the argument, never a segfault inside the library; an empty vector may be NULL."""
    from halo2_lasso_amd import _ffi
    lib, h = ctx.lib, ctx.h
    pp = hl.MultilinearKzg.setup(ctx, [3, 5, 7])
    poly = ctx.upload(b"".join(hl.fr_to_bytes(v) for v in range(8)))
    out, fr = _ffi.lh_g1(), _ffi.lh_fr()
    tr = hl.Keccak256Transcript()
    bad = [
        lib.lh_mkzg_commit(h, pp.h, None, 3, C.byref(out)),
        lib.lh_mkzg_commit(h, pp.h, poly.ptr, 3, None),
        lib.lh_mkzg_commit(h, None, poly.ptr, 3, C.byref(out)),
        lib.lh_mkzg_batch_commit(h, pp.h, None, 1, 3, C.byref(out)),
        lib.lh_mkzg_batch_commit(h, pp.h, (C.c_void_p * 1)(poly.ptr), 1, 3, None),
        lib.lh_mkzg_open(h, pp.h, None, 3, (_ffi.lh_fr * 3)(), tr.p, C.byref(fr)),
        lib.lh_mkzg_open(h, pp.h, poly.ptr, 3, None, tr.p, C.byref(fr)),
        lib.lh_mkzg_open(h, pp.h, poly.ptr, 3, (_ffi.lh_fr * 3)(), None, C.byref(fr)),
        lib.lh_mkzg_batch_open(h, pp.h, 3, None, 1, (_ffi.lh_fr * 3)(), 1, (_ffi.lh_evaluation * 1)(), 1, tr.p),
        lib.lh_mkzg_batch_open(h, pp.h, 3, (C.c_void_p * 1)(poly.ptr), 1, None, 1, (_ffi.lh_evaluation * 1)(), 1, tr.p),
        lib.lh_mkzg_batch_open(h, pp.h, 3, (C.c_void_p * 1)(poly.ptr), 1, (_ffi.lh_fr * 3)(), 1, None, 1, tr.p),
        lib.lh_mkzg_setup(h, None, 3, C.byref(C.c_void_p())),
        lib.lh_srs_upload(h, None, 3, C.byref(C.c_void_p())),
        lib.lh_srs_download(h, pp.h, None),
        lib.lh_msm(h, None, None, 4, C.byref(out)),
        lib.lh_msm_u32(h, poly.ptr, None, 4, C.byref(out)),
        lib.lh_fr_add(h, poly.ptr, None, 8, poly.ptr),
        lib.lh_fr_batch_invert(h, None, 8, poly.ptr),
        lib.lh_fix_var(h, poly.ptr, 8, C.byref(fr), None),
        lib.lh_eq_xy(h, None, 3, poly.ptr),
        lib.lh_evaluate(h, None, 1, 3, (_ffi.lh_fr * 3)(), C.byref(fr)),
        lib.lh_lincomb(h, (C.c_void_p * 1)(poly.ptr), None, 1, 8, poly.ptr),
        lib.lh_upload(h, None, b"1234", 4),
        lib.lh_download(h, None, poly.ptr, 4),
        lib.lh_zeromorph_open(h, None, 8, poly.ptr, 3, (_ffi.lh_fr * 3)(), tr.p),
        lib.lh_zeromorph_batch_commit(h, None, 8, (C.c_void_p * 1)(poly.ptr), 1, 3, C.byref(out)),
    ]
    assert bad == [_ffi.LH_ERR_ARG] * len(bad), bad
    assert b"null argument" in lib.lh_last_error() or b"transcript" in lib.lh_last_error()
    # an empty vector may be NULL: nothing to read, nothing written
    assert lib.lh_mkzg_batch_commit(h, pp.h, None, 0, 3, None) == _ffi.LH_OK
    assert lib.lh_fr_add(h, None, None, 0, None) == _ffi.LH_OK
    assert lib.lh_upload(h, None, None, 0) == _ffi.LH_OK
    # ... and the ctx still proves after all of that
    assert hl.MultilinearKzg.commit(pp, hl.MultilinearPolynomial(ctx, poly, 3)) is not None
