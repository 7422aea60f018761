"""CPU suite: the C-ABI library loads and exports every declared symbol, host-side behaviour without a GPU,
python plumbing of the flat batch format."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from centrolign_amd import capi, synth
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    header = open(os.path.join(ROOT, "include", "centrolign_amd.h")).read()
    declared = set(re.findall(r"\b(cl_[a-z_0-9]+)\s*\(", header))
    declared -= {"cl_t"}
    assert declared, "no declarations found"
    for sym in sorted(declared):
        assert hasattr(lib, sym), "libcentrolign_amd.so does not export %s" % sym
    assert set(capi.EXPORTED_SYMBOLS) == declared
    assert lib.cl_abi_version() == capi.ABI_VERSION


def test_library_exports_nothing_but_the_c_abi():
    """round-5 verdict: C++-mangled internals (cl_big_alloc, cl_stitch_join, kernel stubs ...) used to be visible beside the C entry points; the link now goes
    through centrolign_amd/csrc/exports.map (global: cl_*; local: *)"""
    import subprocess
    path = os.path.join(ROOT, "centrolign_amd", "lib", "libcentrolign_amd.so")
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    header = open(os.path.join(ROOT, "include", "centrolign_amd.h")).read()
    declared = set(re.findall(r"\b(cl_[a-z_0-9]+)\s*\(", header)) - {"cl_t"}
    stray = sorted(n for n in names if n not in declared)
    assert not stray, "symbols exported beside the C ABI: %s" % stray[:10]


def test_default_params_match_cli_values():
    lib = capi.load_library()
    p = capi.StitchParams()
    lib.cl_stitch_params_default(C.byref(p))
    q = capi.default_stitch_params()
    assert bytes(p) == bytes(q)
    assert list(p.alignment_params.gap_open) == [60, 800, 2500] and list(p.alignment_params.gap_extend) == [30, 5, 1]
    assert (p.max_trivial_size, p.min_wfa_size, p.max_wfa_size) == (30000, 40000000, 75000000)


def test_struct_layouts():
    assert C.sizeof(capi.AlignParams) == 32
    assert C.sizeof(capi.GraphSideC) == 88
    assert C.sizeof(capi.StitchBatchC) == 8 + 2 * 88 + 8
    assert C.sizeof(capi.StitchResultC) == 48


def test_no_gpu_fails_loudly():
    """the product has no CPU fallback: without a device, context creation must raise"""
    lib = capi.load_library()
    if lib.cl_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(capi.ClError) as e:
        capi.Context(0)
    assert e.value.code == -3


def test_batch_subset_and_concat_roundtrip():
    b = synth.random_dag_batch(40, seed=3, max_n=20)
    idx = np.array([5, 0, 39, 17])
    s = b.subset(idx)
    for j, k in enumerate(idx):
        for si in range(2):
            assert s.side[si].problem(j) == b.side[si].problem(int(k))
    c = capi.StitchBatch.concat([b.subset(np.arange(0, 10)), b.subset(np.arange(10, 40))])
    for si in range(2):
        for name in ("node_off", "label", "prev_off", "prev_idx", "src_off", "src_idx", "snk_off", "snk_idx"):
            assert np.array_equal(getattr(c.side[si], name), getattr(b.side[si], name)), name


def test_hor_generator_is_deterministic():
    a = synth.hor_sequences(1, 5000, 2)
    b = synth.hor_sequences(1, 5000, 2)
    assert a == b and len(a) == 2 and 2500 < len(a[0]) < 8000
    assert set(a[0]) <= set("ACGT")


def test_c2_fixture_shape():
    b, z = H.c2_batch()
    n1, n2 = b.sizes()
    assert b.n_problems == 13245
    assert int(((n1 + 1) * (n2 + 1)).max()) == 407862
    mx = np.maximum(n1, n2)
    assert np.histogram(mx, bins=[0, 1, 9, 33, 129, 513, 2049, 10 ** 6])[0].tolist() == [933, 3649, 4456, 3587, 616, 0, 4]
