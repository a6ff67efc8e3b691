"""The C-ABI library loads and exports every symbol include/bmbs.h declares (no compute calls)."""
import os
import re

from common import ROOT


def test_library_exports_every_declared_symbol():
    from bitmapperbs_amd import capi
    hdr = open(os.path.join(ROOT, "include", "bmbs.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(bmbs_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = capi.lib()
    for s in declared:
        assert hasattr(lib, s), "libbmbs_hip.so does not export %s" % s
    assert sorted(capi.SYMBOLS) == declared


def test_result_record_is_32_bytes():
    from bitmapperbs_amd import capi
    assert capi.RESULT_DTYPE.itemsize == 32


def test_no_cpu_fallback_without_device():
    """On a box without a HIP device bmbs_create must fail (NULL), never silently fall back."""
    import ctypes as C
    from bitmapperbs_amd import capi
    import torch
    if torch.cuda.is_available():
        return
    assert not capi.lib().bmbs_create(0, C.byref(capi.default_params()))


def test_product_does_not_reference_oracle():
    """nothing under bitmapperbs_amd/ may import, link or execute oracle/"""
    pk = os.path.join(ROOT, "bitmapperbs_amd")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "orc_" not in txt and "import orc" not in txt, os.path.join(dp, f)


def test_library_was_built_from_the_sources_in_this_tree():
    """bmbs_build_id() = sha256 over the library's sources at build time (csrc/Makefile); the .so is git-ignored but travels to the GPU
    box, so every run can (and bench.py does) show that what it loaded matches the sources next to it"""
    from bitmapperbs_amd import capi
    assert capi.build_id() == capi.sources_id()


def test_device_index_builder_fails_loudly_without_a_gpu(tmp_path):
    """no silent fall-back to the host builder: without a HIP device bmbs_index_build_device returns BMBS_ENODEV (-19)"""
    import torch
    from bitmapperbs_amd import capi
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    fa = tmp_path / "g.fa"
    fa.write_text(">c\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    assert capi.lib().bmbs_index_build_device(0, str(fa).encode(), str(fa).encode(), 1) == -19
    assert not (tmp_path / "g.fa.index").exists()


def test_max_cigar_ops_bound_follows_the_penalties():
    """host arithmetic only: 2k + 8 slots with the default penalties, more when gaps are cheaper than mismatches, -1 for bad lengths"""
    import ctypes as C
    from bitmapperbs_amd import capi
    lib = capi.lib()
    assert lib.bmbs_max_cigar_ops(None, 150) == 2 * 12 + 8          # -e 0.08 default: k = 12
    assert lib.bmbs_max_cigar_ops(None, 0) == -1 and lib.bmbs_max_cigar_ops(None, 1001) == -1
    p = capi.default_params(e_f=0.08, mp_max=9, mp_min=8, gap_open=1, gap_ext=3)
    k = int(0.08 * 116)
    assert lib.bmbs_max_cigar_ops(C.byref(p), 116) == 2 * (k * 9 // 4) + 8
    p = capi.default_params(e_f=0.04)
    assert lib.bmbs_max_cigar_ops(C.byref(p), 150) == 2 * 6 + 8
