"""ctypes binding of the C-ABI in include/bmbs.h (libbmbs_hip.so, built in-tree by
``__graft_entry__.build()`` / ``make -C bitmapperbs_amd/csrc``).

The library is the product: if it is missing or cannot create a context on a HIP device the calls
raise -- there is no Python/CPU fallback for any stage.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BMBS_LIB") or os.path.join(_HERE, "libbmbs_hip.so")      # (BMBS_LIB: an experimental build, A/B runs of tools/)

# every symbol include/bmbs.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "bmbs_default_params", "bmbs_create", "bmbs_destroy", "bmbs_last_error", "bmbs_index_attach", "bmbs_index_share",
    "bmbs_locate_batch", "bmbs_vote_order_batch", "bmbs_window_batch", "bmbs_filter_batch", "bmbs_filter_batch_packed", "bmbs_align_batch", "bmbs_seed_batch", "bmbs_map_se", "bmbs_map_se_device",
    "bmbs_map_se_fastq", "bmbs_map_pe_fastq", "bmbs_map_pe", "bmbs_map_pe_device", "bmbs_map_se_var", "bmbs_map_se_var_device", "bmbs_map_pe_var", "bmbs_map_pe_var_device",
    "bmbs_sync", "bmbs_stats_get", "bmbs_stats_reset", "bmbs_stats_allreduce", "bmbs_profile_last",
    "bmbs_counters_last", "bmbs_counters_all", "bmbs_index_file_load", "bmbs_index_file_view", "bmbs_index_file_chrom_name",
    "bmbs_index_file_free", "bmbs_index_build", "bmbs_index_build_device", "bmbs_host_alloc", "bmbs_host_free", "bmbs_build_id",
    "bmbs_max_cigar_ops", "bmbs_host_prefault", "bmbs_reserve", "bmbs_host_alloc_kind", "bmbs_retries", "bmbs_text_times", "bmbs_pack_rows", "bmbs_map_se_packed", "bmbs_map_pe_packed", "bmbs_sam_refs", "bmbs_map_se_text", "bmbs_map_pe_text", "bmbs_profile_total", "bmbs_profile_reset", "bmbs_inflate_bgzf", "bmbs_debug_huff_lengths", "bmbs_text_open_bgzf", "bmbs_text_map_open",
]


class Params(C.Structure):
    _fields_ = [("e_f", C.c_double), ("mp_max", C.c_int32), ("mp_min", C.c_int32), ("np", C.c_int32),
                ("gap_open", C.c_int32), ("gap_ext", C.c_int32), ("q_base", C.c_int32),
                ("seed_len", C.c_int32), ("min_ins", C.c_int32), ("max_ins", C.c_int32),
                ("sensitive", C.c_int32), ("ambiguous_out", C.c_int32)]


class FastqView(C.Structure):
    _fields_ = [("text", C.c_void_p), ("text_bytes", C.c_uint64), ("seq_off", C.c_void_p), ("qual_off", C.c_void_p),
                ("seq_len", C.c_void_p), ("qual_len", C.c_void_p)]


class ZText(C.Structure):
    _fields_ = [("prefix", C.c_void_p), ("prefix_bytes", C.c_uint64), ("comp", C.c_void_p), ("comp_bytes", C.c_uint64),
                ("blk_off", C.c_void_p), ("out_off", C.c_void_p), ("n_blocks", C.c_int64)]


class IndexView(C.Structure):
    _fields_ = [("ref_len", C.c_uint64), ("pac", C.c_void_p), ("pac_bytes", C.c_uint64),
                ("sa_length", C.c_uint64), ("shapline", C.c_uint64), ("nacgt", C.c_uint64 * 5),
                ("bwt", C.c_void_p), ("bwt_words", C.c_uint64), ("high_occ", C.c_void_p),
                ("high_occ_words", C.c_uint64), ("hash_hi", C.c_void_p), ("hash_lo", C.c_void_p),
                ("hash_entries", C.c_uint64), ("sa", C.c_void_p), ("sa_entries", C.c_uint64),
                ("sa_flag", C.c_void_p), ("sa_flag_words", C.c_uint64), ("n_chrom", C.c_int32),
                ("chrom_len", C.c_void_p)]


# numpy view of bmbs_result (32 bytes)
RESULT_DTYPE = np.dtype([("pos", "<u8"), ("cigar_off", "<u4"), ("chrom", "<i4"), ("flag", "<u2"), ("nm", "<u2"),
                         ("score", "<i2"), ("status", "u1"), ("mapq", "u1"), ("n_cigar", "u1"), ("path", "u1"),
                         ("n_cand", "<u2"), ("tlen", "<u4")])
assert RESULT_DTYPE.itemsize == 32

ST_UNMAPPED, ST_UNIQUE, ST_AMBIG, ST_OFFEND = 0, 1, 2, 3

_lib = None


def _torch_runtime_first() -> None:
    """PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, no versioned soname) next to the system one
    this library links (/opt/rocm/lib/libamdhip64.so.7); both sit on ONE libhsa-runtime64.so.1 -- whichever copy the process
    loads first.  With the system ROCr loaded first, torch's older HIP runtime finds no device afterwards
    (torch.cuda.is_available() turns False); the other order works.  So when torch is importable it is initialised before
    libbmbs_hip.so is opened.  Plumbing only: nothing here computes with torch.  BMBS_SKIP_TORCH_PRELOAD=1 skips it."""
    if os.environ.get("BMBS_SKIP_TORCH_PRELOAD"):
        return
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass


def lib() -> C.CDLL:
    """Load libbmbs_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    _torch_runtime_first()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C bitmapperbs_amd/csrc` (the HIP library is mandatory, there is no CPU path)")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    L.bmbs_default_params.argtypes = [C.POINTER(Params)]
    L.bmbs_default_params.restype = None
    L.bmbs_create.argtypes = [C.c_int, C.POINTER(Params)]
    L.bmbs_create.restype = vp
    L.bmbs_destroy.argtypes = [vp]
    L.bmbs_destroy.restype = None
    L.bmbs_last_error.argtypes = [vp]
    L.bmbs_last_error.restype = C.c_char_p
    L.bmbs_index_attach.argtypes = [vp, C.POINTER(IndexView)]
    L.bmbs_index_share.argtypes = [vp, vp]
    L.bmbs_locate_batch.argtypes = [vp, vp, i64, vp]
    L.bmbs_window_batch.argtypes = [vp, vp, i64, i32, vp]
    L.bmbs_vote_order_batch.argtypes = [vp, vp, vp, i64, i32, vp]
    L.bmbs_filter_batch.argtypes = [vp, vp, i32, i32, i64, vp, vp, i64, vp, vp]
    if hasattr(L, "bmbs_filter_batch_packed"):          # (BMBS_LIB may name an older build in same-box comparisons: tools/ab_*.sh)
        L.bmbs_filter_batch_packed.argtypes = [vp, vp, i32, i32, i64, vp, vp, i64, vp, vp]
        L.bmbs_filter_batch_packed.restype = C.c_int
    L.bmbs_align_batch.argtypes = [vp, vp, vp, i32, i32, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, i32]
    L.bmbs_seed_batch.argtypes = [vp, vp, i32, i32, i64, vp, vp, vp, vp, vp, vp, i64, C.POINTER(i64)]
    L.bmbs_map_se.argtypes = [vp, vp, vp, i32, i32, i64, vp, vp, i64, C.POINTER(i64)]
    L.bmbs_map_se_device.argtypes = [vp, u64, u64, i32, i32, i64, u64, u64, i64]
    L.bmbs_map_pe.argtypes = [vp, vp, vp, vp, vp, i32, i32, i64, vp, vp, i64, C.POINTER(i64)]
    L.bmbs_map_pe_device.argtypes = [vp, u64, u64, u64, u64, i32, i32, i64, u64, u64, i64]
    L.bmbs_map_se_var.argtypes = [vp, vp, vp, vp, i32, i32, i64, vp, vp, i64, C.POINTER(i64)]
    L.bmbs_map_se_var_device.argtypes = [vp, u64, u64, u64, i32, i32, i64, u64, u64, i64]
    L.bmbs_map_pe_var.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, vp, vp, i64, C.POINTER(i64)]
    L.bmbs_map_pe_var_device.argtypes = [vp, u64, u64, u64, u64, u64, i32, i32, i64, u64, u64, i64]
    L.bmbs_pack_rows.argtypes = [vp, i32, i32, i64, vp, vp, i32, i32, C.POINTER(i64)]
    L.bmbs_pack_rows.restype = C.c_int
    L.bmbs_map_se_packed.argtypes = [vp, vp, i32, vp, vp, i32, i32, i64, vp, vp, i64, C.POINTER(i64)]
    L.bmbs_map_se_packed.restype = C.c_int
    L.bmbs_map_pe_packed.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i64, vp, vp, i64, C.POINTER(i64)]
    L.bmbs_map_pe_packed.restype = C.c_int
    L.bmbs_map_se_fastq.argtypes = [vp, C.POINTER(FastqView), i64, i32, i32, i32, vp, vp, i64, C.POINTER(i64)]
    L.bmbs_map_pe_fastq.argtypes = [vp, C.POINTER(FastqView), C.POINTER(FastqView), i64, i32, i32, vp, vp, i64, C.POINTER(i64)]
    L.bmbs_sync.argtypes = [vp]
    L.bmbs_stats_get.argtypes = [vp, vp]
    L.bmbs_stats_reset.argtypes = [vp]
    L.bmbs_stats_allreduce.argtypes = [C.POINTER(vp), C.c_int, vp]
    L.bmbs_profile_last.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.bmbs_counters_last.argtypes = [vp, vp]
    L.bmbs_counters_all.argtypes = [vp, vp]
    L.bmbs_index_file_load.argtypes = [C.c_char_p]
    L.bmbs_index_file_load.restype = vp
    L.bmbs_index_file_view.argtypes = [vp, C.POINTER(IndexView)]
    L.bmbs_index_file_view.restype = None
    L.bmbs_index_file_chrom_name.argtypes = [vp, C.c_int]
    L.bmbs_index_file_chrom_name.restype = C.c_char_p
    L.bmbs_index_file_free.argtypes = [vp]
    L.bmbs_index_file_free.restype = None
    L.bmbs_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.bmbs_index_build_device.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int]
    L.bmbs_build_id.argtypes = []
    L.bmbs_build_id.restype = C.c_char_p
    L.bmbs_max_cigar_ops.argtypes = [C.POINTER(Params), C.c_int32]
    L.bmbs_max_cigar_ops.restype = C.c_int32
    L.bmbs_profile_total.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(i64)]
    L.bmbs_profile_total.restype = C.c_int
    L.bmbs_profile_reset.argtypes = [vp]
    L.bmbs_profile_reset.restype = C.c_int
    L.bmbs_sam_refs.argtypes = [vp, C.POINTER(C.c_char_p), i32]
    L.bmbs_sam_refs.restype = C.c_int
    L.bmbs_map_se_text.argtypes = [vp, vp, u64, i64, i32, vp, u64, C.POINTER(u64), C.POINTER(i64)]
    L.bmbs_inflate_bgzf.argtypes = [vp, vp, u64, vp, vp, i64, vp, u64, vp, u64]
    L.bmbs_inflate_bgzf.restype = C.c_int
    L.bmbs_debug_huff_lengths.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
    L.bmbs_debug_huff_lengths.restype = C.c_int
    L.bmbs_text_open_bgzf.argtypes = [vp, C.POINTER(ZText), C.POINTER(ZText), i64, i32, i32, C.POINTER(i64), vp, u64, C.POINTER(u64), vp, C.POINTER(u64)]
    L.bmbs_text_open_bgzf.restype = C.c_int
    L.bmbs_text_map_open.argtypes = [vp, i32, vp, u64, C.POINTER(u64), C.POINTER(i64)]
    L.bmbs_text_map_open.restype = C.c_int
    L.bmbs_map_se_text.restype = C.c_int
    L.bmbs_map_pe_text.argtypes = [vp, vp, u64, vp, u64, i64, i32, vp, u64, C.POINTER(u64), C.POINTER(i64)]
    L.bmbs_map_pe_text.restype = C.c_int
    L.bmbs_retries.argtypes = [vp]
    L.bmbs_retries.restype = i64
    L.bmbs_host_prefault.argtypes = [vp, vp, u64, i32]
    L.bmbs_host_prefault.restype = C.c_int
    L.bmbs_reserve.argtypes = [vp, u64]
    L.bmbs_reserve.restype = C.c_int
    L.bmbs_host_alloc_kind.argtypes = [u64, i32]
    L.bmbs_host_alloc_kind.restype = vp
    L.bmbs_host_alloc.argtypes = [u64]
    L.bmbs_host_alloc.restype = vp
    L.bmbs_host_free.argtypes = [vp]
    L.bmbs_host_free.restype = None
    for name in ("bmbs_index_attach", "bmbs_index_share", "bmbs_locate_batch", "bmbs_vote_order_batch", "bmbs_window_batch", "bmbs_filter_batch", "bmbs_align_batch", "bmbs_seed_batch", "bmbs_map_se",
                 "bmbs_map_se_fastq", "bmbs_map_pe_fastq", "bmbs_map_pe", "bmbs_map_pe_device", "bmbs_map_se_var", "bmbs_map_se_var_device", "bmbs_map_pe_var", "bmbs_map_pe_var_device",
                 "bmbs_map_se_device", "bmbs_sync", "bmbs_stats_get", "bmbs_stats_reset", "bmbs_stats_allreduce",
                 "bmbs_profile_last", "bmbs_counters_last", "bmbs_counters_all", "bmbs_index_build", "bmbs_index_build_device",
                 "bmbs_map_se_fastq", "bmbs_map_pe_fastq"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


LIB_SRCS = ("bmbs_api.hip", "bmbs_kernels.hip", "k_index.hip", "k_rows.hip", "k_attach.hip", "k_scan.hip", "k_seed.hip", "k_vote.hip", "k_filter.hip", "k_reduce.hip", "k_align.hip", "k_finalize.hip", "k_pe_fast.hip", "k_pe_sensitive.hip",
            "bmbs_textpath.hip", "bmbs_text.hip", "bmbs_bam.hip", "bmbs_inflate.hip", "bmbs_bytes.h", "bmbs_host.h", "bmbs_dev.h", "bmbs_sort.h", "../../include/bmbs.h",
            "index_io.cpp", "index_io.h", "index_build_gpu.hip", "build_id.cpp")


def sources_id() -> str:
    """what bmbs_build_id() of a library built from the sources in this tree returns (csrc/Makefile: LIB_SRCS, BUILD_ID)"""
    import hashlib
    h = hashlib.sha256()
    for f in LIB_SRCS:
        h.update(open(os.path.join(_HERE, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def build_id() -> str:
    return lib().bmbs_build_id().decode()


def default_params(**kw) -> Params:
    p = Params()
    lib().bmbs_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def ptr(a: np.ndarray) -> int:
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data
