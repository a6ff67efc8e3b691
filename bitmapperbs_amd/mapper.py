"""Host-side mirror of the reference's `--search` mapping drivers over the C-ABI.

`Index`   == Load_Index / load_index (Index.cpp:940, bwt.cpp:2458): the on-disk index files.
`Mapper`  == Prepare_alignment + Map_Single_Seq (Schema.cpp:639, 26330) for one GPU: batches of
             equal-length reads in, 32-byte result records + CIGAR pool out; SAM text is formatted on
             the host from the records (output_sam_end_to_end, Schema.cpp:11989-12039).
All mapping work happens in libbmbs_hip.so on the GPU; this module only moves buffers and prints.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP[_a] = _b


class Index:
    def __init__(self, prefix: str):
        self._lib = capi.lib()
        self._h = self._lib.bmbs_index_file_load(prefix.encode())
        if not self._h:
            raise FileNotFoundError(f"cannot load BitMapperBS index files at {prefix}.index*")
        self.view = capi.IndexView()
        self._lib.bmbs_index_file_view(self._h, C.byref(self.view))
        self.chrom_names = [self._lib.bmbs_index_file_chrom_name(self._h, i).decode()
                            for i in range(self.view.n_chrom)]
        self.chrom_len = np.ctypeslib.as_array(
            C.cast(self.view.chrom_len, C.POINTER(C.c_uint64)), shape=(self.view.n_chrom,)).copy()
        self.ref_len = int(self.view.ref_len)

    @staticmethod
    def build(fasta: str, prefix: str | None = None, threads: int = 8, device: int | None = None) -> None:
        """createIndex equivalent; device=None: host cores (bmbs_index_build), device=d: the GPU builder (same files)"""
        if device is None:
            rc = capi.lib().bmbs_index_build(fasta.encode(), (prefix or fasta).encode(), threads)
        else:
            rc = capi.lib().bmbs_index_build_device(device, fasta.encode(), (prefix or fasta).encode(), threads)
        if rc:
            raise RuntimeError(f"bmbs_index_build{'' if device is None else '_device'} failed ({rc})")

    def close(self):
        if self._h:
            self._lib.bmbs_index_file_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mapper:
    def __init__(self, index: Index, device: int = 0, share: "Mapper | None" = None, **params):
        """share: another Mapper on the same device whose attached index this one uses (bmbs_index_share) -- for two batches
        in flight from two host threads; `share` has to stay open for as long as this one is used"""
        self._lib = capi.lib()
        self.params = capi.default_params(**params)
        self._ctx = self._lib.bmbs_create(device, C.byref(self.params))
        if not self._ctx:
            raise RuntimeError("bmbs_create failed: no usable HIP device (the mapper has no CPU path)")
        self.index = index
        self._owner = share
        if share is None:
            self._chk(self._lib.bmbs_index_attach(self._ctx, C.byref(index.view)))
        else:
            self._chk(self._lib.bmbs_index_share(self._ctx, share._ctx))

    def _chk(self, rc: int):
        if rc:
            raise RuntimeError(f"bmbs error {rc}: {self._lib.bmbs_last_error(self._ctx).decode()}")

    def close(self):
        if self._ctx:
            self._lib.bmbs_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def threshold(self, L: int) -> int:
        k = int(np.uint64(self.params.e_f * L))   # Schema.cpp:24546
        return min(k, 31)

    def max_cigar_ops(self, L: int) -> int:
        """slots per read the cigar pool needs (bmbs_max_cigar_ops): 2k + 8 with the default penalties"""
        return int(self._lib.bmbs_max_cigar_ops(C.byref(self.params), int(L)))

    # ---- fused single-end mapping ------------------------------------------------------------------
    def map_se(self, seq: np.ndarray, qual: np.ndarray, L: int | None = None):
        """seq/qual: uint8 [n, stride] (ASCII, upper case).  -> (results[n] RESULT_DTYPE, cigar_pool u32)"""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        n, stride = seq.shape
        L = stride if L is None else L
        res = np.zeros(n, dtype=capi.RESULT_DTYPE)
        cap = max(1, n * self.max_cigar_ops(L))
        pool = np.zeros(cap, dtype=np.uint32)
        used = C.c_int64(0)
        self._chk(self._lib.bmbs_map_se(self._ctx, capi.ptr(seq), capi.ptr(qual), L, stride, n, capi.ptr(res),
                                        capi.ptr(pool), cap, C.byref(used)))
        return res, pool[:used.value]

    def map_se_device(self, d_seq: int, d_qual: int, L: int, stride: int, n: int, d_results: int,
                      d_cigar_pool: int, cigar_cap: int):
        self._chk(self._lib.bmbs_map_se_device(self._ctx, d_seq, d_qual, L, stride, n, d_results, d_cigar_pool, cigar_cap))

    # ---- fused paired-end mapping (fast mode) ------------------------------------------------------
    def map_pe(self, seq1, qual1, seq2, qual2, L: int | None = None):
        """seq2/qual2 = mate 2 as in the FASTQ file.  -> (results[2n] (mate1, mate2 interleaved), cigar_pool)"""
        a = [np.ascontiguousarray(x, dtype=np.uint8) for x in (seq1, qual1, seq2, qual2)]
        n, stride = a[0].shape
        L = stride if L is None else L
        res = np.zeros(2 * n, dtype=capi.RESULT_DTYPE)
        cap = max(1, 2 * n * self.max_cigar_ops(L))
        pool = np.zeros(cap, dtype=np.uint32)
        used = C.c_int64(0)
        self._chk(self._lib.bmbs_map_pe(self._ctx, capi.ptr(a[0]), capi.ptr(a[1]), capi.ptr(a[2]), capi.ptr(a[3]), L, stride, n,
                                        capi.ptr(res), capi.ptr(pool), cap, C.byref(used)))
        return res, pool[:used.value]

    def map_pe_device(self, d_seq1, d_qual1, d_seq2, d_qual2, L, stride, n, d_results, d_cigar_pool, cigar_cap):
        self._chk(self._lib.bmbs_map_pe_device(self._ctx, d_seq1, d_qual1, d_seq2, d_qual2, L, stride, n, d_results,
                                               d_cigar_pool, cigar_cap))

    # ---- reads of different lengths in one batch (trimmed libraries) -------------------------------
    def map_se_var(self, seq: np.ndarray, qual: np.ndarray, lens: np.ndarray):
        """seq/qual: uint8 [n, stride]; lens[i] <= stride valid characters in row i"""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint16)
        n, stride = seq.shape
        L = int(lens.max()) if n else 1
        res = np.zeros(n, dtype=capi.RESULT_DTYPE)
        cap = max(1, n * self.max_cigar_ops(L))
        pool = np.zeros(cap, dtype=np.uint32)
        used = C.c_int64(0)
        self._chk(self._lib.bmbs_map_se_var(self._ctx, capi.ptr(seq), capi.ptr(qual), capi.ptr(lens), L, stride, n, capi.ptr(res),
                                            capi.ptr(pool), cap, C.byref(used)))
        return res, pool[:used.value]

    def map_pe_var(self, seq1, qual1, seq2, qual2, lens1, lens2):
        a = [np.ascontiguousarray(x, dtype=np.uint8) for x in (seq1, qual1, seq2, qual2)]
        l1 = np.ascontiguousarray(lens1, dtype=np.uint16); l2 = np.ascontiguousarray(lens2, dtype=np.uint16)
        n, stride = a[0].shape
        L = int(max(l1.max(), l2.max())) if n else 1
        res = np.zeros(2 * n, dtype=capi.RESULT_DTYPE)
        cap = max(1, 2 * n * self.max_cigar_ops(L))
        pool = np.zeros(cap, dtype=np.uint32)
        used = C.c_int64(0)
        self._chk(self._lib.bmbs_map_pe_var(self._ctx, capi.ptr(a[0]), capi.ptr(a[1]), capi.ptr(a[2]), capi.ptr(a[3]), capi.ptr(l1),
                                            capi.ptr(l2), L, stride, n, capi.ptr(res), capi.ptr(pool), cap, C.byref(used)))
        return res, pool[:used.value]

    # ---- packed reads: 2 bits per base + an 'N' plane (bmbs_map_*_packed) -----------------------------
    @staticmethod
    def pack_rows(seq: np.ndarray, L: int, lens: np.ndarray | None = None, pwords: int | None = None, threads: int = 8) -> np.ndarray:
        """ASCII rows [n][stride] -> packed rows [n][pwords] of u64 (bmbs_pack_rows); raises when a row holds something other than ACGTN"""
        a = np.ascontiguousarray(seq, dtype=np.uint8)
        n, stride = a.shape
        if pwords is None:
            pwords = (L + 31) // 32 + (L + 63) // 64
        rows = np.empty((n, pwords), dtype=np.uint64)
        bad = C.c_int64(-1)
        ln = np.ascontiguousarray(lens, dtype=np.uint16) if lens is not None else None
        rc = capi.lib().bmbs_pack_rows(capi.ptr(a), L, stride, n, capi.ptr(ln) if ln is not None else None, capi.ptr(rows), pwords, threads, C.byref(bad))
        if rc:
            raise ValueError("bmbs_pack_rows: row %d cannot be packed (rc %d)" % (bad.value, rc))
        return rows

    def map_se_packed(self, rows: np.ndarray, qual: np.ndarray, L: int, lens: np.ndarray | None = None):
        q = np.ascontiguousarray(qual, dtype=np.uint8)
        n, stride = q.shape
        res = np.zeros(n, dtype=capi.RESULT_DTYPE)
        cap = max(1, n * self.max_cigar_ops(L))
        pool = np.zeros(cap, dtype=np.uint32)
        used = C.c_int64(0)
        ln = np.ascontiguousarray(lens, dtype=np.uint16) if lens is not None else None
        self._chk(self._lib.bmbs_map_se_packed(self._ctx, capi.ptr(rows), rows.shape[1], capi.ptr(q), capi.ptr(ln) if ln is not None else None, L, stride, n,
                                               capi.ptr(res), capi.ptr(pool), cap, C.byref(used)))
        return res, pool[:used.value]

    def map_pe_packed(self, rows1: np.ndarray, rows2: np.ndarray, qual1: np.ndarray, qual2: np.ndarray, L: int, lens1=None, lens2=None):
        q1 = np.ascontiguousarray(qual1, dtype=np.uint8); q2 = np.ascontiguousarray(qual2, dtype=np.uint8)
        n, stride = q1.shape
        res = np.zeros(2 * n, dtype=capi.RESULT_DTYPE)
        cap = max(1, 2 * n * self.max_cigar_ops(L))
        pool = np.zeros(cap, dtype=np.uint32)
        used = C.c_int64(0)
        l1 = np.ascontiguousarray(lens1, dtype=np.uint16) if lens1 is not None else None
        l2 = np.ascontiguousarray(lens2, dtype=np.uint16) if lens2 is not None else None
        self._chk(self._lib.bmbs_map_pe_packed(self._ctx, capi.ptr(rows1), capi.ptr(rows2), rows1.shape[1], capi.ptr(q1), capi.ptr(q2),
                                               capi.ptr(l1) if l1 is not None else None, capi.ptr(l2) if l2 is not None else None, L, stride, n,
                                               capi.ptr(res), capi.ptr(pool), cap, C.byref(used)))
        return res, pool[:used.value]

    # ---- FASTQ text in, SAM text out (newline index and SAM formatting on the device) ---------------
    TEXT_PBAT, TEXT_UNMAPPED, TEXT_BAM = 1, 2, 16

    def _set_refs(self):
        if getattr(self, "_refs_set", False):
            return
        names = (C.c_char_p * len(self.index.chrom_names))(*[n.encode() for n in self.index.chrom_names])
        self._chk(self._lib.bmbs_sam_refs(self._ctx, names, len(self.index.chrom_names)))
        self._refs_set = True

    def map_text(self, text1: bytes, n: int, text2: bytes | None = None, flags: int = 0, cap: int | None = None) -> bytes:
        """FASTQ text of n records (pairs with text2) -> the SAM lines the reference prints for them, in input order (flags & TEXT_BAM:
        their BAM records as complete BGZF blocks).  cap: bytes of the output buffer handed to the library (default: ample)"""
        self._set_refs()
        if cap is None:
            cap = len(text1) + (len(text2) if text2 else 0) + (2 if text2 else 1) * n * 640 + 4096
        out = np.empty(cap, dtype=np.uint8)
        used = C.c_uint64(0); lines = C.c_int64(0)
        a1 = np.frombuffer(text1, dtype=np.uint8)
        if text2 is None:
            self._chk(self._lib.bmbs_map_se_text(self._ctx, capi.ptr(a1), a1.size, n, flags, capi.ptr(out), cap, C.byref(used), C.byref(lines)))
        else:
            a2 = np.frombuffer(text2, dtype=np.uint8)
            self._chk(self._lib.bmbs_map_pe_text(self._ctx, capi.ptr(a1), a1.size, capi.ptr(a2), a2.size, n, flags, capi.ptr(out), cap,
                                                 C.byref(used), C.byref(lines)))
        return out[:used.value].tobytes()

    def inflate_bgzf(self, data: bytes, shift: int | None = None):
        """the text of a BGZF byte string (bgzip's format), inflated on the device (bmbs_inflate_bgzf); raises on a malformed block"""
        import struct
        blk = [0]; out = [0]
        at = 0
        while at < len(data):
            bs = struct.unpack("<H", data[at + 16:at + 18])[0] + 1
            isz = struct.unpack("<I", data[at + bs - 4:at + bs])[0]
            at += bs; blk.append(at); out.append(out[-1] + isz)
        n = len(blk) - 1
        a = np.frombuffer(data, dtype=np.uint8)
        b = np.array(blk, dtype=np.uint64); o = np.array(out, dtype=np.uint64)
        text = np.empty(max(1, out[-1]), dtype=np.uint8)
        if shift is None:
            self._chk(self._lib.bmbs_inflate_bgzf(self._ctx, capi.ptr(a), a.size, capi.ptr(b), capi.ptr(o), n, capi.ptr(text), out[-1], None, 0))
            return text[:out[-1]].tobytes()
        cnt = np.zeros(max(1, (shift + out[-1] + 65535) >> 16), dtype=np.uint32)
        self._chk(self._lib.bmbs_inflate_bgzf(self._ctx, capi.ptr(a), a.size, capi.ptr(b), capi.ptr(o), n, capi.ptr(text), out[-1], capi.ptr(cnt), shift))
        return text[:out[-1]].tobytes(), cnt

    @staticmethod
    def _ztext(prefix: bytes, blocks: bytes, keep):
        """a capi.ZText over `prefix` + the BGZF blocks in `blocks` (objects that must stay alive are appended to `keep`)"""
        import struct
        blk = [0]; out = [0]
        at = 0
        while at < len(blocks):
            bs = struct.unpack("<H", blocks[at + 16:at + 18])[0] + 1
            isz = struct.unpack("<I", blocks[at + bs - 4:at + bs])[0]
            at += bs; blk.append(at); out.append(out[-1] + isz)
        a = np.frombuffer(blocks, dtype=np.uint8) if blocks else np.zeros(1, dtype=np.uint8)
        pf = np.frombuffer(prefix, dtype=np.uint8) if prefix else np.zeros(1, dtype=np.uint8)
        b = np.array(blk, dtype=np.uint64); o = np.array(out, dtype=np.uint64)
        keep += [a, pf, b, o]
        z = capi.ZText()
        z.prefix = capi.ptr(pf) if prefix else None; z.prefix_bytes = len(prefix)
        z.comp = capi.ptr(a); z.comp_bytes = len(blocks); z.blk_off = capi.ptr(b); z.out_off = capi.ptr(o); z.n_blocks = len(blk) - 1
        return z

    def text_open_bgzf(self, w1, w2=None, max_records=1 << 30, last=(False, False), tail_cap=1 << 24):
        """w = (prefix bytes, BGZF blocks): the window(s) are assembled and indexed on the device -> (records, tail1, tail2)"""
        self._set_refs()
        keep = []
        z1 = self._ztext(w1[0], w1[1], keep)
        z2 = self._ztext(w2[0], w2[1], keep) if w2 is not None else None
        t1 = np.empty(tail_cap, dtype=np.uint8); t2 = np.empty(tail_cap, dtype=np.uint8)
        n = C.c_int64(0); b1 = C.c_uint64(0); b2 = C.c_uint64(0)
        self._chk(self._lib.bmbs_text_open_bgzf(self._ctx, C.byref(z1), C.byref(z2) if z2 is not None else None, max_records, int(last[0]), int(last[1]),
                                                C.byref(n), capi.ptr(t1), tail_cap, C.byref(b1), capi.ptr(t2), C.byref(b2)))
        return int(n.value), t1[:b1.value].tobytes(), t2[:b2.value].tobytes()

    def text_map_open(self, flags: int = 0, cap: int = 1 << 28) -> bytes:
        out = np.empty(cap, dtype=np.uint8)
        used = C.c_uint64(0); lines = C.c_int64(0)
        self._chk(self._lib.bmbs_text_map_open(self._ctx, flags, capi.ptr(out), cap, C.byref(used), C.byref(lines)))
        return out[:used.value].tobytes()

    def sync(self):
        self._chk(self._lib.bmbs_sync(self._ctx))

    # ---- stages -------------------------------------------------------------------------------------
    def locate(self, rows: np.ndarray) -> np.ndarray:
        """K5: text position SA[row] of suffix-array rows"""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.zeros(rows.size, dtype=np.uint64)
        self._chk(self._lib.bmbs_locate_batch(self._ctx, capi.ptr(rows), rows.size, capi.ptr(out)))
        return out

    def vote_order(self, vote: np.ndarray, seg_off: np.ndarray, form: int) -> np.ndarray:
        """a9: the visiting order std::sort gives each vote list vote[seg_off[s]:seg_off[s+1]] (form 0: wave kernel, 1: block kernel)"""
        vote = np.ascontiguousarray(vote, dtype=np.uint8)
        seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
        perm = np.zeros(vote.size, dtype=np.uint32)
        self._chk(self._lib.bmbs_vote_order_batch(self._ctx, capi.ptr(vote), capi.ptr(seg_off), seg_off.size - 1, form, capi.ptr(perm)))
        return perm

    def windows(self, site: np.ndarray, length: int) -> np.ndarray:
        """K7: the doubled-genome windows starting at `site` (uint64), uint8 [n, length] (0 bytes = out-of-strand)"""
        site = np.ascontiguousarray(site, dtype=np.uint64)
        out = np.zeros((site.size, length), dtype=np.uint8)
        self._chk(self._lib.bmbs_window_batch(self._ctx, capi.ptr(site), site.size, length, capi.ptr(out)))
        return out

    def filter(self, seq: np.ndarray, L: int, read_of: np.ndarray, site: np.ndarray, packed: bool = False):
        """K7+K8 on (read, site) pairs.  packed: the rows are packed on the device first and the Myers rows read the packed words -- the
        form the mapping calls run (bmbs_filter_batch_packed)"""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        read_of = np.ascontiguousarray(read_of, dtype=np.uint32)
        site = np.ascontiguousarray(site, dtype=np.uint64)
        m = read_of.size
        err = np.zeros(m, dtype=np.uint32)
        end = np.zeros(m, dtype=np.int32)
        fn = self._lib.bmbs_filter_batch_packed if packed else self._lib.bmbs_filter_batch
        self._chk(fn(self._ctx, capi.ptr(seq), L, seq.shape[1], seq.shape[0], capi.ptr(read_of),
                                              capi.ptr(site), m, capi.ptr(err), capi.ptr(end)))
        return err, end

    def align(self, seq, qual, L, read_of, site, end_in, err_in):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        read_of = np.ascontiguousarray(read_of, dtype=np.uint32)
        site = np.ascontiguousarray(site, dtype=np.uint64)
        end_in = np.ascontiguousarray(end_in, dtype=np.int32)
        err_in = np.ascontiguousarray(err_in, dtype=np.uint32)
        m = read_of.size
        max_ops = 2 * self.threshold(L) + 8
        out = {k: np.zeros(m, dtype=t) for k, t in (("start", np.int32), ("end", np.int32), ("nm", np.uint32),
                                                     ("score", np.int32), ("n_ops", np.int32))}
        ops = np.zeros((m, max_ops), dtype=np.uint32)
        self._chk(self._lib.bmbs_align_batch(self._ctx, capi.ptr(seq), capi.ptr(qual), L, seq.shape[1], seq.shape[0],
                                             capi.ptr(read_of), capi.ptr(site), capi.ptr(end_in), capi.ptr(err_in), m,
                                             capi.ptr(out["start"]), capi.ptr(out["end"]), capi.ptr(out["nm"]),
                                             capi.ptr(out["score"]), capi.ptr(ops), capi.ptr(out["n_ops"]), max_ops))
        out["ops"] = ops
        return out

    def seed(self, seq: np.ndarray, L: int, vote_cap: int | None = None):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        n, stride = seq.shape
        cap = vote_cap or max(1024, 64 * n)
        verdict = np.zeros(n, dtype=np.uint8)
        exit_site = np.zeros(n, dtype=np.uint64)
        seg_off = np.zeros(n + 1, dtype=np.uint64)
        n_votes = np.zeros(n, dtype=np.uint32)
        vsite = np.zeros(cap, dtype=np.uint64)
        vcnt = np.zeros(cap, dtype=np.uint32)
        tot = C.c_int64(0)
        self._chk(self._lib.bmbs_seed_batch(self._ctx, capi.ptr(seq), L, stride, n, capi.ptr(verdict), capi.ptr(exit_site),
                                            capi.ptr(seg_off), capi.ptr(n_votes), capi.ptr(vsite), capi.ptr(vcnt), cap,
                                            C.byref(tot)))
        return dict(verdict=verdict, exit_site=exit_site, seg_off=seg_off, n_votes=n_votes, vote_site=vsite[:tot.value],
                    vote_cnt=vcnt[:tot.value])

    # ---- stats / measurement -----------------------------------------------------------------------
    def stats(self) -> np.ndarray:
        s = np.zeros(5, dtype=np.int64)
        self._chk(self._lib.bmbs_stats_get(self._ctx, capi.ptr(s)))
        return s

    def reset_stats(self):
        self._chk(self._lib.bmbs_stats_reset(self._ctx))

    def profile(self) -> list[tuple[str, float]]:
        n = C.c_int(512)
        names = (C.c_char_p * 512)()
        ms = (C.c_float * 512)()
        self._chk(self._lib.bmbs_profile_last(self._ctx, names, ms, C.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]

    def profile_total(self) -> tuple[dict, int]:
        """({kernel: ms summed over the mapping calls since profile_reset()}, number of lane-calls)"""
        n = C.c_int(512)
        names = (C.c_char_p * 512)()
        ms = (C.c_double * 512)()
        calls = C.c_int64(0)
        self._chk(self._lib.bmbs_profile_total(self._ctx, names, ms, C.byref(n), C.byref(calls)))
        return {names[i].decode(): float(ms[i]) for i in range(n.value)}, int(calls.value)

    def profile_reset(self):
        self._chk(self._lib.bmbs_profile_reset(self._ctx))

    def retries(self) -> int:
        """calls that were issued again with exact sizes (a stage count exceeded the learned capacity); diagnostic"""
        return int(self._lib.bmbs_retries(self._ctx))

    def counters(self) -> dict:
        c = np.zeros(32, dtype=np.uint64)
        self._chk(self._lib.bmbs_counters_all(self._ctx, capi.ptr(c)))
        keys = ("n_hash", "n_ext", "n_sa", "n_filter", "n_sw", "n_ungapped", "n_cand_slots", "n_jobs")
        d = {k: int(v) for k, v in zip(keys, c[:8])}
        d["n_jump"] = int(c[8])          # three-letter index steps taken (each counts three in n_ext)
        # what the list kernels handled (round 6): candidates located by the mid / wave / block forms, their lists, the entries
        # filter_pairs read, re-seeded candidates (--sensitive), sites dropped by the paired-end pre-filter
        for j, nm in enumerate(("n_cand_mid", "n_cand_long", "n_cand_big", "n_lists_long", "n_pef_entries", "n_cand_reseed", "n_prefilter_drop")):
            d[nm] = int(c[9 + j])
        for kid, nm in enumerate(("k_seed_first", "k_seed_second", "k_seed_extra")):
            d[nm] = {"n_hash": int(c[16 + 4 * kid]), "n_ext": int(c[17 + 4 * kid]), "n_sa": int(c[18 + 4 * kid])}
        return d


# ---- SAM text (host emit) --------------------------------------------------------------------------
def cigar_text(res_row, pool: np.ndarray, L: int) -> str:
    n = int(res_row["n_cigar"])
    if n == 0:
        return "%dM" % L
    if n == 255:
        raise ValueError("an alignment overflowed its CIGAR slots (bmbs_max_cigar_ops rules this out: internal error)")
    ops = pool[int(res_row["cigar_off"]):int(res_row["cigar_off"]) + n]
    return "".join("%d%s" % (int(o) >> 4, "MDISH"[int(o) & 0xf]) for o in ops)


def sam_header(index: Index, command_line: str = "") -> str:
    """OutPutSAM_Nounheader, Process_sam_out.cpp:1137-1153"""
    out = ["@HD\tVN:1.4\tSO:unsorted"]
    for nm, ln in zip(index.chrom_names, index.chrom_len):
        out.append("@SQ\tSN:%s\tLN:%d" % (nm, int(ln)))
    out.append("@PG\tID:BitMapperBS\tVN:1.0.2.3\tCL:%s" % command_line)
    return "\n".join(out) + "\n"


def sam_lines_se(index: Index, names, seq: np.ndarray, qual: np.ndarray, L: int, res: np.ndarray, pool: np.ndarray):
    """output_sam_end_to_end text branch (Schema.cpp:11989-12039); one line per uniquely mapped read."""
    out = []
    for i in np.nonzero(res["status"] == capi.ST_UNIQUE)[0]:
        r = res[i]
        nm = names[i]
        if isinstance(nm, bytes):
            nm = nm.decode()
        if nm.startswith("@"):
            nm = nm[1:]
        for cut in (" ", "/"):
            j = nm.find(cut)
            if j >= 0:
                nm = nm[:j]
        s, q = seq[i, :L], qual[i, :L]
        if int(r["flag"]) & 16:
            s = _COMP[s][::-1]
            q = q[::-1]
        out.append("%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\tNM:i:%d\n" % (
            nm, int(r["flag"]), index.chrom_names[int(r["chrom"])], int(r["pos"]), int(r["mapq"]),
            cigar_text(r, pool, L), s.tobytes().decode(), q.tobytes().decode(), int(r["nm"])))
    return out


def pe_name(n1, n2) -> str:
    """inputReads_paired_directly (Process_Reads.cpp:296-307): cut at the first differing char, ' ' or '/'"""
    a = n1.decode() if isinstance(n1, bytes) else n1
    b = n2.decode() if isinstance(n2, bytes) else n2
    j = 0
    while j < len(a) and j < len(b) and a[j] == b[j] and a[j] not in " /":
        j += 1
    a = a[:j]
    return a[1:] if a.startswith("@") else a


def sam_lines_pe(index: Index, names1, names2, seq1, qual1, seq2, qual2, L: int, res: np.ndarray, pool: np.ndarray):
    """directly_output_read1 / directly_output_read2 (Schema.cpp:10537-10640, 11494-11590); two lines per unique pair"""
    out = []
    n = seq1.shape[0]
    for i in range(n):
        r1, r2 = res[2 * i], res[2 * i + 1]
        if int(r1["status"]) != capi.ST_UNIQUE:
            continue
        nm = pe_name(names1[i], names2[i])
        tlen = int(r1["tlen"])
        p1, p2 = int(r1["pos"]), int(r2["pos"])
        t1 = "-%d" % tlen if p2 < p1 else "%d" % tlen           # read 1: negative only if the mate lies to the left
        t2 = "%d" % tlen if p1 > p2 else "-%d" % tlen           # read 2: positive only if the mate lies to the right
        s1, q1 = seq1[i, :L], qual1[i, :L]
        if not (int(r1["flag"]) & 32):
            s1 = _COMP[s1][::-1]; q1 = q1[::-1]
        s2, q2 = seq2[i, :L], qual2[i, :L]                       # FASTQ orientation
        if int(r2["flag"]) & 16:
            s2 = _COMP[s2][::-1]; q2 = q2[::-1]
        chrom1 = index.chrom_names[int(r1["chrom"])]; chrom2 = index.chrom_names[int(r2["chrom"])]
        out.append("%s\t%d\t%s\t%d\t%d\t%s\t=\t%d\t%s\t%s\t%s\tNM:i:%d\n" % (
            nm, int(r1["flag"]), chrom1, p1, int(r1["mapq"]), cigar_text(r1, pool, L), p2, t1,
            s1.tobytes().decode(), q1.tobytes().decode(), int(r1["nm"])))
        out.append("%s\t%d\t%s\t%d\t%d\t%s\t=\t%d\t%s\t%s\t%s\tNM:i:%d\n" % (
            nm, int(r2["flag"]), chrom2, p2, int(r2["mapq"]), cigar_text(r2, pool, L), p1, t2,
            s2.tobytes().decode(), q2.tobytes().decode(), int(r2["nm"])))
    return out
