"""ctypes binding of oracle/liboracle.so -- the CHECKER.  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")


class OrcParams(C.Structure):
    _fields_ = [("e_f", C.c_double), ("mp_max", C.c_int), ("mp_min", C.c_int), ("np", C.c_int), ("gap_open", C.c_int),
                ("gap_ext", C.c_int), ("q_base", C.c_int), ("seed_len", C.c_int), ("min_ins", C.c_int),
                ("max_ins", C.c_int), ("sensitive", C.c_int), ("unmapped_out", C.c_int), ("ambiguous_out", C.c_int),
                ("pbat", C.c_int)]


REC_DTYPE = np.dtype([("status", "<i4"), ("chrom", "<i4"), ("pos", "<u8"), ("site", "<u8"), ("start_site", "<i4"),
                      ("end_site", "<i4"), ("flag", "<i4"), ("mapq", "<i4"), ("nm", "<i4"), ("score", "<i4"),
                      ("path", "<i4"), ("n_cand", "<i4"), ("n_votes", "<i4"), ("cigar", "S1024"), ("_pad", "<i4")])
PE_REC_DTYPE = np.dtype([("status", "<i4"), ("n_pairs", "<i4"), ("mapq", "<i4"), ("tlen", "<i4"), ("flag1", "<i4"), ("flag2", "<i4"),
                         ("chrom1", "<i4"), ("chrom2", "<i4"), ("pos1", "<u8"), ("pos2", "<u8"), ("nm1", "<i4"), ("nm2", "<i4"),
                         ("score1", "<i4"), ("score2", "<i4"), ("matched1", "<i4"), ("matched2", "<i4"), ("cigar1", "S1024"),
                         ("cigar2", "S1024")])
COUNTER_KEYS = ("n_reads", "n_hash", "n_ext", "n_lf", "n_sa1", "n_locate_rows", "n_cand", "n_sw", "n_ungapped")

_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ODIR], check=True)


def load():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(ODIR, "liboracle.so")
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".cpp", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        build()
    L = C.CDLL(so)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.orc_default_params.argtypes = [C.POINTER(OrcParams)]
    L.orc_index_build.argtypes = [C.c_char_p, C.c_char_p]
    L.orc_index_load.argtypes = [C.c_char_p]
    L.orc_index_load.restype = vp
    L.orc_index_free.argtypes = [vp]
    L.orc_index_genome_len.argtypes = [vp]
    L.orc_index_genome_len.restype = C.c_uint64
    L.orc_window.argtypes = [vp, C.c_uint64, i32, vp]
    L.orc_bpm.argtypes = [vp, i32, vp, i32, i32, vp]
    L.orc_align.argtypes = [C.POINTER(OrcParams), vp, i32, vp, vp, i32, i32, i32, C.c_uint, i32, i32, vp, vp, vp, vp, vp]
    L.orc_mapq.argtypes = [C.POINTER(OrcParams), C.c_uint, C.c_uint, i32]
    L.orc_sa_at.argtypes = [vp, C.c_uint64]
    L.orc_sa_at.restype = C.c_uint64
    L.orc_map_se.argtypes = [vp, C.POINTER(OrcParams), vp, vp, vp, i32, i64, vp, vp, vp]
    L.orc_map_se_votes.argtypes = [vp, C.POINTER(OrcParams), vp, vp, vp, i32, i64, vp, vp, vp, vp, vp, i64]
    L.orc_map_se_votes.restype = i64
    L.orc_map_pe.argtypes = [vp, C.POINTER(OrcParams), vp, vp, vp, vp, i32, i32, i32, i64, vp, vp, vp]
    L.orc_map_pe_var.argtypes = [vp, C.POINTER(OrcParams), vp, vp, vp, vp, vp, vp, i32, i64, vp, vp, vp]
    L.orc_search_se.argtypes = [vp, C.POINTER(OrcParams), C.c_char_p, C.c_char_p, C.c_char_p, vp]
    L.orc_search_pe.argtypes = [vp, C.POINTER(OrcParams), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, vp]
    assert C.sizeof(OrcParams) == 64
    _lib = L
    return L


def params(**kw):
    p = OrcParams()
    load().orc_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class OrcIndex:
    def __init__(self, prefix):
        self.L = load()
        self.h = self.L.orc_index_load(prefix.encode())
        if not self.h:
            raise FileNotFoundError(prefix)
        self.G = int(self.L.orc_index_genome_len(self.h))

    def window(self, site, n):
        buf = np.zeros(n + 8, dtype=np.uint8)
        self.L.orc_window(self.h, int(site), n, buf.ctypes.data)
        return buf[:n]

    def map_se(self, prm, seq, qual, L):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        n, stride = seq.shape
        ln = np.full(n, L, dtype=np.int32)
        recs = np.zeros(n, dtype=REC_DTYPE)
        assert REC_DTYPE.itemsize == 1088, REC_DTYPE.itemsize
        st = np.zeros(5, dtype=np.int64)
        cnt = np.zeros(len(COUNTER_KEYS), dtype=np.uint64)
        rc = self.L.orc_map_se(self.h, C.byref(prm), seq.ctypes.data, qual.ctypes.data, ln.ctypes.data, stride, n,
                               recs.ctypes.data, st.ctypes.data, cnt.ctypes.data)
        assert rc == 0
        return recs, st, dict(zip(COUNTER_KEYS, (int(x) for x in cnt)))

    def map_se_votes(self, prm, seq, qual, L, cap=None):
        """-> (recs, vote_site, vote_cnt, vote_off[n+1]): the vote lists of the general-path reads in visiting order"""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        n, stride = seq.shape
        ln = np.full(n, L, dtype=np.int32)
        recs = np.zeros(n, dtype=REC_DTYPE)
        st = np.zeros(5, dtype=np.int64)
        cap = cap or max(1024, 256 * n)
        vs = np.zeros(cap, dtype=np.uint64); vc = np.zeros(cap, dtype=np.uint32); vo = np.zeros(n + 1, dtype=np.uint64)
        tot = self.L.orc_map_se_votes(self.h, C.byref(prm), seq.ctypes.data, qual.ctypes.data, ln.ctypes.data, stride, n, recs.ctypes.data,
                                      st.ctypes.data, vs.ctypes.data, vc.ctypes.data, vo.ctypes.data, cap)
        assert tot >= 0, "vote capacity too small"
        return recs, vs[:tot], vc[:tot], vo

    def map_se_var(self, prm, seq, qual, lens):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        n, stride = seq.shape
        ln = np.ascontiguousarray(lens, dtype=np.int32)
        recs = np.zeros(n, dtype=REC_DTYPE)
        st = np.zeros(5, dtype=np.int64)
        cnt = np.zeros(len(COUNTER_KEYS), dtype=np.uint64)
        rc = self.L.orc_map_se(self.h, C.byref(prm), seq.ctypes.data, qual.ctypes.data, ln.ctypes.data, stride, n,
                               recs.ctypes.data, st.ctypes.data, cnt.ctypes.data)
        assert rc == 0
        return recs, st, dict(zip(COUNTER_KEYS, (int(x) for x in cnt)))

    def map_pe_var(self, prm, seq1, qual1, seq2_fastq, qual2, lens1, lens2):
        """per-pair mate lengths; seq2_fastq = mate 2 as in the FASTQ (left-aligned rows)"""
        comp = np.arange(256, dtype=np.uint8)
        for a, b in zip(b"ACGT", b"TGCA"):
            comp[a] = b
        a = [np.ascontiguousarray(x, dtype=np.uint8) for x in (seq1, qual1, seq2_fastq, qual2)]
        n, stride = a[0].shape
        l1 = np.ascontiguousarray(lens1, dtype=np.int32); l2 = np.ascontiguousarray(lens2, dtype=np.int32)
        s2 = np.zeros_like(a[2])
        for i in range(n):
            s2[i, :l2[i]] = comp[a[2][i, :l2[i]]][::-1]
        recs = np.zeros(n, dtype=PE_REC_DTYPE)
        st = np.zeros(5, dtype=np.int64)
        cnt = np.zeros(len(COUNTER_KEYS), dtype=np.uint64)
        rc = self.L.orc_map_pe_var(self.h, C.byref(prm), a[0].ctypes.data, a[1].ctypes.data, s2.ctypes.data, a[3].ctypes.data,
                                   l1.ctypes.data, l2.ctypes.data, stride, n, recs.ctypes.data, st.ctypes.data, cnt.ctypes.data)
        assert rc == 0, rc
        return recs, st, dict(zip(COUNTER_KEYS, (int(x) for x in cnt)))

    def map_pe(self, prm, seq1, qual1, seq2_fastq, qual2, L):
        """seq2_fastq = mate 2 as in the FASTQ; the reverse complement the reference's reader builds is made here"""
        comp = np.arange(256, dtype=np.uint8)
        for a, b in zip(b"ACGT", b"TGCA"):
            comp[a] = b
        a = [np.ascontiguousarray(x, dtype=np.uint8) for x in (seq1, qual1, seq2_fastq, qual2)]
        n, stride = a[0].shape
        s2 = np.zeros_like(a[2]); s2[:, :L] = comp[a[2][:, :L]][:, ::-1]
        recs = np.zeros(n, dtype=PE_REC_DTYPE)
        assert PE_REC_DTYPE.itemsize == 2120, PE_REC_DTYPE.itemsize
        st = np.zeros(5, dtype=np.int64)
        cnt = np.zeros(len(COUNTER_KEYS), dtype=np.uint64)
        rc = self.L.orc_map_pe(self.h, C.byref(prm), a[0].ctypes.data, a[1].ctypes.data, s2.ctypes.data, a[3].ctypes.data, L, L, stride, n,
                               recs.ctypes.data, st.ctypes.data, cnt.ctypes.data)
        assert rc == 0, rc
        return recs, st, dict(zip(COUNTER_KEYS, (int(x) for x in cnt)))

    def close(self):
        if self.h:
            self.L.orc_index_free(self.h)
            self.h = None


def bpm(window, read, k):
    L = load()
    w = np.ascontiguousarray(window, dtype=np.uint8)
    r = np.ascontiguousarray(read, dtype=np.uint8)
    wbuf = np.concatenate([w, np.zeros(8, np.uint8)])
    err = C.c_uint(0)
    end = L.orc_bpm(wbuf.ctypes.data, w.size, r.ctypes.data, r.size, k, C.byref(err))
    return err.value, end


def align(prm, window, read, qual, k, end_site, err, is_forward, reverse_quality=0):
    L = load()
    w = np.concatenate([np.ascontiguousarray(window, dtype=np.uint8), np.zeros(8, np.uint8)])
    r = np.ascontiguousarray(read, dtype=np.uint8)
    q = np.ascontiguousarray(qual, dtype=np.uint8)
    s, e, nm, sc = C.c_int(0), C.c_int(0), C.c_uint(0), C.c_int(0)
    cg = C.create_string_buffer(1024)
    L.orc_align(C.byref(prm), w.ctypes.data, w.size - 8, r.ctypes.data, q.ctypes.data, r.size, k, end_site, err, is_forward,
                reverse_quality, C.byref(s), C.byref(e), C.byref(nm), C.byref(sc), cg)
    return dict(start=s.value, end=e.value, nm=nm.value, score=sc.value, cigar=cg.value.decode())
