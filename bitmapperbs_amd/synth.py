"""Seeded synthetic genomes and bisulfite reads (SURVEY.md §8d workload definition).

Own generator (numpy, vectorised): uniform-random N-free chromosomes, reads drawn uniformly, 50 %
each strand, directional-protocol C->T conversion after strand selection, substitutions, at most
one indel per read, optional random qualities and N bases.  Reads have one fixed length per set,
so a read set is a dense SoA: ``seq[n, L]``, ``qual[n, L]`` (uint8 ASCII) plus generated names.

This is workload tooling for tests/ and bench.py; it is not on the mapping path.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def make_genome(total_len: int, n_chrom: int = 4, seed: int = 20240229):
    """-> (names, [uint8 ASCII arrays]); N-free so that the index build is deterministic
    (the reference randomises N with srand(time(0)), Index.cpp:703)."""
    rng = np.random.default_rng(seed)
    per = total_len // n_chrom
    chroms = [_ACGT[rng.integers(0, 4, per, dtype=np.uint8)] for _ in range(n_chrom)]
    names = ["chr%d" % (i + 1) for i in range(n_chrom)]
    return names, chroms


def write_fasta(path, names, chroms, width: int = 60):
    with open(path, "wb") as f:
        for nm, s in zip(names, chroms):
            f.write(b">" + nm.encode() + b"\n")
            n = s.size
            full = (n // width) * width
            if full:
                body = np.empty((full // width, width + 1), dtype=np.uint8)
                body[:, :width] = s[:full].reshape(-1, width)
                body[:, width] = 10
                f.write(body.tobytes())
            if full < n:
                f.write(s[full:].tobytes() + b"\n")


def revcomp(a: np.ndarray) -> np.ndarray:
    """reverse complement along the last axis"""
    return _COMP[a][..., ::-1]


def _hash01(a: np.ndarray, b: np.ndarray, salt: int) -> np.ndarray:
    """deterministic per-(a,b) uniform in [0,1) (used so overlapping mates share conversions)"""
    x = (a.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (
        b.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)) ^ np.uint64(salt)
    x ^= x >> np.uint64(29)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(32)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(29)
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _mutate(rng, W: np.ndarray, L: int, sub: float, indel: float, n_rate: float) -> np.ndarray:
    """W: [n, L+8] converted template windows -> reads [n, L]"""
    n = W.shape[0]
    R = W[:, :L].copy()
    if sub > 0:
        m = rng.random((n, L)) < sub
        R[m] = _ACGT[rng.integers(0, 4, int(m.sum()), dtype=np.uint8)]
    if indel > 0:
        who = np.nonzero(rng.random(n) < indel * L)[0]
        pos = rng.integers(5, L - 5, who.size)
        isdel = rng.random(who.size) < 0.5
        ins_b = _ACGT[rng.integers(0, 4, who.size, dtype=np.uint8)]
        for j, i in enumerate(who):
            p = int(pos[j])
            if isdel[j]:   # one template base skipped
                R[i, p:] = W[i, p + 1:L + 1]
                # substitutions already drawn for the tail are dropped; harmless
            else:          # one extra base in the read
                R[i, p + 1:] = R[i, p:L - 1].copy()
                R[i, p] = ins_b[j]
    if n_rate > 0:
        m = rng.random((n, L)) < n_rate
        R[m] = ord("N")
    return R


def _quals(rng, n: int, L: int, mode: str) -> np.ndarray:
    if mode == "const":
        return np.full((n, L), ord("I"), dtype=np.uint8)
    if mode == "random":      # Phred 2..40 inclusive, +33
        return (rng.integers(2, 41, (n, L), dtype=np.uint8) + 33).astype(np.uint8)
    raise ValueError(mode)


def _draw(rng, chroms, n: int, span: np.ndarray):
    """pick chromosome + start for fragments of per-read length `span` (+8 slack)"""
    lens = np.array([c.size for c in chroms], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)])[:-1]
    cat = np.concatenate(chroms)
    c = rng.integers(0, len(chroms), n)
    hi = lens[c] - span - 8
    p = (rng.random(n) * hi).astype(np.int64)
    return cat, c, p, offs[c] + p


def make_reads_se(chroms, n: int, L: int, seed: int = 7, sub: float = 0.005, indel: float = 0.0002,
                  conv: float = 0.99, qual: str = "const", n_rate: float = 0.0, name_prefix="r"):
    """-> dict(seq[n,L], qual[n,L], names(list[bytes]), truth=(chrom, pos0, strand))"""
    rng = np.random.default_rng(seed)
    cat, c, p, g = _draw(rng, chroms, n, np.full(n, L, dtype=np.int64))
    idx = g[:, None] + np.arange(L + 8, dtype=np.int64)[None, :]
    W = cat[idx]
    minus = rng.random(n) < 0.5
    W[minus] = revcomp(W[minus])
    m = (W == ord("C")) & (rng.random(W.shape) < conv)
    W[m] = ord("T")
    seq = _mutate(rng, W, L, sub, indel, n_rate)
    q = _quals(rng, n, L, qual)
    names = [b"%s%d_chr%d_%d_%s" % (name_prefix.encode(), i, c[i] + 1, p[i] + 1, b"-" if minus[i] else b"+")
             for i in range(n)] if n <= 2_000_000 else None
    return {"seq": seq, "qual": q, "names": names, "truth": (c, p, minus)}


def make_reads_pe(chroms, n: int, L: int, seed: int = 7, sub: float = 0.005, indel: float = 0.0002,
                  conv: float = 0.99, qual: str = "const", ins_lo: int | None = None, ins_hi: int = 400):
    """-> (mate1 dict, mate2 dict); insert uniform in [L+20, ins_hi); mate 2 is the reverse
    complement end of the converted fragment (as sequenced, i.e. as it appears in the FASTQ)."""
    rng = np.random.default_rng(seed)
    ins_lo = L + 20 if ins_lo is None else ins_lo
    ins = rng.integers(ins_lo, ins_hi, n).astype(np.int64)
    cat, c, p, g = _draw(rng, chroms, n, ins)
    minus = rng.random(n) < 0.5
    ar = np.arange(L + 8, dtype=np.int64)[None, :]
    rid = np.arange(n, dtype=np.int64)[:, None]
    # fragment coordinates f in [0, ins): plus-strand fragment base f = cat[g+f];
    # minus-strand fragment base f = comp(cat[g+ins-1-f])
    def frag(fpos):
        fpos_c = np.clip(fpos, 0, (ins[:, None] + 7))
        gi = np.where(minus[:, None], g[:, None] + ins[:, None] - 1 - fpos_c, g[:, None] + fpos_c)
        gi = np.clip(gi, 0, cat.size - 1)
        b = cat[gi]
        b = np.where(minus[:, None], _COMP[b], b)
        cv = (b == ord("C")) & (_hash01(rid + 0 * fpos_c, fpos_c, seed) < conv)
        return np.where(cv, np.uint8(ord("T")), b).astype(np.uint8)
    W1 = frag(ar + 0 * rid)                       # fragment positions 0..L+7
    # mate 2 reads the fragment from its 3' end on the opposite strand: base j = comp(frag[ins-1-j])
    W2 = _COMP[frag(ins[:, None] - 1 - ar)]
    s1 = _mutate(rng, W1, L, sub, indel, 0.0)
    s2 = _mutate(rng, W2, L, sub, indel, 0.0)
    q1 = _quals(rng, n, L, qual)
    q2 = _quals(rng, n, L, qual)
    base = [b"p%d_chr%d_%d_%d_%s" % (i, c[i] + 1, p[i] + 1, ins[i], b"-" if minus[i] else b"+")
            for i in range(n)] if n <= 2_000_000 else None
    m1 = {"seq": s1, "qual": q1, "names": [b + b"/1" for b in base] if base else None}
    m2 = {"seq": s2, "qual": q2, "names": [b + b"/2" for b in base] if base else None}
    return m1, m2


def write_fastq(path, reads):
    seq, qual, names = reads["seq"], reads["qual"], reads["names"]
    n, L = seq.shape
    if names is None:
        names = [b"r%d" % i for i in range(n)]
    with open(path, "wb") as f:
        chunk = 100_000
        for a in range(0, n, chunk):
            b = min(n, a + chunk)
            parts = []
            for i in range(a, b):
                parts.append(b"@" + names[i] + b"\n" + seq[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n")
            f.write(b"".join(parts))
