"""shared helpers for the test-suite"""
import gzip
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")


def gunzip_to(src, dst):
    with gzip.open(src, "rb") as f, open(dst, "wb") as g:
        g.write(f.read())


def golden_args():
    return json.load(open(os.path.join(GOLD, "se_args.json")))


def e_of(args):
    return float(args[args.index("-e") + 1]) if "-e" in args else 0.08


def read_fastq(path):
    """-> names(list[bytes]), seq[n,L] u8, qual[n,L] u8 (all reads equal length)"""
    names, seqs, quals = [], [], []
    with open(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().rstrip(b"\n"); f.readline(); q = f.readline().rstrip(b"\n")
            names.append(h.rstrip(b"\n")[1:]); seqs.append(s.upper()); quals.append(q)
    L = len(seqs[0])
    assert all(len(s) == L for s in seqs)
    return names, np.frombuffer(b"".join(seqs), dtype=np.uint8).reshape(-1, L).copy(), \
        np.frombuffer(b"".join(quals), dtype=np.uint8).reshape(-1, L).copy()


def sha_file(path, drop_tail=0):
    b = open(path, "rb").read()
    if drop_tail:
        b = b[:-drop_tail]
    return hashlib.sha256(b).hexdigest()


def plant_repeats(chroms, seed=202, spec=((400, 40, 0.03), (1500, 6, 0.01), (150, 60, 0.0), (60, 80, 0.0))):
    from bitmapperbs_amd import synth
    rng = np.random.default_rng(seed)
    for (elen, copies, div) in spec:
        el = synth._ACGT[rng.integers(0, 4, elen)]
        for c in range(copies):
            ch = chroms[rng.integers(0, len(chroms))]
            p = int(rng.integers(0, ch.size - elen))
            e = el.copy(); m = rng.random(elen) < rng.random() * div
            e[m] = synth._ACGT[rng.integers(0, 4, int(m.sum()))]
            if rng.random() < 0.5:
                e = synth.revcomp(e)
            ch[p:p + elen] = e


def orc_sam_lines(index_names, names, seq, qual, L, recs):
    """SAM lines from ORACLE records, for comparisons with host-side emit"""
    comp = np.arange(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    out = []
    for i in np.nonzero(recs["status"] == 1)[0]:
        r = recs[i]
        nm = names[i].decode()
        for cut in (" ", "/"):
            j = nm.find(cut)
            if j >= 0:
                nm = nm[:j]
        s, q = seq[i, :L], qual[i, :L]
        if int(r["flag"]) & 16:
            s = comp[s][::-1]; q = q[::-1]
        out.append("%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\tNM:i:%d\n" % (
            nm, int(r["flag"]), index_names[int(r["chrom"])], int(r["pos"]), int(r["mapq"]), r["cigar"].decode(),
            s.tobytes().decode(), q.tobytes().decode(), int(r["nm"])))
    return out


def pbat_fastq(src, dst):
    """FASTQ of the reverse-complemented reads with reversed qualities (what a pbat library of the same fragments yields)"""
    comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
    with open(src, "rb") as f, open(dst, "wb") as o:
        lines = f.read().split(b"\n")
        for i in range(0, len(lines) - 3, 4):
            o.write(lines[i] + b"\n" + lines[i + 1].translate(comp)[::-1] + b"\n" + lines[i + 2] + b"\n" + lines[i + 3][::-1] + b"\n")


def trim_fastq(src, dst, seed, lo=30):
    """every record cut to a seeded random length in [lo, len] at its 3' end (what an adapter/quality trimmer leaves)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    with open(src, "rb") as f, open(dst, "wb") as o:
        lines = f.read().split(b"\n")
        for i in range(0, len(lines) - 3, 4):
            n = len(lines[i + 1])
            k = int(rng.integers(min(lo, n), n + 1)) if rng.random() < 0.7 else n
            o.write(lines[i] + b"\n" + lines[i + 1][:k] + b"\n" + lines[i + 2] + b"\n" + lines[i + 3][:k] + b"\n")


def bam_payload(path):
    """BGZF-decompressed BAM split into (reference dictionary bytes, record stream bytes); the header text (it holds the
    command line) is dropped"""
    import gzip, struct
    d = gzip.open(path, "rb").read()
    assert d[:4] == b"BAM\x01"
    l_text = struct.unpack("<i", d[4:8])[0]
    p = 8 + l_text
    n_ref = struct.unpack("<i", d[p:p + 4])[0]
    q = p + 4
    for _ in range(n_ref):
        ln = struct.unpack("<i", d[q:q + 4])[0]
        q += 8 + ln
    return d[p:q], d[q:]


def write_bgzf(path, data: bytes, block: int = 60000, level: int = 6):
    """what `bgzip` writes: independent gzip members of at most 64 KiB of input with their compressed size in a 'BC' extra field
    (SAM specification section 4.1), closed by the empty end-of-file block"""
    import struct
    import zlib
    with open(path, "wb") as f:
        for a in list(range(0, len(data), block)) + [None]:
            chunk = b"" if a is None else data[a:a + block]
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            z = co.compress(chunk) + co.flush()
            bsize = len(z) + 25
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize))
            f.write(z)
            f.write(struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))


_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def _reg2bin(beg, end):
    end -= 1
    for sh, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh:
            return off + (beg >> sh)
    return 0


def sam_to_bam_records(sam: bytes, ref_names) -> bytes:
    """the BAM record stream htslib's sam_parse1 + bam_write1 give for SAM lines of the shape this path prints (11 columns + NM:i):
    the checker of the device-side BAM encoder (test infrastructure; SAM specification section 4.2)"""
    import struct
    ids = {n: i for i, n in enumerate(ref_names)}
    out = []
    for line in sam.split(b"\n"):
        if not line or line.startswith(b"@"):
            continue
        f = line.decode("latin-1").split("\t")
        flag = int(f[1]); refid = -1 if f[2] == "*" else ids[f[2]]; pos = int(f[3]) - 1; mapq = int(f[4])
        cig = []; reflen = 0
        if f[5] != "*":
            num = ""
            for ch in f[5]:
                if ch.isdigit():
                    num += ch
                else:
                    op = "MIDNSHP=X".index(ch); ln = int(num); num = ""
                    cig.append(ln << 4 | op)
                    if op in (0, 2, 3, 7, 8):
                        reflen += ln
        nref = refid if f[6] == "=" else (-1 if f[6] == "*" else ids[f[6]])
        npos = int(f[7]) - 1; tlen = int(f[8])
        seq = "" if f[9] == "*" else f[9]
        name = f[0].encode("latin-1") + b"\0"
        body = struct.pack("<iiBBHHHIiii", refid, pos, len(name), mapq, _reg2bin(pos, pos + (reflen if cig and reflen else 1)), len(cig), flag,
                           len(seq), nref, npos, tlen) + name + b"".join(struct.pack("<I", c) for c in cig)
        nib = [_NT16.get(c.upper(), 15) if not c.isdigit() else (1, 2, 4, 8, 15, 15, 15, 15, 15, 15)[int(c)] for c in seq]
        if len(nib) & 1:
            nib.append(0)
        body += bytes((nib[i] << 4) | nib[i + 1] for i in range(0, len(nib), 2))
        body += bytes(0xff for _ in seq) if f[10] == "*" else bytes((ord(c) - 33) & 0xff for c in f[10])
        if len(f) > 11 and f[11].startswith("NM:i:"):
            v = int(f[11][5:])
            body += b"NM" + (b"C" + struct.pack("<B", v) if v <= 0xff else b"S" + struct.pack("<H", v) if v <= 0xffff else b"I" + struct.pack("<I", v))
        out.append(struct.pack("<i", len(body)) + body)
    return b"".join(out)


def bgzf_blocks(data: bytes):
    """the members of a BGZF byte string as (compressed size, inflated bytes); checks the BC field, CRC-32 and ISIZE of each"""
    import struct
    import zlib
    at = 0
    out = []
    while at < len(data):
        assert data[at:at + 4] == b"\x1f\x8b\x08\x04" and data[at + 10:at + 16] == b"\x06\x00BC\x02\x00", at
        bsize = struct.unpack("<H", data[at + 16:at + 18])[0] + 1
        raw = zlib.decompressobj(-15).decompress(data[at + 18:at + bsize - 8])
        crc, isize = struct.unpack("<II", data[at + bsize - 8:at + bsize])
        assert isize == len(raw) and crc == (zlib.crc32(raw) & 0xffffffff) and len(raw) <= 0xff00, at
        out.append((bsize, raw))
        at += bsize
    return out
