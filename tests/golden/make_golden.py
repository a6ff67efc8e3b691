#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory by RUNNING THE REAL REFERENCE here.

  python tests/golden/make_golden.py [--psascan /path/to/psascan] [--only small|big]

Needs oracle/_ref/bitmapperBS (oracle/build_ref.sh; /root/reference present) and, for the index pins, the reference's external
suffix sorter oracle/_ref/psascan (oracle/build_psascan.sh; used by default when it is there).  Fixtures are data only:
  genome.fa.gz                   seeded synthetic genome (2 x 150 kb + planted repeats)
  se_<name>.fq.gz                seeded synthetic reads
  se_<name>.ref.sam.gz           SAM written by the reference (`bitmapperBS --search ... -t 1`)
  se_<name>.ref.stats            the reference's mapstats text
  index_ref_sha256.json          sha256 of the index files the REFERENCE's own `--index` wrote for
                                 genome.fa (needs the reference's external ./psascan binary, which is
                                 built from its vendored pSAscan/libdivsufsort with cmake -- not part of
                                 oracle/build_ref.sh; pass --psascan to refresh this file only).
The index the search runs use is built by the oracle builder (its byte-identity with the reference-built
index is exactly what index_ref_sha256.json pins).

The BIG family (big_*.json / big_*.ref.fields.gz): a 5 Mb genome with thousands of planted repeat copies, 20 000 reads / pairs
per mode, so that long candidate lists, vote-order ties and the --sensitive rescue paths are pinned by the reference itself.
Genome and reads are NOT stored: they are regenerated from their seeds (numpy PCG64 streams) and checked against the sha256
recorded here; of the reference's SAM the fixture keeps every column except QNAME / SEQ / QUAL (derivable from the inputs) plus
the sha256 of the complete SAM body, which is what the tests compare the product's complete SAM text with.  The big index is
written by the REFERENCE's own `--index` (psascan) and its file hashes are recorded as a second pin of our builders.
"""
import argparse, gzip, hashlib, json, os, shutil, subprocess, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bitmapperbs_amd import synth
import orc

SETS = {
    "a100": dict(reads=dict(n=1500, L=100, seed=1, sub=0.005, indel=0.0002, qual="const"), args=[]),
    "b150": dict(reads=dict(n=1500, L=150, seed=2, sub=0.02, indel=0.002, qual="random", n_rate=0.002), args=["-e", "0.04"]),
    "c150": dict(reads=dict(n=1200, L=150, seed=3, sub=0.03, indel=0.003, qual="random", conv=0.9), args=[]),
    "d250": dict(reads=dict(n=600, L=250, seed=4, sub=0.03, indel=0.001, qual="random"), args=[]),
    "e75": dict(reads=dict(n=1500, L=75, seed=5, sub=0.04, indel=0.004, qual="random", n_rate=0.01), args=[]),
    # reads of 401 .. 998 bases: the threshold is capped at 31 (Schema.cpp:24546-24550), 25 seeds.  998 is the longest read the
    # reference itself handles: at 999 and 1000 (Auxiliary.h:15 SEQ_MAX_LENGTH 1000) it prints 222 of 300 records of the same
    # reads, one of them with bytes outside ASCII -- its fixed 1000-byte buffers have no room for the terminators -- so there is
    # nothing to pin there
    "f600": dict(reads=dict(n=400, L=600, seed=6, sub=0.02, indel=0.002, qual="random"), args=[]),
    "g998": dict(reads=dict(n=300, L=998, seed=7, sub=0.015, indel=0.0015, qual="random", n_rate=0.001), args=[]),
}

PE_SETS = {
    "p150": dict(reads=dict(n=800, L=150, seed=11, sub=0.01, indel=0.001, qual="random"), args=[]),
    "p100": dict(reads=dict(n=800, L=100, seed=12, sub=0.02, indel=0.002, qual="random", ins_hi=560), args=["-e", "0.04", "--max", "520"]),
    "p75": dict(reads=dict(n=800, L=75, seed=13, sub=0.005, indel=0.0, qual="const", ins_lo=60, ins_hi=300), args=["--min", "100", "--max", "250"]),
    # --sensitive (Map_Pair_Seq_end_to_end rescue + reseed_filter): high-error pairs so that the rescue paths run
    "s100": dict(reads=dict(n=1200, L=100, seed=14, sub=0.06, indel=0.003, qual="random"), args=["--sensitive"]),
    "s150": dict(reads=dict(n=800, L=150, seed=15, sub=0.07, indel=0.004, qual="random", ins_hi=450), args=["--sensitive", "-e", "0.1", "--max", "450"]),
    "p998": dict(reads=dict(n=250, L=998, seed=16, sub=0.015, indel=0.001, qual="random", ins_lo=1020, ins_hi=1500), args=["--max", "1600"]),
    "s600": dict(reads=dict(n=300, L=600, seed=17, sub=0.05, indel=0.002, qual="random", ins_lo=620, ins_hi=1100), args=["--sensitive", "--max", "1200"]),
}

# output variants (Process_CommandLines.cpp:93-105) run on the read sets above; only the reference's SAM + mapstats are stored.
# "pbat": the FASTQ handed to the reference is the set's reverse complement with reversed qualities (common.pbat_fastq).
VARIANTS = {
    "se_e75_ua": dict(kind="se", base="e75", args=["--unmapped_out", "--ambiguous_out"]),
    "se_b150_pbat": dict(kind="se", base="b150", pbat=True, args=["-e", "0.04", "--pbat", "--unmapped_out", "--ambiguous_out"]),
    "pe_p75_ua": dict(kind="pe", base="p75", args=["--min", "100", "--max", "250", "--unmapped_out", "--ambiguous_out"]),
    "pe_s100_ua": dict(kind="pe", base="s100", args=["--sensitive", "--unmapped_out", "--ambiguous_out"]),
    # PE pbat = the two files swap roles (exchange_two_reads): hand them over swapped so that the pairs map
    # trimmed libraries: every read its own length (common.trim_fastq); mates of a pair trimmed independently
    "se_b150_trim": dict(kind="se", base="b150", trim=[7], args=["-e", "0.04", "--unmapped_out"]),
    "pe_p100_trim": dict(kind="pe", base="p100", trim=[8, 9], args=["-e", "0.04", "--max", "520", "--ambiguous_out"]),
    "pe_s100_trim": dict(kind="pe", base="s100", trim=[10, 11], args=["--sensitive"]),
    # --bam: the reference's BAM file itself is the fixture (compared after BGZF decompression, minus the @PG command line)
    "se_e75_bam": dict(kind="se", base="e75", bam=True, args=["--bam", "--unmapped_out", "--ambiguous_out"]),
    "pe_p75_bam": dict(kind="pe", base="p75", bam=True, args=["--bam", "--min", "100", "--max", "250", "--unmapped_out"]),
    "pe_p100_pbat": dict(kind="pe", base="p100", swap=True, args=["-e", "0.04", "--max", "520", "--pbat", "--unmapped_out"]),
}

def genome():
    names, chroms = synth.make_genome(300_000, 2, seed=101)
    rng = np.random.default_rng(202)
    for (elen, copies, div) in [(400, 40, 0.03), (1500, 6, 0.01), (150, 60, 0.0), (60, 80, 0.0)]:
        el = synth._ACGT[rng.integers(0, 4, elen)]
        for c in range(copies):
            ch = chroms[rng.integers(0, len(chroms))]
            p = int(rng.integers(0, ch.size - elen))
            e = el.copy(); m = rng.random(elen) < rng.random() * div
            e[m] = synth._ACGT[rng.integers(0, 4, int(m.sum()))]
            if rng.random() < 0.5: e = synth.revcomp(e)
            ch[p:p + elen] = e
    return names, chroms

def sha(path, drop_tail=0):
    b = open(path, "rb").read()
    if drop_tail: b = b[:-drop_tail]
    return hashlib.sha256(b).hexdigest()

BIG_SE = {
    "se150": dict(reads=dict(n=20000, L=150, seed=31, sub=0.02, indel=0.002, qual="random"), args=[]),
    "se100": dict(reads=dict(n=20000, L=100, seed=32, sub=0.03, indel=0.003, qual="random", n_rate=0.002), args=["-e", "0.06"]),
}
BIG_PE = {
    "pe150": dict(reads=dict(n=20000, L=150, seed=33, sub=0.015, indel=0.002, qual="random"), args=[]),
    "pes100": dict(reads=dict(n=20000, L=100, seed=34, sub=0.05, indel=0.003, qual="random"), args=["--sensitive"]),
}

def big_genome():
    names, chroms = synth.make_genome(5_000_000, 3, seed=303)
    rng = np.random.default_rng(404)
    for (elen, copies, div) in [(300, 2000, 0.06), (1000, 200, 0.02), (6000, 20, 0.01), (100, 3000, 0.0), (45, 2000, 0.0)]:
        el = synth._ACGT[rng.integers(0, 4, elen)]
        for c in range(copies):
            ch = chroms[rng.integers(0, len(chroms))]
            p = int(rng.integers(0, ch.size - elen))
            e = el.copy(); m = rng.random(elen) < rng.random() * div
            e[m] = synth._ACGT[rng.integers(0, 4, int(m.sum()))]
            if rng.random() < 0.5: e = synth.revcomp(e)
            ch[p:p + elen] = e
    return names, chroms

def sam_fields(sam_path):
    """-> (sha256 of the SAM body without @PG, text of every record line without QNAME / SEQ / QUAL)"""
    h = hashlib.sha256(); out = []
    for l in open(sam_path):
        if l.startswith("@PG"): continue
        h.update(l.encode())
        if l.startswith("@"): continue
        c = l.rstrip("\n").split("\t")
        out.append("\t".join(c[1:9] + c[11:]) + "\n")
    return h.hexdigest(), "".join(out)

def big_family(ref, psascan):
    assert psascan, "the big family lets the reference build its own index: oracle/build_psascan.sh first"
    wd = tempfile.mkdtemp(prefix="golden_big_")
    names, chroms = big_genome()
    fa = os.path.join(wd, "big.fa")
    synth.write_fasta(fa, names, chroms)
    shutil.copy(psascan, os.path.join(wd, "psascan"))
    subprocess.run([ref, "--index", "big.fa"], cwd=wd, check=True, capture_output=True)
    meta = {"genome_sha256": sha(fa), "index": {}, "sets": {}}
    for s_ in ("index", "index.bs.pac", "index.bs.index", "index.bs.index.occ", "index.bs.index.bwt"):
        meta["index"][s_] = sha(fa + "." + s_)
    meta["index"]["index.bs.index.sa[:-8]"] = sha(fa + ".index.bs.index.sa", 8)
    for name, cfg in list(BIG_SE.items()) + list(BIG_PE.items()):
        pe = name in BIG_PE
        sam = os.path.join(wd, name + ".sam")
        if pe:
            m1, m2 = synth.make_reads_pe(chroms, **cfg["reads"])
            f1 = os.path.join(wd, name + "_1.fq"); f2 = os.path.join(wd, name + "_2.fq")
            synth.write_fastq(f1, m1); synth.write_fastq(f2, m2)
            inp = ["--seq1", f1, "--seq2", f2]; insha = [sha(f1), sha(f2)]
        else:
            r = synth.make_reads_se(chroms, **cfg["reads"])
            fq = os.path.join(wd, name + ".fq"); synth.write_fastq(fq, r)
            inp = ["--seq", fq]; insha = [sha(fq)]
        p = subprocess.run([ref, "--search", fa] + inp + ["-t", "1", "-o", sam] + cfg["args"], capture_output=True, text=True, cwd=wd)
        assert p.returncode == 0, p.stderr
        digest, fields = sam_fields(sam)
        with gzip.GzipFile(os.path.join(HERE, "big_%s.ref.fields.gz" % name), "wb", mtime=0) as g: g.write(fields.encode())
        stats = "".join(l for l in p.stderr.splitlines(True) if l.startswith("No. of") or l.startswith("Mismatch"))
        meta["sets"][name] = {"kind": "pe" if pe else "se", "reads": cfg["reads"], "args": cfg["args"], "fastq_sha256": insha,
                              "sam_body_sha256": digest, "stats": stats, "records": fields.count("\n")}
        print("BIG", name, "records", fields.count("\n"), stats.splitlines()[1], stats.splitlines()[2])
    json.dump(meta, open(os.path.join(HERE, "big_family.json"), "w"), indent=1)
    shutil.rmtree(wd)

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--psascan"); ap.add_argument("--only", choices=["small", "big"])
    a = ap.parse_args()
    ref = os.path.join(ROOT, "oracle", "_ref", "bitmapperBS")
    assert os.path.exists(ref), "build the reference first: oracle/build_ref.sh"
    if not a.psascan and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "psascan")):
        a.psascan = os.path.join(ROOT, "oracle", "_ref", "psascan")
    if a.only != "small":
        big_family(ref, a.psascan)
        if a.only == "big":
            return
    wd = tempfile.mkdtemp(prefix="golden_")
    names, chroms = genome()
    fa = os.path.join(wd, "genome.fa")
    synth.write_fasta(fa, names, chroms)
    with open(fa, "rb") as f, gzip.GzipFile(os.path.join(HERE, "genome.fa.gz"), "wb", mtime=0) as g: g.write(f.read())
    if a.psascan:
        iw = os.path.join(wd, "refidx"); os.makedirs(iw)
        shutil.copy(fa, os.path.join(iw, "genome.fa")); shutil.copy(a.psascan, os.path.join(iw, "psascan"))
        subprocess.run([ref, "--index", "genome.fa"], cwd=iw, check=True, capture_output=True)
        h = {}
        for s in ("index", "index.bs.pac", "index.bs.index", "index.bs.index.occ", "index.bs.index.bwt"):
            h[s] = sha(os.path.join(iw, "genome.fa." + s))
        # the reference writes one uninitialised trailing SA_flag word (bwt.cpp:1690-1700): hash without it
        h["index.bs.index.sa[:-8]"] = sha(os.path.join(iw, "genome.fa.index.bs.index.sa"), 8)
        json.dump(h, open(os.path.join(HERE, "index_ref_sha256.json"), "w"), indent=1)
    orc.load().orc_index_build(fa.encode(), fa.encode())
    for name, cfg in SETS.items():
        r = synth.make_reads_se(chroms, **cfg["reads"])
        fq = os.path.join(wd, "se_%s.fq" % name)
        synth.write_fastq(fq, r)
        sam = os.path.join(wd, "se_%s.sam" % name)
        p = subprocess.run([ref, "--search", fa, "--seq", fq, "-t", "1", "-o", sam] + cfg["args"], capture_output=True, text=True, cwd=wd)
        assert p.returncode == 0, p.stderr
        with open(fq, "rb") as f, gzip.GzipFile(os.path.join(HERE, "se_%s.fq.gz" % name), "wb", mtime=0) as g: g.write(f.read())
        body = "".join(l for l in open(sam) if not l.startswith("@PG"))
        with gzip.GzipFile(os.path.join(HERE, "se_%s.ref.sam.gz" % name), "wb", mtime=0) as g: g.write(body.encode())
        stats = "".join(l for l in p.stderr.splitlines(True) if l.startswith("No. of") or l.startswith("Mismatch"))
        open(os.path.join(HERE, "se_%s.ref.stats" % name), "w").write(stats)
        print(name, "lines", body.count("\n"), stats.splitlines()[1])
    json.dump({k: v["args"] for k, v in SETS.items()}, open(os.path.join(HERE, "se_args.json"), "w"))
    # paired-end (default = fast mode; s* sets run --sensitive)
    for name, cfg in PE_SETS.items():
        m1, m2 = synth.make_reads_pe(chroms, **cfg["reads"])
        f1 = os.path.join(wd, "pe_%s_1.fq" % name); f2 = os.path.join(wd, "pe_%s_2.fq" % name)
        synth.write_fastq(f1, m1); synth.write_fastq(f2, m2)
        sam = os.path.join(wd, "pe_%s.sam" % name)
        p = subprocess.run([ref, "--search", fa, "--seq1", f1, "--seq2", f2, "-t", "1", "-o", sam] + cfg["args"], capture_output=True, text=True, cwd=wd)
        assert p.returncode == 0, p.stderr
        for src, dst in ((f1, "pe_%s_1.fq.gz" % name), (f2, "pe_%s_2.fq.gz" % name)):
            with open(src, "rb") as f, gzip.GzipFile(os.path.join(HERE, dst), "wb", mtime=0) as g: g.write(f.read())
        body = "".join(l for l in open(sam) if not l.startswith("@PG"))
        with gzip.GzipFile(os.path.join(HERE, "pe_%s.ref.sam.gz" % name), "wb", mtime=0) as g: g.write(body.encode())
        stats = "".join(l for l in p.stderr.splitlines(True) if l.startswith("No. of") or l.startswith("Mismatch"))
        open(os.path.join(HERE, "pe_%s.ref.stats" % name), "w").write(stats)
        print("PE", name, "lines", body.count("\n"), stats.splitlines()[1])
    json.dump({k: v["args"] for k, v in PE_SETS.items()}, open(os.path.join(HERE, "pe_args.json"), "w"))
    from common import pbat_fastq, trim_fastq
    for name, v in VARIANTS.items():
        sam = os.path.join(wd, name + ".sam")
        if v["kind"] == "se":
            fq = os.path.join(wd, "se_%s.fq" % v["base"])
            if v.get("pbat"):
                pbat_fastq(fq, fq[:-3] + "_pbat.fq"); fq = fq[:-3] + "_pbat.fq"
            if v.get("trim"):
                trim_fastq(fq, fq[:-3] + "_trim.fq", v["trim"][0]); fq = fq[:-3] + "_trim.fq"
            inp = ["--seq", fq]
        else:
            a, b = (2, 1) if v.get("swap") else (1, 2)
            fa_, fb_ = os.path.join(wd, "pe_%s_%d.fq" % (v["base"], a)), os.path.join(wd, "pe_%s_%d.fq" % (v["base"], b))
            if v.get("trim"):
                trim_fastq(fa_, fa_[:-3] + "_trim.fq", v["trim"][0]); trim_fastq(fb_, fb_[:-3] + "_trim.fq", v["trim"][1])
                fa_, fb_ = fa_[:-3] + "_trim.fq", fb_[:-3] + "_trim.fq"
            inp = ["--seq1", fa_, "--seq2", fb_]
        p = subprocess.run([ref, "--search", fa] + inp + ["-t", "1", "-o", sam] + v["args"], capture_output=True, text=True, cwd=wd)
        assert p.returncode == 0, p.stderr
        stats = "".join(l for l in p.stderr.splitlines(True) if l.startswith("No. of") or l.startswith("Mismatch"))
        open(os.path.join(HERE, "var_%s.ref.stats" % name), "w").write(stats)
        if v.get("bam"):
            shutil.copy(sam, os.path.join(HERE, "var_%s.ref.bam" % name))
            print("VARIANT", name, "bam bytes", os.path.getsize(sam), stats.splitlines()[1])
            continue
        body = "".join(l for l in open(sam) if not l.startswith("@PG"))
        with gzip.GzipFile(os.path.join(HERE, "var_%s.ref.sam.gz" % name), "wb", mtime=0) as g: g.write(body.encode())
        print("VARIANT", name, "lines", body.count("\n"), stats.splitlines()[1], stats.splitlines()[2])
    json.dump(VARIANTS, open(os.path.join(HERE, "variants.json"), "w"), indent=1)
    shutil.rmtree(wd)

if __name__ == "__main__":
    main()
