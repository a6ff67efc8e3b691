"""The BIG golden family (tests/golden/big_family.json, written by tests/golden/make_golden.py with the real reference):
a 5 Mb genome with thousands of planted repeat copies, 20 000 reads / pairs per mode.  Long candidate lists, vote-order ties,
ambiguity decisions and the --sensitive rescue paths are pinned here by the reference itself, on an index the REFERENCE built
with its own psascan.  Inputs are regenerated from their seeds and checked against the recorded sha256 before anything is
compared; the expected output is the sha256 of the reference's complete SAM body plus its per-record columns.

CPU: the oracle and the host index builder against the fixture.  GPU: the product (bmbs_search, file to file; GPU index
builder) against the same fixture."""
import gzip
import hashlib
import json
import os
import subprocess
import sys

import pytest

from common import GOLD, ROOT, sha_file

sys.path.insert(0, os.path.join(GOLD))

META = json.load(open(os.path.join(GOLD, "big_family.json")))
IDX = ("index", "index.bs.pac", "index.bs.index", "index.bs.index.occ", "index.bs.index.bwt")


def _make_golden():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def big(tmp_path_factory):
    """regenerated genome + reads (sha256-checked against what the reference was given); index by the host builder"""
    from bitmapperbs_amd import synth, mapper
    mg = _make_golden()
    wd = str(tmp_path_factory.mktemp("big"))
    names, chroms = mg.big_genome()
    fa = os.path.join(wd, "big.fa")
    synth.write_fasta(fa, names, chroms)
    assert sha_file(fa) == META["genome_sha256"], "the seeded genome no longer regenerates (numpy stream changed?)"
    mapper.Index.build(fa, fa, threads=8)
    files = {}
    for name, st in META["sets"].items():
        if st["kind"] == "pe":
            m1, m2 = synth.make_reads_pe(chroms, **st["reads"])
            f1 = os.path.join(wd, name + "_1.fq"); f2 = os.path.join(wd, name + "_2.fq")
            synth.write_fastq(f1, m1); synth.write_fastq(f2, m2)
            assert [sha_file(f1), sha_file(f2)] == st["fastq_sha256"], name
            files[name] = ["--seq1", f1, "--seq2", f2]
        else:
            r = synth.make_reads_se(chroms, **st["reads"])
            fq = os.path.join(wd, name + ".fq"); synth.write_fastq(fq, r)
            assert [sha_file(fq)] == st["fastq_sha256"], name
            files[name] = ["--seq", fq]
    return fa, files, wd


def _check(name, sam_path, stats_text):
    st = META["sets"][name]
    mg = _make_golden()
    digest, fields = mg.sam_fields(sam_path)
    if digest != st["sam_body_sha256"]:
        want = gzip.open(os.path.join(GOLD, "big_%s.ref.fields.gz" % name), "rt").read().splitlines()
        got = fields.splitlines()
        assert len(got) == len(want), (name, len(got), len(want))
        for i, (a, b) in enumerate(zip(got, want)):
            assert a == b, (name, "record", i, a, b)
        raise AssertionError("%s: columns equal but the SAM text differs (QNAME / SEQ / QUAL)" % name)
    assert stats_text == st["stats"], name


def test_host_builder_matches_reference_built_big_index(big):
    fa, _, _ = big
    for s in IDX:
        assert sha_file(fa + "." + s) == META["index"][s], s
    assert sha_file(fa + ".index.bs.index.sa", 8) == META["index"]["index.bs.index.sa[:-8]"]


@pytest.mark.parametrize("name", sorted(META["sets"]))
def test_oracle_reproduces_reference_on_big_family(name, big, oracle):
    fa, files, wd = big
    out = os.path.join(wd, name + ".orc.sam")
    q = subprocess.run([os.path.join(ROOT, "oracle", "bmbs_oracle"), "search", fa] + files[name] + ["-o", out] + META["sets"][name]["args"],
                       capture_output=True, text=True)
    assert q.returncode == 0, q.stderr
    stats = "".join(l + "\n" for l in q.stderr.splitlines() if l.startswith("No. of") or l.startswith("Mismatch"))
    _check(name, out, stats)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(META["sets"]))
def test_product_reproduces_reference_on_big_family(name, big):
    fa, files, wd = big
    out = os.path.join(wd, name + ".gpu.sam"); ms = os.path.join(wd, name + ".ms")
    p = subprocess.run([os.path.join(ROOT, "bitmapperbs_amd", "bmbs_search"), "--search", fa] + files[name] +
                       ["-o", out, "--mapstats", ms, "--batch", "7000"] + META["sets"][name]["args"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    _check(name, out, open(ms).read())


@pytest.mark.gpu
def test_device_builder_matches_reference_built_big_index(big, tmp_path):
    import shutil
    from bitmapperbs_amd import mapper
    fa, _, _ = big
    fd = str(tmp_path / "big.fa")
    shutil.copy(fa, fd)
    mapper.Index.build(fd, fd, threads=8, device=0)
    for s in IDX:
        assert sha_file(fd + "." + s) == META["index"][s], s
    assert sha_file(fd + ".index.bs.index.sa", 8) == META["index"]["index.bs.index.sa[:-8]"]
