"""The reference-side binding of INTEGRATION.md, compiled and linked for real (north_star: "C host code (Process_Reads.cpp driver,
Index/bwt lookup, Process_sam_out emit) calls through a thin C-ABI into hand-written HIP kernels").

oracle/bind_check.cpp is the code a BitMapperBS maintainer adds: the loaded index handed over from the reference's globals
(bwt.h:34-177, Index.cpp:29-34) and batch replacements of Map_Single_Seq / Map_Pair_Seq (Schema.h:375-388) that read with the reference's
reader and print with the reference's emitters (Schema.cpp:11928, 10537, 11494).

CPU (build container, where /root/reference exists): the file compiles against the reference's real headers; INTEGRATION.md quotes it
verbatim; the linked binary stops without a HIP device instead of mapping on the CPU.
GPU: oracle/_ref/bitmapperBS_hip -- the reference's main, CLI, Load_Index, reader, SAM / BAM writers around libbmbs_hip.so -- writes the
committed goldens byte for byte."""
import gzip
import json
import os
import re
import subprocess

import pytest

from common import GOLD, ROOT, gunzip_to

REF = os.environ.get("BMBS_REFERENCE_DIR", "/root/reference")
BIND = os.path.join(ROOT, "oracle", "bind_check.cpp")
BIN = os.path.join(ROOT, "oracle", "_ref", "bitmapperBS_hip")


def doc_blocks():
    """the `//[doc:name]` ... `//[doc:end]` blocks of bind_check.cpp, marker lines dropped"""
    out, cur, name = {}, None, None
    for line in open(BIND).read().split("\n"):
        m = re.match(r"\s*//\[doc:([a-z_]+)\]\s*$", line)
        if m and m.group(1) == "end":
            out[name] = "\n".join(cur); cur = None
        elif m:
            name, cur = m.group(1), []
        elif cur is not None:
            cur.append(line)
    assert cur is None, "unterminated doc block %s" % name
    return out


def test_integration_md_quotes_the_compiled_binding_verbatim():
    blocks = doc_blocks()
    assert set(blocks) >= {"globals", "attach", "cigar", "map_se", "map_pe", "stats", "entry"}
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name, text in blocks.items():
        assert text.strip("\n") in doc, "INTEGRATION.md does not carry block '%s' of oracle/bind_check.cpp (tools/sync_integration.py)" % name


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only in the build container")
def test_binding_compiles_against_the_reference_headers():
    """g++ -fsyntax-only against the reference's own bwt.h / Auxiliary.h / Schema.h / Process_Reads.h / Process_sam_out.h / bam_prase.h"""
    p = subprocess.run(["g++", "-fsyntax-only", "-w", "-mavx2", "-mpopcnt", "-D__AVX2__", "-iquote", REF, "-I", REF,
                        "-I", os.path.join(REF, "htslib"), "-I", os.path.join(ROOT, "include"), BIND], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    # and every global the binding names is one the reference defines
    src = open(BIND).read()
    for g in ("bitmapper_index_params", "_ih_refGen", "refGenLength_2_bit", "_ih_refGenName", "refChromeCont", "_msf_refGenLength",
              "unique_mapped_read", "ambiguous_mapped_read", "mapped_bases", "error_mapped_bases", "completedSeqCnt", "thread_e_f",
              "over_all_seed_length", "minDistance_pair", "maxDistance_pair", "is_local", "ambiguous_out", "unmapped_out", "bam_output"):
        assert g in src
        hits = subprocess.run("grep -l -w %s %s/*.cpp %s/*.h" % (g, REF, REF), shell=True, capture_output=True, text=True).stdout
        assert hits.strip(), "%s is not a name of the reference" % g


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only in the build container")
def test_linked_binary_is_the_reference_program_around_the_hip_library(tmp_path):
    """oracle/build_ref_hip.sh: the reference's objects + bind_check.o + libbmbs_hip.so.  The mapping entry points resolve to the binding
    (the reference's own loops are still in Schema.o, weak and unused), the library is a dynamic dependency, and without a HIP device
    the program loads the index with the reference's Load_Index and then STOPS -- there is no CPU path behind the binding."""
    p = subprocess.run([os.path.join(ROOT, "oracle", "build_ref_hip.sh")], capture_output=True, text=True)
    assert p.returncode == 0 and os.path.exists(BIN), p.stderr[-2000:]
    syms = subprocess.run(["nm", "-C", BIN], capture_output=True, text=True).stdout
    assert re.search(r" t bind_map_single\(bool\)", syms) and re.search(r" T Map_Single_Seq\(int\)", syms)
    assert re.search(r" U bmbs_map_pe_var", syms) and re.search(r" U bmbs_index_attach", syms)
    assert "libbmbs_hip.so" in subprocess.run(["ldd", BIN], capture_output=True, text=True).stdout
    import torch
    if torch.cuda.is_available():
        return
    fa = str(tmp_path / "genome.fa"); fq = str(tmp_path / "r.fq")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa); gunzip_to(os.path.join(GOLD, "se_e75.fq.gz"), fq)
    from bitmapperbs_amd import mapper
    mapper.Index.build(fa, fa, threads=4)
    r = subprocess.run([BIN, "--search", fa, "--seq", fq, "-o", str(tmp_path / "o.sam")], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "hash table has been loaded" in r.stderr and "no HIP device" in r.stderr, r.stderr[-1500:]


# ---- GPU: the linked program against the goldens -----------------------------------------------------------------------------------
def _run_bound(tmp_path, inputs, args, out):
    assert os.path.exists(BIN), "oracle/_ref/bitmapperBS_hip not built (oracle/build_ref_hip.sh, build container)"
    p = subprocess.run([BIN, "--search", str(tmp_path / "genome.fa")] + inputs + ["-o", out] + args, capture_output=True, text=True,
                       cwd=str(tmp_path), env=dict(os.environ, BMBS_BIND_BATCH="257"))      # several batches per golden
    assert p.returncode == 0, p.stderr[-3000:]
    return "".join(l + "\n" for l in p.stderr.splitlines() if l.startswith("No. of") or l.startswith("Mismatch"))


def _index(tmp_path):
    from bitmapperbs_amd import mapper
    fa = str(tmp_path / "genome.fa")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa)
    mapper.Index.build(fa, fa, threads=4)      # our builder's files, loaded by the reference's Load_Index
    return fa


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["b150", "e75", "a100", "f600"])
def test_reference_program_bound_to_the_hip_library_writes_the_se_golden(name, tmp_path):
    _index(tmp_path)
    fq = str(tmp_path / "r.fq"); out = str(tmp_path / "o.sam")
    gunzip_to(os.path.join(GOLD, "se_%s.fq.gz" % name), fq)
    args = json.load(open(os.path.join(GOLD, "se_args.json")))[name]
    stats = _run_bound(tmp_path, ["--seq", fq], args, out)
    mine = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert mine == gzip.open(os.path.join(GOLD, "se_%s.ref.sam.gz" % name), "rt").read()
    assert stats == open(os.path.join(GOLD, "se_%s.ref.stats" % name)).read()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["p100", "s100", "p150", "s150"])
def test_reference_program_bound_to_the_hip_library_writes_the_pe_golden(name, tmp_path):
    _index(tmp_path)
    f1 = str(tmp_path / "1.fq"); f2 = str(tmp_path / "2.fq"); out = str(tmp_path / "o.sam")
    gunzip_to(os.path.join(GOLD, "pe_%s_1.fq.gz" % name), f1); gunzip_to(os.path.join(GOLD, "pe_%s_2.fq.gz" % name), f2)
    args = json.load(open(os.path.join(GOLD, "pe_args.json")))[name]
    stats = _run_bound(tmp_path, ["--seq1", f1, "--seq2", f2], args, out)
    mine = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert mine == gzip.open(os.path.join(GOLD, "pe_%s.ref.sam.gz" % name), "rt").read()
    assert stats == open(os.path.join(GOLD, "pe_%s.ref.stats" % name)).read()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(__import__("test_oracle").variants()))
def test_reference_program_bound_to_the_hip_library_writes_the_variant_goldens(name, tmp_path):
    """--unmapped_out / --ambiguous_out / --pbat / trimmed reads (bmbs_map_*_var) / --bam (the reference's htslib writer fed by the
    reference's emitters from bmbs_result records)"""
    from test_oracle import variants, variant_inputs
    v = variants()[name]
    _index(tmp_path)
    out = str(tmp_path / ("o.bam" if v.get("bam") else "o.sam"))
    stats = _run_bound(tmp_path, variant_inputs(v, tmp_path), v["args"], out)
    if v.get("bam"):
        from common import bam_payload
        assert bam_payload(out) == bam_payload(os.path.join(GOLD, "var_%s.ref.bam" % name))
    else:
        mine = "".join(l for l in open(out) if not l.startswith("@PG"))
        assert mine == gzip.open(os.path.join(GOLD, "var_%s.ref.sam.gz" % name), "rt").read()
    assert stats == open(os.path.join(GOLD, "var_%s.ref.stats" % name)).read()
