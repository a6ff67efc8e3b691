#!/usr/bin/env python3
"""tools/text_bench.py [pairs=525000] [L=150] -- on the GPU box: the two ends of the file path alone, on one window of FASTQ text
(configs[2]-shaped pairs over a 20 Mb genome): bmbs_map_pe_text to SAM and to BAM, bmbs_map_se_text, and the same window as BGZF
through bmbs_text_open_bgzf; phase times from the library's own trace (BMBS_TEXT_TRACE, stderr) and per kernel from its event
profile.  The output of every form is compared with the first call's (run-to-run identical bytes)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BMBS_TEXT_TRACE"] = "1"
import bench  # noqa: E402
from bitmapperbs_amd import synth, mapper, capi  # noqa: E402


def fastq_text(seq, qual, L):
    import tempfile
    f = tempfile.mktemp(prefix="tb_", dir="/tmp")
    bench.write_fastq_sample(f, seq, qual, L)
    with open(f, "rb") as fh:
        t = fh.read()
    os.unlink(f)
    return t


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 525_000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    wd = os.environ.get("BMBS_BENCH_DIR", "/tmp/bmbs_textbench")
    os.makedirs(wd, exist_ok=True)
    names, chroms = synth.make_genome(20_000_000, 4, seed=3)
    fa = os.path.join(wd, "g20.fa")
    if not os.path.exists(fa + ".index"):
        synth.write_fasta(fa, names, chroms)
        mapper.Index.build(fa, fa, 8, device=0)
    ix = mapper.Index(fa)
    m1, m2 = synth.make_reads_pe(chroms, n=pairs, L=L, seed=5)
    t1 = fastq_text(m1["seq"], m1["qual"], L); t2 = fastq_text(m2["seq"], m2["qual"], L)
    m = mapper.Mapper(ix, 0, e_f=0.08)
    first = {}
    only = os.environ.get("TEXT_BENCH_ONLY")
    for tag, fn in (("pe_sam", lambda: m.map_text(t1, pairs, t2)),
                    ("pe_bam", lambda: m.map_text(t1, pairs, t2, flags=mapper.Mapper.TEXT_BAM)),
                    ("se_sam", lambda: m.map_text(t1, pairs)),
                    ("pe_sam_unmapped", lambda: m.map_text(t1, pairs, t2, flags=mapper.Mapper.TEXT_UNMAPPED))):
        if only and tag not in only.split(","):
            continue
        for rep in range(4):
            sys.stderr.write("== %s rep %d\n" % (tag, rep)); sys.stderr.flush()
            t0 = time.time()
            out = fn()
            dt = time.time() - t0
            if rep == 0:
                first[tag] = out
            assert out == first[tag], tag
        print("%s: %d bytes, last call %.1f ms" % (tag, len(out), dt * 1e3), flush=True)
        if tag == "pe_bam" and hasattr(capi.lib(), "bmbs_debug_bgzf_prof"):       # a -DBGZF_PROFILE build (tools/bgzf_prof.sh)
            import ctypes as C
            v = (C.c_uint64 * 8)()
            capi.lib().bmbs_debug_bgzf_prof(v)
            tot = float(sum(v)) or 1.0
            names = ["load + tables", "parse + CRC + counts", "ranks", "trees + header (lane 0)", "token bits + prefix", "emit", "copy out"]
            print("  k_bgzf_block phases (thread 0's cycles, %d calls): " % 4 + ", ".join("%s %.1f%%" % (n_, 100.0 * x / tot) for n_, x in zip(names, v)), flush=True)
    m.close()


main()
