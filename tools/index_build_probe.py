#!/usr/bin/env python3
"""tools/index_build_probe.py <genome_bp> [--host] -- time bmbs_index_build_device on a synthetic genome (and, with --host,
bmbs_index_build next to it, comparing the files)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bitmapperbs_amd import synth, mapper

ap = argparse.ArgumentParser()
ap.add_argument("genome", type=int)
ap.add_argument("--host", action="store_true")
ap.add_argument("--workdir", default="/tmp/bmbs_ibp")
a = ap.parse_args()
os.makedirs(a.workdir + "/d", exist_ok=True); os.makedirs(a.workdir + "/h", exist_ok=True)
t = time.time()
names, chroms = synth.make_genome(a.genome, 4, seed=20240229)
print("genome synth %.1f s" % (time.time() - t), flush=True)
fa = a.workdir + "/d/g.fa"
t = time.time(); synth.write_fasta(fa, names, chroms); print("fasta write %.1f s" % (time.time() - t), flush=True)
os.environ["BMBS_BUILD_VERBOSE"] = "1"
t = time.time(); mapper.Index.build(fa, fa, threads=min(64, os.cpu_count()), device=0); td = time.time() - t
print("device build %.1f s" % td, flush=True)
if a.host:
    fh = a.workdir + "/h/g.fa"
    os.link(fa, fh) if not os.path.exists(fh) else None
    t = time.time(); mapper.Index.build(fh, fh, threads=min(64, os.cpu_count())); th = time.time() - t
    print("host build %.1f s" % th, flush=True)
    import hashlib
    for s in ("index", "index.bs.pac", "index.bs.index", "index.bs.index.occ", "index.bs.index.bwt", "index.bs.index.sa"):
        ha = hashlib.sha256(open(fa + "." + s, "rb").read()).hexdigest(); hb = hashlib.sha256(open(fh + "." + s, "rb").read()).hexdigest()
        print(s, "SAME" if ha == hb else "DIFFERENT", flush=True)
