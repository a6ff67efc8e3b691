#!/usr/bin/env python3
"""tools/scale_debug.py <genome_bp> -- which reads does the GPU path lose on a very large genome?  (diagnostic)"""
import os, sys, subprocess, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
from bitmapperbs_amd import mapper, gpusynth, capi

class A: pass
a = A(); a.workdir = os.environ.get("BMBS_BENCH_DIR", "/tmp/bmbs_bench"); a.genome = int(sys.argv[1]); a.repeats = 0
fa, names, chroms = bench.ensure_index(a, 0, 1, None)
L, n, stride = 150, 1_000_000, 160
ix = mapper.Index(fa)
g, l = gpusynth.upload_genome(chroms)
s, q = gpusynth.make_reads_se(g, l, n, L, stride, seed=7)
seq, qual = s.cpu().numpy(), q.cpu().numpy()
del g
results = {}
for tag, env in (("default", {}),):
    for k_, v_ in env.items(): os.environ[k_] = v_
    m = mapper.Mapper(ix, 0, e_f=0.04)
    r_, pool = m.map_se(seq, qual, L)
    print(tag, "GPU stats", m.stats().tolist(), "status hist", np.bincount(r_["status"], minlength=4).tolist(), "path hist", np.bincount(r_["path"], minlength=5).tolist(), flush=True)
    m.close()
    for k_ in env: del os.environ[k_]
    results[tag] = r_
res = results["default"]
fq = os.path.join(a.workdir, "dbg.fq"); sam = os.path.join(a.workdir, "dbg.sam")
bench.write_fastq_sample(fq, seq, qual, L)
ref = os.path.join(ROOT, "oracle", "_ref", "bitmapperBS")
subprocess.run([ref, "--search", fa, "--seq", fq, "-e", "0.04", "-t", "32", "-o", sam], capture_output=True, cwd=a.workdir)
refmap = {}
for line in open(sam):
    if line[0] == "@": continue
    f = line.split("\t", 6)
    refmap[int(f[0][1:])] = (f[2], int(f[3]), int(f[1]), f[5])
print("reference mapped", len(refmap))
lost = [i for i in range(n) if res[i]["status"] != 1 and i in refmap]
extra = [i for i in range(n) if res[i]["status"] == 1 and i not in refmap]
print("lost on GPU", len(lost), "extra on GPU", len(extra))
h = collections.Counter((refmap[i][0], refmap[i][2]) for i in lost)
print("lost by (chrom, flag):", sorted(h.items()))
pos = np.array([refmap[i][1] for i in lost]); print("lost pos quantiles", np.quantile(pos, [0, .1, .5, .9, 1]).tolist() if len(pos) else None)
for i in lost[:15]:
    print(i, refmap[i], "gpu status/path/n_cand", int(res[i]["status"]), int(res[i]["path"]), int(res[i]["n_cand"]))
diff = [i for i in range(n) if res[i]["status"] == 1 and i in refmap and (ix.chrom_names[int(res[i]["chrom"])], int(res[i]["pos"])) != refmap[i][:2]]
print("mapped elsewhere", len(diff)); 
for i in diff[:10]: print(i, refmap[i], ix.chrom_names[int(res[i]["chrom"])], int(res[i]["pos"]), int(res[i]["flag"]))

# the CPU restatement on the first reads: does it side with the reference or with the GPU?
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
oix = orc.OrcIndex(fa)
nn = 3000
recs, ost, cnt = oix.map_se(orc.params(e_f=0.04), seq[:nn], qual[:nn], L)
print("oracle stats on first", nn, ost.tolist())
for i in [x for x in lost if x < nn][:25]:
    print(i, "ref", refmap[i], "| oracle status/path/n_cand/n_votes", int(recs[i]["status"]), int(recs[i]["path"]), int(recs[i]["n_cand"]), int(recs[i]["n_votes"]), "| gpu status/path/n_cand", int(res[i]["status"]), int(res[i]["path"]), int(res[i]["n_cand"]))
agree_ref = sum(1 for i in range(nn) if (int(recs[i]["status"]) == 1) == (i in refmap))
agree_gpu = sum(1 for i in range(nn) if int(recs[i]["status"]) == int(res[i]["status"]))
print("oracle agrees with reference on", agree_ref, "of", nn, "; with the GPU on", agree_gpu)
# stage check for one lost read: seeds and votes of the GPU vs the oracle's candidate counts

# the HBM suffix array against the index files: random rows
G = ix.ref_len
rng = np.random.default_rng(3)
rows = rng.integers(0, 2 * G + 1, 300000).astype(np.uint64)
m = mapper.Mapper(ix, 0, e_f=0.04)
got = m.locate(rows)
exp = np.array([oix.L.orc_sa_at(oix.h, int(r_)) for r_ in rows], dtype=np.uint64)
bad = np.nonzero(got != exp)[0]
print("rows compared", rows.size, "differ", bad.size)
if bad.size:
    print("bad row quantiles", np.quantile(rows[bad].astype(np.float64), [0, .25, .5, .75, 1]).tolist())
    print("bad expected-position quantiles", np.quantile(exp[bad].astype(np.float64), [0, .25, .5, .75, 1]).tolist())
    for j in bad[:12]:
        print("row", int(rows[j]), "gpu", int(got[j]), "oracle", int(exp[j]), "diff", int(got[j]) - int(exp[j]), "exp mod 8", int(exp[j]) % 8, "exp>>3", int(exp[j]) >> 3)
m.close()
