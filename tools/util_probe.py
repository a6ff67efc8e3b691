#!/usr/bin/env python3
"""tools/util_probe.py -- lane utilisation of the seeding engine (library built with -DBMBS_UTIL):
counters 8/9, 10/11, 12/13 = (wave step iterations, active lane-steps) of k_seed_first / second / extra."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
from bitmapperbs_amd import mapper, gpusynth, capi

class A: pass
a = A(); a.workdir = os.environ.get("BMBS_BENCH_DIR", "/tmp/bmbs_bench"); a.genome = 46_000_000
fa, names, chroms = bench.ensure_index(a, 0, 1, None)
L, n = 150, int(os.environ.get("N", "10000000")); stride = 160
ix = mapper.Index(fa); m = mapper.Mapper(ix, 0, e_f=0.04)
g, l = gpusynth.upload_genome(chroms)
s, q = gpusynth.make_reads_se(g, l, n, L, stride, seed=7)
res = torch.empty((n, 32), dtype=torch.uint8, device="cuda"); cig = torch.empty((n * 20,), dtype=torch.int32, device="cuda")
for _ in range(2):
    m.map_se_device(s.data_ptr(), q.data_ptr(), L, stride, n, res.data_ptr(), cig.data_ptr(), n * 20); m.sync()
c = np.zeros(32, dtype=np.uint64)
m._chk(m._lib.bmbs_counters_all(m._ctx, capi.ptr(c)))
for i, nm in enumerate(("k_seed_first", "k_seed_second", "k_seed_extra")):
    w, act = int(c[8 + 2 * i]), int(c[9 + 2 * i])
    print("%-14s wave-steps %12d  active lane-steps %14d  lanes active per step %.1f / 64" % (nm, w, act, act / max(1, w)))
print(dict(m.profile()))
