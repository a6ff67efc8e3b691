#!/usr/bin/env python3
"""tools/wavelog.py <file> [kernel_id] -- summary of a BMBS_WAVELOG dump (bmbs_kernels.hip: wavelog_begin/_end): for the LAST
launch of the kernel, how many waves ran, on how many CUs / SIMDs, in how many occupancy rounds, how long a wave lives, the shader
clock (cycles per 100 MHz tick), and the SIMD-level timeline (busy SIMDs over time)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
kid = int(sys.argv[2]) if len(sys.argv) > 2 else None
k = (a[:, 0] >> np.uint64(56)).astype(int)
if kid is not None:
    a = a[k == kid]
if not len(a):
    sys.exit("no records")
# split into launches: gaps of > 50 us between consecutive start times
o = np.argsort(a[:, 1]); a = a[o]
t0 = a[:, 1].astype(np.int64); t1 = a[:, 2].astype(np.int64)
brk = np.flatnonzero(np.diff(t0) > 100 * 200) + 1          # 100 ticks per us
groups = np.split(np.arange(len(a)), brk)
groups = [g for g in groups if len(g) > 16]
print("records %d, launches %d" % (len(a), len(groups)))
g = max(groups[-4:], key=len) if groups else np.arange(len(a))
a = a[g]; t0 = t0[g]; t1 = t1[g]
hw = (a[:, 0] & np.uint64(0xffffffff)).astype(np.int64); xcc = ((a[:, 0] >> np.uint64(32)) & np.uint64(0xff)).astype(np.int64)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
simdid = cuid * 4 + simd
base = t0.min()
life = (t1 - t0) / 100.0
cyc = a[:, 3].astype(np.float64)
print("waves %d  span %.1f us  life us: min %.1f med %.1f max %.1f   cycles/wave med %.0f  clock %.2f GHz" % (
    len(a), (t1.max() - base) / 100.0, life.min(), np.median(life), life.max(), np.median(cyc), np.median(cyc / np.maximum(t1 - t0, 1)) / 10.0))
print("distinct XCC %d, CUs %d, SIMDs %d;  waves per SIMD: min %d med %d max %d" % (
    len(set(xcc)), len(set(cuid)), len(set(simdid)), np.bincount(simdid).min(), int(np.median(np.bincount(simdid)[np.bincount(simdid) > 0])), np.bincount(simdid).max()))
# timeline: resident waves and busy SIMDs in 20 slices
T = t1.max() - base
print(" t(us)  resident_waves  busy_SIMDs  started")
for sidx in range(20):
    ta = base + T * sidx // 20; tb = base + T * (sidx + 1) // 20; tm = (ta + tb) // 2
    res = (t0 <= tm) & (t1 > tm)
    print("%6.0f  %8d  %8d  %8d" % ((tm - base) / 100.0, res.sum(), len(set(simdid[res])), ((t0 >= ta) & (t0 < tb)).sum()))
