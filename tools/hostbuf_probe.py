#!/usr/bin/env python3
"""tools/hostbuf_probe.py [pairs=2000000] [packed=1] [reps=4] (PROBE_GAP_MS=<ms of idle time between calls>) -- on the GPU box: bmbs_map_pe / bmbs_map_pe_packed on page-locked host buffers
(20 Mb genome, 150 bp pairs), the rate per call; run it under `rocprofv3 --kernel-trace --memory-copy-trace` and tools/timeline_summary.py
shows how uploads, kernels and downloads of a call's chunks overlapped."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bitmapperbs_amd import synth, mapper, capi  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    packed = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    wd = os.environ.get("BMBS_BENCH_DIR", "/tmp/bmbs_textbench")
    os.makedirs(wd, exist_ok=True)
    names, chroms = synth.make_genome(20_000_000, 4, seed=3)
    fa = os.path.join(wd, "g20.fa")
    if not os.path.exists(fa + ".index"):
        synth.write_fasta(fa, names, chroms)
        mapper.Index.build(fa, fa, 8, device=0)
    ix = mapper.Index(fa)
    L, stride = 150, 160
    m1, m2 = synth.make_reads_pe(chroms, n=min(n, 500_000), L=L, seed=5)
    rep = (n + m1["seq"].shape[0] - 1) // m1["seq"].shape[0]
    def rows(a):
        o = np.zeros((a.shape[0], stride), dtype=np.uint8); o[:, :L] = a
        return np.tile(o, (rep, 1))[:n]
    host = [rows(m1["seq"]), rows(m1["qual"]), rows(m2["seq"]), rows(m2["qual"])]
    lib = capi.lib()
    m = mapper.Mapper(ix, 0, e_f=0.08)
    pw = (L + 31) // 32 + (L + 63) // 64
    pin = []
    for i, h in enumerate(host):
        if packed and i % 2 == 0:
            p_ = lib.bmbs_host_alloc(n * pw * 8)
            assert lib.bmbs_pack_rows(h.ctypes.data, L, stride, n, None, p_, pw, 16, None) == 0
        else:
            p_ = lib.bmbs_host_alloc(h.nbytes); C.memmove(p_, h.ctypes.data, h.nbytes)
        pin.append(p_)
    ops = m.max_cigar_ops(L)
    cap = 2 * n * ops
    res = lib.bmbs_host_alloc(2 * n * 32); pool = lib.bmbs_host_alloc(cap * 4)
    used = C.c_int64(0)
    def call():
        if packed:
            rc = lib.bmbs_map_pe_packed(m._ctx, pin[0], pin[2], pw, pin[1], pin[3], None, None, L, stride, n, res, pool, cap, C.byref(used))
        else:
            rc = lib.bmbs_map_pe(m._ctx, pin[0], pin[1], pin[2], pin[3], L, stride, n, res, pool, cap, C.byref(used))
        assert rc == 0, lib.bmbs_last_error(m._ctx)
    call()
    gap = float(os.environ.get("PROBE_GAP_MS", "0")) / 1e3        # idle time between calls (is a call right behind another one slower?)
    for r in range(reps):
        if gap:
            time.sleep(gap)
        t = time.perf_counter(); call(); dt = time.perf_counter() - t
        up = (n * pw * 8 + n * stride) * 2 if packed else n * stride * 4
        print("%s: %d pairs in %.2f ms = %.1f M reads/s, upload %.1f GB/s" % ("packed" if packed else "ascii", n, dt * 1e3, 2 * n / dt / 1e6, up / dt / 1e9), flush=True)
    m.close()


main()
