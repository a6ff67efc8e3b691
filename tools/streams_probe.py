#!/usr/bin/env python3
"""tools/streams_probe.py -- one batch of the bench workload mapped as S concurrent parts (S contexts sharing one index, S host threads):
throughput against S = 1.  python tools/streams_probe.py [reads] [S ...]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from bitmapperbs_amd import mapper, gpusynth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    Ss = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 4]
    class A: pass
    a = A(); a.genome = 46_000_000; a.workdir = os.environ.get("BMBS_BENCH_DIR", "/tmp/bmbs_bench"); a.repeats = 0
    fa, names, chroms = bench.ensure_index(a, 0, 1, None)
    L, stride = 150, 160
    ix = mapper.Index(fa)
    m0 = mapper.Mapper(ix, 0, e_f=0.04)
    g, lens = gpusynth.upload_genome(chroms)
    seq, qual = gpusynth.make_reads_se(g, lens, n, L, stride, seed=7)
    del g
    max_ops = 2 * m0.threshold(L) + 8
    res = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    cig = torch.empty((n * max_ops,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ms = [m0]
    for S in Ss:
        while len(ms) < S:
            ms.append(mapper.Mapper(ix, 0, share=m0, e_f=0.04))
        cuts = [i * n // S for i in range(S + 1)]
        def part(i):
            a0, b0 = cuts[i], cuts[i + 1]
            ms[i].map_se_device(seq.data_ptr() + a0 * stride, qual.data_ptr() + a0 * stride, L, stride, b0 - a0,
                                res.data_ptr() + a0 * 32, cig.data_ptr() + a0 * max_ops * 4, (b0 - a0) * max_ops)
            ms[i].sync()
        with ThreadPoolExecutor(S) as ex:
            for _ in range(3):
                list(ex.map(part, range(S)))
            for mm in ms[:S]:
                mm.reset_stats()
            torch.cuda.synchronize()
            t = time.perf_counter()
            K = 10
            for _ in range(K):
                list(ex.map(part, range(S)))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
        st = sum(mm.stats() for mm in ms[:S])
        print("S=%d  %.3f ms/step  %.1f M reads/s  stats %s" % (S, dt / K * 1e3, n * K / dt / 1e6, st.tolist()), flush=True)


if __name__ == "__main__":
    main()
