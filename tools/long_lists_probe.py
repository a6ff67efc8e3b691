#!/usr/bin/env python3
"""tools/long_lists_probe.py [pairs] -- on the GRCh38-like stress genome of bench.py: how long the candidate lists are (histogram of
bmbs_result.n_cand over the reads of one launch) and what the vote / pair-filter kernels cost on them.
Environment: PROBE_CONFIG=4 the 250-base pairs of configs[4] (default 2), PROBE_SE=1 single-end reads, PROBE_SENSITIVE=1 pairs in --sensitive mode, PROBE_AB="KNOB=v1,v2,..." the same launch
under each value of a knob the context reads when it is created.  tools/trace_probe.sh runs it under rocprofv3 --kernel-trace (a
kernel's own duration: the HIP-event sums printed here include what the other lane ran beside it)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from bitmapperbs_amd import mapper, capi
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    se = os.environ.get("PROBE_SE") == "1"           # single-end reads on the same genome (the SE vote / reduce kernels)
    args = bench.parse(["--config", os.environ.get("PROBE_CONFIG", "2"), "--units", str(pairs), "--launches", "1"] + (["--se"] if se else []))
    cfg = args.cfg
    fa, names, chroms, built = bench.ensure_index(args, cfg, 0, 0, 1, None, grch38_like=True)
    ix = mapper.Index(fa)
    m = mapper.Mapper(ix, device=0, e_f=cfg["e"], sensitive=1 if os.environ.get("PROBE_SENSITIVE") == "1" else 0)
    job = bench.Job(m, cfg, chroms, 0, args.sub, args.indel, args.qual)
    for rep in range(3):
        m.profile_reset()
        job.launch(0); m.sync()
        prof, calls = m.profile_total()
        print("call %d: " % rep + ", ".join("%s=%.2f" % (k, v) for k, v in sorted(prof.items(), key=lambda kv: -kv[1])[:10]))
    # A/B of a knob the context reads when it is created: PROBE_AB="BMBS_SW=reg2,reg"
    ab = os.environ.get("PROBE_AB")
    if ab:
        key, vals = ab.split("=")
        for v in vals.split(","):
            os.environ[key] = v
            m2 = mapper.Mapper(ix, device=0, share=m, e_f=cfg["e"])
            job.m = m2
            for rep in range(3):
                m2.profile_reset()
                job.launch(0); m2.sync()
                prof, calls = m2.profile_total()
            print("%s=%s: " % (key, v) + ", ".join("%s=%.2f" % (k, x) for k, x in sorted(prof.items(), key=lambda kv: -kv[1])[:8]))
            m2.close()
        job.m = m
        os.environ.pop(key, None)
    res = job.res_d.cpu().numpy().view(capi.RESULT_DTYPE).reshape(-1)
    nc = res["n_cand"].astype(np.int64)
    edges = [0, 1, 2, 17, 33, 65, 257, 513, 1025, 4097, 65535, 1 << 30]
    h, _ = np.histogram(nc, bins=edges)
    for a, b, c in zip(edges[:-1], edges[1:], h):
        print("n_cand in [%d, %d): %d reads (%.2f %%), candidates %d" % (a, b, c, 100.0 * c / nc.size, int(nc[(nc >= a) & (nc < b)].sum())))
    print("counters", m.counters())


if __name__ == "__main__":
    main()
