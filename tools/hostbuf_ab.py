#!/usr/bin/env python3
"""tools/hostbuf_ab.py [chunk ...] -- e2e.host_buffers_overlapped of bench.py alone, once per BMBS_CHUNK value (0 = the default
500 k units), on the main configuration's index; a context per value on the shared index.  GPU box only."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    chunks = [int(x) for x in sys.argv[1:]] or [0, 250000, 125000]
    args = bench.parse(["--no-cpu", "--no-secondary"])
    cfg = args.cfg
    import torch
    from bitmapperbs_amd import mapper
    torch.cuda.set_device(0)
    fa, names, chroms, built_s = bench.ensure_index(args, cfg, 0, 0, 1, None)
    ix = mapper.Index(fa)
    m = mapper.Mapper(ix, device=0, e_f=cfg["e"], sensitive=0)
    m.sync()
    cfg1 = dict(cfg, units=2_000_000, launches=1)
    job = bench.Job(m, cfg1, chroms, 0, args.sub, args.indel, args.qual)
    del chroms
    for ch in chunks:
        if ch:
            os.environ["BMBS_CHUNK"] = str(ch)
            os.environ["BMBS_SPLIT_MIN"] = str(min(ch, 250000))
        else:
            os.environ.pop("BMBS_CHUNK", None)
        mc = mapper.Mapper(ix, device=0, share=m, e_f=cfg["e"], sensitive=0)
        job.m = mc
        for rep in range(2):
            r = bench.host_buffer_rate(mc, job, torch)
            print(json.dumps({"chunk": ch, "rep": rep, "value": r["value"], "upload_GBps": r["upload_GBps"]}), flush=True)
        mc.close()
    job.m = m
    m.close(); ix.close()


if __name__ == "__main__":
    main()
