#!/usr/bin/env python3
"""tools/e2e_setup.py [pairs] -- on the GPU box: the bench's 3.1 Gb index (built on the GPU, cached under $BMBS_BENCH_DIR) and a
paired-end FASTQ sample of the configs[2] workload written `rep` times over, for tools/e2e_probe.sh.  Prints the three paths."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    rep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    import torch
    from bitmapperbs_amd import gpusynth
    args = bench.parse(["--config", "2"])
    cfg = args.cfg
    fa, names, chroms, built = bench.ensure_index(args, cfg, 0, 0, 1, None)
    L = cfg["read_len"]; stride = (L + 15) // 16 * 16
    g, lens = gpusynth.upload_genome(chroms)
    t = gpusynth.make_reads_pe(g, lens, pairs, L, stride, seed=7, sub=args.sub, indel=args.indel, qual=args.qual)
    host = [x.cpu().numpy() for x in t]
    del g, lens, t
    torch.cuda.empty_cache()
    f1 = os.path.join(args.workdir, "e2e_1.fq"); f2 = os.path.join(args.workdir, "e2e_2.fq")
    bench.write_fastq_sample(f1 + ".one", host[0], host[1], L)
    bench.write_fastq_sample(f2 + ".one", host[2], host[3], L)
    for f in (f1, f2):
        with open(f, "wb") as o:
            for _ in range(rep):
                with open(f + ".one", "rb") as i:
                    while True:
                        blk = i.read(1 << 26)
                        if not blk:
                            break
                        o.write(blk)
        os.unlink(f + ".one")
    print(fa, f1, f2, pairs * rep)


if __name__ == "__main__":
    main()
