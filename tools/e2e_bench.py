#!/usr/bin/env python3
"""tools/e2e_bench.py -- file-to-file measurement of SURVEY.md section 8(f) row 1 (host I/O path).

FASTQ on disk -> SAM on disk, `bitmapperbs_amd/bmbs_search` (GPU) next to the reference binary
(`oracle/_ref/bitmapperBS`, when it has travelled to this box) on the same files, same options.
The SAM files are compared as sorted line sets (the reference's -t N output order is not the input order).

  python tools/e2e_bench.py [--reads 10000000] [--pe] [--ref-threads 8,32] [--io-threads 32] [--out gpurun_out/e2e.json]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def sorted_digest(path):
    p = subprocess.run("grep -v '^@' %s | LC_ALL=C sort --parallel=16 -S 8G | sha256sum" % path, shell=True, capture_output=True, text=True)
    return p.stdout.split()[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--genome", type=int, default=46_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("-e", type=float, default=0.04)
    ap.add_argument("--pe", action="store_true")
    ap.add_argument("--sensitive", action="store_true", help="with --pe: --sensitive for both programs")
    ap.add_argument("--sub", type=float, default=0.005, help="substitution rate of the synthetic reads")
    ap.add_argument("--repeats", type=int, default=0, help="plant this many diverged repeat copies into the genome (as bench.py --repeats)")
    ap.add_argument("--extra", default="", help="further options handed to both programs, e.g. '--unmapped_out --ambiguous_out'")
    ap.add_argument("--ref-threads", default="8")
    ap.add_argument("--io-threads", default="32")
    ap.add_argument("--batch", type=int, default=1_000_000)
    ap.add_argument("--driver-args", default="", help="further options for bmbs_search only, e.g. '--contexts 3 --devices 0'")
    ap.add_argument("--no-ref", action="store_true", help="skip the reference run (timing experiments)")
    ap.add_argument("--workdir", default=os.environ.get("BMBS_BENCH_DIR", "/tmp/bmbs_bench"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "e2e.json"))
    args = ap.parse_args()
    import torch
    from bitmapperbs_amd import gpusynth
    cfg = dict(genome=args.genome, n_chrom=4 if args.genome < 1_000_000_000 else 24)
    args.host_index = False
    fa, names, chroms, _ = bench.ensure_index(args, cfg, 0, 0, 1, None, repeats=args.repeats)
    L = args.read_len
    stride = (L + 15) // 16 * 16
    genome_d, lens_d = gpusynth.upload_genome(chroms)
    wd = args.workdir
    files = []
    if not args.pe:
        s, q = gpusynth.make_reads_se(genome_d, lens_d, args.reads, L, stride, seed=7, sub=args.sub)
        fq = os.path.join(wd, "e2e.fq")
        bench.write_fastq_sample(fq, s.cpu().numpy(), q.cpu().numpy(), L)
        files = [fq]
        in_args = ["--seq", fq]
    else:
        s1, q1, s2, q2 = gpusynth.make_reads_pe(genome_d, lens_d, args.reads, L, stride, seed=7, sub=args.sub)
        f1 = os.path.join(wd, "e2e_1.fq"); f2 = os.path.join(wd, "e2e_2.fq")
        bench.write_fastq_sample(f1, s1.cpu().numpy(), q1.cpu().numpy(), L)
        bench.write_fastq_sample(f2, s2.cpu().numpy(), q2.cpu().numpy(), L)
        files = [f1, f2]
        in_args = ["--seq1", f1, "--seq2", f2] + (["--sensitive"] if args.sensitive else [])
    in_args += args.extra.split()
    del genome_d
    torch.cuda.empty_cache()
    in_bytes = sum(os.path.getsize(f) for f in files)
    n_reads = args.reads * (2 if args.pe else 1)
    res = {"workload": "%d %s of %d bp, genome %d bp%s, -e %.2f, substitutions %.3f%s%s, FASTQ %.2f GB on %s" % (
               args.reads, "pairs" if args.pe else "SE reads", L, args.genome, " + %d repeat copies" % args.repeats if args.repeats else "", args.e, args.sub,
               ", --sensitive" if args.sensitive else "", " " + args.extra if args.extra else "", in_bytes / 1e9, wd),
           "runs": []}
    drv = os.path.join(ROOT, "bitmapperbs_amd", "bmbs_search")
    digests = {}
    for t in [int(x) for x in args.io_threads.split(",") if x]:
        out = os.path.join(wd, "e2e_gpu.sam")
        for rep in range(2):                      # second run: page cache warm, index files cached
            if os.path.exists(out):
                os.unlink(out)                    # freeing a multi-GB file is not part of either program's work
            t0 = time.time()
            p = subprocess.run([drv, "--search", fa] + in_args + ["-e", str(args.e), "-o", out, "-t", str(t), "--batch", str(args.batch), "--verbose"] +
                               args.driver_args.split(), capture_output=True, text=True)
            dt = time.time() - t0
            if p.returncode:
                print(p.stderr[-2000:]); sys.exit(1)
        line = [x for x in p.stderr.splitlines() if x.startswith("[bmbs_search]") and "mapping wall" in x][-1]
        map_wall = float(line.split("mapping wall")[1].split("s")[0])
        res["runs"].append({"program": "bmbs_search (1 MI355X)", "io_threads": t, "wall_s": round(dt, 3), "mapping_wall_s": map_wall,
                            "Mreads_per_s_wall": round(n_reads / dt / 1e6, 3), "Mreads_per_s_mapping": round(n_reads / map_wall / 1e6, 3),
                            "sam_bytes": os.path.getsize(out), "detail": line,
                            "detail2": ([x for x in p.stderr.splitlines() if x.startswith("[bmbs_search] read stage")] or [""])[-1]})
        digests["gpu"] = sorted_digest(out)
    # the same pipeline with the SAM text handed to write() on /dev/null: what the host side sustains when the file system
    # (one file, buffered writes: ~10.5 GB/s on the MI355X boxes, tools/host_mem_probe) is taken out
    t = int(args.io_threads.split(",")[0])
    p = subprocess.run([drv, "--search", fa] + in_args + ["-e", str(args.e), "-o", "/dev/null", "-t", str(t), "--batch", str(args.batch), "--verbose"] +
                       args.driver_args.split(), capture_output=True, text=True)
    if p.returncode == 0:
        line = [x for x in p.stderr.splitlines() if x.startswith("[bmbs_search]") and "mapping wall" in x][-1]
        map_wall = float(line.split("mapping wall")[1].split("s")[0])
        res["runs"].append({"program": "bmbs_search (1 MI355X), -o /dev/null", "io_threads": t, "mapping_wall_s": map_wall,
                            "Mreads_per_s_mapping": round(n_reads / map_wall / 1e6, 3), "detail": line})
    ref = os.path.join(ROOT, "oracle", "_ref", "bitmapperBS")
    if os.path.exists(ref) and not args.no_ref:
        for t in [int(x) for x in args.ref_threads.split(",") if x]:
            out = os.path.join(wd, "e2e_ref.sam")
            if os.path.exists(out):
                os.unlink(out)
            t0 = time.time()
            p = subprocess.run([ref, "--search", fa] + in_args + ["-e", str(args.e), "-t", str(t), "-o", out], capture_output=True, text=True, cwd=wd)
            dt = time.time() - t0
            secs = None
            for ln in p.stderr.splitlines():
                if ln.strip().startswith("Total:"):
                    secs = float(ln.split()[2])
            res["runs"].append({"program": "reference bitmapperBS", "threads": t, "wall_s": round(dt, 3), "mapping_s_as_printed": secs,
                                "Mreads_per_s_wall": round(n_reads / dt / 1e6, 3),
                                "Mreads_per_s_mapping": round(n_reads / secs / 1e6, 3) if secs else None, "sam_bytes": os.path.getsize(out)})
            digests["ref_t%d" % t] = sorted_digest(out)
    res["sorted_sam_sha256"] = digests
    res["sam_identical"] = len(set(digests.values())) == 1 if len(digests) > 1 else None
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
