#!/usr/bin/env python3
"""bench.py -- headline measurement of the MI355X bisulfite mapping hot path.

Workload (BASELINE.json configs[1]): synthetic 150 bp single-end bisulfite reads against a
chr21-size (46 Mb, 4 chromosomes, N-free) synthetic genome, -e 0.04; one "step" = one pass of the
whole device pipeline (seed -> locate -> vote -> Myers filter -> reduce -> align -> finalize) over one
batch of reads that is already resident in HBM; results (32-byte records + CIGAR pool) stay in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--genome G]

N > 1 is launched by torch.distributed.run (one rank per GPU); reads shard by rank (weak scaling,
index replicated per GPU, no data-path collective); the only collective is the RCCL all-reduce of
the five mapstats counters (Schema.cpp:451-476).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GATHER_CEILING_GREQ = 54.0   # measured on this chip: dependent divergent gathers/s (tools/gather_bench.hip, profiles/r01_gather_bench.txt)
SECTOR_CEILING_GBS = 3100.0  # the same measurement in bytes: 48.4 G gathers/s over a 34 GB table x one 64-byte sector each


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_pmc_fetch_write.csv: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, KB units as
    rocprofv3 reports them).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide (>=16 B/lane) coalesced
    streams by 2x; the mapping kernels issue 4-16 B per-lane gathers, for which the guide gives no calibration,
    so the raw (FETCH_SIZE + WRITE_SIZE) * 1024 is reported.  None when no profile has been committed."""
    import csv
    path = os.path.join(ROOT, "profiles", "r01_pmc_fetch_write.csv")
    if not os.path.exists(path):
        return None
    for row in csv.DictReader(open(path)):
        if row["kernel"].split("<")[0] == kernel:
            return int((float(row["FETCH_SIZE_KB_last_launch"]) + float(row["WRITE_SIZE_KB_last_launch"])) * 1024)
    return None


def gather_roofline(kernel, cnt, kern_ms):
    c = cnt.get(kernel)
    if not isinstance(c, dict) or kern_ms.get(kernel, 0) <= 0:
        return None
    req = c["n_hash"] + c["n_ext"] + c["n_sa"]
    ach = req / (kern_ms[kernel] * 1e-3) / 1e9
    return {"index_requests_per_launch": int(req), "achieved_Greq_s": round(ach, 2), "ceiling_Greq_s": GATHER_CEILING_GREQ,
            "frac": round(ach / GATHER_CEILING_GREQ, 4),
            "note": "counts index gathers only; read-character loads and result stores are further requests of the same kind"}


def sector_roofline(kernel, kern_ms):
    t = pmc_traffic(kernel)
    if t is None or kern_ms.get(kernel, 0) <= 0:
        return None
    gbs = t / (kern_ms[kernel] * 1e-3) / 1e9
    return {"traffic_GBps": round(gbs, 1), "random_sector_ceiling_GBps": SECTOR_CEILING_GBS, "frac": round(gbs / SECTOR_CEILING_GBS, 4)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--genome", type=int, default=46_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("-e", type=float, default=0.04)
    ap.add_argument("--cpu-sample", type=int, default=8_000_000, help="reads timed on the host CPU baseline (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pe", action="store_true", help="paired-end fast mode: --reads pairs per GPU per step (value counts 2 reads per pair)")
    ap.add_argument("--sensitive", action="store_true", help="with --pe: Map_Pair_Seq_end_to_end (--sensitive) instead of fast mode")
    ap.add_argument("--sub", type=float, default=0.005, help="substitution rate of the synthetic reads (SURVEY.md 8d: 0.5 %%)")
    ap.add_argument("--repeats", type=int, default=0, help="stress: plant this many diverged copies of 300-bp elements (5 families, 2-8 %% divergence) into the genome")
    ap.add_argument("--workdir", default=os.environ.get("BMBS_BENCH_DIR", "/tmp/bmbs_bench"))
    return ap.parse_args()


def ensure_index(args, rank, world, dist):
    from bitmapperbs_amd import synth, mapper
    rep = getattr(args, "repeats", 0)
    wd = os.path.join(args.workdir, "g%d%s" % (args.genome, "_r%d" % rep if rep else ""))
    fa = os.path.join(wd, "g.fa")
    names, chroms = synth.make_genome(args.genome, 4, seed=20240229)
    if rep:
        # interspersed-repeat stress (Alu-like): seeds inside a copy hit hundreds of places, candidate lists get long
        rng = np.random.default_rng(77)
        for fam in range(5):
            el = synth._ACGT[rng.integers(0, 4, 300)]
            for _ in range(rep // 5):
                ch = chroms[int(rng.integers(0, len(chroms)))]
                p = int(rng.integers(0, ch.size - 300))
                e = el.copy()
                m = rng.random(300) < rng.uniform(0.02, 0.08)
                e[m] = synth._ACGT[rng.integers(0, 4, int(m.sum()))]
                ch[p:p + 300] = synth.revcomp(e) if rng.random() < 0.5 else e
    if rank == 0:
        os.makedirs(wd, exist_ok=True)
        if not os.path.exists(fa + ".index.bs.index.sa"):
            synth.write_fasta(fa, names, chroms)
            t = time.time()
            mapper.Index.build(fa, fa, threads=min(64, os.cpu_count() or 1))
            sys.stderr.write("[bench] index built in %.1fs\n" % (time.time() - t))
    if world > 1:
        dist.barrier()
    return fa, names, chroms


def write_fastq_sample(path, seq, qual, L):
    """vectorised FASTQ writer: fixed-width names s%08d -> fixed record length"""
    n = seq.shape[0]
    rec = np.empty((n, 1 + 9 + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("s")
    idx = np.arange(n, dtype=np.int64)
    for d in range(8):
        rec[:, 2 + d] = (idx // 10 ** (7 - d)) % 10 + ord("0")
    rec[:, 10] = 10
    rec[:, 11:11 + L] = seq[:, :L]
    rec[:, 11 + L] = 10; rec[:, 12 + L] = ord("+"); rec[:, 13 + L] = 10
    rec[:, 14 + L:14 + 2 * L] = qual[:, :L]
    rec[:, 14 + 2 * L] = 10
    rec.tofile(path)


def cpu_baseline(args, fa, seq_h, qual_h, L):
    """Time the reference's own CPU path (oracle/_ref/bitmapperBS, kind 'reference') -- or, when it has
    not been built, the scalar restatement (oracle/liboracle.so, kind 'port') -- on a bounded sample."""
    n = seq_h.shape[0]
    ref = os.path.join(ROOT, "oracle", "_ref", "bitmapperBS")
    sample = "first %d reads of the step batch (150 bp SE, e=%.2f)" % (n, args.e)
    if os.path.exists(ref):
        fq = os.path.join(args.workdir, "cpu_sample.fq")
        write_fastq_sample(fq, seq_h, qual_h, L)
        cores = min(8, os.cpu_count() or 1)
        out = os.path.join(args.workdir, "cpu_sample.sam")
        p = subprocess.run([ref, "--search", fa, "--seq", fq, "-e", str(args.e), "-t", str(cores), "-o", out],
                           capture_output=True, text=True, cwd=args.workdir)
        secs = None
        for line in p.stderr.splitlines():
            if line.strip().startswith("Total:"):
                secs = float(line.split()[2])          # "Total: <load s> <map s>" (Bitmapper_main.cpp:262)
        if p.returncode == 0 and secs and secs > 0:
            return {"value": n / secs / 1e6, "unit": "Mreads/s", "cores": cores, "kind": "reference",
                    "sample": sample + ", bitmapperBS -t %d, mapping seconds as printed by main" % cores}, out
    # port
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    oix = orc.OrcIndex(fa)
    m = min(n, 300_000)
    t = time.time()
    oix.map_se(orc.params(e_f=args.e), seq_h[:m], qual_h[:m], L)
    dt = time.time() - t
    return {"value": m / dt / 1e6, "unit": "Mreads/s", "cores": 1, "kind": "port",
            "sample": "first %d reads of the step batch, scalar CPU restatement (oracle/)" % m}, None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the mapping path has no CPU fallback)")
    torch.cuda.set_device(local)
    from bitmapperbs_amd import mapper, gpusynth, capi

    fa, names, chroms = ensure_index(args, rank, world, dist)
    L = args.read_len
    stride = (L + 15) // 16 * 16
    n = args.reads
    ix = mapper.Index(fa)
    m = mapper.Mapper(ix, device=local, e_f=args.e, sensitive=1 if args.sensitive else 0)
    k = m.threshold(L)
    genome_d, lens_d = gpusynth.upload_genome(chroms)
    max_ops = 2 * k + 8
    if not args.pe:
        seq_d, qual_d = gpusynth.make_reads_se(genome_d, lens_d, n, L, stride, seed=7 + 1000 * rank, sub=args.sub)
        cig_cap = n * max_ops
        res_d = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    else:
        seq_d, qual_d, seq2_d, qual2_d = gpusynth.make_reads_pe(genome_d, lens_d, n, L, stride, seed=7 + 1000 * rank, sub=args.sub)
        cig_cap = 2 * n * max_ops
        res_d = torch.empty((2 * n, 32), dtype=torch.uint8, device="cuda")
    del genome_d
    cig_d = torch.empty((cig_cap,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def step():
        if not args.pe:
            m.map_se_device(seq_d.data_ptr(), qual_d.data_ptr(), L, stride, n, res_d.data_ptr(), cig_d.data_ptr(), cig_cap)
        else:
            m.map_pe_device(seq_d.data_ptr(), qual_d.data_ptr(), seq2_d.data_ptr(), qual2_d.data_ptr(), L, stride, n,
                            res_d.data_ptr(), cig_d.data_ptr(), cig_cap)
        m.sync()

    for _ in range(args.warmup):
        step()
    m.reset_stats()
    m.sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kern_ms: dict[str, float] = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for name, ms in m.profile():
            kern_ms[name] = kern_ms.get(name, 0.0) + ms
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    stats = torch.from_numpy(m.stats()).cuda()
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)        # the only collective: 5 x int64 mapstats
    dt = float(tt.item())
    stats = stats.cpu().numpy()
    cnt = m.counters()
    for kname in kern_ms:
        kern_ms[kname] /= max(1, args.steps)

    if rank == 0:
        reads_per_unit = 2 if args.pe else 1
        nr = n * reads_per_unit                     # reads seeded per launch
        total_reads = nr * world * args.steps
        value = total_reads / dt / 1e6
        mapping = {kn: v for kn, v in kern_ms.items() if kn.startswith("k_")}
        dom = max(mapping, key=mapping.get) if mapping else "k_seed_first"
        # ALGORITHMIC bytes of one launch = SURVEY.md §8d per-unit figures x this launch's event counts
        # (10 B per 16-mer lookup, 80 B per backward extension = two 40-B Occ blocks, 4 B per SA read,
        #  ceil(len/4)+1 B per fetched window, read characters actually consumed, 16 B per recorded seed)
        win = (L + 2 * k + 3) // 4 + 1
        def seed_bytes(c):
            return 10 * c["n_hash"] + 80 * c["n_ext"] + 16 * c["n_hash"] + c["n_ext"] + 16 * c["n_hash"]
        alg = {
            "k_seed_first": seed_bytes(cnt["k_seed_first"]) + 14 * nr,
            # every read row once (staged through LDS), carry-in + verdict records, one SA word and one window per verified read
            "k_seed_decide": (L + 14 + 40) * nr + 4 * cnt["n_sa"] + ((L + 3) // 4 + 1) * cnt["n_ungapped"],
            "k_seed_second": seed_bytes(cnt["k_seed_second"]) + 4 * cnt["k_seed_second"]["n_sa"] + 24 * cnt["k_seed_second"]["n_hash"],
            "k_seed_extra": seed_bytes(cnt["k_seed_extra"]),
            "k_locate": (4 + 8) * cnt["n_cand_slots"],
            "k_vote": (8 + 16 + 4) * cnt["n_cand_slots"],
            "k_vote_fused": (4 + 16 + 4) * cnt["n_cand_slots"],
            "k_filter": (win + L + 16 + 8) * cnt["n_filter"],
            "k_align_ungapped": (win + 2 * L + 16) * cnt["n_jobs"],
            "k_align_sw": (win + 2 * L + 16) * cnt["n_sw"],
            "k_finalize": 32 * nr,
        }
        bytes_dom = alg.get(dom, 0)
        ach = bytes_dom / (kern_ms[dom] * 1e-3) / 1e9 if kern_ms.get(dom, 0) > 0 else 0.0
        rl_all = {kn: round(alg[kn] / (kern_ms[kn] * 1e-3) / 1e9, 2) for kn in alg if kern_ms.get(kn, 0) > 0}
        out = {
            "metric": "M 150bp %s reads aligned/s" % ("PE" if args.pe else "SE"), "value": round(value, 4), "unit": "Mreads/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: %d synthetic %d bp SE bisulfite reads" % (n, L) if not args.pe else
                                    "%d synthetic %d bp read PAIRS (%s PE mode, insert 170-400, substitutions %.3f)" % (n, L, "sensitive" if args.sensitive else "fast", args.sub)) +
                                   " per GPU per step vs %d bp 4-chromosome synthetic (chr21-size) genome, -e %.2f (k=%d), inputs and "
                                   "results resident in HBM" % (args.genome, args.e, k),
                       "reads_per_gpu_per_step": n, "read_len": L, "genome_bp": args.genome,
                       "parallelism": "reads sharded by rank, index replicated, RCCL all-reduce of 5 mapstats counters"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": pmc_traffic(dom),
                         "algorithmic_bytes_per_launch": int(bytes_dom), "avg_launch_ms": round(kern_ms.get(dom, 0.0), 4),
                         # the same kernel against the bound that really applies to an index walk: divergent gather requests/s
                         # (index lookups only: one per 16-mer table access, backward extension, SA read) vs the measured ceiling
                         "gather": gather_roofline(dom, cnt, kern_ms),
                         # and in bytes: the HBM-side traffic of the committed PMC pass over the live launch time, against
                         # what the chip delivers when every 64-byte sector is a random access of its own
                         "sectors": sector_roofline(dom, kern_ms)},
            "kernels_ms_per_step": {a: round(b, 4) for a, b in kern_ms.items()},
            "kernels_algorithmic_GBps": rl_all,
            "counters_per_step": cnt,
            "mapstats": {"reads": int(stats[0]), "unique": int(stats[1]), "ambiguous": int(stats[2]),
                         "unmapped": int(stats[0] - stats[1] - stats[2]), "mapped_bases": int(stats[3]), "error_bases": int(stats[4])},
        }
        if world == 1 and not args.no_cpu and not args.pe:
            ns = min(n, args.cpu_sample)
            seq_h = seq_d[:ns].cpu().numpy()
            qual_h = qual_d[:ns].cpu().numpy()
            cb, ref_sam = cpu_baseline(args, fa, seq_h, qual_h, L)
            out["cpu_baseline"] = cb
            if ref_sam:
                # bonus check: the GPU records of the same sample print the same SAM lines as the reference
                nchk = min(ns, 200_000)
                res_h = res_d[:nchk].cpu().numpy().view(capi.RESULT_DTYPE).reshape(-1)
                cig_h = cig_d.cpu().numpy().view(np.uint32)
                names_s = [b"s%08d" % i for i in range(nchk)]
                mine = set(mapper.sam_lines_se(ix, names_s, seq_h[:nchk], qual_h[:nchk], L, res_h, cig_h))
                with open(ref_sam) as f:
                    theirs = set(x for x in f if not x.startswith("@") and int(x[1:x.index("\t")]) < nchk)
                out["sample_sam_identical_to_reference"] = (mine == theirs)
                out["sample_sam_lines_compared"] = len(theirs)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    m.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
