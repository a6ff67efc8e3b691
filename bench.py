#!/usr/bin/env python3
"""bench.py -- headline measurement of the MI355X bisulfite mapping hot path.

Default workload = BASELINE.json configs[2]: 50 M synthetic 150 bp read PAIRS against a GRCh38-size index (3.1 Gb random
genome, 24 chromosomes, N-free: repeat-poor, unlike the real GRCh38 -- stated in `config.workload`), default (fast) paired-end
mode, default -e 0.08 (k = 12), one MI355X.  One "step" = the whole 50 M-pair job: five launches of 10 M pairs, each one pass of
the device pipeline (prepare -> seed -> locate/vote -> pair filter -> Myers -> pairing -> align -> finalize) over pairs that
are already resident in HBM; the 32-byte records and the CIGAR pool stay in HBM.  `--config 1` = configs[1] (10 M 150 bp SE
reads, 46 Mb genome, -e 0.04), `--config 3` = configs[3]'s per-GPU share (--sensitive), `--config 4` = configs[4] (250 bp).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C] ...

N > 1 without WORLD_SIZE in the environment: bench.py starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a CHILD process before anything touches the GPU and exits with its code (a process that has initialised the GPU is never
re-exec'ed).  One rank per GPU; pairs shard by rank (weak scaling, index replicated per GPU, no data-path collective); the only
collective is the RCCL all-reduce of the five mapstats counters (Schema.cpp:451-476).  Rank 0 prints ONE JSON line.

The index is built on the GPU (bmbs_index_build_device: 6.2 G suffixes in seconds) and cached under --workdir.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and two streams on one queue run
# one behind the other: with four, the kernel streams of a context's two lanes shared a queue in some processes (lanes not overlapping:
# the same binary 1108 or 1248 M reads/s from one run to the next).  The library asks for eight when it is loaded; torch may touch the
# GPU before that, so the bench says it here, before anything else.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GATHER_CEILING_GREQ = 54.0   # measured on this chip: dependent divergent gathers/s (tools/gather_bench.hip, profiles/r01_gather_bench.txt)
SECTOR_CEILING_GBS = 3100.0  # the same measurement in bytes: 48.4 G gathers/s over a 34 GB table x one 64-byte sector each
MIN_TIMED_S = 2.0            # the timed region is stretched to at least this (whole extra passes per step) whatever --steps says
SECONDARY_MIN_S = 1.0        # ... and that of every secondary key to this

CONFIGS = {
    1: dict(pe=False, sensitive=False, genome=46_000_000, n_chrom=4, read_len=150, e=0.04, units=10_000_000, launches=1,
            label="BASELINE configs[1]: 10 M synthetic 150 bp SE reads vs chr21-size genome, -e 0.04"),
    2: dict(pe=True, sensitive=False, genome=3_100_000_000, n_chrom=24, read_len=150, e=0.08, units=10_000_000, launches=5,
            label="BASELINE configs[2]: 50 M synthetic 150 bp PE read pairs vs GRCh38-size genome, default (fast) mode"),
    3: dict(pe=True, sensitive=True, genome=3_100_000_000, n_chrom=24, read_len=150, e=0.08, units=5_000_000, launches=5,
            label="BASELINE configs[3] per-GPU share: 25 M synthetic 150 bp PE read pairs vs GRCh38-size genome, --sensitive"),
    4: dict(pe=True, sensitive=False, genome=3_100_000_000, n_chrom=24, read_len=250, e=0.08, units=2_500_000, launches=5,
            label="BASELINE configs[4] share: 12.5 M synthetic 250 bp PE read pairs, -e 0.08 (k = 20), vs GRCh38-size genome"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--units", type=int, default=None, help="reads (SE) or pairs (PE) per launch per GPU")
    ap.add_argument("--launches", type=int, default=None, help="launches per step (step = launches x units)")
    ap.add_argument("--genome", type=int, default=None)
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("-e", type=float, default=None)
    ap.add_argument("--pe", action="store_true", default=None)
    ap.add_argument("--se", dest="pe", action="store_false")
    ap.add_argument("--sensitive", action="store_true", default=None)
    ap.add_argument("--sub", type=float, default=0.005, help="substitution rate of the synthetic reads (SURVEY.md 8d: 0.5 %%)")
    ap.add_argument("--indel", type=float, default=0.0002, help="indel rate per base (SURVEY.md 8d: 0.02 %%; at most one per read)")
    ap.add_argument("--qual", default="const", choices=["const", "random"])
    ap.add_argument("--grch38-like", action="store_true", help="the main workload on the repeat-rich genome of secondary.grch38_like (profiles of the representative case)")
    ap.add_argument("--repeats", type=int, default=0, help="stress: plant this many diverged copies of 300-bp elements into the genome")
    ap.add_argument("--cpu-sample", type=int, default=None, help="reads / pairs timed on the host CPU baseline (rank 0, N=1)")
    ap.add_argument("--cpu-threads", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-single-lane", action="store_true", help="skip the extra single-lane pass that times the kernels alone")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (configs[1] SE line, stress keys)")
    ap.add_argument("--no-stress", action="store_true", help="keep the PCIe-inclusive and file-to-file keys, skip the stress keys behind them (other genomes / read sets)")
    ap.add_argument("--min-seconds", type=float, default=MIN_TIMED_S)
    ap.add_argument("--host-index", action="store_true", help="build the index with the host builder (bmbs_index_build)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / collective check without a GPU: gloo, no mapping (tests)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="collective backend of the N > 1 run (gloo: tests)")
    ap.add_argument("--same-device", action="store_true", help="every rank on GPU 0 (a functional test of the N > 1 path on a 1-GPU box; use with --dist-backend gloo)")
    ap.add_argument("--fasta", default=os.environ.get("BMBS_BENCH_FASTA"),
                    help="a real assembly (e.g. GRCh38) instead of the synthetic genome of the main configuration: indexed by the GPU builder, "
                         "reads drawn from its N-free parts (SURVEY.md 8d); also BMBS_BENCH_FASTA")
    ap.add_argument("--workdir", default=os.environ.get("BMBS_BENCH_DIR", "/tmp/bmbs_bench"))
    a = ap.parse_args(argv)
    cfg = dict(CONFIGS[a.config])
    for key, attr in (("units", "units"), ("launches", "launches"), ("genome", "genome"), ("read_len", "read_len"), ("e", "e"),
                      ("pe", "pe"), ("sensitive", "sensitive")):
        v = getattr(a, attr)
        if v is not None:
            if cfg[key] != v:
                cfg["label"] = "custom (from configs[%d])" % a.config
            cfg[key] = v
    if a.genome is not None and a.genome < 1_000_000_000:
        cfg["n_chrom"] = 4
    if a.grch38_like and a.launches is None and cfg["launches"] > 2:
        # the repeat-rich genome's candidate lists take 30-38 GB of work buffers per lane at this launch size: with five launches' inputs
        # and results resident beside the index, the trigram table and the single-lane pass's context, 288 GB do not always hold them
        cfg["launches"] = 2
    a.cfg = cfg
    return a


# ---- N > 1 launcher ------------------------------------------------------------------------------------------------------------
def launch_children(args) -> int:
    """--gpus N without a torchrun environment: start the N ranks as a child process tree.  Nothing in this process has
    touched the GPU yet (torch is not even imported), and the child is a fresh interpreter."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


# ---- workload ------------------------------------------------------------------------------------------------------------------
# repeat content of the GRCh38-like stress genome: (name, element length, copies per Gb, divergence range, what)
REPEAT_FAMILIES = [
    ("SINE/Alu-like", 300, 350_000, (0.05, 0.18), "interspersed"),
    ("SINE/MIR-like", 260, 180_000, (0.15, 0.30), "interspersed"),
    ("LINE/L1-like 5'-truncated", 900, 170_000, (0.04, 0.20), "interspersed"),
    ("LINE/L1-like full length", 6000, 1_600, (0.02, 0.10), "interspersed"),
    ("LTR-like", 450, 140_000, (0.08, 0.22), "interspersed"),
    ("DNA-transposon-like", 280, 110_000, (0.12, 0.28), "interspersed"),
    ("segmental duplication", 40_000, 700, (0.005, 0.03), "interspersed"),
    ("alpha-satellite-like arrays (171 bp monomer)", 171, 160_000, (0.01, 0.06), "tandem"),
    ("microsatellite tracts (2-6 bp unit)", 120, 90_000, (0.0, 0.03), "simple"),
]


def plant_repeat_families(chroms, seed=99, device="cuda"):
    """GRCh38-like repeat structure planted into the uniform-random chromosomes, on the GPU: about 45 % of the bases come from
    nine families of 10^3-10^6 copies each with per-copy divergence drawn from the family's range (half of the copies reverse
    complemented), satellite monomers in head-to-tail arrays of 300-3000 units, microsatellites as short unit repeats.
    -> fraction of the genome covered (overlaps counted once is not attempted: copies may land on one another, as in a real genome)"""
    import torch
    g = torch.Generator(device=device); g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    comp = torch.arange(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    planted = 0
    total = sum(c.size for c in chroms)
    for ci, ch in enumerate(chroms):
        t = torch.from_numpy(ch).to(device)
        n = t.numel()
        for fi, (name, elen, per_gb, (d_lo, d_hi), kind) in enumerate(REPEAT_FAMILIES):
            fg = torch.Generator(device=device); fg.manual_seed(seed * 1000 + fi)        # the family's consensus is the same on every chromosome
            copies = max(1, int(per_gb * n / 1e9))
            if kind == "simple":
                unit_len = torch.randint(2, 7, (copies,), generator=g, device=device)
                units = acgt[torch.randint(0, 4, (copies, 6), generator=g, device=device)]
                j = torch.arange(elen, device=device)[None, :]
                vals = torch.gather(units, 1, j % unit_len[:, None])
            else:
                cons = acgt[torch.randint(0, 4, (elen,), generator=fg, device=device)]
                vals = None
            if kind == "tandem":
                arrays = max(1, copies // 1500)
                for _ in range(arrays):
                    units_n = int(torch.randint(300, 3000, (1,), generator=g, device=device).item())
                    L_arr = units_n * elen
                    if L_arr >= n - 1:
                        continue
                    p = int(torch.randint(0, n - L_arr, (1,), generator=g, device=device).item())
                    arr = cons.repeat(units_n)
                    div = d_lo + (d_hi - d_lo) * float(torch.rand(1, generator=g, device=device).item())
                    m = torch.rand(L_arr, generator=g, device=device) < div
                    arr = torch.where(m, acgt[torch.randint(0, 4, (L_arr,), generator=g, device=device)], arr)
                    t[p:p + L_arr] = arr
                    planted += L_arr
                continue
            step = max(1, (1 << 26) // elen)
            for a in range(0, copies, step):
                b = min(copies, a + step)
                k = b - a
                v = vals[a:b] if vals is not None else cons[None, :].expand(k, elen).clone()
                div = d_lo + (d_hi - d_lo) * torch.rand((k, 1), generator=g, device=device)
                m = torch.rand((k, elen), generator=g, device=device) < div
                v = torch.where(m, acgt[torch.randint(0, 4, (k, elen), generator=g, device=device)], v)
                rc = torch.rand(k, generator=g, device=device) < 0.5
                v = torch.where(rc[:, None], comp[v.flip(1).long()], v)
                p = torch.randint(0, n - elen, (k,), generator=g, device=device)
                idx = p[:, None] + torch.arange(elen, device=device)[None, :]
                t[idx.reshape(-1)] = v.reshape(-1)
                planted += k * elen
        ch[:] = t.cpu().numpy()
        del t
    torch.cuda.empty_cache()
    return planted / total


def make_genome(cfg, repeats=0, grch38_like=False):
    from bitmapperbs_amd import synth
    names, chroms = synth.make_genome(cfg["genome"], cfg["n_chrom"], seed=20240229)
    if grch38_like:
        frac = plant_repeat_families(chroms)
        sys.stderr.write("[bench] GRCh38-like repeat families planted: %.1f %% of the bases\n" % (100 * frac))
    if repeats:
        # interspersed-repeat stress (Alu-like): seeds inside a copy hit hundreds of places, candidate lists get long
        rng = np.random.default_rng(77)
        for fam in range(5):
            el = synth._ACGT[rng.integers(0, 4, 300)]
            for _ in range(repeats // 5):
                ch = chroms[int(rng.integers(0, len(chroms)))]
                p = int(rng.integers(0, ch.size - 300))
                e = el.copy()
                m = rng.random(300) < rng.uniform(0.02, 0.08)
                e[m] = synth._ACGT[rng.integers(0, 4, int(m.sum()))]
                ch[p:p + 300] = synth.revcomp(e) if rng.random() < 0.5 else e
    return names, chroms


def load_fasta(path):
    """-> (names, chroms): the sequences of a FASTA file as upper-case uint8 arrays with everything outside ACGT REMOVED (reads are
    drawn from these; the index is built from the file itself, where the builder puts a fixed pseudo-random base for such letters)"""
    data = np.fromfile(path, dtype=np.uint8)
    nl = np.flatnonzero(data == 10)
    starts = np.concatenate([[0], nl[:-1] + 1]) if nl.size else np.array([0])
    heads = starts[data[starts] == ord(">")]
    names, chroms = [], []
    for i, h in enumerate(heads):
        e = int(nl[np.searchsorted(nl, h)])
        stop = int(heads[i + 1]) if i + 1 < heads.size else data.size
        names.append(bytes(data[h + 1:e]).split()[0].decode())
        seg = data[e + 1:stop] & 0xDF                       # upper case (newline 10 -> 10 & 0xDF = 10: dropped below with the rest)
        keep = (seg == 65) | (seg == 67) | (seg == 71) | (seg == 84)
        chroms.append(np.ascontiguousarray(seg[keep]))
    return names, chroms


def ensure_index(args, cfg, rank, local, world, dist, repeats=0, grch38_like=False):
    """-> (fasta/prefix path, names, chroms, seconds spent building or 0 when cached)"""
    from bitmapperbs_amd import synth, mapper
    if getattr(args, "fasta", None) and not repeats and not grch38_like and cfg["genome"] >= 1_000_000_000:
        # a real assembly supplied at run time takes the place of the synthetic genome of the GRCh38-size configurations
        src = os.path.abspath(args.fasta)
        wd = os.path.join(args.workdir, "fa_%s_%d" % (os.path.basename(src).replace(".", "_"), os.path.getsize(src)))
        fa = os.path.join(wd, "g.fa")
        built = 0.0
        if rank == 0:
            os.makedirs(wd, exist_ok=True)
            if not os.path.exists(fa):
                os.symlink(src, fa)
            if not os.path.exists(fa + ".index.bs.index.sa.ok"):
                t = time.time()
                mapper.Index.build(fa, fa, threads=min(64, os.cpu_count() or 1), device=local)
                open(fa + ".index.bs.index.sa.ok", "w").write("ok\n")
                built = time.time() - t
                sys.stderr.write("[bench] index of %s built in %.1f s (GPU builder)\n" % (src, built))
        if world > 1:
            dist.barrier()
        names, chroms = load_fasta(fa)
        cfg["genome"] = int(sum(c.size for c in chroms)); cfg["n_chrom"] = len(chroms)
        cfg["label"] += " [real assembly: %s, %d sequences, %d ACGT bases]" % (os.path.basename(src), len(chroms), cfg["genome"])
        cfg["real_fasta"] = True
        return fa, names, chroms, built
    wd = os.path.join(args.workdir, "g%d_c%d%s%s" % (cfg["genome"], cfg["n_chrom"], "_r%d" % repeats if repeats else "", "_hg" if grch38_like else ""))
    fa = os.path.join(wd, "g.fa")
    names, chroms = make_genome(cfg, repeats, grch38_like)
    built = 0.0
    if rank == 0:
        os.makedirs(wd, exist_ok=True)
        if not os.path.exists(fa + ".index.bs.index.sa.ok"):
            t = time.time()
            synth.write_fasta(fa, names, chroms)
            threads = min(64, os.cpu_count() or 1)
            if args.host_index:
                mapper.Index.build(fa, fa, threads=threads)
            else:
                mapper.Index.build(fa, fa, threads=threads, device=local)
            open(fa + ".index.bs.index.sa.ok", "w").write("ok\n")
            built = time.time() - t
            sys.stderr.write("[bench] %d bp index built in %.1f s (%s builder, FASTA write included)\n" %
                             (cfg["genome"], built, "host" if args.host_index else "GPU"))
    if world > 1:
        dist.barrier()
    return fa, names, chroms, built


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


def cpu_baseline(args, cfg, fa, host_reads, L, tag="cpu_sample"):
    """Time the reference's own CPU path (oracle/_ref/bitmapperBS, kind 'reference') -- or, when it has not been built, the
    scalar restatement (oracle/liboracle.so, kind 'port') -- on a bounded sample of the step's first launch.
    host_reads = (seq, qual) or (seq1, qual1, seq2, qual2) numpy arrays."""
    n = host_reads[0].shape[0]
    pe = cfg["pe"]
    ref = os.path.join(ROOT, "oracle", "_ref", "bitmapperBS")
    unit = "pairs" if pe else "reads"
    sample = "first %d %s of the step's first launch (%d bp %s, -e %.2f%s)" % (
        n, unit, L, "PE" if pe else "SE", cfg["e"], ", --sensitive" if cfg["sensitive"] else "")
    if os.path.exists(ref):
        cores = args.cpu_threads or min(32, os.cpu_count() or 1)
        out = os.path.join(args.workdir, tag + ".sam")
        cmd = [ref, "--search", fa]
        if pe:
            f1 = os.path.join(args.workdir, tag + "_1.fq"); f2 = os.path.join(args.workdir, tag + "_2.fq")
            write_fastq_sample(f1, host_reads[0], host_reads[1], L)
            write_fastq_sample(f2, host_reads[2], host_reads[3], L)
            cmd += ["--seq1", f1, "--seq2", f2]
            if cfg["sensitive"]:
                cmd += ["--sensitive"]
        else:
            fq = os.path.join(args.workdir, tag + ".fq")
            write_fastq_sample(fq, host_reads[0], host_reads[1], L)
            cmd += ["--seq", fq]
        cmd += ["-e", str(cfg["e"]), "-t", str(cores), "-o", out]
        t = time.time()
        p = subprocess.run(cmd, capture_output=True, text=True, cwd=args.workdir)
        wall = time.time() - t
        secs = load = None
        for line in p.stderr.splitlines():
            if line.strip().startswith("Total:"):
                load = float(line.split()[1]); secs = float(line.split()[2])   # "Total: <load s> <map s>" (Bitmapper_main.cpp:262)
        if p.returncode == 0 and secs and secs > 0:
            nr = n * (2 if pe else 1)
            return {"value": round(nr / secs / 1e6, 4), "unit": "Mreads/s", "cores": cores, "cpu_quota_cores": cpu_quota_cores(), "kind": "reference",
                    "sample": sample + ", bitmapperBS -t %d, mapping seconds as printed by main (%.1f s; index load %.1f s; wall %.1f s)" % (
                        cores, secs, load, wall)}, out
        sys.stderr.write("[bench] reference run failed (rc %d): %s\n" % (p.returncode, p.stderr[-400:]))
    if pe:
        return None, None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    oix = orc.OrcIndex(fa)
    m = min(n, 300_000)
    t = time.time()
    oix.map_se(orc.params(e_f=cfg["e"]), host_reads[0][:m], host_reads[1][:m], L)
    dt = time.time() - t
    return {"value": round(m / dt / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
            "sample": "first %d reads of the step's first launch, scalar CPU restatement (oracle/)" % m}, None


def sam_identity(ix, m, job, host, L, pe, ref_sam, nchk):
    """the GPU records of the first nchk reads / pairs of the job's first launch print the same SAM lines as the reference wrote for
    them (ref_sam: the reference's output for a sample that starts with those reads) -> (identical, lines compared)"""
    from bitmapperbs_amd import mapper, capi
    job.launch(0); m.sync()
    nrec = nchk * (2 if pe else 1)
    res_h = job.res_d[:nrec].cpu().numpy().view(capi.RESULT_DTYPE).reshape(-1)
    cig_h = job.cig_d.cpu().numpy().view(np.uint32)
    names_s = [b"s%08d" % i for i in range(nchk)]
    if pe:
        mine = set(mapper.sam_lines_pe(ix, names_s, names_s, host[0][:nchk], host[1][:nchk], host[2][:nchk], host[3][:nchk], L, res_h, cig_h))
    else:
        mine = set(mapper.sam_lines_se(ix, names_s, host[0][:nchk], host[1][:nchk], L, res_h, cig_h))
    with open(ref_sam) as f:
        theirs = set(x for x in f if not x.startswith("@") and int(x[1:x.index("\t")]) < nchk)
    return mine == theirs, len(theirs)


# ---- roofline accounting --------------------------------------------------------------------------------------------------------
# FETCH_SIZE correction per kernel (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports exactly half the bytes of a wide
# coalesced streaming read, 16 B per lane; WRITE_SIZE is exact for 16-byte streaming stores).  "x2": the kernel's reads are such
# streams -- whole rows / text fetched 16 bytes per lane by consecutive lanes; "raw": per-lane gathers of 4-16 bytes at unrelated
# addresses (index walks, candidate lists), for which tools/gather_bench with a known byte count reads 1.0x.
FETCH_RULE = {
    "k_pe_prepare": "x2", "k_pe_prepare_p": "x2", "k_pack_rows": "x2", "k_fastq_rows": "x2", "k_seed_decide_p": "x2 on the staged rows, raw on the window gathers: reported raw (lower bound)",
    "k_finalize_pe": "x2", "k_finalize": "x2", "k_fq_count": "x2", "k_fq_lines": "x2", "k_sam_write": "raw (byte gathers from the text)",
}


# kernel name in the rocprofv3 CSV -> the label its launches carry in the HIP-event profile (and in the algorithmic-byte tables)
PROF_NAME = {"k_pe_prepare_p": "k_pe_prepare", "k_seed_decide_p": "k_seed_decide", "k_align_sw2": "k_align_sw", "k_align_ungapped_p": "k_align_ungapped"}


def fetch_rule(kernel):
    r = FETCH_RULE.get(kernel.split("<")[0], "raw")
    return r, (2.0 if r == "x2" else 1.0)


def event_label(csv_name):
    """kernel name of a rocprofv3 CSV row -> the label its launches carry in the library's HIP-event profile (and in the byte tables):
    the size-class instances of the list kernels share one C++ name and differ in their template arguments"""
    base = csv_name.split("<")[0].split("(")[0]
    targ = csv_name.split("<")[1] if "<" in csv_name else ""
    if base in ("k_vote_long", "k_vote_pe_long"):
        wave_form = targ.startswith("256;") or targ.startswith("256,")
        return base if wave_form else base.replace("_long", "_big")
    if base == "k_pes_vote_long":
        return "k_pes_vote"
    if base == "k_pe_filter_pairs_long":
        return "k_pe_filter_pairs"
    return PROF_NAME.get(base, base)


def pmc_table(tag):
    """{event label: (fetch bytes corrected, write bytes, rule)} per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/<tag>_pmc_fetch_write.csv: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, KB units as rocprofv3 reports
    them); {} when no profile of this configuration has been committed"""
    import csv
    path = os.path.join(ROOT, "profiles", "%s_pmc_fetch_write.csv" % tag)
    out = {}
    if not tag or not os.path.exists(path):
        return out
    for row in csv.DictReader(open(path)):
        if not (row["FETCH_SIZE_KB_last_launch"] and row["WRITE_SIZE_KB_last_launch"]):
            continue
        rule, f = fetch_rule(row["kernel"].split("<")[0].split("(")[0])
        name = event_label(row["kernel"])
        lanes = int(row.get("launches_per_call") or 1)
        fe = float(row["FETCH_SIZE_KB_last_launch"]) * 1024 * f * lanes
        wr = float(row["WRITE_SIZE_KB_last_launch"]) * 1024 * lanes
        a, b, _ = out.get(name, (0.0, 0.0, rule))
        out[name] = (a + fe, b + wr, rule)
    return out


def pmc_traffic(kernel, tag):
    t = pmc_table(tag).get(kernel)
    return int(t[0] + t[1]) if t else None


def workload_tag(config, cfg, grch38_like, base_cfg=None):
    """name of the committed profile family of the workload that ACTUALLY runs: c<config> for a BASELINE configuration as it stands,
    grch38_like_{pe,se,sensitive} for the repeat-rich genome; anything else (a configuration changed by flags) has no committed PMC pass"""
    if grch38_like:
        return "grch38_like_" + ("se" if not cfg["pe"] else "sensitive" if cfg["sensitive"] else "pe")
    base = base_cfg or CONFIGS[config]
    if cfg["pe"] != base["pe"] or cfg["sensitive"] != base["sensitive"] or cfg["read_len"] != base["read_len"] or cfg["genome"] != base["genome"]:
        return None
    return "c%d" % config


def profile_tag(wtag):
    """the newest committed PMC pass of that workload: profiles/r<NN>_<wtag>_pmc_fetch_write.csv -> 'r<NN>_<wtag>' or None"""
    if not wtag:
        return None
    for rnd in range(9, 1, -1):
        t = "r%02d_%s" % (rnd, wtag)
        if os.path.exists(os.path.join(ROOT, "profiles", "%s_pmc_fetch_write.csv" % t)):
            return t
    return None


def algorithmic_bytes(cnt, nr, L, k, pe):
    """per-launch ALGORITHMIC bytes of every mapping kernel, two models side by side:
    `s8d`  = SURVEY.md section 8(d)'s terms only: 10 B per 16-mer lookup, 80 B per backward extension (two 40-B Occ blocks), 4 B
             per suffix-array read, (ceil((L+2k)/4)+1) B per fetched window, the read characters a kernel consumes (1 B each;
             8(d)'s "2L seq+qual in"), 32 B per result record;
    `own`  = the builder's model of round 1: the same plus the 16-byte seed record written per lookup and the carry /
             verdict records between the seeding kernels (traffic this design adds and has to pay for)."""
    win = (L + 2 * k + 3) // 4 + 1
    def seed_idx(c):
        return 10 * c["n_hash"] + 80 * c["n_ext"] + 4 * c["n_sa"]
    def seed_chars(c):
        return 16 * c["n_hash"] + c["n_ext"]
    s8d, own = {}, {}
    for kn in ("k_seed_first", "k_seed_second", "k_seed_extra"):
        c = cnt[kn]
        s8d[kn] = seed_idx(c) + seed_chars(c)
        own[kn] = s8d[kn] + 16 * c["n_hash"] + (14 * nr if kn == "k_seed_first" else 24 * c["n_hash"] if kn == "k_seed_second" else 0)
    s8d["k_seed_decide"] = L * nr + 4 * cnt["n_sa"] + ((L + 3) // 4 + 1) * cnt["n_ungapped"]
    own["k_seed_decide"] = s8d["k_seed_decide"] + (14 + 40) * nr
    # candidates by the kernel class that located them (device counters, round 6): lists of <= 16 (the fused kernels), 17..32 (mid),
    # 33..256 (the wave form: k_vote[_pe]_long), beyond (the block forms: k_vote[_pe]_big)
    cand = cnt["n_cand_slots"]
    c_mid, c_long, c_big = cnt.get("n_cand_mid", 0), cnt.get("n_cand_long", 0), cnt.get("n_cand_big", 0)
    c_fused = max(0, cand - c_mid - c_long - c_big)
    for kn in ("k_vote_fused", "k_vote_pe_fused"):
        s8d[kn] = 4 * c_fused; own[kn] = (4 + 16 + 4) * c_fused
    # the list kernels: 4 B per suffix-array read + the 16-byte candidate record (site, err, end / site, vote) per located candidate
    for kn, c_ in (("k_vote_mid", c_mid), ("k_vote_pe_mid", c_mid), ("k_vote_long", c_long), ("k_vote_pe_long", c_long),
                   ("k_vote_big", c_big), ("k_vote_pe_big", c_big), ("k_pes_vote", cnt.get("n_cand_reseed", 0))):
        s8d[kn] = 20 * c_; own[kn] = 24 * c_
    # filter_pairs: the 16-byte entries of both mates' lists read once (survivors written: at most as many again, not counted)
    s8d["k_pe_filter_pairs"] = 16 * cnt.get("n_pef_entries", 0); own["k_pe_filter_pairs"] = 32 * cnt.get("n_pef_entries", 0)
    # K9 / K10: the 16-byte vote record and the 8 bytes of (err, end) of every filtered candidate
    s8d["k_reduce"] = 24 * cnt["n_filter"]; own["k_reduce"] = 24 * cnt["n_filter"] + 40 * nr
    for kn in ("k_filter", "k_filter_pe_r1"):
        s8d[kn] = (win + L) * cnt["n_filter"]; own[kn] = (win + L + 24) * cnt["n_filter"]
    s8d["k_align_ungapped"] = (win + 2 * L) * cnt["n_jobs"]; own["k_align_ungapped"] = (win + 2 * L + 16) * cnt["n_jobs"]
    s8d["k_align_sw"] = (win + 2 * L) * cnt["n_sw"]; own["k_align_sw"] = (win + 2 * L + 16) * cnt["n_sw"]
    for kn in ("k_finalize", "k_finalize_pe"):
        s8d[kn] = 32 * nr; own[kn] = 32 * nr
    if pe:
        s8d["k_pe_prepare"] = 2 * L * nr; own["k_pe_prepare"] = 2 * L * nr       # both mates read, the working copy written
    return s8d, own


def native_bytes(cnt, wide):
    """the bytes THIS layout has to move for the same events (the reference's 40-byte Occ blocks and 10-byte hash entries are not
    what is in HBM here): 8 B per lookup (one outcome-table entry answers the 16-mer lookup and the first 4-5 extensions), 32 B per
    backward extension (two 16-byte Occ blocks), 4 / 8 B per suffix-array read (64-bit SA on texts >= 2^32), 16 B of packed row per
    seed start, 16 B per seed record written"""
    out = {}
    for kn in ("k_seed_first", "k_seed_second", "k_seed_extra"):
        c = cnt[kn]
        out[kn] = (8 + 16 + 16) * c["n_hash"] + 32 * c["n_ext"] + (8 if wide else 4) * c["n_sa"]
    return out


def gather_roofline(kernel, cnt, kern_ms):
    c = cnt.get(kernel)
    if not isinstance(c, dict) or kern_ms.get(kernel, 0) <= 0:
        return None
    req = c["n_hash"] + c["n_ext"] + c["n_sa"]
    ach = req / (kern_ms[kernel] * 1e-3) / 1e9
    return {"index_requests_per_launch": int(req), "achieved_Greq_s": round(ach, 2), "ceiling_Greq_s": GATHER_CEILING_GREQ,
            "frac": round(ach / GATHER_CEILING_GREQ, 4),
            "note": "counts index gathers only; read-character loads and result stores are further requests of the same kind"}


def roofline_block(kern_ms, cnt, nr, L, k, pe, genome, wtag, single=None):
    """the `roofline` object of a bench line for the workload that ran: dominant kernel by HIP-event time, SURVEY 8(d) bytes of the
    last launch over that time, PMC traffic from the committed profile of THIS workload (wtag: workload_tag) -> (dict, s8d, tag)"""
    mapping = {kn: v for kn, v in kern_ms.items() if kn.startswith("k_")}
    dom = max(mapping, key=mapping.get) if mapping else "k_seed_first"
    s8d, own = algorithmic_bytes(cnt, nr, L, k, pe)
    nat = native_bytes(cnt, 2 * genome + 1 >= (1 << 32))
    tag = profile_tag(wtag)

    def rl(model):
        b = model.get(dom, 0)
        a = b / (kern_ms[dom] * 1e-3) / 1e9 if kern_ms.get(dom, 0) > 0 else 0.0
        return b, a
    b8, a8 = rl(s8d)
    bo, ao = rl(own)
    traffic = pmc_traffic(dom, tag) if tag else None
    out = {"bound": "hbm", "kernel": dom, "achieved": round(a8, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(a8 / HBM_PEAK_GBS, 6), "traffic": traffic,
           "algorithmic_bytes_per_launch": int(b8), "avg_launch_ms": round(kern_ms.get(dom, 0.0), 4),
           "model": "SURVEY 8(d) terms: 10 B/16-mer lookup, 80 B/extension, 4 B/SA read, window bytes, read characters consumed, 32 B/record; "
                    "list kernels: 4 B/SA read + the 16-byte candidate record per located candidate; filter_pairs: 16 B/list entry read",
           # the builder's own model (adds the seed / carry records this design writes) next to it
           "builder_model": {"algorithmic_bytes_per_launch": int(bo), "achieved": round(ao, 3), "frac": round(ao / HBM_PEAK_GBS, 6)},
           # ... and what THIS layout has to move for the same events (16-byte Occ blocks, 8-byte outcome-table entries)
           "design_native_model": ({"algorithmic_bytes_per_launch": int(nat[dom]), "achieved": round(nat[dom] / (kern_ms[dom] * 1e-3) / 1e9, 3),
                                    "frac": round(nat[dom] / (kern_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                    "model": "8 B/lookup (outcome-table entry) + 16 B packed row + 16 B seed record per seed start, 32 B/extension, 4|8 B/SA read"}
                                   if dom in nat and kern_ms.get(dom, 0) > 0 else None),
           "traffic_rule": (pmc_table(tag).get(dom) or (0, 0, fetch_rule(dom)[0]))[2], "traffic_profile": tag,
           # the same kernel timed ALONE (one lane, same launches, outside the timed region): what round 2's figure was
           "single_lane": ({"avg_launch_ms": round(single["kernels_ms"].get(dom, 0.0), 4),
                            "achieved": round(s8d.get(dom, 0) / (single["kernels_ms"][dom] * 1e-3) / 1e9, 3),
                            "frac": round(s8d.get(dom, 0) / (single["kernels_ms"][dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                            "value_M_reads_s": single["value"], "ms_per_launch": single["ms_per_launch"]}
                           if single and single["kernels_ms"].get(dom, 0) > 0 else None),
           "traffic_over_algorithmic": round(traffic / b8, 3) if traffic and b8 else None,
           # the same kernel against the bound that applies to an index walk: divergent gather requests/s (index gathers only: table
           # lookups, Occ blocks, SA rows -- ONE definition, also in DESIGN.md section 3)
           "gather": gather_roofline(dom, cnt, kern_ms),
           "sectors": ({"traffic_GBps": round(traffic / (kern_ms[dom] * 1e-3) / 1e9, 1), "random_sector_ceiling_GBps": SECTOR_CEILING_GBS,
                        "frac": round(traffic / (kern_ms[dom] * 1e-3) / 1e9 / SECTOR_CEILING_GBS, 4)} if traffic else None)}
    # the same bytes over the WHOLE launch: with several lanes a kernel's event time includes what the other lanes ran beside it (three
    # lanes since round 6: the dominant kernel's figure above fell from 0.16 to 0.13 while the job kept its rate) -- all kernels' 8(d)
    # bytes of a launch over the sum of their event times divided by the lanes that overlap is not measurable from events alone, so
    # the launch's wall time is used by the caller (whole_launch is filled in there)
    out["whole_launch"] = {"algorithmic_bytes_per_launch": int(sum(v for kn, v in s8d.items() if kern_ms.get(kn, 0) > 0))}
    return out, s8d, tag


def kernels_traffic_table(tag, s8d):
    """PMC traffic per launch of every mapping kernel from the committed profile of this workload next to its 8(d) bytes"""
    if not tag:
        return None
    return {kn: {"fetch_plus_write_bytes": int(t[0] + t[1]), "rule": t[2],
                 "over_algorithmic": (round((t[0] + t[1]) / s8d[kn], 2) if s8d.get(kn) else None)}
            for kn, t in sorted(pmc_table(tag).items()) if kn.startswith("k_")}


# ---- one measured configuration ---------------------------------------------------------------------------------------------------
class Job:
    """reads of one configuration resident in HBM + the launch closure"""

    def __init__(self, m, cfg, chroms, rank, sub, indel, qual, genome_d=None, trimmed=False):
        import torch
        from bitmapperbs_amd import gpusynth
        self.m, self.cfg = m, cfg
        L = cfg["read_len"]
        self.L = L
        self.stride = (L + 15) // 16 * 16
        self.n = cfg["units"]
        self.k = m.threshold(L)
        self.max_ops = m.max_cigar_ops(L)
        own = genome_d is None
        if own:
            genome_d = gpusynth.upload_genome(chroms)
        g, lens = genome_d
        self.batches = []
        for b in range(cfg["launches"]):
            seed = 7 + 1000 * rank + 101 * b
            if cfg["pe"]:
                self.batches.append(gpusynth.make_reads_pe(g, lens, self.n, L, self.stride, seed=seed, sub=sub, indel=indel, qual=qual))
            else:
                self.batches.append(gpusynth.make_reads_se(g, lens, self.n, L, self.stride, seed=seed, sub=sub, indel=indel, qual=qual))
        if own:
            del g, lens, genome_d
            torch.cuda.empty_cache()
        nrec = self.n * (2 if cfg["pe"] else 1)
        # a trimmed library: 70 % of the reads cut to a uniform-random length in [30, L] at their 3' end (tests/common.py:trim_fastq);
        # the rows keep their stride, the lengths travel beside them (u16: first mates, then second mates)
        self.lens = None
        if trimmed:
            g = torch.Generator(device="cuda"); g.manual_seed(4242 + rank)
            self.lens = []
            for _ in self.batches:
                ln = torch.randint(30, L + 1, (nrec,), generator=g, device="cuda")
                keep = torch.rand(nrec, generator=g, device="cuda") >= 0.7
                self.lens.append(torch.where(keep, torch.full_like(ln, L), ln).to(torch.int16))
        self.reads_per_launch = nrec
        self.cig_cap = nrec * self.max_ops
        # every launch of a step has result records and a CIGAR pool of its own: the launches of a step are in flight together (two
        # lanes, up to eight calls each), and a shared pair of buffers would be written by all of them at once
        self.res_all = [torch.empty((nrec, 32), dtype=torch.uint8, device="cuda") for _ in self.batches]
        self.cig_all = [torch.empty((self.cig_cap,), dtype=torch.int32, device="cuda") for _ in self.batches]
        self.res_d, self.cig_d = self.res_all[0], self.cig_all[0]            # (launch 0's: what the identity checks read)
        torch.cuda.synchronize()

    def launch(self, b):
        t = self.batches[b]
        res_d, cig_d = self.res_all[b], self.cig_all[b]
        if self.lens is not None:
            from bitmapperbs_amd import capi
            lib = capi.lib()
            if self.cfg["pe"]:
                rc = lib.bmbs_map_pe_var_device(self.m._ctx, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), self.lens[b].data_ptr(),
                                                self.L, self.stride, self.n, res_d.data_ptr(), cig_d.data_ptr(), self.cig_cap)
            else:
                rc = lib.bmbs_map_se_var_device(self.m._ctx, t[0].data_ptr(), t[1].data_ptr(), self.lens[b].data_ptr(), self.L, self.stride, self.n,
                                                res_d.data_ptr(), cig_d.data_ptr(), self.cig_cap)
            self.m._chk(rc)
        elif self.cfg["pe"]:
            self.m.map_pe_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), self.L, self.stride, self.n,
                                 res_d.data_ptr(), cig_d.data_ptr(), self.cig_cap)
        else:
            self.m.map_se_device(t[0].data_ptr(), t[1].data_ptr(), self.L, self.stride, self.n, res_d.data_ptr(),
                                 cig_d.data_ptr(), self.cig_cap)

    def step(self):
        # nothing waits inside a step: the calls go to the context's lanes one behind the other (per-kernel HIP-event times
        # are summed by the library as the calls complete: Mapper.profile_total)
        for b in range(len(self.batches)):
            self.launch(b)
        self.m.sync()

    def reads_per_step(self):
        return self.reads_per_launch * len(self.batches)


def timed(job, steps, warmup, min_seconds, world, dist, torch, cdev="cuda"):
    """W warm-up steps, then K timed steps between barrier + synchronize; a step is repeated `passes` whole times when K steps
    would take less than min_seconds.  -> (seconds, passes, per-launch kernel ms averages)"""
    t_w = None
    # (the FIRST step of a context allocates its work buffers and learns its capacities -- several times a settled step: the length of
    # a step is read off a later one, so a lone warm-up step is followed by a second)
    for _ in range(max(2, warmup)):
        torch.cuda.synchronize()
        t = time.perf_counter()
        job.step()
        torch.cuda.synchronize()
        t_w = time.perf_counter() - t
    passes = 1
    if min_seconds > 0 and t_w * steps < min_seconds:
        passes = int(np.ceil(min_seconds / (t_w * steps)))
    if world > 1:
        pt = torch.tensor([passes], dtype=torch.int64, device=cdev)
        dist.all_reduce(pt, op=dist.ReduceOp.MAX)
        passes = int(pt.item())
    job.m.reset_stats()
    job.m.profile_reset()
    job.m.sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for _ in range(passes):
            job.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    launches = steps * passes * len(job.batches)
    # HIP events around every kernel of every call of the timed region (recorded on the lanes' own streams), summed per kernel name
    # over the chunks of a call, averaged per launch = per API call
    kern_ms, _calls = job.m.profile_total()
    for kname in kern_ms:
        kern_ms[kname] /= max(1, launches)
    return dt, passes, kern_ms


def two_context_rate(m, ix, job, cfg, local, torch, steps=6, n_ctx=2):
    """the main configuration's launches dealt to TWO contexts on one attached index (bmbs_index_share: own stream and work buffers
    each), driven by two host threads -- how bmbs_search runs a device by default.  Kernels of different launches overlap (the DP
    and Myers kernels are issue-bound, the seeding kernels wait on memory), and the host round trips of one call hide behind the
    other.  A secondary key: `value` and the roofline are the single-context numbers."""
    import threading
    from bitmapperbs_amd import mapper
    extra = [mapper.Mapper(ix, device=local, share=m, e_f=cfg["e"], sensitive=1 if cfg["sensitive"] else 0) for _ in range(n_ctx - 1)]
    # (every further context brings work buffers of its own for a 10 M-pair call; the result buffers are the job's per-launch ones, and
    # those -- and the input launches -- beyond three are given back first: index 170 GB + trigram table 28 GB + inputs 32 GB leave no room for both)
    del job.res_all[3:], job.cig_all[3:], job.batches[3:]
    torch.cuda.empty_cache()
    free_b, _total_b = torch.cuda.mem_get_info()
    need_b = 38e9 * (n_ctx - 1)                      # work buffers of a further context for a 10 M-pair call (two lanes): 30-38 GB measured
    if free_b < need_b:
        return {"skipped": "%.0f GB of HBM free, a further context's work buffers for calls of this size need 30-38 GB each (index + full suffix array + "
                           "outcome table hold ~140 GB at this genome size, the trigram table 28 GB more once a repeat-rich input has asked for it); bmbs_search runs its contexts on 0.5 M-pair batches" % (free_b / 1e9)}
    while len(job.res_all) < n_ctx:
        job.res_all.append(torch.empty_like(job.res_d)); job.cig_all.append(torch.empty_like(job.cig_d))
    ctxs = [(m, job.res_d, job.cig_d)] + [(x, job.res_all[1 + i], job.cig_all[1 + i]) for i, x in enumerate(extra)]

    def launch(c, b):
        mm, res, cig = c
        t = job.batches[b]
        if cfg["pe"]:
            mm.map_pe_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), job.L, job.stride, job.n, res.data_ptr(), cig.data_ptr(), job.cig_cap)
        else:
            mm.map_se_device(t[0].data_ptr(), t[1].data_ptr(), job.L, job.stride, job.n, res.data_ptr(), cig.data_ptr(), job.cig_cap)
    for c in ctxs:                                   # the second context allocates its work buffers here, not in the timed part
        launch(c, 0); c[0].sync()
    torch.cuda.synchronize()
    total = steps * len(job.batches)
    nxt = [0]
    lock = threading.Lock()

    def worker(c):
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= total:
                break
            launch(c, i % len(job.batches))
        c[0].sync()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(c,)) for c in ctxs]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for x in extra:
        x.close()
    del ctxs
    torch.cuda.empty_cache()
    return {"what": "main configuration, launches dealt to %d contexts sharing one index (a stream, work buffers and a host thread each; bmbs_search runs --contexts 4 per device by default)" % n_ctx,
            "value": round(job.reads_per_launch * total / dt / 1e6, 2), "unit": "Mreads/s", "timed_s": round(dt, 3),
            "ms_per_launch": round(dt / total * 1e3, 3)}


def secondary(args, label, cfg, rank, local, env=None, sub=None, qual="const", repeats=0, steps=3, trimmed=False, grch38_like=False, ref_check=0,
              roofline=False):
    """a short secondary measurement on its own index / mapper: -> dict(value, ms_per_launch, ...)"""
    import torch
    from bitmapperbs_amd import mapper
    old = {}
    for k_, v_ in (env or {}).items():
        old[k_] = os.environ.get(k_); os.environ[k_] = v_
    try:
        fa, names, chroms, built = ensure_index(args, cfg, rank, local, 1, None, repeats=repeats, grch38_like=grch38_like)
        ix = mapper.Index(fa)
        m = mapper.Mapper(ix, device=local, e_f=cfg["e"], sensitive=1 if cfg["sensitive"] else 0)
        job = Job(m, cfg, chroms, rank, args.sub if sub is None else sub, args.indel, qual, trimmed=trimmed)
        # every key over >= 1 s of mapping (passes from a settled step: timed())
        dt, passes, kern_ms = timed(job, steps, 1, SECONDARY_MIN_S, 1, None, torch)
        nreads = job.reads_per_step() * steps * passes
        top = sorted(((v, k_) for k_, v in kern_ms.items() if k_.startswith("k_")), reverse=True)[:3]
        st = m.stats()
        out = {"what": label, "value": round(nreads / dt / 1e6, 2), "unit": "Mreads/s", "timed_s": round(dt, 3),
               "launches_timed": steps * passes * len(job.batches),
               "ms_per_launch": round(dt / (steps * passes * len(job.batches)) * 1e3, 3),
               "top_kernels_ms": {k_: round(v, 3) for v, k_ in top},
               "mapstats": {"unique_pct": round(100.0 * float(st[1]) / max(1.0, float(st[0])), 2), "ambiguous_pct": round(100.0 * float(st[2]) / max(1.0, float(st[0])), 2)}}
        if roofline:
            # the key's own roofline: its dominant kernel, the event counters of its last launch, the committed PMC pass of THIS workload
            cnt = m.counters()
            L_ = cfg["read_len"]
            roof, s8d, tag = roofline_block(kern_ms, cnt, job.reads_per_launch, L_, m.threshold(L_), cfg["pe"], cfg["genome"],
                                            workload_tag(2, cfg, grch38_like, base_cfg=cfg) if grch38_like else None)
            wl_ms = dt / (steps * passes * len(job.batches)) * 1e3
            roof["whole_launch"].update({"ms_per_launch": round(wl_ms, 4), "achieved": round(roof["whole_launch"]["algorithmic_bytes_per_launch"] / (wl_ms * 1e-3) / 1e9, 3),
                                         "frac": round(roof["whole_launch"]["algorithmic_bytes_per_launch"] / (wl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)})
            out["roofline"] = roof
            out["kernels_ms_per_launch"] = {a: round(b, 4) for a, b in kern_ms.items() if a.startswith("k_")}
            out["kernels_traffic"] = kernels_traffic_table(tag, s8d)
            out["counters_last_launch"] = {a: b for a, b in cnt.items() if not isinstance(b, dict)}
        if ref_check and not args.no_cpu:
            # the reference binary on the first ref_check reads / pairs of this key's first launch: the same SAM lines?
            try:
                L = cfg["read_len"]
                host = [x[:ref_check].cpu().numpy() for x in job.batches[0]]
                cb, ref_sam = cpu_baseline(args, cfg, fa, host, L, tag="chk_sample")
                if ref_sam:
                    same, nl = sam_identity(ix, m, job, host, L, cfg["pe"], ref_sam, ref_check)
                    out["sample_sam_identical_to_reference"] = same
                    out["sample_sam_lines_compared"] = nl
                    out["reference_on_the_sample"] = cb
                for f_ in ("chk_sample.sam", "chk_sample_1.fq", "chk_sample_2.fq", "chk_sample.fq"):
                    if os.path.exists(os.path.join(args.workdir, f_)):
                        os.unlink(os.path.join(args.workdir, f_))
            except Exception as ex:
                out["sample_sam_identical_to_reference"] = "error: %r" % (ex,)
        m.close(); ix.close()
        del job
        torch.cuda.empty_cache()
        return out
    finally:
        for k_, v_ in old.items():
            if v_ is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v_


def host_buffer_rate(m, job, torch, jobs_per_read=0.0, packed=False):
    """PCIe-inclusive rate through the host-pointer entry points: page-locked host buffers in, records and CIGAR pool back out,
    copy-in -> kernels -> copy-out of different chunks overlapped on one context (never `value`).  packed: bmbs_map_*_packed -- the
    sequences as 2 bits per base + an 'N' plane (bmbs_pack_rows, timed apart), the qualities as bytes."""
    import ctypes as C
    from bitmapperbs_amd import capi
    lib = capi.lib()
    n = min(job.n, 4_000_000)       # (r6: 4 M units per call; the first upload and the last chunk's kernels and download hide behind nothing -- at 2 M
    host = [x[:n].cpu().numpy() for x in job.batches[0]]     # units they were a seventh of the call: 186-192 against 212 M reads/s, same box)
    nbytes = host[0].nbytes
    L, stride, pe = job.L, job.stride, job.cfg["pe"]
    pwords = (L + 31) // 32 + (L + 63) // 64
    pin = []
    pack_s = None
    for i, h in enumerate(host):
        if packed and i % 2 == 0:                  # sequence rows -> packed rows, straight into page-locked memory
            p_ = lib.bmbs_host_alloc(n * pwords * 8)
            t0 = time.perf_counter()
            rc = lib.bmbs_pack_rows(h.ctypes.data, L, stride, n, None, p_, pwords, 16, None)
            pack_s = (pack_s or 0.0) + time.perf_counter() - t0
            if rc:
                raise RuntimeError("bmbs_pack_rows: %d" % rc)
        else:
            p_ = lib.bmbs_host_alloc(nbytes)
            C.memmove(p_, h.ctypes.data, nbytes)
        pin.append(p_)
    nrec = n * (2 if pe else 1)
    cap = nrec * job.max_ops
    res = lib.bmbs_host_alloc(nrec * 32); pool = lib.bmbs_host_alloc(cap * 4)
    used = C.c_int64(0)
    def call():
        if packed and pe:
            rc = lib.bmbs_map_pe_packed(m._ctx, pin[0], pin[2], pwords, pin[1], pin[3], None, None, L, stride, n, res, pool, cap, C.byref(used))
        elif packed:
            rc = lib.bmbs_map_se_packed(m._ctx, pin[0], pwords, pin[1], None, L, stride, n, res, pool, cap, C.byref(used))
        elif pe:
            rc = lib.bmbs_map_pe(m._ctx, pin[0], pin[1], pin[2], pin[3], L, stride, n, res, pool, cap, C.byref(used))
        else:
            rc = lib.bmbs_map_se(m._ctx, pin[0], pin[1], L, stride, n, res, pool, cap, C.byref(used))
        if rc:
            raise RuntimeError(lib.bmbs_last_error(m._ctx).decode())
    call()
    first = np.ctypeslib.as_array(C.cast(res, C.POINTER(C.c_uint8)), shape=(nrec * 32,)).copy()
    reps = 7                        # (r6: each call timed, the MEDIAN call reported with the spread beside it: a 45 ms call now and then
    per_call = []                   # loses a fifth of its rate to a host thread scheduled late, and a mean of three moved by 10 % with it)
    for _ in range(reps):
        t = time.perf_counter()
        call()
        per_call.append(time.perf_counter() - t)
    dt = float(np.median(per_call)) * reps
    again = np.ctypeslib.as_array(C.cast(res, C.POINTER(C.c_uint8)), shape=(nrec * 32,))
    same = bool((first == again).all())
    for p_ in pin + [res, pool]:
        lib.bmbs_host_free(p_)
    up = ((n * pwords * 8 + nbytes) * (2 if pe else 1) if packed else nbytes * len(pin)) * reps
    # what the library copies back: a 32-byte record per read and the CIGAR slots of the DP jobs a chunk produced (max_ops x 4 bytes
    # per job; `used` is the EXTENT of the host pool that was written to, not the bytes moved)
    down = nrec * 32 * reps + int(jobs_per_read * nrec) * job.max_ops * 4 * reps
    LINK = 56.0           # GB/s one direction, page-locked, measured on these boxes (tools/pcie_probe; PCIe Gen5 x16 spec 63)
    out = {"what": "bmbs_map_%s%s on page-locked HOST buffers, %d %s per call: the call is cut into chunks of n/8 (250 k .. 500 k) units dealt to the context's lanes, "
                   "uploads, kernels and downloads of different chunks overlap (copies on streams that carry no kernel); PCIe-inclusive, never `value`" % (
                "pe" if pe else "se", "_packed" if packed else "", n, "pairs" if pe else "reads"),
           "value": round(nrec * reps / dt / 1e6, 2), "unit": "Mreads/s", "statistic": "median of %d calls" % reps,
           "calls_Mreads_s": [round(nrec / x / 1e6, 1) for x in per_call],
           "bytes_up_per_read": round(up / (nrec * reps), 1), "bytes_down_per_read": round(down / (nrec * reps), 1),
           "host_cigar_pool_extent_per_read": round(int(used.value) * 4 / nrec, 1),
           "upload_GBps": round(up / dt / 1e9, 1), "download_GBps": round(down / dt / 1e9, 1), "link_GBps_one_direction": LINK,
           "frac_of_link": round(up / dt / 1e9 / LINK, 3), "repeat_calls_identical": same}
    if packed:
        out["pack_rows_host_GBps"] = round(nbytes * (2 if pe else 1) / pack_s / 1e9, 2)
        out["pack_rows_threads"] = 16
    return out, first


def write_bgzf(path, data, level=1, threads=16):
    """bgzip's format (SAM spec 4.1): independent gzip members of <= 64 KiB of input, compressed size in a 'BC' extra field"""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    block = 65280

    def one(a):
        chunk = data[a:a + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        z = co.compress(chunk) + co.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(z) + 25) + z +
                struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    with open(path, "wb") as f, ThreadPoolExecutor(threads) as ex:
        for blk in ex.map(one, range(0, len(data), block)):
            f.write(blk)
        f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00")


def write_gzip_one_member(path, src, level=1, threads=16, piece=8 << 20):
    """ONE gzip member (one deflate stream) holding the bytes of file `src`, written the way pigz does it: pieces deflated side by side,
    each ended by a sync flush (an empty stored block: the piece ends on a byte boundary), the last one by the final block; CRC-32
    over the whole text in the trailer.  A single `gzip -1` process would take minutes for the bench's 6 GB files."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    size = os.path.getsize(src)
    offs = list(range(0, size, piece)) or [0]

    def one(a):
        with open(src, "rb") as f:
            f.seek(a)
            chunk = f.read(piece)
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        z = co.compress(chunk) + (co.flush(zlib.Z_FINISH) if a + piece >= size else co.flush(zlib.Z_SYNC_FLUSH))
        return z, chunk
    crc = 0
    with open(path, "wb") as o, ThreadPoolExecutor(threads) as ex:
        o.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x04\xff")
        for z, chunk in ex.map(one, offs):
            o.write(z)
            crc = zlib.crc32(chunk, crc)
        o.write(struct.pack("<II", crc & 0xffffffff, size & 0xffffffff))


def driver_run(drv, fa, inp, cfg, extra, env=None):
    """one bmbs_search run -> (mapping wall seconds, the driver's own busy fractions, its stage lines) from --verbose"""
    p = subprocess.run([drv, "--search", fa] + inp + ["-e", str(cfg["e"]), "--verbose"] + extra, capture_output=True, text=True, env=env)
    if p.returncode:
        raise RuntimeError(p.stderr[-300:])
    lines = [x[len("[bmbs_search] "):] for x in p.stderr.splitlines() if x.startswith("[bmbs_search]")]
    rec_line = [x for x in lines if x.startswith("records") and "mapping wall" in x][-1]
    wall = float(rec_line.split("mapping wall")[1].split("s")[0])
    records = int(rec_line.split()[1])            # what the driver mapped (reads, or pairs): --loop-input is not taken by every source kind
    busy = {}
    for x in lines:
        if x.startswith("busy fractions"):
            t = x.split(":", 1)[1]
            for key, tag in (("link_up", "link up "), ("link_down", "link down "), ("gpu_workers", "gpu workers "), ("readers", "readers "), ("writers", "writers ")):
                busy[key] = float(t.split(tag)[1].split(",")[0].split(" ")[0])
    return wall, busy, " | ".join(lines)[:1100], records


def e2e_key(n_reads, wall, busy, stages, loops=1, records=None, pe=False):
    # the rate is over the records the DRIVER says it mapped (its --verbose line), not over what the key expected
    if records is not None:
        got = records * (2 if pe else 1)
        if got != int(n_reads):
            sys.stderr.write("[bench] the driver mapped %d reads where %d were expected (input passes %d): rate over the driver's count\n" % (got, n_reads, loops))
        n_reads = got
    d = {"value": round(n_reads / wall / 1e6, 2), "unit": "Mreads/s", "mapping_wall_s": wall, "reads": int(n_reads), "input_passes": loops, "busy": busy}
    if busy:
        d["bound"] = max(busy, key=busy.get)
    if stages:
        d["stages"] = stages
    return d


def gz_input_rate(args, drv, fa, cfg, big, inp_big, rec_bytes, loops):
    """compressed input, the SAME records for every key (the sample REP times over): `bgzf` = bgzip-style blocks, inflated on the device
    (read `loops` times over: --loop-input), `plain_gzip` = one gzip member per file, inflated by the host's block-parallel inflater"""
    out = {}
    n = os.path.getsize(big[0]) // rec_bytes * (2 if cfg["pe"] else 1)
    for label in ("bgzf", "plain_gzip"):
        gzf = []
        for src in big:
            dst = src + (".bgzf.gz" if label == "bgzf" else ".plain.gz")
            if label == "bgzf":
                with open(src, "rb") as f:
                    write_bgzf(dst, f.read())
            else:
                write_gzip_one_member(dst, src)
            gzf.append(dst)
        a = [gzf[big.index(x)] if x in big else x for x in inp_big]
        k = loops if label == "bgzf" else 1
        try:
            wall, busy, stages, recs = driver_run(drv, fa, a, cfg, ["-o", "/dev/null"] + (["-t", "32", "--loop-input", str(k)] if label == "bgzf" else []))
            out[label] = e2e_key(n * k, wall, busy, None, k, recs, cfg["pe"])
            if label == "bgzf":         # the reference's documented invocation (README.md:44,81): .fastq.gz in, --bam out
                wall, busy, stages, recs = driver_run(drv, fa, a, cfg, ["-o", "/dev/null", "--bam", "-t", "32", "--loop-input", str(k)])
                out["bgzf_in_bam_out"] = e2e_key(n * k, wall, busy, None, k, recs, cfg["pe"])
        except RuntimeError as ex:
            out[label] = {"error": str(ex)}
        for f in gzf:
            os.unlink(f)
    return out


def cpu_quota_cores():
    """cores' worth of CPU time the container grants a process group (cgroup v2 cpu.max, v1 cfs quota), None when unlimited / unknown"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except Exception:
        return None


def fs_write_ceiling(workdir, gb=6, threads=(4, 8), block=4 << 20):
    """what ONE file takes on this box's scratch file system: `threads` writers pwrite()-ing 4 MiB blocks at disjoint, interleaved offsets
    of a new file (buffered, the file growing as they go -- the overlay file system of the boxes refuses fallocate; O_DIRECT measured
    slower, profiles/r02_write_probe.txt) -> best GB/s.  The `file` key cannot be faster than this whatever the mapper does."""
    import threading
    buf = bytes(block)
    nblk = (gb << 30) // block
    best = 0.0
    for T in threads:
        path = os.path.join(workdir, "fs_probe.bin")
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        def work(t):
            for b in range(t, nblk, T):
                os.pwrite(fd, buf, b * block)
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        dt = time.perf_counter() - t0
        os.close(fd)
        os.unlink(path)
        best = max(best, nblk * block / dt / 1e9)
    return round(best, 2)


def file_to_file_rate(args, cfg, fa, L):
    """FASTQ file(s) -> SAM / BAM file through bitmapperbs_amd/bmbs_search (the drop-in driver).  Input: the cpu_baseline sample REP times
    over (20 M pairs on configs[2]), which the driver reads `input_passes` times (--loop-input, a measurement aid) so that every key
    maps for seconds, not for the half second in which the pipeline's fill and the contexts' first calls are a fifth of the wall; the
    keys that write a real file take fewer passes (the box's disk) and report the median of three runs."""
    drv = os.path.join(ROOT, "bitmapperbs_amd", "bmbs_search")
    if cfg["pe"]:
        files = [os.path.join(args.workdir, "cpu_sample_1.fq"), os.path.join(args.workdir, "cpu_sample_2.fq")]
        inp = ["--seq1", files[0], "--seq2", files[1]] + (["--sensitive"] if cfg["sensitive"] else [])
    else:
        files = [os.path.join(args.workdir, "cpu_sample.fq")]
        inp = ["--seq", files[0]]
    if not os.path.exists(drv) or not all(os.path.exists(f) for f in files):
        return None
    rec_bytes = 2 * L + 15                      # write_fastq_sample: '@s%08d\n' + L + '\n+\n' + L + '\n'
    REP = 4
    big = [f[:-3] + "_x%d.fq" % REP for f in files]
    for src, dst in zip(files, big):
        with open(dst, "wb") as o:
            for _ in range(REP):
                with open(src, "rb") as i:
                    while True:
                        blk = i.read(1 << 26)
                        if not blk:
                            break
                        o.write(blk)
    os.sync()           # the inputs just written are dirty pages: left there, they count against the writers of the runs below
    inp = [big[files.index(x)] if x in files else x for x in inp]
    n = os.path.getsize(big[0]) // rec_bytes * (2 if cfg["pe"] else 1)
    out = {}
    parts = 8
    sam = os.path.join(args.workdir, "f2f.sam")
    # (the run that writes most files goes first: 14 GB of dirty pages from an earlier run made the box throttle the next writer)
    for label, dst, extra, loops, runs in (("file_%d_parts" % parts, sam, ["--out-parts", str(parts)], 1, 3),
                                          ("null_sink", "/dev/null", [], 8, 1),
                                          ("file", sam, [], 1, 3),
                                          # --bam (the reference's documented invocation, README.md:44): records and BGZF blocks made on the device
                                          ("bam", sam, ["--bam"], 3, 1),
                                          ("bam_null_sink", "/dev/null", ["--bam"], 8, 1)):
        got = []
        try:
            for _ in range(runs):
                g_ = driver_run(drv, fa, inp, cfg, ["-o", dst, "-t", "32", "--loop-input", str(loops)] + extra)
                written = sum(os.path.getsize(f) for f in [sam] + [os.path.join(args.workdir, "f2f.sam.part%03d" % k) for k in range(parts)] if os.path.exists(f))
                got.append(g_ + (written,))
                for f in [sam] + [os.path.join(args.workdir, "f2f.sam.part%03d" % k) for k in range(parts)]:
                    if os.path.exists(f):
                        os.unlink(f)            # (the boxes' scratch disk holds one such output beside the inputs, not two)
        except RuntimeError as ex:
            return {"error": str(ex)}
        got.sort(key=lambda g: g[0])
        wall, busy, stages, recs, written = got[len(got) // 2]
        out[label] = e2e_key(n * loops, wall, busy, stages, loops, recs, cfg["pe"])
        if written:
            out[label]["written_GBps"] = round(written / wall / 1e9, 2)
        if runs > 1:
            out[label]["runs_Mreads_s"] = [round(n * loops / g[0] / 1e6, 2) for g in got]
    try:
        # the one-file sink against what one inode takes here
        out["file"]["fs_ceiling_GBps"] = fs_write_ceiling(args.workdir)
    except Exception as ex:
        out["file"]["fs_ceiling_GBps"] = repr(ex)
    try:
        out["gz_input"] = gz_input_rate(args, drv, fa, cfg, big, inp, rec_bytes, 6)
    except Exception as ex:
        out["gz_input"] = {"error": repr(ex)}
    for f in [sam] + big:
        if os.path.exists(f):
            os.unlink(f)
    out["what"] = ("bmbs_search, FASTQ -> SAM / BAM, 1 GPU, 32 host I/O threads, the cpu_baseline sample %d times over read `input_passes` times (--loop-input), "
                   "index load + attach excluded (as the reference's own 'mapping time'); newline index and SAM text / BAM blocks on the device, the host only "
                   "reads and writes; `file` = one output file, `file_%d_parts` = --out-parts %d, `null_sink` = -o /dev/null, `bam` / `bam_null_sink` = --bam; "
                   "gz_input (same records, /dev/null): `bgzf` inflated on the device, `plain_gzip` (one deflate stream per file, written pigz-style) by the "
                   "host's block-parallel inflater; `busy` = fraction of the mapping wall the link was held per direction (copies of the text calls), the "
                   "GPU worker threads were inside calls, the readers / writers were at work; `bound` = the largest of them; keys with "
                   "`runs_Mreads_s` are the median of three runs" % (REP, parts, parts))
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_children(args))
    cfg = args.cfg
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("[bench] --gpus %d but WORLD_SIZE=%d: reporting the world size the process group has\n" % (args.gpus, world))
    import torch
    dist = None
    if args.dry_run:
        # launcher / collective plumbing only (CPU tests): gloo, no device, no mapping
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        st = torch.tensor([rank + 1, 1, 0, 0, 0], dtype=torch.int64)
        dist.all_reduce(st, op=dist.ReduceOp.SUM)
        tt = torch.tensor([1.0 + rank], dtype=torch.float64)
        every = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(every, tt)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "dry-run", "n_gpus": dist.get_world_size(), "mapstats_sum": st.tolist(),
                              "per_rank_s": [float(x.item()) for x in every]}), flush=True)
        dist.destroy_process_group()
        return
    if args.same_device:
        local = 0
    cdev = "cuda" if args.dist_backend == "nccl" else "cpu"          # where the tensors of the collectives live
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
        world = dist.get_world_size()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the mapping path has no CPU fallback)")
    torch.cuda.set_device(local)
    from bitmapperbs_amd import mapper, capi

    t_all = time.time()
    fa, names, chroms, built_s = ensure_index(args, cfg, rank, local, world, dist, repeats=args.repeats, grch38_like=args.grch38_like)
    if args.grch38_like:
        cfg["label"] = cfg["label"] + " on the GRCh38-like repeat-rich genome"
    L = cfg["read_len"]
    t = time.time()
    ix = mapper.Index(fa)
    load_s = time.time() - t
    t = time.time()
    m = mapper.Mapper(ix, device=local, e_f=cfg["e"], sensitive=1 if cfg["sensitive"] else 0)
    m.sync()
    attach_s = time.time() - t
    k = m.threshold(L)
    t = time.time()
    job = Job(m, cfg, chroms, rank, args.sub, args.indel, args.qual)
    synth_s = time.time() - t
    del chroms

    dt, passes, kern_ms = timed(job, args.steps, args.warmup, args.min_seconds, world, dist, torch, cdev)
    stats = torch.from_numpy(m.stats()).to(cdev)
    # The same launches once more on ONE lane (a context of its own on the same index, BMBS_LANES=1, outside the timed region): with
    # two lanes the kernels of two chunks share the chip, so a kernel's own duration -- the roofline's denominator -- is longer than
    # it is alone although the job is faster; this gives the duration alone beside it.  Rank 0, N = 1 only.
    single = None
    if world == 1 and not args.no_single_lane and os.environ.get("BMBS_LANES", "2") != "1":
        old_l = os.environ.get("BMBS_LANES")
        os.environ["BMBS_LANES"] = "1"
        try:
            m1 = mapper.Mapper(ix, device=local, share=m, e_f=cfg["e"], sensitive=1 if cfg["sensitive"] else 0)
        finally:
            if old_l is None:
                os.environ.pop("BMBS_LANES", None)
            else:
                os.environ["BMBS_LANES"] = old_l
        keep = job.m
        job.m = m1
        dt1, p1, k1 = timed(job, 2, 1, 0.0, 1, None, torch, cdev)
        job.m = keep
        single = {"ms_per_launch": round(dt1 / (2 * p1 * len(job.batches)) * 1e3, 3), "value": round(job.reads_per_step() * 2 * p1 / dt1 / 1e6, 2), "kernels_ms": k1,
                  "counters": m1.counters()}
        m1.close()
    tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
    per_rank_s = [dt]
    if world > 1:
        # every rank's own timed seconds (for the per-GPU table of BASELINE.md section 4), then the job's: the slowest rank's
        every = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank_s = [float(x.item()) for x in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)        # the only collective on the data path's results: 5 x int64 mapstats
    dt = float(tt.item())
    stats = stats.cpu().numpy()
    cnt = m.counters()                                       # event counters of the LAST launch

    if rank == 0:
        pe = cfg["pe"]
        nr = job.reads_per_launch                            # reads seeded per launch
        reads_per_step = job.reads_per_step() * passes
        total_reads = reads_per_step * world * args.steps
        value = total_reads / dt / 1e6
        wtag = workload_tag(args.config, cfg, args.grch38_like) if not (args.repeats or cfg.get("real_fasta")) else None
        roof, s8d, tag = roofline_block(kern_ms, cnt, nr, L, k, pe, cfg["genome"], wtag, single)
        wl_ms = dt / (args.steps * passes * len(job.batches)) * 1e3          # wall per launch over the timed region
        roof["whole_launch"].update({"ms_per_launch": round(wl_ms, 4), "achieved": round(roof["whole_launch"]["algorithmic_bytes_per_launch"] / (wl_ms * 1e-3) / 1e9, 3),
                                     "frac": round(roof["whole_launch"]["algorithmic_bytes_per_launch"] / (wl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                     "what": "SURVEY 8(d) bytes of every kernel of a launch over the launch's wall time (all lanes overlapping)"})
        out = {
            "metric": "M %dbp %s reads aligned/s" % (L, "PE" if pe else "SE"), "value": round(value, 4), "unit": "Mreads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("%s; one step = %d launch(es) x %d %s%s per GPU; %d bp %d-chromosome " + ("REAL assembly (--fasta)" if cfg.get("real_fasta") else
                                    "uniform-random N-free synthetic genome (repeat-poor: NOT representative of GRCh38's repeat structure -- secondary.grch38_like is "
                                    "the same launch on a repeat-rich genome)") + ", -e %.2f (k=%d), %s, substitutions %.3f, "
                                    "indels %.4f/bp, qualities %s; inputs and results resident in HBM") % (
                                       cfg["label"], len(job.batches), job.n, "pairs" if pe else "reads",
                                       " x %d passes (timed region stretched to >= %.1f s)" % (passes, args.min_seconds) if passes > 1 else "",
                                       cfg["genome"], cfg["n_chrom"], cfg["e"], k,
                                       ("PE --sensitive" if cfg["sensitive"] else "PE fast mode") if pe else "SE", args.sub, args.indel, args.qual),
                       "reads_per_gpu_per_step": reads_per_step, "units_per_launch": job.n, "launches_per_step": len(job.batches) * passes,
                       "read_len": L, "genome_bp": cfg["genome"], "timed_s": round(dt, 3),
                       "parallelism": "pairs sharded by rank, index replicated, %s all-reduce of 5 mapstats counters" % ("RCCL" if args.dist_backend == "nccl" else "gloo")},
            "roofline": roof,
            "kernels_ms_per_launch": {a: round(b, 4) for a, b in kern_ms.items()},
            # the same table from the single-lane pass: each kernel alone on the chip (what a kernel change should be read against)
            "kernels_ms_per_launch_single_lane": ({a: round(b, 4) for a, b in single["kernels_ms"].items()} if single else None),
            "kernels_algorithmic_GBps": {kn: round(s8d[kn] / (kern_ms[kn] * 1e-3) / 1e9, 2) for kn in s8d if kern_ms.get(kn, 0) > 0},
            # PMC traffic per launch of every mapping kernel (committed profile of this command), FETCH_SIZE corrected by the rule named,
            # next to the kernel's 8(d) algorithmic bytes
            "kernels_traffic": kernels_traffic_table(tag, s8d),
            "counters_last_launch": cnt,
            # one entry per rank: what each GPU mapped per second over its own timed region (value = all reads / the slowest rank's time)
            "per_rank_Mreads_s": [round(reads_per_step * args.steps / t_ / 1e6, 2) for t_ in per_rank_s],
            "mapstats": {"reads_or_pairs": int(stats[0]), "unique": int(stats[1]), "ambiguous": int(stats[2]),
                         "unmapped": int(stats[0] - stats[1] - stats[2]), "mapped_bases": int(stats[3]), "error_bases": int(stats[4])},
            "library": {"build_id": capi.build_id(), "sources_id": capi.sources_id(), "built_from_these_sources": capi.build_id() == capi.sources_id()},
            "index": {"build_s": round(built_s, 1), "builder": "host" if args.host_index else "gpu", "files_load_s": round(load_s, 1),
                      "attach_s": round(attach_s, 1), "read_synthesis_s": round(synth_s, 1)},
        }
        if world == 1 and not args.no_cpu:
            ns = min(job.n, args.cpu_sample or (5_000_000 if pe else 8_000_000))
            host = [x[:ns].cpu().numpy() for x in job.batches[0]]
            cb, ref_sam = cpu_baseline(args, cfg, fa, host, L)
            out["cpu_baseline"] = cb
            if ref_sam:
                # the GPU records of the same sample print the same SAM lines as the reference
                nchk = min(ns, 100_000 if pe else 200_000)
                same, nl = sam_identity(ix, m, job, host, L, pe, ref_sam, nchk)
                out["sample_sam_identical_to_reference"] = same
                out["sample_sam_lines_compared"] = nl
            if not args.no_secondary:
                try:
                    jpr = float(cnt.get("n_jobs", 0)) / max(1, nr)
                    hb, rec_ascii = host_buffer_rate(m, job, torch, jobs_per_read=jpr)
                    hp, rec_packed = host_buffer_rate(m, job, torch, jobs_per_read=jpr, packed=True)
                    hp["records_identical_to_ascii_call"] = bool((rec_ascii == rec_packed).all())
                    out["e2e"] = {"host_buffers_overlapped": hb, "host_buffers_packed": hp}
                    # SURVEY 8(d) defines the metric "incl. H2D/D2H": the same mapping with inputs and results in HOST memory -- through the
                    # packed entry point (2 bits per base over the link), and through the ASCII one beside it
                    out["value_incl_pcie"] = {"value": hp["value"], "unit": "Mreads/s", "frac_of_link": hp["frac_of_link"],
                                              "statistic": hp["statistic"], "calls_Mreads_s": hp["calls_Mreads_s"],
                                              "bytes_up_per_read": hp["bytes_up_per_read"], "bytes_down_per_read": hp["bytes_down_per_read"],
                                              "entry": "bmbs_map_pe_packed" if pe else "bmbs_map_se_packed",
                                              "ascii_rows": {"value": hb["value"], "frac_of_link": hb["frac_of_link"], "bytes_up_per_read": hb["bytes_up_per_read"],
                                                             "entry": "bmbs_map_pe" if pe else "bmbs_map_se"},
                                              "records_identical_to_ascii_call": hp["records_identical_to_ascii_call"],
                                              "what": "page-locked host buffers in, records back out (e2e.host_buffers_packed / _overlapped); `value` is the device-resident rate the roofline describes"}
                except Exception as ex:
                    out["e2e"] = {"error": repr(ex)}
                try:
                    out["two_contexts"] = two_context_rate(m, ix, job, cfg, local, torch, n_ctx=2)
                except Exception as ex:
                    out["two_contexts"] = {"error": repr(ex)}
                try:
                    out["three_contexts"] = two_context_rate(m, ix, job, cfg, local, torch, n_ctx=3)
                except Exception as ex:
                    out["three_contexts"] = {"error": repr(ex)}
        else:
            out["cpu_baseline"] = None
    m.close(); ix.close()
    del job
    torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_secondary:
            # secondary keys: the configs[1] line of round 1 and the stress cases (each on its own short timed region)
            sec = {}
            try:        # the driver attaches its own copy of the index: run it now that this process has released its own
                if "e2e" in out and "error" not in out["e2e"] and out.get("cpu_baseline"):
                    out["e2e"]["file_to_file"] = file_to_file_rate(args, cfg, fa, L)
            except Exception as ex:
                out["e2e"]["file_to_file"] = {"error": repr(ex)}
            if args.no_stress:
                out["wall_s"] = round(time.time() - t_all, 1)
                print(json.dumps(out), flush=True)
                return
            c1 = dict(CONFIGS[1])
            small = dict(cfg, genome=46_000_000, n_chrom=4, launches=1, units=min(cfg["units"], 5_000_000))
            def one_key(name, *a, **kw):      # a secondary key must never lose the headline line, nor the keys behind it
                try:
                    sec[name] = secondary(args, *a, **kw)
                except Exception as ex:
                    sec[name] = {"error": repr(ex)}
            if args.config != 1:
                one_key("configs1_se_chr21", CONFIGS[1]["label"], c1, rank, local)
            one = dict(cfg, launches=1)
            one_key("random_qualities", "main configuration, Phred 2..40 uniform-random qualities", one, rank, local, qual="random")
            one_key("trimmed_library", "main configuration, 70 % of the reads trimmed to a uniform-random length in [30, L] (mates independently)", one, rank, local, trimmed=True)
            one_key("sub_5pct", "main configuration, 5 % substitutions", one, rank, local, sub=0.05)
            one_key("no_20mer_table", "main configuration, BMBS_T20=0 (16-mer table + Occ walk only)", one, rank, local, env={"BMBS_T20": "0"})
            if cfg["genome"] >= 1_000_000_000:
                one_key("grch38_like", "main configuration on a genome of the same size with GRCh38-like repeat content: ~45 % of the bases from nine "
                                               "families (Alu / MIR / L1 / LTR / DNA-transposon-like interspersed copies at 1-30 % divergence, segmental "
                                               "duplications, alpha-satellite arrays, microsatellites)", one, rank, local, grch38_like=True, ref_check=100_000, roofline=True)
                one_key("grch38_like_se", "the same GRCh38-like genome, mate 1 alone as 150 bp single-end reads, -e 0.08 (every candidate of a "
                                                  "single-end read is verified -- no mate prunes the list first)", dict(one, pe=False), rank, local, grch38_like=True, roofline=True)
                one_key("grch38_like_sensitive", "the same GRCh38-like genome, pairs in --sensitive mode (configs[3]'s launch size: 5 M pairs)",
                                                         dict(one, sensitive=True, units=min(cfg["units"], 5_000_000)), rank, local, grch38_like=True, roofline=True)
            one_key("repeats_50000", "46 Mb genome with 50 000 planted diverged 300-bp repeat copies, same mode", small, rank, local, repeats=50000)
            out["secondary"] = sec
            # the workload BASELINE configs[2]-[4] name is "vs GRCh38": the same launch on the repeat-rich genome, with its own roofline,
            # at the top level beside `value` (the secondary keys hold the details)
            for top_key, sec_key in (("value_grch38_like", "grch38_like"), ("value_grch38_like_se", "grch38_like_se"),
                                     ("value_grch38_like_sensitive", "grch38_like_sensitive")):
                g_ = sec.get(sec_key)
                if isinstance(g_, dict) and "value" in g_:
                    out[top_key] = {"value": g_["value"], "unit": g_["unit"], "timed_s": g_["timed_s"], "ms_per_launch": g_["ms_per_launch"],
                                    "roofline": g_.get("roofline"), "what": g_["what"]}
        out["wall_s"] = round(time.time() - t_all, 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
