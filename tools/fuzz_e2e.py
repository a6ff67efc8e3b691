#!/usr/bin/env python3
"""tools/fuzz_e2e.py [--trials N] [--seed S] -- file-level randomised parity against the REAL reference binary
(oracle/_ref/bitmapperBS, which travels to the GPU box): random FASTQ files (read length 30-250, trimmed records, letters outside
ACGT, lower case) and random `--search` options (-e, --mp_max/--mp_min/--np, --gap_open/--gap_extension, --min/--max, --sensitive,
--pbat, --unmapped_out, --ambiguous_out) are mapped by the reference on the host cores (-t 1: input order) and by bmbs_search on
the GPU; the SAM bodies (everything but @PG) and the --mapstats files must be the same bytes.  The oracle is not involved."""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.path.join(ROOT, "oracle", "_ref", "bitmapperBS")
DRV = os.path.join(ROOT, "bitmapperbs_amd", "bmbs_search")
ORC = os.path.join(ROOT, "oracle", "bmbs_oracle")
BOUND = os.path.join(ROOT, "oracle", "_ref", "bitmapperBS_hip")        # the reference program linked around libbmbs_hip.so (oracle/build_ref_hip.sh)
# --oracle: the second program is the oracle's command line instead of bmbs_search (no GPU needed): the same random files and options
# pin the ORACLE to the real reference; no --bam / gzipped input there (the oracle writes SAM from plain FASTQ)
USE_ORACLE = False
USE_BOUND = False


def body(path):
    return [l for l in open(path, "rb") if not l.startswith(b"@PG")]


def bam_body(path):
    """the BAM record stream: BGZF blocks inflated, header (whose text holds @PG) parsed away"""
    import gzip
    import struct
    raw = gzip.open(path, "rb").read()
    assert raw[:4] == b"BAM\1", raw[:4]
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]; o += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, o)[0]; o += 4
        refs.append(raw[o:o + l_name + 4]); o += l_name + 4
    head = [l for l in text.split(b"\n") if l and not l.startswith(b"@PG")] + refs
    recs = []
    while o < len(raw):
        bs = struct.unpack_from("<i", raw, o)[0]
        recs.append(raw[o:o + 4 + bs]); o += 4 + bs
    return head + recs


def write_fq(path, reads, lens, rng, lower, bgzf=False):
    import gzip
    seq, qual = reads["seq"], reads["qual"]
    if bgzf:
        # bgzip's block format (blocks of a drawn size; the driver inflates them on the device)
        from common import write_bgzf
        text = []
        for i in range(seq.shape[0]):
            s = seq[i, :lens[i]].tobytes()
            if lower and rng.random() < 0.3:
                s = s.lower()
            text.append(b"@r%d some comment\n" % i + s + b"\n+\n" + qual[i, :lens[i]].tobytes() + b"\n")
        write_bgzf(path, b"".join(text), block=int(rng.choice([700, 5000, 40000, 65280])), level=int(rng.choice([1, 6])))
        return
    with (gzip.open(path, "wb", compresslevel=int(rng.choice([1, 6, 9]))) if path.endswith(".gz") else open(path, "wb")) as f:
        for i in range(seq.shape[0]):
            s = seq[i, :lens[i]].tobytes()
            if lower and rng.random() < 0.3:
                s = s.lower()
            f.write(b"@r%d some comment\n" % i + s + b"\n+\n" + qual[i, :lens[i]].tobytes() + b"\n")


def draw(rng):
    mode = ["se", "se_pbat", "pe", "pes"][int(rng.integers(0, 4))]
    L = int(rng.choice([int(rng.integers(30, 70)), int(rng.integers(70, 160)), int(rng.integers(160, 251))]))
    long_reads = bool(rng.integers(0, 10) == 0)
    if long_reads:
        L = int(rng.integers(251, 999))            # one trial in ten: up to 998 bases, the longest read the reference itself handles
    mp_max = int(rng.integers(2, 9))
    goe_open = int(rng.integers(2, 9))
    opt = ["-e", str(float(rng.choice([0.02, 0.04, 0.08, 0.1, 0.12])))]
    if rng.random() < 0.6:
        opt += ["--mp_max", str(mp_max), "--mp_min", str(int(rng.integers(0, mp_max + 1))), "--np", str(int(rng.integers(0, 3)))]
    if rng.random() < 0.5:
        opt += ["--gap_open", str(goe_open), "--gap_extension", str(int(rng.integers(1, 5)))]
    if rng.random() < 0.4:
        opt += ["--unmapped_out"]
    if rng.random() < 0.4:
        opt += ["--ambiguous_out"]
    t = dict(mode=mode, L=L, n=int(rng.integers(1500, 5000)) if not long_reads else int(rng.integers(200, 800)), seed=int(rng.integers(1, 1 << 30)), sub=float(rng.choice([0.0, 0.01, 0.04, 0.07])),
             indel=float(rng.choice([0.0, 0.001, 0.003])), n_rate=float(rng.choice([0.0, 0.002, 0.02])), mixed=bool(rng.integers(0, 2)),
             lower=bool(rng.integers(0, 4) == 0), opt=opt)
    if mode in ("pe", "pes"):
        # Not drawn: pairs whose mates have threshold k = 0.  new_faster_verify_pairs (Schema.cpp:15773-15950) starts from
        # best_sum_err = 4k + 2 and, for a pair whose error sum EQUALS that start value, counts the pair without ever setting
        # best_pair_1/2_index: the reference then reads two uninitialised stack words.  Reachable only with k = 0 and both mates
        # on the 1-mismatch exit (err = 1 whatever k is); in the first 400-trial run the reference printed nothing for four such
        # pairs in one trial and died with SIGSEGV in another (profiles/r02_fuzz_e2e.txt, trials 52 and 58).  Undefined in the
        # reference, so not pinnable; the oracle and the product take indices 0, 0 and print the pair.
        shortest = max(20, L // 3) if t["mixed"] else L
        e = float(t["opt"][1])
        while int(e * shortest) < 1:
            e = round(e + 0.02, 2)
        t["opt"][1] = str(e)
        mx = int(rng.choice([400, 500, 700])) if not long_reads else L + int(rng.choice([200, 500]))
        t["opt"] += ["--min", str(int(rng.choice([0, 0, 80]))), "--max", str(mx)]
        t["ins_hi"] = max(L + 40, mx + int(rng.integers(-50, 60)))
        if mode == "pes":
            t["opt"] += ["--sensitive"]
    if mode == "se_pbat":
        t["opt"] += ["--pbat"]
    t["gz"] = bool(rng.integers(0, 4) == 0) and not USE_ORACLE            # gzipped FASTQ input ...
    t["bgzf"] = bool(t["gz"] and rng.integers(0, 2) == 0)                 # ... half of it in bgzip's block format (inflated on the device)
    if rng.integers(0, 5) == 0 and not USE_ORACLE:
        t["opt"] += ["--bam"]
    return t


def run_trial(t, env, wd):
    from bitmapperbs_amd import synth
    rng = np.random.default_rng(t["seed"])
    L, n = t["L"], t["n"]
    if t["mode"].startswith("se"):
        r = synth.make_reads_se(env["chroms"], n=n, L=L, seed=t["seed"], sub=t["sub"], indel=t["indel"], qual="random", n_rate=t["n_rate"])
        if t["mode"] == "se_pbat":
            # a PBAT library reads the strand complementary to the bisulfite-converted one: hand over the reverse complement of the
            # ordinary synthetic read so that --pbat (which maps the record's reverse complement) has something to find
            comp = np.arange(256, dtype=np.uint8); comp[ord("A")] = ord("T"); comp[ord("T")] = ord("A"); comp[ord("C")] = ord("G"); comp[ord("G")] = ord("C")
            r = dict(seq=comp[r["seq"][:, ::-1]].copy(), qual=r["qual"][:, ::-1].copy())
        lens = rng.integers(max(20, L // 3), L + 1, n) if t["mixed"] else np.full(n, L)
        fq = os.path.join(wd, "r.fq" + (".gz" if t.get("gz") else "")); write_fq(fq, r, lens, rng, t["lower"], t.get("bgzf", False))
        inp = ["--seq", fq]
    else:
        m1, m2 = synth.make_reads_pe(env["chroms"], n=n, L=L, seed=t["seed"], sub=t["sub"], indel=t["indel"], qual="random", ins_hi=t["ins_hi"])
        m2f = m2                                   # make_reads_pe returns mate 2 as sequenced, i.e. as the FASTQ file holds it
        if t["n_rate"]:
            for mm in (m1, m2f):
                pos = rng.random(mm["seq"].shape) < t["n_rate"]
                mm["seq"][pos] = np.frombuffer(b"NNNRY", dtype=np.uint8)[rng.integers(0, 5, int(pos.sum()))]
        l1 = rng.integers(max(20, L // 3), L + 1, n) if t["mixed"] else np.full(n, L)
        l2 = rng.integers(max(20, L // 3), L + 1, n) if t["mixed"] else np.full(n, L)
        f1 = os.path.join(wd, "r_1.fq" + (".gz" if t.get("gz") else "")); f2 = os.path.join(wd, "r_2.fq" + (".gz" if t.get("gz") else ""))
        write_fq(f1, m1, l1, rng, t["lower"], t.get("bgzf", False)); write_fq(f2, m2f, l2, rng, t["lower"], t.get("bgzf", False))
        inp = ["--seq1", f1, "--seq2", f2]
    outs = {}
    # the driver's own knobs vary too: batch size (records per library call), host threads, contexts per device
    drv_extra = ["-t", str(int(rng.choice([1, 3, 8, 16]))), "--batch", str(int(rng.choice([97, 333, 1777, 50000]))), "--contexts", str(int(rng.choice([1, 2, 3])))]
    # ... and the number of output parts (every part a pipeline of its own over its record range; `cat` of the parts is compared)
    n_parts = int(rng.choice([1, 1, 2, 3, 5]))
    if n_parts > 1 and not USE_ORACLE and not USE_BOUND:
        drv_extra += ["--out-parts", str(n_parts)]
    # --bound: the second program is the reference's own main / CLI / reader / writers with its mapping loops bound to the library
    # (oracle/bind_check.cpp); its batch size varies like the driver's
    bound_env = dict(os.environ, BMBS_BIND_BATCH=str(int(rng.choice([97, 333, 1777, 50000]))))
    second = (ORC, []) if USE_ORACLE else (BOUND, ["-t", "1"]) if USE_BOUND else (DRV, drv_extra)
    for who, exe, extra in (("ref", REF, ["-t", "1"]), ("gpu", second[0], second[1])):
        out = os.path.join(wd, who + ".sam"); ms = os.path.join(wd, who + ".ms")
        for f in (out, ms):
            if os.path.exists(f):
                os.unlink(f)
        head = [exe, "search", env["fa"]] if exe == ORC else [exe, "--search", env["fa"]]
        p = subprocess.run(head + inp + t["opt"] + ["-o", out, "--mapstats", ms] + extra, capture_output=True, text=True, cwd=wd,
                           env=bound_env if (who == "gpu" and USE_BOUND) else None)
        if p.returncode:
            return ["%s exit code %d: %s" % (who, p.returncode, p.stderr[-300:])], 0
        if who == "gpu" and "--out-parts" in extra:
            with open(out, "wb") as o:
                for k in range(n_parts):
                    o.write(open(out + ".part%03d" % k, "rb").read())
        outs[who] = (bam_body(out) if "--bam" in t["opt"] else body(out), open(ms).read() if os.path.exists(ms) else "")
    a, b = outs["ref"], outs["gpu"]
    bad = []
    if len(a[0]) != len(b[0]):
        bad.append("line count %d vs %d" % (len(a[0]), len(b[0])))
    for i, (x, y) in enumerate(zip(a[0], b[0])):
        if x != y:
            bad.append("line %d:\n  ref %r\n  gpu %r" % (i, x[:300], y[:300]))
            if len(bad) >= 3:
                break
    if a[1] != b[1]:
        bad.append("mapstats differ:\n%s---\n%s" % (a[1], b[1]))
    return bad, sum(1 for l in a[0] if not l.startswith(b"@")) if "--bam" not in t["opt"] else len(a[0])


def make_env(wd, big=False):
    from bitmapperbs_amd import synth
    if not USE_ORACLE:
        from bitmapperbs_amd import mapper
    from common import plant_repeats
    if big:
        # the BIG golden family's genome: 5 Mb with ~7 000 planted repeat copies (long candidate lists, vote-order ties, ambiguity)
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
        mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
        names, chroms = mg.big_genome()
    else:
        names, chroms = synth.make_genome(1_500_000, 3, seed=77)
        plant_repeats(chroms, seed=78)
    fa = os.path.join(wd, "g.fa")
    synth.write_fasta(fa, names, chroms)
    if USE_ORACLE:
        subprocess.run([ORC, "index", fa], check=True, capture_output=True)      # the oracle's own index builder (pinned to the reference's files)
    else:
        mapper.Index.build(fa, fa, threads=8)
    return dict(fa=fa, chroms=chroms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--oracle", action="store_true", help="compare the ORACLE's command line (CPU) with the reference instead of bmbs_search")
    ap.add_argument("--bound", action="store_true", help="compare oracle/_ref/bitmapperBS_hip (the reference program bound to the library: INTEGRATION.md) with the reference instead of bmbs_search")
    ap.add_argument("--big", action="store_true", help="the repeat-rich 5 Mb genome of the BIG golden family instead of the 1.5 Mb one")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz_e2e.json"))
    ap.add_argument("--only", type=int, default=-1, help="run this one trial of the sequence (the draws before it are made and dropped)")
    ap.add_argument("--keep", default="", help="directory that receives the output files of a trial that fails or raises")
    a = ap.parse_args()
    global USE_ORACLE, USE_BOUND
    USE_ORACLE = a.oracle
    USE_BOUND = a.bound
    if USE_BOUND and not os.path.exists(BOUND):
        sys.exit("oracle/_ref/bitmapperBS_hip is not here (oracle/build_ref_hip.sh builds it where /root/reference exists)")
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/bitmapperBS is not here (oracle/build_ref.sh builds it where /root/reference exists)")
    rng = np.random.default_rng(a.seed)
    fails = []
    with tempfile.TemporaryDirectory() as wd:
        env = make_env(wd, a.big)
        for i in range(a.trials):
            t = draw(rng)
            if a.only >= 0 and i != a.only:
                continue
            try:
                bad, lines = run_trial(t, env, wd)
            except Exception as ex:           # (an output file that cannot even be read: kept for the post-mortem)
                bad, lines = ["exception: %r" % (ex,)], 0
            if bad and a.keep:
                import shutil
                os.makedirs(a.keep, exist_ok=True)
                for f in os.listdir(wd):
                    if f.startswith("gpu.") or f.startswith("ref.sam"):
                        shutil.copy(os.path.join(wd, f), os.path.join(a.keep, "t%d_%s" % (i, f)))
            print("trial %3d %-7s L=%3d n=%4d mixed=%d records=%5d %s  %s" % (i, t["mode"], t["L"], t["n"], t["mixed"], lines, "SAME" if not bad else "DIFF", " ".join(t["opt"])), flush=True)
            if bad:
                print("\n".join(bad)[:1500], flush=True)
                fails.append(dict(trial=t, first=bad[:3]))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(dict(trials=a.trials, seed=a.seed, failures=fails), open(a.out, "w"), indent=1)
    print("%d trials, %d with differences" % (a.trials, len(fails)))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
