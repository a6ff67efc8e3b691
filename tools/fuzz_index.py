#!/usr/bin/env python3
"""tools/fuzz_index.py [--trials N] [--seed S] -- random FASTA files (1 to 300 sequences of 50 bp to 2 Mb, runs of N, lower case,
IUPAC letters, planted repeats and tandem arrays, random line widths, names with spaces) through the host index builder
(bmbs_index_build, pinned to the reference's own --index output by tests/test_index_build.py) and through the device builder
(bmbs_index_build_device): all six files must be the same bytes."""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SUFFIXES = ("index", "index.bs.pac", "index.bs.index", "index.bs.index.occ", "index.bs.index.bwt", "index.bs.index.sa")


def draw_fasta(rng, path):
    n_seq = int(rng.choice([1, 2, 3, int(rng.integers(4, 40)), int(rng.integers(40, 300))]))
    total = int(rng.choice([int(rng.integers(2_000, 50_000)), int(rng.integers(50_000, 500_000)), int(rng.integers(500_000, 2_500_000))]))
    cuts = np.sort(rng.integers(0, total, n_seq - 1)) if n_seq > 1 else np.array([], dtype=np.int64)
    lens = np.diff(np.concatenate([[0], cuts, [total]])).astype(np.int64)
    lens = np.maximum(lens, 50)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    width = int(rng.choice([50, 60, 70, 80, 61, 1000]))
    desc = []
    with open(path, "wb") as f:
        for i, L in enumerate(lens):
            s = acgt[rng.integers(0, 4, int(L))].copy()
            feats = []
            if rng.random() < 0.4 and L > 400:                     # run(s) of N
                for _ in range(int(rng.integers(1, 4))):
                    a = int(rng.integers(0, L - 100)); b = a + int(rng.integers(1, min(5000, L - a)))
                    s[a:b] = ord("N"); feats.append("N")
            if rng.random() < 0.3 and L > 2000:                    # tandem array / homopolymer
                u = acgt[rng.integers(0, 4, int(rng.integers(1, 9)))]
                a = int(rng.integers(0, L - 1500)); reps = int(rng.integers(50, 1400 // len(u)))
                s[a:a + reps * len(u)] = np.tile(u, reps); feats.append("tandem")
            if rng.random() < 0.3 and L > 5000:                    # dispersed repeat copies
                el = s[100:100 + int(rng.integers(100, 900))].copy()
                for _ in range(int(rng.integers(2, 30))):
                    a = int(rng.integers(0, L - len(el))); s[a:a + len(el)] = el
                feats.append("repeat")
            if rng.random() < 0.2:                                 # IUPAC letters
                pos = rng.random(int(L)) < 0.001
                s[pos] = np.frombuffer(b"RYKMSWBDHV", dtype=np.uint8)[rng.integers(0, 10, int(pos.sum()))]; feats.append("iupac")
            if rng.random() < 0.3:                                 # soft-masked stretches
                a = int(rng.integers(0, max(1, L - 60))); b = a + int(rng.integers(1, max(2, L - a)))
                s[a:b] = np.frombuffer(bytes(s[a:b]).lower(), dtype=np.uint8); feats.append("lower")
            f.write(b">seq%d some description %d\n" % (i, L))
            b = s.tobytes()
            for o in range(0, len(b), width):
                f.write(b[o:o + width] + b"\n")
            desc.append(",".join(feats))
    return n_seq, int(lens.sum()), width


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from bitmapperbs_amd import mapper
    rng = np.random.default_rng(a.seed)
    fails = 0
    for i in range(a.trials):
        with tempfile.TemporaryDirectory() as wd:
            os.makedirs(os.path.join(wd, "h")); os.makedirs(os.path.join(wd, "d"))
            fh = os.path.join(wd, "h", "g.fa"); fd = os.path.join(wd, "d", "g.fa")
            n_seq, total, width = draw_fasta(rng, fh)
            open(fd, "wb").write(open(fh, "rb").read())
            mapper.Index.build(fh, fh, threads=8)
            mapper.Index.build(fd, fd, threads=8, device=0)
            bad = [s for s in SUFFIXES if open(fh + "." + s, "rb").read() != open(fd + "." + s, "rb").read()]
            print("trial %3d  %4d sequences  %8d bp  width %4d  %s" % (i, n_seq, total, width, "SAME" if not bad else "DIFF " + ",".join(bad)), flush=True)
            fails += bool(bad)
    print("%d trials, %d with differences" % (a.trials, fails))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
