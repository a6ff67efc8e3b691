"""Synthetic bisulfite reads generated directly in HBM with torch (workload tooling for bench.py).

Same model as synth.py: uniform start, 50 % each strand, C->T at `conv` after strand
selection, substitutions, at most one single-base indel per read, constant or random qualities.
torch is plumbing here (device memory + RNG); nothing in this file is on the mapping path.
"""
from __future__ import annotations

import numpy as np
import torch


def upload_genome(chroms) -> tuple[torch.Tensor, torch.Tensor]:
    cat = np.concatenate(chroms)
    lens = np.array([c.size for c in chroms], dtype=np.int64)
    return torch.from_numpy(cat).cuda(), torch.from_numpy(lens).cuda()


@torch.no_grad()
def make_reads_se(genome: torch.Tensor, lens: torch.Tensor, n: int, L: int, stride: int, seed: int,
                  sub: float = 0.005, indel: float = 0.0002, conv: float = 0.99, qual: str = "const"):
    """-> (seq[n, stride] uint8, qual[n, stride] uint8) on the current CUDA/HIP device"""
    dev = genome.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    offs = torch.cumsum(lens, 0) - lens
    c = torch.randint(0, lens.numel(), (n,), generator=g, device=dev)
    hi = (lens[c] - L - 8).to(torch.float64)
    p = (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * hi).to(torch.int64)
    start = offs[c] + p
    minus = torch.rand(n, generator=g, device=dev) < 0.5
    # single indel per read: kind 0 none, 1 deletion (template base skipped), 2 insertion
    has = torch.rand(n, generator=g, device=dev) < indel * L
    kind = torch.where(has, torch.randint(1, 3, (n,), generator=g, device=dev), torch.zeros(n, dtype=torch.int64, device=dev))
    ipos = torch.randint(5, L - 5, (n,), generator=g, device=dev)
    seq = torch.full((n, stride), 0, dtype=torch.uint8, device=dev)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    comp = torch.arange(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    chunk = 1 << 20
    j = torch.arange(L, device=dev)[None, :]
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        k_ = kind[a:b, None]
        ip = ipos[a:b, None]
        # template offset of read position j
        t = j + ((k_ == 1) & (j >= ip)).to(torch.int64) - ((k_ == 2) & (j > ip)).to(torch.int64)
        m = minus[a:b, None]
        # minus strand: read = revcomp of the window [start, start+L+8)
        gi = torch.where(m, start[a:b, None] + (L + 7) - t, start[a:b, None] + t)
        base = genome[gi]
        base = torch.where(m, comp[base.long()], base)
        cv = (base == ord("C")) & (torch.rand((b - a, L), generator=g, device=dev) < conv)
        base = torch.where(cv, torch.tensor(ord("T"), dtype=torch.uint8, device=dev), base)
        sm = torch.rand((b - a, L), generator=g, device=dev) < sub
        rb = acgt[torch.randint(0, 4, (b - a, L), generator=g, device=dev)]
        base = torch.where(sm, rb, base)
        ins = (k_ == 2) & (j == ip)
        base = torch.where(ins, acgt[torch.randint(0, 4, (b - a, L), generator=g, device=dev)], base)
        seq[a:b, :L] = base
    if qual == "const":
        q = torch.full((n, stride), ord("I"), dtype=torch.uint8, device=dev)
    else:
        q = (torch.randint(2, 41, (n, stride), generator=g, device=dev) + 33).to(torch.uint8)
    return seq, q


@torch.no_grad()
def make_reads_pe(genome: torch.Tensor, lens: torch.Tensor, n: int, L: int, stride: int, seed: int,
                  sub: float = 0.005, conv: float = 0.99, ins_lo: int | None = None, ins_hi: int = 400,
                  indel: float = 0.0, qual: str = "const"):
    """-> (seq1, qual1, seq2, qual2) [n, stride] uint8 on the device; mate 2 as it would appear in the FASTQ
    (reverse-complement end of the converted fragment); substitutions and at most one single-base indel per mate."""
    dev = genome.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    ins_lo = L + 20 if ins_lo is None else ins_lo
    ins_hi = max(ins_hi, ins_lo + 150)
    offs = torch.cumsum(lens, 0) - lens
    c = torch.randint(0, lens.numel(), (n,), generator=g, device=dev)
    ins = torch.randint(ins_lo, ins_hi, (n,), generator=g, device=dev)
    hi = (lens[c] - ins - 8).to(torch.float64)
    p = (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * hi).to(torch.int64)
    start = offs[c] + p
    minus = torch.rand(n, generator=g, device=dev) < 0.5
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    comp = torch.arange(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    s1 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    s2 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    j = torch.arange(L, device=dev)[None, :]
    T = torch.tensor(ord("T"), dtype=torch.uint8, device=dev)
    chunk = 1 << 20
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        m = minus[a:b, None]; st = start[a:b, None]; il = ins[a:b, None]
        def frag(f):                                   # fragment base f of the sequenced strand, bisulfite converted
            gi = torch.where(m, st + il - 1 - f, st + f)
            base = genome[gi]
            base = torch.where(m, comp[base.long()], base)
            # conversion decided per (pair, fragment position) so that overlapping mates agree
            h = ((torch.arange(a, b, device=dev)[:, None] * 2654435761 + f * 40503 + seed) % 1000003).to(torch.float32) / 1000003.0
            return torch.where((base == ord("C")) & (h < conv), T, base)
        # single indel per mate: kind 0 none, 1 deletion (template base skipped), 2 insertion; t = template offset of read base j
        ts = []
        for _mate in range(2):
            if indel > 0:
                has = torch.rand((b - a, 1), generator=g, device=dev) < indel * L
                kind = torch.where(has, torch.randint(1, 3, (b - a, 1), generator=g, device=dev), torch.zeros((b - a, 1), dtype=torch.int64, device=dev))
                ip = torch.randint(5, L - 5, (b - a, 1), generator=g, device=dev)
                t = j + ((kind == 1) & (j >= ip)).to(torch.int64) - ((kind == 2) & (j > ip)).to(torch.int64)
                ts.append((t, (kind == 2) & (j == ip)))
            else:
                ts.append((j + 0 * il, None))
        r1 = frag(ts[0][0])
        r2 = comp[frag(il - 1 - ts[1][0]).long()]
        for r, dst, (t_, insm) in ((r1, s1, ts[0]), (r2, s2, ts[1])):
            sm = torch.rand((b - a, L), generator=g, device=dev) < sub
            rb = acgt[torch.randint(0, 4, (b - a, L), generator=g, device=dev)]
            r = torch.where(sm, rb, r)
            if insm is not None:
                r = torch.where(insm, acgt[torch.randint(0, 4, (b - a, L), generator=g, device=dev)], r)
            dst[a:b, :L] = r
    if qual == "const":
        q = torch.full((n, stride), ord("I"), dtype=torch.uint8, device=dev)
        q2 = q.clone()
    else:
        q = (torch.randint(2, 41, (n, stride), generator=g, device=dev) + 33).to(torch.uint8)
        q2 = (torch.randint(2, 41, (n, stride), generator=g, device=dev) + 33).to(torch.uint8)
    return s1, q, s2, q2
