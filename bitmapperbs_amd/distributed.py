"""Multi-GPU host logic: one process per GPU, reads shard by contiguous record ranges, the index is
replicated in every GPU's HBM, no data-path collective.  The only exchange is the sum of the five
mapstats counters (get_mapping_informations, Schema.cpp:451-476) -- one 40-byte all-reduce over RCCL
(`nccl` backend) on GPUs, or gloo in the CPU tests -- and the ordered concatenation of the per-rank SAM
parts, which reproduces the reference's `-t 1` record order."""
from __future__ import annotations

import os

import numpy as np


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous range of records for `rank` (range i -> GPU i); sizes differ by at most one"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_stats(stats: np.ndarray, device: str | None = None) -> np.ndarray:
    """sum int64[5] mapstats over all ranks (no-op when torch.distributed is not initialised)"""
    import torch
    import torch.distributed as dist
    s = np.ascontiguousarray(stats, dtype=np.int64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return s.copy()
    t = torch.from_numpy(s.copy())
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def concat_parts(out_path: str, header: str, world: int, rank: int) -> None:
    """rank 0 concatenates <out>.part<r> in rank order behind the header (call after a barrier)"""
    if rank != 0:
        return
    with open(out_path, "w") as o:
        o.write(header)
        for r in range(world):
            p = "%s.part%d" % (out_path, r)
            with open(p) as f:
                for chunk in iter(lambda: f.read(1 << 24), ""):
                    o.write(chunk)
            os.remove(p)


def mapstats_text(st) -> str:
    """the reference's --mapstats / stderr block (Bitmapper_main.cpp:275-307)"""
    reads, uniq, amb = int(st[0]), int(st[1]), int(st[2])
    unm = reads - uniq - amb
    pct = lambda x: (float(x) / float(reads)) * 100 if reads else float("nan")
    rate = (float(st[4]) / float(st[3])) * 100 if st[3] else float("nan")
    return ("%-48s%d\n" % ("No. of Reads:", reads) +
            "%-48s%d (%0.2f%%)\n" % ("No. of Unique Mapped Reads:", uniq, pct(uniq)) +
            "%-48s%d (%0.2f%%)\n" % ("No. of Ambiguous Mapped Reads:", amb, pct(amb)) +
            "%-48s%d (%0.2f%%)\n" % ("No. of Unmapped Reads:", unm, pct(unm)) +
            "%-47s %0.2f%%\n" % ("Mismatch and Indel Rate:", rate))
