#!/usr/bin/env python3
"""tools/lane_overlap.py <rocprofv3 --kernel-trace output dir> [last N ms] -- how the kernels of a context's lanes shared the GPU: over
the last N ms of the trace (the timed steps of a bench run), the time no kernel, one, two, three or more kernels were running, and per
kernel the time it ran with nothing beside it against the time it shared the chip.
GPU box:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lanes -- python3 $R/bench.py --grch38-like --no-cpu
          --no-secondary --no-single-lane --steps 3;  python3 tools/lane_overlap.py gpurun_out/lanes 400"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows]
    iv = [x for x in iv if x[2].startswith("k_")]
    end = max(b for _, b, _ in iv)
    start = end - int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else min(a for a, _, _ in iv)
    iv = [(max(a, start), b, n) for a, b, n in iv if b > start]
    ev = []
    for i, (a, b, n) in enumerate(iv):
        ev.append((a, 1, i)); ev.append((b, -1, i))
    ev.sort()
    level = defaultdict(int)              # ns at each number of running kernels
    alone = defaultdict(int); shared = defaultdict(int)
    running = set()
    t = start
    for ts, kind, i in ev:
        dt = ts - t
        if dt > 0:
            level[min(len(running), 4)] += dt
            for j in running:
                (alone if len(running) == 1 else shared)[iv[j][2]] += dt
        t = ts
        if kind == 1:
            running.add(i)
        else:
            running.discard(i)
    wall = end - start
    print("window %.1f ms, %d dispatches; kernels running at once: none %.1f %%, one %.1f %%, two %.1f %%, three %.1f %%, four or more %.1f %%; sum of durations / wall = %.2f" % (
        wall / 1e6, len(iv), *(100.0 * level[k] / wall for k in range(5)), sum(b - a for a, b, _ in iv) / wall))
    names = sorted(set(alone) | set(shared), key=lambda n: -(alone[n] + shared[n]))
    print("%-34s %10s %10s %10s" % ("kernel", "alone ms", "shared ms", "% of wall"))
    for n in names[:18]:
        print("%-34s %10.1f %10.1f %10.1f" % (n[:34], alone[n] / 1e6, shared[n] / 1e6, 100.0 * (alone[n] + shared[n]) / wall))


main()
