#!/usr/bin/env python3
"""tools/timeline_summary.py <rocprofv3 output dir> [last N ms] -- how the kernels and the copies of a traced run overlapped: busy time
(union of intervals) of the kernels, of the copies per direction, of both at once and of neither, over the last N ms of the trace
(the mapping phase of a bmbs_search run; the whole trace when N is omitted), and the kernels by total time."""
import csv
import glob
import sys


def load(pat, start, end):
    rows = []
    for f in glob.glob(pat, recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    return rows


def union(iv):
    iv = sorted(iv)
    tot = 0; cur_a = cur_b = None
    out = []
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                out.append((cur_a, cur_b))
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        out.append((cur_a, cur_b))
    return out


def total(u):
    return sum(b - a for a, b in u)


def inter(u, v):
    i = j = 0; t = 0
    while i < len(u) and j < len(v):
        a = max(u[i][0], v[j][0]); b = min(u[i][1], v[j][1])
        if b > a:
            t += b - a
        if u[i][1] < v[j][1]:
            i += 1
        else:
            j += 1
    return t


def main():
    d = sys.argv[1]
    k = load(d + "/**/*kernel_trace.csv", 0, 0)
    c = load(d + "/**/*memory_copy_trace.csv", 0, 0)
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in k]
    cs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), int(r.get("Bytes", 0) or 0)) for r in c]
    t1 = max([x[1] for x in ks] + [x[1] for x in cs])
    t0 = min([x[0] for x in ks] + [x[0] for x in cs])
    if len(sys.argv) > 2:
        t0 = t1 - int(float(sys.argv[2]) * 1e6)
    ks = [x for x in ks if x[1] > t0]; cs = [x for x in cs if x[1] > t0]
    uk = union([(max(a, t0), b) for a, b, _ in ks])
    print("span %.1f ms: %d kernels busy %.1f ms" % ((t1 - t0) / 1e6, len(ks), total(uk) / 1e6))
    dirs = sorted(set(x[2] for x in cs))
    uc_all = union([(max(a, t0), b) for a, b, _, _ in cs])
    for dname in dirs:
        sel = [x for x in cs if x[2] == dname]
        u = union([(max(a, t0), b) for a, b, _, _ in sel])
        print("  copies %-28s %5d  busy %7.1f ms  %8.1f MB  -> %.1f GB/s while busy; %.1f ms of it with a kernel running" %
              (dname, len(sel), total(u) / 1e6, sum(x[3] for x in sel) / 1e6, sum(x[3] for x in sel) / max(1, total(u)), inter(u, uk) / 1e6))
    both = inter(uk, uc_all)
    anyb = total(uk) + total(uc_all) - both
    print("  kernels and copies at once %.1f ms; neither %.1f ms" % (both / 1e6, ((t1 - t0) - anyb) / 1e6))
    agg = {}
    for a, b, n in ks:
        n = n.split("(")[0][:60]
        x = agg.setdefault(n, [0, 0]); x[0] += 1; x[1] += b - max(a, t0)
    for n, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-60s %5d  %8.2f ms" % (n, cnt, t / 1e6))


main()
