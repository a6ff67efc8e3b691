#!/usr/bin/env python3
"""tools/baseline_table.py [tag=r04] -- the rows of BASELINE.md section 4 from the committed bench lines and PMC passes of a round
(profiles/<tag>_c<k>_bench.json, _pmc_fetch_write.csv): whole-launch figures, one GPU."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
for table_cfg, k in ((2, 1), (3, 2), (4, 3), (5, 4)):
    f = os.path.join(ROOT, "profiles", "%s_c%d_bench.json" % (tag, k))
    if not os.path.exists(f):
        continue
    d = json.load(open(f))
    ms = d.get("kernels_ms_per_launch_single_lane") or d.get("kernels_ms_per_launch") or {}
    alg = d.get("kernels_algorithmic_GBps") or {}
    launch_ms = sum(v for v in ms.values())
    alg_bytes = sum(alg[kn] * 1e9 * (d["kernels_ms_per_launch"].get(kn, 0.0) * 1e-3) for kn in alg)
    alg_gbs = alg_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms else 0.0
    sect = ((d.get("roofline") or {}).get("sectors") or {}).get("traffic_GBps")
    fetch = None
    p = os.path.join(ROOT, "profiles", "%s_c%d_pmc_fetch_write.csv" % (tag, k))
    if os.path.exists(p) and launch_ms:
        tot = 0.0
        for r in csv.DictReader(open(p)):
            if r["kernel"].startswith("k_occ3") or r["kernel"].startswith("k_call") or not r["FETCH_SIZE_KB_last_launch"]:
                continue
            tot += float(r["FETCH_SIZE_KB_last_launch"]) * 1024
        fetch = tot / (launch_ms * 1e-3) / 1e9
    same = d.get("sample_sam_identical_to_reference")
    print("| %d | 1 | %.0f | %.0f | %s | %s | %.1f %% | %s |" % (table_cfg, d["value"], alg_gbs, "%.0f (dominant kernel)" % sect if sect else "", "%.0f" % fetch if fetch else "",
                                                              100.0 * alg_gbs / 8000.0, "yes (%s lines)" % d.get("sample_sam_lines_compared") if same is True else ("not run" if same is None else str(same))))
