#!/usr/bin/env python3
"""tools/bench_show2.py <bench json> -- the headline, PCIe-inclusive and e2e keys of a bench line, one per row"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["kernel"], "single_lane", (d["roofline"].get("single_lane") or {}).get("frac"))
print("incl_pcie", {k: v for k, v in d.get("value_incl_pcie", {}).items() if k != "what"})
e = d.get("e2e", {}).get("file_to_file", {}) or {}
for k, v in e.items():
    if isinstance(v, dict) and "value" in v:
        print("%-16s %7.2f  wall %.3f s  passes %s  bound %s  busy %s %s" % (k, v["value"], v["mapping_wall_s"], v.get("input_passes"), v.get("bound"), v.get("busy"), v.get("runs_Mreads_s", "")))
for k, v in (e.get("gz_input") or {}).items():
    print("gz.%-13s %s" % (k, {a: b for a, b in v.items() if a != "stages"} if isinstance(v, dict) else v))
for k, v in (d.get("secondary") or {}).items():
    if isinstance(v, dict):
        print("sec.%-24s %s" % (k, v.get("value")))
print("cpu_baseline", {k: v for k, v in d.get("cpu_baseline", {}).items() if k != "sample"})
kms = d.get("kernels_ms_per_launch_single_lane") or {}
print("single-lane kernels:", {k: round(v, 2) for k, v in sorted(kms.items(), key=lambda kv: -kv[1])[:14]})
