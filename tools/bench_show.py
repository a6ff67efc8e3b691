#!/usr/bin/env python3
"""tools/bench_show.py <bench.json> -- the keys of a bench line a reader looks at first"""
import json
import sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("value %.1f M reads/s  ms/step %.1f  wall %.0f s" % (d["value"], d["ms_per_step"], d.get("wall_s", 0)))
print("roofline %s frac %.3f (native %s) traffic %s x%s rule=%s" % (r["kernel"], r["frac"], (r.get("design_native_model") or {}).get("frac"), r.get("traffic"),
                                                                    r.get("traffic_over_algorithmic"), r.get("traffic_rule")))
print("single_lane", r.get("single_lane"))
print("cpu_baseline", (d.get("cpu_baseline") or {}).get("value"), "sam identical", d.get("sample_sam_identical_to_reference"))
e = d.get("e2e", {})
print("host_buffers_overlapped", json.dumps(e.get("host_buffers_overlapped")))
for k, v in (e.get("file_to_file") or {}).items():
    if isinstance(v, dict):
        print("f2f", k, v.get("value"), v.get("mapping_wall_s"))
    elif k == "error":
        print("f2f error", v)
print("two_contexts", (d.get("two_contexts") or {}).get("value"), "three", (d.get("three_contexts") or {}).get("value"))
for k, v in (d.get("secondary") or {}).items():
    print("sec", k, v if isinstance(v, str) else (v.get("value"), v.get("top_kernels_ms")))
print({k: round(v, 3) for k, v in d["kernels_ms_per_launch"].items()})
if d.get("kernels_ms_per_launch_single_lane"):
    k1 = d["kernels_ms_per_launch_single_lane"]
    print("single lane:", {k: round(v, 3) for k, v in sorted(k1.items(), key=lambda x: -x[1])}, "sum %.2f" % sum(k1.values()))
