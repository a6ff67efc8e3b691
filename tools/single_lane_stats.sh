#!/bin/bash
# tools/single_lane_stats.sh "<bench args>" tag -- GPU box: rocprofv3 kernel stats of the bench command on ONE lane (BMBS_LANES=1): with
# nothing running beside a kernel its duration is what it costs, which the three-lane profile's durations (stretched by whatever shares
# the CUs) are not.  Output: gpurun_out/<tag>_single_lane_kernel_stats.csv (per-launch ms and share, largest first).
ARGS=$1; TAG=${2:-sl}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/sl_$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export BMBS_LANES=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS --no-cpu --no-secondary --no-single-lane --launches 2 --steps 3 --min-seconds 0 > $O/bench.json 2>/dev/null
cd $R
python3 - <<PY
import csv, glob, os, json
O = "$O"
d = json.loads(open(os.path.join(O, "bench.json")).read().strip().splitlines()[-1])
launches = (d["steps"] + max(2, d["warmup"])) * d["config"]["launches_per_step"]
f = glob.glob(os.path.join(O, "stats", "**", "*kernel_stats.csv"), recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "k_" in r["Name"] and not r["Name"].startswith("void at::") and "k_ib_" not in r["Name"] and int(r["Calls"]) >= launches]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("$R/gpurun_out/${TAG}_single_lane_kernel_stats.csv", "w") as o:
    w = csv.writer(o); w.writerow(["kernel", "calls", "ms_per_launch", "share", "avg_us"])
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        w.writerow([r["Name"].split("(")[0], r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6 / launches), "%.4f" % (float(r["TotalDurationNs"]) / tot), "%.1f" % (float(r["AverageNs"]) / 1e3)])
print("value", d["value"], "ms/launch", d["ms_per_step"] / d["config"]["launches_per_step"], "kernel ms/launch", tot / 1e6 / launches, "launches", launches)
PY
rm -rf $O
head -24 $R/gpurun_out/${TAG}_single_lane_kernel_stats.csv
