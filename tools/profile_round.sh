#!/bin/bash
# tools/profile_round.sh [config] [tag] [extra bench.py arguments, e.g. "--grch38-like --se"] -- on the MI355X box: the bench line and the rocprofv3 evidence behind it (kernel stats of the
# SAME command, then separate --pmc passes: FETCH_SIZE, WRITE_SIZE, SQ stall counters).  Output under gpurun_out/prof_<tag>/.
set -u
CFG=${1:-2}
TAG=${2:-r06_c$CFG}
X=${3:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --config $CFG $X ${BENCH_FIRST:---no-stress} > $O/bench.json 2> $O/bench.err
Q="--config $CFG $X --no-cpu --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $Q > $O/bench_under_rocprof.json 2>/dev/null
# the counter passes run the call on ONE lane (BMBS_LANES=1): a kernel's last dispatch is then the whole launch, as the algorithmic bytes of
# the bench line are; the counters of a kernel do not depend on what runs beside it
export BMBS_LANES=1
P="$Q --steps 1 --warmup 1 --min-seconds 0 --no-single-lane --launches 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $P > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py $P > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py $P > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py $O/pmc_fetch > $O/pmc_fetch.csv
python3 tools/pmc_summary.py $O/pmc_write > $O/pmc_write.csv
python3 tools/pmc_summary.py $O/pmc_sq > $O/pmc_sq.csv
python3 - <<PY
import csv, glob, os
O = "$O"
# kernel stats: keep the mapping kernels, sum the rest
f = glob.glob(os.path.join(O, "stats", "**", "*kernel_stats.csv"), recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    keep = [r for r in rows if "k_" in r["Name"] and not r["Name"].startswith("void at::") and "k_ib_" not in r["Name"]]
    other = [r for r in rows if r not in keep]
    with open(os.path.join(O, "kernel_stats.csv"), "w") as o:
        w = csv.writer(o); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for r in keep: w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
        w.writerow(["(torch / rocPRIM / index-builder kernels that set the workload up)", sum(int(r["Calls"]) for r in other), sum(int(r["TotalDurationNs"]) for r in other), "", "", ""])
# FETCH_SIZE + WRITE_SIZE of the last launch of every kernel, one file
fe = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(O, "pmc_fetch.csv")))}
wr = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(O, "pmc_write.csv")))}
with open(os.path.join(O, "pmc_fetch_write.csv"), "w") as o:
    w = csv.writer(o); w.writerow(["kernel", "launches", "FETCH_SIZE_KB_last_launch", "WRITE_SIZE_KB_last_launch"])
    for k in sorted(fe):
        w.writerow([k, fe[k]["dispatches"], fe[k].get("FETCH_SIZE", ""), wr.get(k, {}).get("WRITE_SIZE", "")])
PY
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
ls -la $O | head -30
tail -c 400 $O/bench.err
