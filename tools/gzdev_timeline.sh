#!/bin/bash
# tools/gzdev_timeline.sh -- on the GPU box: when the kernels of one pair of gzip windows ran (rocprofv3 kernel trace of the device gzip path, one context)
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
read FA F1 F2 NP < <(python3 tools/e2e_setup.py 2500000 2 | tail -1)
( gzip -1 -c $F1 > $W/g_1.fq.gz ) & ( gzip -1 -c $F2 > $W/g_2.fq.gz ) & wait
O=/tmp/gztl; rm -rf $O
cd /tmp && export TMPDIR=/tmp
BMBS_GZ_DEVICE=2 rocprofv3 --kernel-trace --output-format csv -d $O -- $R/bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --seq1 $W/g_1.fq.gz --seq2 $W/g_2.fq.gz -o /dev/null -t 32 --contexts 1 > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_gz" in r["Kernel_Name"] or "k_crc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[14:42]:
    print("%-20s queue %-4s  start %8.3f ms  end %8.3f ms  (%.3f)" % (r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
