#!/bin/bash
# tools/r5_text_probe.sh -- on the GPU box: (1) phase times of the text calls inside the driver, 4 contexts; (2) cycles per phase of
# k_bgzf_inflate (needs bitmapperbs_amd/libbmbs_hip_prof.so, tools/inflate_prof.sh); (3) rocprofv3 kernel stats of a BGZF -> SAM run
O=gpurun_out/r5text; mkdir -p $O
bash tools/e2e_trace.sh 4 > $O/trace.txt 2>&1
[ -f bitmapperbs_amd/libbmbs_hip_prof.so ] && python3 tools/inflate_bench.py 160 bitmapperbs_amd/libbmbs_hip_prof.so > $O/inflate_prof.txt 2>&1
bash tools/ztimeline_probe.sh 2500000 4 > $O/ztimeline.txt 2>&1
python3 - <<PY > $O/kernel_summary.txt
import csv, collections, glob
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/ztimeline/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print("%-62s n=%5d avg=%8.3f ms tot=%9.1f ms" % (k, len(v), sum(v) / len(v), sum(v)))
PY
