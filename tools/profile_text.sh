#!/bin/bash
# tools/profile_text.sh [tag] -- on the MI355X box: rocprofv3 evidence for the kernels of the FILE path (the two ends of a bmbs_search run:
# k_fq_*, k_fastq_rows, k_bgzf_inflate, k_sam_len, k_line_write<SAM|BAM>, k_bam_len, k_bgzf_block, k_bgzf_gather), which the bench's
# device-resident launches never run: kernel stats of one BGZF -> SAM run and one FASTQ -> BAM run, then FETCH_SIZE / WRITE_SIZE passes of
# the same commands (the program itself after `--`: no shell in between).
TAG=${1:-r05_text}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
read FA F1 F2 NP < <(python3 $R/tools/e2e_setup.py 2500000 2 | tail -1)
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import bench
for k in (1, 2):
    with open("$W/e2e_%d.fq" % k, "rb") as f: data = f.read()
    bench.write_bgzf("$W/b_%d.fq.gz" % k, data, level=1, threads=16)
PY
D=$R/bitmapperbs_amd/bmbs_search
cd /tmp && export TMPDIR=/tmp
A_SAM="--search $FA -e 0.08 --seq1 $W/b_1.fq.gz --seq2 $W/b_2.fq.gz -o /dev/null -t 32 --verbose"
A_BAM="--search $FA -e 0.08 --seq1 $F1 --seq2 $F2 -o /dev/null --bam -t 32 --verbose"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sam -- $D $A_SAM > $O/run_bgzf_sam.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bam -- $D $A_BAM > $O/run_fastq_bam.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_sam_$C -- $D $A_SAM --contexts 1 > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_bam_$C -- $D $A_BAM --contexts 1 > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv, glob, os
O = "$O"
TEXT = ("k_fq_", "k_fastq_rows", "k_bgzf_", "k_sam_len", "k_line_write", "k_bam_len", "k_nl_count", "k_close_last_line", "k_pack_rows", "k_rows_from")
def short(n): return n.split("(")[0].replace("void ", "")
with open(os.path.join(O, "text_kernel_stats.csv"), "w") as o:
    w = csv.writer(o); w.writerow(["run", "kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "share_of_gpu_kernel_time"])
    for run in ("sam", "bam"):
        f = glob.glob(os.path.join(O, "stats_" + run, "**", "*kernel_stats.csv"), recursive=True)
        if not f: continue
        rows = [r for r in csv.DictReader(open(f[0])) if not any(x in r["Name"] for x in ("k_build_t20", "k_expand_sa", "k_occ3", "k_build_gen2", "k_repack", "k_attach"))]
        tot = sum(int(r["TotalDurationNs"]) for r in rows)
        agg = {}
        for r in rows:
            k = short(r["Name"]) if any(t in r["Name"] for t in TEXT) else "(mapping kernels, scans, copies)"
            a = agg.setdefault(k, [0, 0, 1 << 62, 0])
            a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"]); a[2] = min(a[2], int(r["MinNs"])); a[3] = max(a[3], int(r["MaxNs"]))
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow(["bgzf->sam" if run == "sam" else "fastq->bam", k, a[0], round(a[1] / 1e6, 3), round(a[1] / a[0] / 1e6, 4), round(a[2] / 1e6, 4), round(a[3] / 1e6, 4), round(a[1] / tot, 4)])
# PMC: bytes per dispatch of the text kernels (median over dispatches)
import statistics
with open(os.path.join(O, "text_pmc_fetch_write.csv"), "w") as o:
    w = csv.writer(o); w.writerow(["run", "kernel", "dispatches", "FETCH_SIZE_KB_median", "WRITE_SIZE_KB_median"])
    for run in ("sam", "bam"):
        vals = {}
        for C in ("FETCH_SIZE", "WRITE_SIZE"):
            for f in glob.glob(os.path.join(O, "pmc_%s_%s" % (run, C), "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") != C or not any(t in r["Kernel_Name"] for t in TEXT): continue
                    vals.setdefault(short(r["Kernel_Name"]), {}).setdefault(C, {}).setdefault(r["Dispatch_Id"], 0.0)
                    vals[short(r["Kernel_Name"])][C][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, v in sorted(vals.items()):
            fe = list(v.get("FETCH_SIZE", {}).values()); wr = list(v.get("WRITE_SIZE", {}).values())
            w.writerow(["bgzf->sam" if run == "sam" else "fastq->bam", k, len(fe), round(statistics.median(fe), 1) if fe else "", round(statistics.median(wr), 1) if wr else ""])
PY
grep -h "mapping wall" $O/run_*.log | cut -c1-160
rm -rf $O/stats_* $O/pmc_*
cat $O/text_kernel_stats.csv
cat $O/text_pmc_fetch_write.csv
