#!/bin/bash
# tools/gz_pmc_probe.sh -- on the GPU box: SQ counters of k_gz_spans / k_bgzf_inflate (one launch alone on the chip), groups of counters in passes of their own
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
F=$(python3 $R/tools/e2e_setup.py 1000000 1 | tail -1 | cut -d" " -f2)
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  rm -rf /tmp/pmx
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmx -- python3 $R/tools/gzip_dev_bench.py $F 64 > /tmp/pmx.log 2>&1 || { echo "group failed: $grp"; tail -2 /tmp/pmx.log; continue; }
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for ff in glob.glob("/tmp/pmx/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(ff)):
        k = r["Kernel_Name"].split("(")[0]
        if k in ("k_gz_spans", "k_gz_starts"):
            agg[k][r["Counter_Name"]] = max(agg[k][r["Counter_Name"]], float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: "%.4g" % x for c, x in v.items()})
PY
done
