#!/bin/bash
# tools/ztimeline_probe.sh [pairs] -- on the GPU box: rocprofv3 kernel + copy trace of a BGZF run through the device inflater; how kernels and
# copies overlapped during the mapping phase (tools/timeline_summary.py)
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
N=${1:-2500000}; REP=${2:-2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
read FA F1 F2 NP < <(python3 tools/e2e_setup.py $N $REP | tail -1)
python3 - <<PY
import sys; sys.path.insert(0, ".")
import bench
for k in (1, 2):
    with open("$W/e2e_%d.fq" % k, "rb") as f: data = f.read()
    bench.write_bgzf("$W/b_%d.fq.gz" % k, data, level=1, threads=16)
PY
O=$R/gpurun_out/ztimeline; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- $R/bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --seq1 $W/b_1.fq.gz --seq2 $W/b_2.fq.gz -o /dev/null -t 32 --verbose > $O/run.log 2>&1
grep "mapping wall" $O/run.log | cut -c1-200
W_MS=$(grep "mapping wall" $O/run.log | sed 's/.*mapping wall \([0-9.]*\)s.*/\1/' | awk '{print $1*1000}')
python3 $R/tools/timeline_summary.py $O $W_MS
find $O -name "*.csv" -size +20M -delete
