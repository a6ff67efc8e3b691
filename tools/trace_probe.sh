#!/bin/bash
# tools/trace_probe.sh [kernel-regex] [pairs] -- rocprofv3 kernel-trace durations of the kernels that match, on the GRCh38-like stress genome
PAT=${1:-k_vote_pe}; N=${2:-2000000}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/trace_probe; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/long_lists_probe.py $N > $O/probe.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
head -1 $f; grep -E "$PAT" $f
grep -E "^call" $O/probe.log
