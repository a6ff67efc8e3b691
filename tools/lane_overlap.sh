#!/bin/bash
# tools/lane_overlap.sh "<bench args>" [last ms] -- GPU box: a kernel trace of the bench command and tools/lane_overlap.py on its last ms
ARGS=$1; MS=${2:-300}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/lanes_trace; rm -rf $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py $ARGS --no-cpu --no-secondary --no-single-lane --launches 2 --steps 3 --min-seconds 0 > /dev/null 2>&1
cd $R && python3 tools/lane_overlap.py $O $MS; rm -rf $O
