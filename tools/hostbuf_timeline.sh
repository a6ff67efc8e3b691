#!/bin/bash
# tools/hostbuf_timeline.sh -- on the GPU box: timeline of the host-buffer calls (packed and ASCII): copies and kernels busy / overlapped
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $R/tools/hostbuf_probe.py 2000000 1 4
python3 $R/tools/hostbuf_probe.py 2000000 0 4
for P in 1 0; do
  O=$R/gpurun_out/hb_timeline_$P; rm -rf $O; mkdir -p $O
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 $R/tools/hostbuf_probe.py 2000000 $P 3 > $O/run.log 2>&1)
  tail -2 $O/run.log
  python3 $R/tools/timeline_summary.py $O 60 | head -12
  find $O -name "*.csv" -size +20M -delete
done
