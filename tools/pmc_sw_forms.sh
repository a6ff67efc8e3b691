#!/bin/bash
# tools/pmc_sw_forms.sh -- FETCH_SIZE / WRITE_SIZE of the three DP kernel forms on the configs[1] batch (one launch each)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_sw; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for form in reg reg2 wave; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    BMBS_SW=$form rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${form}_$ctr -- python3 $R/bench.py --config 1 --no-cpu --no-secondary --steps 1 --warmup 1 --min-seconds 0 > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py $O/${form}_$ctr | grep -E "kernel|k_align_sw" | sed "s/^/$form $ctr /"
  done
done
rm -rf $O
