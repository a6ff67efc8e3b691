#!/bin/bash
# tools/inflate_pmc.sh [MB] -- GPU box: where k_bgzf_inflate's fetched bytes come from (tools/inflate_kbench, the kernel alone on one
# window): FETCH_SIZE / WRITE_SIZE, L2 hits and misses, L1 -> L2 read requests, in separate rocprofv3 --pmc passes
MB=${1:-160}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=/tmp/inf_pmc_$$; rm -rf $O; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p -- $R/tools/inflate_kbench $MB > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $O/p | head -1; python3 $R/tools/pmc_summary.py $O/p | grep k_bgzf_inflate
  rm -rf $O/p
done
