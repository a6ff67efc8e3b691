#!/bin/bash
# tools/pmc_sq_probe.sh [kernel-regex] [pairs] -- SQ counters of one kernel on the GRCh38-like stress genome (tools/long_lists_probe.py)
PAT=${1:-k_vote_pe_long}; N=${2:-2000000}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_sqp; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/long_lists_probe.py $N > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/b -- python3 $R/tools/long_lists_probe.py $N > /dev/null 2>&1
for p in a b; do python3 $R/tools/pmc_summary.py $O/$p | grep -E "^kernel|$PAT"; done
rm -rf $O
