#!/bin/bash
# tools/vote_kbench_pmc.sh <binary> [args] -- GPU box: instruction mix of k_vote_long in the microbench (rocprofv3 --pmc, separate passes)
B=$1; shift
O=/tmp/vk_pmc_$$; rm -rf $O; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p -- $R/tools/$B "$@" > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $O/p | grep -v "^kernel" | head -3
  python3 $R/tools/pmc_summary.py $O/p | head -1
  rm -rf $O/p
done
