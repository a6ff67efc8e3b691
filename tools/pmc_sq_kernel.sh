#!/bin/bash
# tools/pmc_sq_kernel.sh [config] [kernel-regex] -- SQ counters of one kernel (wave life, waits, instruction mix) on a bench config
CFG=${1:-2}; PAT=${2:-k_align_sw}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_sqk; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="--config $CFG --no-cpu --no-secondary --steps 1 --warmup 1 --min-seconds 0"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/a -- python3 $R/bench.py $P > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/b -- python3 $R/bench.py $P > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH --kernel-trace --output-format csv -d $O/c -- python3 $R/bench.py $P > /dev/null 2>&1
for p in a b c; do python3 $R/tools/pmc_summary.py $O/$p | grep -E "^kernel|$PAT"; done
rm -rf $O
