#!/bin/bash
# tools/pmc_tcp_kernel.sh [kernel-regex] [bench args] -- L1 (TCP) / L2 (TCC) request counters of one kernel, single lane
PAT=${1:-k_filter}; X=${2:---grch38-like --se}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_tcp; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export BMBS_LANES=1
P="$X --no-cpu --no-secondary --steps 1 --warmup 1 --min-seconds 0 --no-single-lane --launches 1"
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z_]*\|TCC_[A-Z_]*" | sort -u | tr '\n' ' ' > $O/names.txt
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $O/a -- python3 $R/bench.py $P > $O/a.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/b -- python3 $R/bench.py $P > $O/b.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/c -- python3 $R/bench.py $P > $O/c.log 2>&1
cd $R
for p in a b c; do python3 tools/pmc_summary.py $O/$p | grep -E "^kernel|$PAT" > $O/$p.csv; cat $O/$p.csv; tail -2 $O/$p.log | cut -c1-200; rm -rf $O/$p; done
