#!/bin/bash
# tools/ab_sw.sh -- GPU box: the DP kernels' single-lane times of the main config under A/B environments (env assignments as the first argument of run)
run() { echo "== $1 $2"; env $1 python bench.py $2 --no-cpu --no-secondary --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_launch']
print(d['value'], k['k_align_sw'], d['counters_last_launch']['n_sw'])"; }
run A=1 "--config 2"
run BMBS_SW=reg "--config 2"
run A=1 "--config 1"
run BMBS_SW=reg "--config 1"
run A=1 "--config 4"
run BMBS_SW=reg "--config 4"
