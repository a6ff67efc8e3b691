#!/bin/bash
# tools/profile_all.sh [tag prefix, default r06] -- GPU box: the four profile sets of a round, one after the other (the default bench line with
# everything; the three GRCh38-like launches without the CPU baseline and the secondary keys).  ~25 minutes.
P=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
BENCH_FIRST=" " bash tools/profile_round.sh 2 ${P}_c2 > gpurun_out/${P}_profile_all.log 2>&1
BENCH_FIRST="--no-cpu --no-secondary" bash tools/profile_round.sh 2 ${P}_grch38_like_pe "--grch38-like" >> gpurun_out/${P}_profile_all.log 2>&1
BENCH_FIRST="--no-cpu --no-secondary" bash tools/profile_round.sh 2 ${P}_grch38_like_se "--grch38-like --se" >> gpurun_out/${P}_profile_all.log 2>&1
BENCH_FIRST="--no-cpu --no-secondary" bash tools/profile_round.sh 2 ${P}_grch38_like_sensitive "--grch38-like --sensitive --units 5000000" >> gpurun_out/${P}_profile_all.log 2>&1
for t in c2 grch38_like_pe grch38_like_se grch38_like_sensitive; do
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/prof_${P}_$t/bench.json').read().strip().splitlines()[-1])
print('$t', d['value'], d['library'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('traffic'))"
done
