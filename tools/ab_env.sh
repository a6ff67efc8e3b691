#!/bin/bash
# tools/ab_env.sh "<bench args>" VAR=v1 VAR=v2 ... -- GPU box: the same bench.py command under each environment setting in turn, twice
# (a b a b): value, ms per launch and the largest kernels.  e.g.  tools/ab_env.sh "--grch38-like --se" BMBS_TDEPTH=21 BMBS_TDEPTH=20
ARGS=$1; shift
for rep in 1 2; do
  for kv in "$@"; do
    env $kv timeout 900 python bench.py $ARGS --launches 2 --steps 3 --no-cpu --no-secondary --no-single-lane 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernels_ms_per_launch',{})
top=sorted(((v,a) for a,v in k.items() if a.startswith('k_')), reverse=True)[:6]
print('$kv', d['value'], 'ms/launch %.2f' % (d['ms_per_step']/d['config']['launches_per_step']), ' '.join('%s=%.2f'%(a,v) for v,a in top))"
  done
done
