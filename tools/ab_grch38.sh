#!/bin/bash
# tools/ab_grch38.sh [modes...] -- GPU box: the GRCh38-like genome (bench.py --grch38-like), old build (bitmapperbs_amd/libbmbs_hip_old.so)
# against the new one in the same call, per mode (pe, se, sensitive): value and the largest kernels.  The index is built once (cached
# under the bench's work directory); every run is >= 2 s of mapping.
MODES=${*:-pe se sensitive}
for mode in $MODES; do
  case $mode in pe) X="";; se) X="--se";; sensitive) X="--sensitive --units 5000000";; esac
  for tag in old new old new; do
    lib=$PWD/bitmapperbs_amd/libbmbs_hip.so; [ $tag = old ] && lib=$PWD/bitmapperbs_amd/libbmbs_hip_old.so
    BMBS_LIB=$lib timeout 900 python bench.py --grch38-like $X --launches 2 --steps 3 --no-cpu --no-secondary --no-single-lane 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernels_ms_per_launch',{})
top=sorted(((v,a) for a,v in k.items() if a.startswith('k_')), reverse=True)[:7]
print('$mode $tag', d['value'], 'ms/launch %.2f' % (d['ms_per_step']/d['config']['launches_per_step']), ' '.join('%s=%.2f'%(a,v) for v,a in top), 'drop', d['counters_last_launch'].get('n_prefilter_drop'))"
  done
done
