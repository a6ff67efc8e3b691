#!/bin/bash
# tools/ab_c2.sh [kernel keys...] -- the main config, old build (libbmbs_hip_old.so) against the new one on the same box: value and the
# single-lane times of the kernels named
for tag in old new old new; do
  lib=$PWD/bitmapperbs_amd/libbmbs_hip.so; [ $tag = old ] && lib=$PWD/bitmapperbs_amd/libbmbs_hip_old.so
  BMBS_LIB=$lib timeout 600 python bench.py --steps 6 --no-cpu --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernels_ms_per_launch',{})
print('$tag', d['value'], d['ms_per_step'], ' '.join('%s=%s'%(x,k.get(x)) for x in '$*'.split()), 'sum=%.2f'%sum(k.values()))"
done
