#!/bin/bash
# tools/ab_c4.sh -- the 250-bp config (64-bit Myers bands), old build against new on the same box
for tag in old new old new; do
  lib=$PWD/bitmapperbs_amd/libbmbs_hip.so; [ $tag = old ] && lib=$PWD/bitmapperbs_amd/libbmbs_hip_old.so
  BMBS_LIB=$lib timeout 600 python bench.py --config 4 --steps 6 --no-cpu --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernels_ms_per_launch',{})
print('$tag', d['value'], d['ms_per_step'], ' '.join('%s=%s'%(x,k.get(x)) for x in ['k_filter_pe_r1','k_filter_pe_r2']), 'sum=%.2f'%sum(k.values()))"
done
