for tag in old new; do
  lib=$PWD/bitmapperbs_amd/libbmbs_hip.so; [ $tag = old ] && lib=$PWD/bitmapperbs_amd/libbmbs_hip_old.so
  BMBS_LIB=$lib BMBS_LANES=1 timeout 900 python bench.py --grch38-like --launches 2 --steps 3 --no-cpu --no-secondary --no-single-lane 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernels_ms_per_launch',{})
print('$tag single lane', d['value'], {a:v for a,v in k.items() if 'vote_pe' in a})"
done
