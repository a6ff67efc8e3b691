#!/bin/bash
# tools/ab_lanes.sh -- GRCh38-like single-end, old and new build of the library at 1, 2 and 3 lanes (same box)
out=gpurun_out/abl; mkdir -p $out; : > $out/ab.txt
for lanes in 1 2 3; do for tag in old new; do
  lib=$PWD/bitmapperbs_amd/libbmbs_hip.so; [ $tag = old ] && lib=$PWD/bitmapperbs_amd/libbmbs_hip_old.so
  BMBS_LANES=$lanes BMBS_LIB=$lib timeout 600 python bench.py --grch38-like --se --launches 1 --steps 6 --no-cpu --no-secondary --no-single-lane 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes $lanes $tag', d['value'], d['ms_per_step'])" >> $out/ab.txt
done; done
cat $out/ab.txt
