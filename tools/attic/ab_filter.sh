#!/bin/bash
# tools/ab_filter.sh -- same-box A/B of two builds of the library on the workloads where the Myers filter weighs most
# (GRCh38-like SE / PE / sensitive, then the main config); old build = bitmapperbs_amd/libbmbs_hip_old.so
out=gpurun_out/abf; mkdir -p $out
for tag in old new old new; do
  lib=$PWD/bitmapperbs_amd/libbmbs_hip.so; [ $tag = old ] && lib=$PWD/bitmapperbs_amd/libbmbs_hip_old.so
  for w in "--se" "--pe" "--pe --sensitive"; do
    BMBS_LIB=$lib timeout 600 python bench.py --grch38-like $w --launches 1 --steps 6 --no-cpu --no-secondary --no-single-lane 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag','$w', d['value'], d['ms_per_step'])" >> $out/ab.txt
  done
  BMBS_LIB=$lib timeout 600 python bench.py --steps 6 --no-cpu --no-secondary --no-single-lane 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag','c2', d['value'], d['ms_per_step'])" >> $out/ab.txt
done
cat $out/ab.txt
