#!/bin/bash
# tools/bound_speed.sh -- GPU box: the reference program bound to the library (oracle/_ref/bitmapperBS_hip, INTEGRATION.md sections 2-3) and the
# unmodified reference (-t 32) on the same 5 M pairs: mapping seconds as their own main prints them ("Total: <load> <map>"), and the SAM compared
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
read FA F1 F2 NP < <(python3 tools/e2e_setup.py 5000000 1 | tail -1)
for exe in bitmapperBS_hip bitmapperBS; do
  T="-t 32"; [ $exe = bitmapperBS_hip ] && T="-t 1"
  t0=$(date +%s)
  oracle/_ref/$exe --search $FA --seq1 $F1 --seq2 $F2 -o $W/$exe.sam $T 2>&1 | grep -E "Total:|No. of Reads|bitmapperBS_hip:"
  echo "$exe wall $(( $(date +%s) - t0 )) s"
done
# (the reference's -t 32 writes its records in whatever order its threads finish: compared as sorted lines)
grep -v "^@PG" $W/bitmapperBS_hip.sam | LC_ALL=C sort | md5sum; grep -v "^@PG" $W/bitmapperBS.sam | LC_ALL=C sort | md5sum
rm -f $W/bitmapperBS_hip.sam $W/bitmapperBS.sam
