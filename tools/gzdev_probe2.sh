#!/bin/bash
# tools/gzdev_probe2.sh -- on the GPU box: the device gzip path with 1 and 4 contexts, every text call's phase times
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
read FA F1 F2 NP < <(python3 tools/e2e_setup.py 2500000 2 | tail -1)
( gzip -1 -c $F1 > $W/g_1.fq.gz ) & ( gzip -1 -c $F2 > $W/g_2.fq.gz ) & wait
for c in 1 4; do
echo "== contexts $c"
BMBS_GZ_DEVICE=2 BMBS_TEXT_TRACE=1 ./bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --seq1 $W/g_1.fq.gz --seq2 $W/g_2.fq.gz -o /dev/null -t 32 --contexts $c --verbose 2>&1 | grep -E "text open gzip\] buf|mapping wall" | cut -c1-200 | tail -${TAILN:-8}
done
