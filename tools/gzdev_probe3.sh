#!/bin/bash
# tools/gzdev_probe3.sh -- the bench's plain_gzip_device case (2.5 M records per file) with the phase times of every window
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
read FA F1 F2 NP < <(python3 tools/e2e_setup.py 2500000 1 | tail -1)
( gzip -1 -c $F1 > $W/g_1.fq.gz ) & ( gzip -1 -c $F2 > $W/g_2.fq.gz ) & wait
for rep in 1 2; do
BMBS_GZ_DEVICE=2 BMBS_TEXT_TRACE=1 ./bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --seq1 $W/g_1.fq.gz --seq2 $W/g_2.fq.gz -o /dev/null -t 32 --verbose 2>&1 | grep -E "text open gzip|mapping wall" | cut -c1-200
done
