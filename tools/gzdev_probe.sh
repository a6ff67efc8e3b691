#!/bin/bash
# tools/gzdev_probe.sh [pairs] -- on the GPU box: ordinary one-member .gz FASTQ (gzip -1) through the driver's device inflater (BMBS_GZ_DEVICE=2),
# pairs and one file alone, with the phase times of every window, and through the host's block-parallel inflater (the default)
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
N=${1:-2500000}
read FA F1 F2 NP < <(python3 tools/e2e_setup.py $N 2 | tail -1)
( gzip -1 -c $F1 > $W/g_1.fq.gz ) & ( gzip -1 -c $F2 > $W/g_2.fq.gz ) & wait
ls -la $W/g_1.fq.gz
echo "== pairs, device"
BMBS_GZ_DEVICE=2 BMBS_TEXT_TRACE=1 ./bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --seq1 $W/g_1.fq.gz --seq2 $W/g_2.fq.gz -o /dev/null -t 32 --verbose 2>&1 | grep -E "text open gzip|mapping wall" | tail -4 | cut -c1-330
echo "== one file, device"
BMBS_GZ_DEVICE=2 BMBS_TEXT_TRACE=1 ./bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --seq $W/g_1.fq.gz -o /dev/null -t 32 --verbose 2>&1 | grep -E "text open gzip|mapping wall" | tail -4 | cut -c1-330
echo "== pairs, host"
./bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --seq1 $W/g_1.fq.gz --seq2 $W/g_2.fq.gz -o /dev/null --verbose 2>&1 | grep -E "mapping wall" | cut -c1-330
rm -f $W/g_1.fq.gz $W/g_2.fq.gz
