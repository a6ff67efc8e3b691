#!/bin/bash
# tools/inflate_probe.sh [pairs] -- on the GPU box: BGZF FASTQ pairs through the device inflater, with the phase times of every window
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
read FA F1 F2 NP < <(python3 tools/e2e_setup.py ${1:-2500000} 2 | tail -1)
python3 - <<PY
import sys; sys.path.insert(0, ".")
import bench
for k in (1, 2):
    with open("$W/e2e_%d.fq" % k, "rb") as f: data = f.read()
    bench.write_bgzf("$W/b_%d.fq.gz" % k, data, level=1, threads=16)
PY
BMBS_TEXT_TRACE=1 ./bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --seq1 $W/b_1.fq.gz --seq2 $W/b_2.fq.gz -o /dev/null -t 32 --verbose 2>&1 | grep -E "text open|inflate|mapping wall" | head -16
