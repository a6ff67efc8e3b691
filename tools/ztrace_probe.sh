#!/bin/bash
# tools/ztrace_probe.sh [pairs] -- on the GPU box: phase times of every text call of a BGZF run (device inflater) and of a plain-text run
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
N=${1:-2500000}; REP=${2:-2}
read FA F1 F2 NP < <(python3 tools/e2e_setup.py $N $REP | tail -1)
python3 - <<PY
import sys; sys.path.insert(0, ".")
import bench
for k in (1, 2):
    with open("$W/e2e_%d.fq" % k, "rb") as f: data = f.read()
    bench.write_bgzf("$W/b_%d.fq.gz" % k, data, level=1, threads=16)
PY
for c in 4 1; do
echo "== bgzf, device inflater, $c context(s)"
BMBS_TEXT_TRACE=1 ./bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --seq1 $W/b_1.fq.gz --seq2 $W/b_2.fq.gz -o /dev/null -t 32 --contexts $c --verbose 2>&1 | grep -E "^\[text|mapping wall" | tail -9
echo "== plain text, $c context(s)"
BMBS_TEXT_TRACE=1 ./bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --seq1 $F1 --seq2 $F2 -o /dev/null -t 32 --contexts $c --verbose 2>&1 | grep -E "^\[text|mapping wall" | tail -5
done
