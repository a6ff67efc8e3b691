#!/bin/bash
# tools/e2e_trace.sh -- phase times of every bmbs_map_pe_text call (BMBS_TEXT_TRACE=1), one context and three
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
read FA F1 F2 NP < <(python3 tools/e2e_setup.py 5000000 2 | tail -1)
for c in 1 3; do
  echo "== contexts $c"
  BMBS_TEXT_TRACE=1 ./bitmapperbs_amd/bmbs_search --search $FA --seq1 $F1 --seq2 $F2 -e 0.08 -t 32 --verbose -o /dev/null --contexts $c 2>&1 | grep -E "^\[text\]|mapping wall" | cut -c1-400
done
