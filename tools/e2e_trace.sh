#!/bin/bash
# tools/e2e_trace.sh [contexts] -- phase times of every bmbs_map_pe_text call (BMBS_TEXT_TRACE=1); then the copies alone
read FA F1 F2 NP < <(python3 tools/e2e_setup.py 5000000 4 | tail -1)
D="./bitmapperbs_amd/bmbs_search --search $FA --seq1 $F1 --seq2 $F2 -e 0.08 -t 32 --verbose -o /dev/null"
for c in ${1:-3}; do
  echo "== contexts $c"
  BMBS_TEXT_TRACE=1 $D --contexts $c 2>&1 | grep -E "^\[text\]|mapping wall" | cut -c1-330 | tail -6
  echo "== contexts $c, copies only"
  BMBS_TEXT_TRACE=1 BMBS_TEXT_COPY_ONLY=1 $D --contexts $c 2>&1 | grep -E "^\[text\]|mapping wall" | cut -c1-330 | tail -6
done
