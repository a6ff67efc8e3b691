#!/bin/bash
# tools/e2e_numa.sh -- start-up effects of a short run: buffers touched by the device while loading or not; -o /dev/null
read FA F1 F2 NP < <(python3 tools/e2e_setup.py 5000000 4 | tail -1)
run() { local label=$1; shift; "$@" 2>&1 | grep "mapping wall" | sed "s/^/$label: /" | cut -c1-250; }
D="./bitmapperbs_amd/bmbs_search --search $FA --seq1 $F1 --seq2 $F2 -e 0.08 -t 32 --verbose -o /dev/null"
run prefault_c3 $D --contexts 3
run prefault_c3_again $D --contexts 3
run noprefault_c3 env BMBS_NO_PREFAULT=1 $D --contexts 3
run prefault_c4 $D --contexts 4
echo "== trace, 3 contexts, first calls"
BMBS_TEXT_TRACE=1 $D --contexts 3 2>&1 | grep "^\[text\]" | cut -c1-260 | head -8
