#!/bin/bash
# tools/vote_kbench.sh -- here: builds tools/vote_kbench (this tree's k_vote.hip) and, when tools/_ab/k_vote_r05.hip exists (git show <round-5
# commit>:bitmapperbs_amd/csrc/k_vote.hip), tools/vote_kbench_old, each also with the phase cycle counters (-DVOTE_PROF).  GPU box: run them.
cd "$(dirname "$0")/.." || exit 1
H="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -w"
$H -o tools/vote_kbench tools/vote_kbench.hip && $H -DVOTE_PROF -o tools/vote_kbench_prof tools/vote_kbench.hip || exit 1
if [ -f tools/_ab/k_vote_r05.hip ]; then
  $H -DVB_OLD -DKVOTE_FILE='"../tools/_ab/k_vote_r05.hip"' -o tools/vote_kbench_old tools/vote_kbench.hip
  $H -DVB_OLD -DVOTE_PROF -DKVOTE_FILE='"../tools/_ab/k_vote_r05.hip"' -o tools/vote_kbench_old_prof tools/vote_kbench.hip
fi
