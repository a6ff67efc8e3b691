#!/bin/bash
# tools/ab_e2e_env.sh "<keys of e2e_quick.sh>" VAR=a "VAR=b OTHER=c" ... -- GPU box: the driver's file-to-file keys under several environment
# settings, twice each (a b a b)
KEYS=$1; shift
for rep in 1 2; do for kv in "$@"; do echo "== $kv"; env $kv tools/e2e_quick.sh $KEYS 2>&1 | grep -E "sink|bgzf|file|bam"; done; done
