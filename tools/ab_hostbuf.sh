#!/bin/bash
# tools/ab_hostbuf.sh [pairs=2000000] VAR=a "VAR=b ..." ... -- GPU box: bmbs_map_pe_packed on page-locked host buffers (tools/hostbuf_probe.py, 6 calls
# each) under several environment settings, twice in turn: the calls' rates and their median
N=${1:-2000000}; shift
for rep in 1 2; do for kv in "$@"; do
  env $kv python3 tools/hostbuf_probe.py $N ${PACKED:-1} 6 2>/dev/null | python3 -c "
import sys,re,statistics
v=[float(re.search(r'= ([0-9.]+) M reads/s', l).group(1)) for l in sys.stdin if 'M reads/s' in l]
print('$kv', 'median %.1f' % statistics.median(v), ' '.join('%.0f' % x for x in v))"
done; done
