#!/bin/bash
# tools/kf_quick.sh -- the Myers filter after a change: parity cases that reach it, then GRCh38-like single-end at one and two lanes
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "filter or se_ or pe_ or mixed or packed or sensitive" 2>&1 | tail -3
for lanes in 1 2; do
  BMBS_LANES=$lanes timeout 600 python bench.py --grch38-like --se --launches 1 --steps 6 --no-cpu --no-secondary --no-single-lane 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes $lanes', d['value'], d['ms_per_step'], d.get('kernels_ms_per_launch',{}).get('k_filter'))"
done
